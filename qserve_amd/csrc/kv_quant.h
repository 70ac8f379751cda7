// kv_quant.h -- the ONE statement of the rotary embedding and of the KV-cache quantiser.
//
// Everything here must stay byte-identical between the prefill writer (attention.hip), the KV4 decode kernels
// (attention.hip, attention_mfma.hip) and the KV8 decode kernels (attention.hip, attention_mfma8.hip): the cache a decode
// step reads was written by any of them.  Reference statements followed (behaviour, not code):
//   RoPE coefficients / rotation ....... decoderMaskedMultiheadAttentionUtils.h:1147-1167, 2536-2557
//   scale / zero / inverse scale ....... decoderMaskedMultiheadAttentionTemplate.hpp:1051-1082, 1227-1258;
//                                        applyBiasRopeUpdateKVCache.h:288-331
//   float -> u8 (rni, saturate 0..255) . ...Utils.h:1687-1697;  nibble keep-low-4-bits (16 wraps to 0): :1838-1852
// Conventions where the reference is ambiguous: coefficients by pow / cos / sin in double on the float32
// inputs, rounded once; the rotation in fp32 WITHOUT contraction; the quantiser's `x * inv + zero` as ONE fmaf.
// Degenerate vectors follow IEEE as the reference's statements do: max == min gives scale = 0, zero = +-inf (NaN when the
// value is 0), inv = inf, and every byte 0 (the fma is NaN, cvt.rni.sat maps NaN to 0).
#pragma once
#include "common.h"

namespace {

struct RopeCS {
    float c, s;
};
// cos/sin of pos / base^(2i/dim): every step rounded to float32 from a double evaluation, so that host oracle and
// device agree bit for bit (the reference's fast-math __powf/__cosf are not reproducible anyway).
__device__ __forceinline__ RopeCS rope_coef(int pair, int pos, float base, int dim) {
    const float expo = (float)(2 * pair) / (float)dim;
    const float denom = (float)pow((double)base, (double)expo);
    const float ang = (float)pos / denom;
    RopeCS r;
    r.c = (float)cos((double)ang);
    r.s = (float)sin((double)ang);
    return r;
}
__device__ __forceinline__ void rope_pair(float a, float b, RopeCS cs, _Float16& oa, _Float16& ob) {
#pragma clang fp contract(off)
    const float ra = cs.c * a - cs.s * b;   // Utils.h:1157-1158
    const float rb = cs.c * b + cs.s * a;
    oa = (_Float16)ra;
    ob = (_Float16)rb;
}

struct QParams {
    _Float16 scale, zero;
    float inv;
};
// scale / zero / 1/scale of one (token, head) vector from its min and max (Template.hpp:1067, 1078-1079)
template <bool INT4>
__device__ __forceinline__ QParams make_qparams(float mn, float mx) {
    constexpr float levels = INT4 ? 15.f : 255.f;
    QParams p;
    const float rng = mx - mn;
    p.scale = (_Float16)(rng / levels);
    p.zero = (_Float16)((-levels * mn) / rng);
    p.inv = 1.0f / (float)p.scale;
    return p;
}
__device__ __forceinline__ unsigned quant_u8(_Float16 x, const QParams& p) {
    return rni_sat_u8(fmaf((float)x, p.inv, (float)p.zero));   // Utils.h:2045-2077 (nvcc contracts mul+add)
}

}  // namespace
