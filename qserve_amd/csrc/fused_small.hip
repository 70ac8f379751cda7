// fused_small.hip -- activation-side kernels adjacent to the W4A8 GEMMs (they produce A / ascales / a_ssums).
//
// Behaviour follows (not code):
//   invoke_quant(_fuse_sum) ......... kernels/csrc/fused_kernels.cu:52-137
//   rms_norm_general(_fuse_sum) ..... kernels/csrc/layernorm_kernels.cu:20-29,189-326,427-508  (mean-subtracting
//                                     generalLayerNorm with beta = nullptr, per-token dynamic int8 scaling)
//   rms_norm ........................ kernels/csrc/layernorm_kernels.cu:330-362
//   silu_and_mul .................... kernels/csrc/activation_kernels.cu:7-30
// HBM-bound row kernels: one 256-thread workgroup per token, 16-byte (8 x fp16) accesses, wave64 shuffles +
// one LDS hop for the block reductions.  hidden % 8 == 0.
#include "common.h"
#include "row_ops.h"

namespace {

constexpr int TPB = 256;

template <int NT = TPB>
__device__ __forceinline__ float block_reduce(float v, float* sm, int op) {
    // op 0 = sum, 1 = max ; returns the result to every thread (NT threads per workgroup)
    v = op == 0 ? wave_sum(v) : wave_max(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();   // protect sm reuse
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    float r = sm[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) r = op == 0 ? r + sm[w] : fmaxf(r, sm[w]);
    return r;
}
// One-barrier variants: every reduction round of a kernel gets its OWN LDS slots (`sm`, `sm2` never reused), so the
// leading "protect reuse" barrier is unnecessary, and a max and a sum that are needed at the same point travel together.
// Same shuffle trees and the same left-to-right combination over waves as block_reduce: bit-identical results.
template <int NT = TPB>
__device__ __forceinline__ float block_reduce_once(float v, float* sm, int op) {
    v = op == 0 ? wave_sum(v) : wave_max(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    float r = sm[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) r = op == 0 ? r + sm[w] : fmaxf(r, sm[w]);
    return r;
}
// rows wider than this are handled by 1024-thread workgroups (same rule in invoke_quant and silu_and_mul_quant, so the
// two associate their fp32 statistics identically)
constexpr int WIDE_ROW = qs_row::WIDE_ROW;

__device__ __forceinline__ h8 load8(const _Float16* p) { return *reinterpret_cast<const h8*>(p); }

// silu(x) = x / (1 + exp(-x)) rounded to fp16 (activation_kernels.cu:9-13).  The reference is compiled with
// --use_fast_math (kernels/setup.py:33): expf is ex2.approx(x * log2 e) and the division rcp.approx + multiply - the same
// hardware forms here (v_exp_f32, v_rcp_f32; ~1 ulp each, far below the fp16 rounding that follows).  The IEEE forms
// (libm expf + correctly rounded division) cost 37 VALU instructions per element and made silu_and_mul(+quant) the
// one VALU-bound row kernel: 7.0 us per decode launch, 1.13 ms per 65 536-token prompt layer.
__device__ __forceinline__ _Float16 silu_h(float xf) { return qs_silu_h(xf); }

__device__ __forceinline__ void store_q8(int8_t* p, const float (&v)[8], float mul) { qs_store_q8(p, v, mul); }

// ---------------------------------------------------------------------------------------------------------
// Row kernels keep the whole token row in registers (NC chunks of 8 fp16 per thread, hidden <= NC*2048): ONE global
// read, all statistics from registers.  At decode batch sizes these kernels are pure latency (64 workgroups), so
// every removed round trip to memory is ~1-2 us.
template <int NC, int NT>
__global__ __launch_bounds__(NT) void quant_kernel(int8_t* __restrict__ out, const _Float16* __restrict__ in,
                                                    __half* __restrict__ sum_out, __half* __restrict__ scale_out,
                                                    int hidden) {
    __shared__ float sm[(1 + NC) * (NT / 64)];
    const size_t base = (size_t)blockIdx.x * hidden;
    qs_row::quant_row<NC, NT / 64, NT / 64, false>(out + base, in + base, sum_out ? sum_out + blockIdx.x : nullptr,
                                                   scale_out + blockIdx.x, hidden, sm, (int)threadIdx.x);
}

// ---------------------------------------------------------------------------------------------------------

// REFSUM instantiations (qs_set_row_sum_order(1)): the row sum in the reference's order, `hidden` halves of dynamic LDS.
extern __shared__ __attribute__((aligned(16))) _Float16 s_hv_dyn[];

template <int NC, bool REFSUM = false>
__global__ __launch_bounds__(TPB) void general_norm_quant_kernel(int8_t* __restrict__ out,
                                                                 const _Float16* __restrict__ in,
                                                                 const _Float16* __restrict__ gamma,
                                                                 __half* __restrict__ sum_out,
                                                                 __half* __restrict__ scale_out, float eps,
                                                                 int hidden) {
    __shared__ float sm[4 * (TPB / 64) + (REFSUM ? 32 : 0)];
    const size_t base = (size_t)blockIdx.x * hidden;
    qs_row::norm_quant_row<NC, TPB / 64, TPB / 64, false, false, qs_row::NoHook, qs_row::FromRow, false, REFSUM>(
        out + base, const_cast<_Float16*>(in) + base, nullptr, gamma, sum_out ? sum_out + blockIdx.x : nullptr,
        scale_out + blockIdx.x, eps, hidden, sm, (int)threadIdx.x, qs_row::NoHook(), qs_row::FromRow(),
        REFSUM ? s_hv_dyn : nullptr);
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void rms_norm_kernel(_Float16* __restrict__ out, const _Float16* __restrict__ in,
                                                       const _Float16* __restrict__ w, float eps, int hidden) {
    __shared__ float sm[TPB / 64];
    const size_t base = (size_t)blockIdx.x * hidden;
    float vs = 0.f;
    for (int i = threadIdx.x * 8; i < hidden; i += TPB * 8) {
        h8 v = load8(in + base + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) vs += (float)v[j] * (float)v[j];
    }
    const float rstd = 1.0f / sqrtf(block_reduce_once(vs, sm, 0) / hidden + eps);   // :347
    for (int i = threadIdx.x * 8; i < hidden; i += TPB * 8) {
        h8 v = load8(in + base + i);
        h8 g = load8(w + i);
        h8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            _Float16 t = (_Float16)((float)v[j] * rstd);                 // (scalar_t)(x * s_variance), :358
            o[j] = (_Float16)((float)t * (float)g[j]);                   // half * half
        }
        *reinterpret_cast<h8*>(out + base + i) = o;
    }
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void silu_and_mul_kernel(_Float16* __restrict__ out,
                                                           const _Float16* __restrict__ in, int d) {
    const size_t ib = (size_t)blockIdx.x * 2 * d, ob = (size_t)blockIdx.x * d;
    for (int i = (blockIdx.y * TPB + threadIdx.x) * 8; i < d; i += gridDim.y * TPB * 8) {
        h8 x = load8(in + ib + i);
        h8 y = load8(in + ib + d + i);
        h8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xf = (float)x[j];
            _Float16 s = silu_h(xf);                                     // activation_kernels.cu:11
            o[j] = (_Float16)((float)s * (float)y[j]);
        }
        *reinterpret_cast<h8*>(out + ob + i) = o;
    }
}

__global__ __launch_bounds__(TPB) void residual_add_kernel(_Float16* __restrict__ a, const _Float16* __restrict__ b,
                                                           int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n8; i += (int64_t)gridDim.x * TPB) {
        h8 x = reinterpret_cast<const h8*>(a)[i];
        h8 y = reinterpret_cast<const h8*>(b)[i];
        reinterpret_cast<h8*>(a)[i] = x + y;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Pair fusions for the decode loop.  At batch 64 every one of the row kernels above is a ~4.7 us latency chain
// (launch -> load -> reduce -> store), so two adjacent ops in one launch cost the same as one.  Both fusions replicate
// the intermediate fp16 rounding of the op pair they replace, i.e. they are bit-identical to calling the two ops.
//
//   add_residual_norm_quant : hidden += delta (fp16 add, written back) ; rms_norm_general(_fuse_sum)(hidden)
template <int NC, bool FULL, bool REFSUM = false>
__global__ __launch_bounds__(TPB) void add_residual_norm_quant_kernel(int8_t* __restrict__ out,
                                                                      _Float16* __restrict__ hidden_io,
                                                                      const _Float16* __restrict__ delta,
                                                                      const _Float16* __restrict__ gamma,
                                                                      __half* __restrict__ sum_out,
                                                                      __half* __restrict__ scale_out, float eps,
                                                                      int hidden) {
    __shared__ float sm[4 * (TPB / 64) + (REFSUM ? 32 : 0)];
    const size_t base = (size_t)blockIdx.x * hidden;
    qs_row::norm_quant_row<NC, TPB / 64, TPB / 64, true, false, qs_row::NoHook, qs_row::FromRow, FULL, REFSUM>(
        out + base, hidden_io + base, delta + base, gamma, sum_out ? sum_out + blockIdx.x : nullptr,
        scale_out + blockIdx.x, eps, hidden, sm, (int)threadIdx.x, qs_row::NoHook(), qs_row::FromRow(),
        REFSUM ? s_hv_dyn : nullptr);
}

//   add_residual_norm_quant over K-slice PLANES (round 4): the residual branch arrives as the int32 planes a W4A8 GEMM left
//   (qs_w4a8_*_gemm_planes) instead of its fp16 output; delta = fp16(GEMM epilogue(sum of the planes)) is formed here, bit for
//   bit what the GEMM would have stored, then everything is add_residual_norm_quant.  `ascale` / `asum` are the activation scale /
//   sum the GEMM's INPUT was quantised with; they may alias scale_out / sum_out (a row reads its own values before it writes).
template <int NC, int KS, int MODE, bool FULL, bool REFSUM = false>
__global__ __launch_bounds__(TPB) void add_residual_norm_quant_planes_kernel(
    int8_t* __restrict__ out, _Float16* __restrict__ hidden_io, const int* __restrict__ planes, size_t pstride,
    const _Float16* __restrict__ wscales, const _Float16* __restrict__ wszs, const __half* ascale, const __half* asum,
    const _Float16* __restrict__ gamma, __half* sum_out, __half* scale_out, float eps, int hidden, int epi_fma) {
    __shared__ float sm[4 * (TPB / 64) + (REFSUM ? 32 : 0)];
    const size_t base = (size_t)blockIdx.x * hidden;
    qs_row::FromPlanes<KS, MODE> dfn;
    dfn.fma = epi_fma;
    dfn.row0 = planes + base;
    dfn.pstride = pstride;
    dfn.ws = wscales;
    dfn.wz = wszs;
    dfn.sa = __half2float(ascale[blockIdx.x]);
    dfn.ss = MODE == 0 ? __half2float(asum[blockIdx.x]) : 0.f;
    qs_row::norm_quant_row<NC, TPB / 64, TPB / 64, true, false, qs_row::NoHook, qs_row::FromPlanes<KS, MODE>, FULL, REFSUM>(
        out + base, hidden_io + base, nullptr, gamma, sum_out ? sum_out + blockIdx.x : nullptr, scale_out + blockIdx.x, eps,
        hidden, sm, (int)threadIdx.x, qs_row::NoHook(), dfn, REFSUM ? s_hv_dyn : nullptr);
}

//   silu_mul_quant : act = silu_and_mul(input) rounded to fp16 (never written) ; invoke_quant(_fuse_sum)(act)
// Same thread -> element mapping and reduction order as quant_kernel, so the fp32 statistics associate identically.
template <int NC, int NT>
__global__ __launch_bounds__(NT) void silu_mul_quant_kernel(int8_t* __restrict__ out, const _Float16* __restrict__ in,
                                                             __half* __restrict__ sum_out,
                                                             __half* __restrict__ scale_out, int d) {
    __shared__ float sm[(1 + NC) * (NT / 64)];
    const size_t ib = (size_t)blockIdx.x * 2 * d, ob = (size_t)blockIdx.x * d;
    h8 x[NC], y[NC], o[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {                     // all loads first: 2*NC 16-byte requests in flight per thread
        const int i = (c * NT + threadIdx.x) * 8;
        if (i < d) {
            x[c] = load8(in + ib + i);
            y[c] = load8(in + ib + d + i);
        }
    }
    float amax1[1] = {0.f}, sum1[1][NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        sum1[0][c] = 0.f;
        const int i = (c * NT + threadIdx.x) * 8;
        if (i < d) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xf = (float)x[c][j];
                const _Float16 sl = silu_h(xf);                             // silu_and_mul_kernel
                o[c][j] = (_Float16)((float)sl * (float)y[c][j]);
                const float f = (float)o[c][j];
                sum1[0][c] += f;                                            // quant_kernel's statistics (block order, row_ops.h)
                QS_SEQ(sum1[0][c]);
                amax1[0] = fmaxf(amax1[0], fabsf(f));
            }
        }
    }
    float amax, sum;
    qs_row::reduce_max_blocksum<NC, NT / 64, NT / 64>(amax1, sum1, sm, sm + NT / 64, sum_out != nullptr, (d + 511) / 512,
                                                      (int)(threadIdx.x >> 6), (int)(threadIdx.x & 63), amax, sum);
    if (threadIdx.x == 0) {
        scale_out[blockIdx.x] = __float2half_rn(amax / 127.0f);
        if (sum_out) sum_out[blockIdx.x] = __float2half_rn(sum);
    }
    const float mul = 127.0f / amax;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int i = (c * NT + threadIdx.x) * 8;
        if (i < d) {
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = (float)o[c][j];
            store_q8(out + ob + i, f, mul);
        }
    }
}

qs_flag g_row_refsum = 0;

}  // namespace

extern "C" int qs_set_row_sum_order(int order) {
    QS_REQUIRE(order == 0 || order == 1, "qs_set_row_sum_order: %d not in {0 = this library's fp32 chains, 1 = the reference's order}", order);
    g_row_refsum = order;
    return QS_OK;
}
extern "C" int qs_get_row_sum_order(void) { return g_row_refsum; }

extern "C" int qs_invoke_quant(int8_t* out, const void* input, void* input_sum, void* scale, int num_tokens,
                               int hidden, qs_stream_t stream) {
    QS_REQUIRE(out && input && scale, "invoke_quant: null pointer");
    QS_REQUIRE(hidden > 0 && hidden % 8 == 0, "invoke_quant: hidden=%d must be a positive multiple of 8", hidden);
    if (num_tokens <= 0) return QS_OK;
    QS_REQUIRE(hidden <= 32768, "invoke_quant: hidden=%d larger than 32768 is not supported", hidden);
#define QS_Q(NC, NT)                                                                                      \
    hipLaunchKernelGGL((quant_kernel<NC, NT>), dim3(num_tokens), dim3(NT), 0, (hipStream_t)stream, out,     \
                       (const _Float16*)input, (__half*)input_sum, (__half*)scale, hidden)
    if (hidden > WIDE_ROW) {
        if (hidden <= 8192) QS_Q(1, 1024);
        else if (hidden <= 16384) QS_Q(2, 1024);
        else QS_Q(4, 1024);
    } else {
        const int nc = (hidden + TPB * 8 - 1) / (TPB * 8);
        if (nc == 1) QS_Q(1, TPB);
        else QS_Q(2, TPB);
    }
#undef QS_Q
    return qs_launch_status("invoke_quant");
}

extern "C" int qs_rms_norm_general(int8_t* out, const void* input, const void* weight, void* input_sum, void* scaling,
                                   float epsilon, int num_tokens, int hidden, qs_stream_t stream) {
    QS_REQUIRE(out && input && weight && scaling, "rms_norm_general: null pointer");
    QS_REQUIRE(hidden > 0 && hidden % 8 == 0, "rms_norm_general: hidden=%d must be a positive multiple of 8", hidden);
    if (num_tokens <= 0) return QS_OK;
    QS_REQUIRE(hidden <= 8 * TPB * 8, "rms_norm_general: hidden=%d larger than %d is not supported", hidden, 8 * TPB * 8);
    const int nc = (hidden + TPB * 8 - 1) / (TPB * 8);
    const bool refsum = g_row_refsum && input_sum;   // (the order of the row sum matters only where a sum is asked for)
#define QS_N(NC)                                                                                                         \
    do {                                                                                                                 \
        if (refsum)                                                                                                      \
            hipLaunchKernelGGL((general_norm_quant_kernel<NC, true>), dim3(num_tokens), dim3(TPB), hidden * 2,           \
                               (hipStream_t)stream, out, (const _Float16*)input, (const _Float16*)weight,                \
                               (__half*)input_sum, (__half*)scaling, epsilon, hidden);                                   \
        else                                                                                                             \
            hipLaunchKernelGGL((general_norm_quant_kernel<NC, false>), dim3(num_tokens), dim3(TPB), 0,                   \
                               (hipStream_t)stream, out, (const _Float16*)input, (const _Float16*)weight,                \
                               (__half*)input_sum, (__half*)scaling, epsilon, hidden);                                   \
    } while (0)
    switch (nc) {
        case 1: QS_N(1); break;
        case 2: QS_N(2); break;
        case 3: case 4: QS_N(4); break;
        default: QS_N(8); break;
    }
#undef QS_N
    return qs_launch_status("rms_norm_general");
}

extern "C" int qs_rms_norm(void* out, const void* input, const void* weight, float epsilon, int num_tokens, int hidden,
                           qs_stream_t stream) {
    QS_REQUIRE(out && input && weight, "rms_norm: null pointer");
    QS_REQUIRE(hidden > 0 && hidden % 8 == 0, "rms_norm: hidden=%d must be a positive multiple of 8", hidden);
    if (num_tokens <= 0) return QS_OK;
    hipLaunchKernelGGL(rms_norm_kernel, dim3(num_tokens), dim3(TPB), 0, (hipStream_t)stream, (_Float16*)out,
                       (const _Float16*)input, (const _Float16*)weight, epsilon, hidden);
    return qs_launch_status("rms_norm");
}

extern "C" int qs_silu_and_mul(void* out, const void* input, int num_tokens, int d, qs_stream_t stream) {
    QS_REQUIRE(out && input, "silu_and_mul: null pointer");
    QS_REQUIRE(d > 0 && d % 8 == 0, "silu_and_mul: d=%d must be a positive multiple of 8", d);
    if (num_tokens <= 0) return QS_OK;
    int chunks = (d + TPB * 8 - 1) / (TPB * 8);
    if (chunks > 16) chunks = 16;
    hipLaunchKernelGGL(silu_and_mul_kernel, dim3(num_tokens, chunks), dim3(TPB), 0, (hipStream_t)stream, (_Float16*)out,
                       (const _Float16*)input, d);
    return qs_launch_status("silu_and_mul");
}

extern "C" int qs_residual_add(void* a, const void* b, int64_t numel, qs_stream_t stream) {
    QS_REQUIRE(a && b, "residual_add: null pointer");
    QS_REQUIRE(numel % 8 == 0, "residual_add: numel must be a multiple of 8");
    if (numel <= 0) return QS_OK;
    const int64_t n8 = numel / 8;
    int blocks = (int)((n8 + TPB - 1) / TPB);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(residual_add_kernel, dim3(blocks), dim3(TPB), 0, (hipStream_t)stream, (_Float16*)a,
                       (const _Float16*)b, n8);
    return qs_launch_status("residual_add");
}

/* ---- greedy sampling helper: row-wise argmax of fp16 logits (first maximum wins; NaN-free input) -------------------- */
namespace {
__device__ __forceinline__ void argmax_pick(float& bv, int& bi, float v, int i) {
    if (v > bv || (v == bv && i < bi)) {
        bv = v;
        bi = i;
    }
}
// one workgroup of 1024 threads per row; every thread requests all its 16-byte chunks before looking at any
template <int MAXC>
__global__ __launch_bounds__(1024) void argmax_rows_kernel(const _Float16* __restrict__ x, int64_t* __restrict__ out,
                                                            int n, int64_t row_stride) {
    const _Float16* row = x + (size_t)blockIdx.x * row_stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nchunk = n >> 3;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c0 = 0; c0 < nchunk; c0 += MAXC * 1024) {       // one trip up to MAXC * 8192 columns
        h8 v[MAXC];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {               // unconditional (clamped) loads: a load under a branch makes the compiler
            const int ch = c0 + tid + c * 1024;        // drain the queue at every merge point - one load in flight at a time
            v[c] = load8(row + (size_t)(ch < nchunk ? ch : nchunk - 1) * 8);
        }
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ch = c0 + tid + c * 1024;
            if (ch < nchunk) {
#pragma unroll
                for (int e = 0; e < 8; ++e) argmax_pick(bv, bi, (float)v[c][e], ch * 8 + e);
            }
        }
    }
    for (int i = (nchunk << 3) + tid; i < n; i += 1024) argmax_pick(bv, bi, (float)row[i], i);   // n % 8 tail
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float ov = __shfl_xor(bv, m, 64);
        const int oi = __shfl_xor(bi, m, 64);
        argmax_pick(bv, bi, ov, oi);
    }
    __shared__ float s_v[16];
    __shared__ int s_i[16];
    if (lane == 0) {
        s_v[wave] = bv;
        s_i[wave] = bi;
    }
    __syncthreads();
    if (wave == 0) {
        bv = lane < 16 ? s_v[lane] : -INFINITY;
        bi = lane < 16 ? s_i[lane] : 0x7fffffff;
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) {
            const float ov = __shfl_xor(bv, m, 64);
            const int oi = __shfl_xor(bi, m, 64);
            argmax_pick(bv, bi, ov, oi);
        }
        if (lane == 0) out[blockIdx.x] = bi == 0x7fffffff ? 0 : bi;
    }
}
// Few long rows (the greedy sampler of a 64-sequence step: 64 rows x 128 256 logits) leave 3/4 of the CUs idle and
// make the row's 16 waves VALU-bound: SPLIT workgroups per row, each reducing a contiguous range of 16-byte chunks; the
// candidates meet in a 64-bit key (orderable value bits << 32 | ~index: the maximum key is the first maximum) by a
// device-scope atomic max, an arrival ticket tells the last workgroup of a row to decode the key and to put key and
// ticket back to zero (nothing for the host to reset between launches or graph replays).
template <int MAXC>
__global__ __launch_bounds__(1024) void argmax_rows_split_kernel(const _Float16* __restrict__ x, int64_t* __restrict__ out,
                                                                  int n, int64_t row_stride,
                                                                  unsigned long long* __restrict__ keys,
                                                                  unsigned* __restrict__ tickets) {
    const int r = blockIdx.x, part = blockIdx.y, nsplit = gridDim.y;
    const _Float16* row = x + (size_t)r * row_stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nchunk = n >> 3;
    const int per = (nchunk + nsplit - 1) / nsplit;
    const int c_lo = part * per, c_hi = min(nchunk, c_lo + per);
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c0 = c_lo; c0 < c_hi; c0 += MAXC * 1024) {
        h8 v[MAXC];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {               // unconditional (clamped) loads, see argmax_rows_kernel
            const int ch = c0 + tid + c * 1024;
            v[c] = load8(row + (size_t)(ch < c_hi ? ch : c_hi - 1) * 8);
        }
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ch = c0 + tid + c * 1024;
            if (ch < c_hi) {
#pragma unroll
                for (int e = 0; e < 8; ++e) argmax_pick(bv, bi, (float)v[c][e], ch * 8 + e);
            }
        }
    }
    if (part == nsplit - 1)
        for (int i = (nchunk << 3) + tid; i < n; i += 1024) argmax_pick(bv, bi, (float)row[i], i);   // n % 8 tail
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float ov = __shfl_xor(bv, m, 64);
        const int oi = __shfl_xor(bi, m, 64);
        argmax_pick(bv, bi, ov, oi);
    }
    __shared__ float s_v[16];
    __shared__ int s_i[16];
    if (lane == 0) {
        s_v[wave] = bv;
        s_i[wave] = bi;
    }
    __syncthreads();
    if (wave != 0) return;
    bv = lane < 16 ? s_v[lane] : -INFINITY;
    bi = lane < 16 ? s_i[lane] : 0x7fffffff;
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) {
        const float ov = __shfl_xor(bv, m, 64);
        const int oi = __shfl_xor(bi, m, 64);
        argmax_pick(bv, bi, ov, oi);
    }
    if (lane != 0) return;
    u32 u = __builtin_bit_cast(u32, bv);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);                 // monotone in the float order, never 0
    const unsigned long long key = ((unsigned long long)u << 32) | (0xFFFFFFFFu - (u32)bi);
    // The ticket must not overtake the key: keys[r] and tickets[r] are different addresses (possibly different L2 channels)
    // and relaxed atomics are not ordered among themselves.  A RETURNING atomic has been performed at the device-coherent
    // level once its value is back, so the maximum's return value is consumed (the compiler waits for it) before the
    // ticket is drawn; the last arriver's exchange then sees every part's maximum.
    const unsigned long long prev = __hip_atomic_fetch_max(keys + r, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::"v"((u32)prev), "v"((u32)(prev >> 32)) : "memory");
    const unsigned t = __hip_atomic_fetch_add(tickets + r, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t != (unsigned)(nsplit - 1)) return;
    const unsigned long long k = __hip_atomic_exchange(keys + r, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(tickets + r, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u32 idx = 0xFFFFFFFFu - (u32)(k & 0xFFFFFFFFu);
    out[r] = idx == 0x7fffffffu ? 0 : (int64_t)idx;
}

// per-device keys / tickets of the split form (zero between launches); never allocated while a stream is capturing
struct ArgmaxWs {
    unsigned long long* keys = nullptr;
    unsigned* tickets = nullptr;
    bool tried = false;
};
constexpr int ARGMAX_WS_ROWS = 8192;
ArgmaxWs* argmax_ws(hipStream_t stream) {
    static ArgmaxWs ws[QS_MAX_DEVICES][QS_MAX_STREAM_SLOTS];
    ArgmaxWs& w = ws[qs_device_slot()][qs_scratch_slot(stream)];
    if (!w.tried) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
            (void)hipGetLastError();
            return nullptr;                            // first use inside a capture: one workgroup per row this time
        }
        w.tried = true;
        void* a = nullptr;
        // (the memset runs on the NULL stream: synchronise, or a launch on a non-blocking stream could overtake it)
        if (hipMalloc(&a, ARGMAX_WS_ROWS * 12) == hipSuccess && hipMemset(a, 0, ARGMAX_WS_ROWS * 12) == hipSuccess &&
            hipDeviceSynchronize() == hipSuccess) {
            w.keys = reinterpret_cast<unsigned long long*>(a);
            w.tickets = reinterpret_cast<unsigned*>(w.keys + ARGMAX_WS_ROWS);
        } else {
            (void)hipGetLastError();
        }
    }
    return w.keys ? &w : nullptr;
}
}  // namespace

bool qs_argmax_scratch_prealloc(hipStream_t stream) { return argmax_ws(stream) != nullptr; }
qs_flag g_argmax_split = -1;   // qs_debug_argmax_split: -1 heuristic, 1 one workgroup per row, >= 2 forced split (tests, A/B)
extern "C" void qs_debug_argmax_split(int split) { g_argmax_split = split; }

extern "C" int qs_argmax_rows(const void* x, int64_t* out, int rows, int n, int64_t row_stride, qs_stream_t stream) {
    QS_REQUIRE(x && out, "argmax_rows: null pointer");
    QS_REQUIRE(n >= 8 && row_stride >= n && row_stride % 8 == 0, "argmax_rows: n=%d (>= 8), row stride %lld (must be >= n, multiple of 8)",
               n, (long long)row_stride);
    if (rows <= 0) return QS_OK;
    // split rows over several workgroups while the grid stays within about one round of the chip and every part keeps
    // at least 2 chunks per thread
    int split = 1;
    if (g_argmax_split >= 2) split = g_argmax_split;
    else if (g_argmax_split < 0)
        while (split < 8 && rows * split * 2 <= qs_num_cus() && (n >> 3) / (split * 2) >= 2048) split *= 2;
    if (split > 1 && rows <= ARGMAX_WS_ROWS) {
        if (ArgmaxWs* w = argmax_ws((hipStream_t)stream)) {
            const int per = ((n >> 3) + split - 1) / split, pc = (per + 1023) / 1024;
            auto launch_s = [&](auto k) {
                hipLaunchKernelGGL(k, dim3(rows, split), dim3(1024), 0, (hipStream_t)stream, (const _Float16*)x, out, n,
                                   row_stride, w->keys, w->tickets);
            };
            if (pc <= 2) launch_s(argmax_rows_split_kernel<2>);
            else if (pc <= 4) launch_s(argmax_rows_split_kernel<4>);
            else if (pc <= 8) launch_s(argmax_rows_split_kernel<8>);
            else launch_s(argmax_rows_split_kernel<16>);
            return qs_launch_status("argmax_rows");
        }
    }
    const int chunks = ((n >> 3) + 1023) / 1024;
    auto launch = [&](auto k) {
        hipLaunchKernelGGL(k, dim3(rows), dim3(1024), 0, (hipStream_t)stream, (const _Float16*)x, out, n, row_stride);
    };
    if (chunks <= 4) launch(argmax_rows_kernel<4>);
    else if (chunks <= 8) launch(argmax_rows_kernel<8>);
    else launch(argmax_rows_kernel<16>);
    return qs_launch_status("argmax_rows");
}

/* ---- pair fusions (bit-identical to the two ops they replace; see the kernels) ---------------------------------- */
extern "C" int qs_add_residual_rms_norm_general(int8_t* out, void* hidden_io, const void* delta, const void* weight,
                                                void* input_sum, void* scaling, float epsilon, int num_tokens,
                                                int hidden, qs_stream_t stream) {
    QS_REQUIRE(out && hidden_io && delta && weight && scaling, "add_residual_rms_norm_general: null pointer");
    QS_REQUIRE(hidden > 0 && hidden % 8 == 0, "add_residual_rms_norm_general: hidden=%d must be a positive multiple of 8",
               hidden);
    if (num_tokens <= 0) return QS_OK;
    QS_REQUIRE(hidden <= 8 * TPB * 8, "add_residual_rms_norm_general: hidden=%d larger than %d is not supported", hidden,
               8 * TPB * 8);
    const int nc = (hidden + TPB * 8 - 1) / (TPB * 8);
    const bool refsum = g_row_refsum && input_sum;
#define QS_N(NC)                                                                                                      \
    do {                                                                                                              \
        if (refsum)                                                                                                   \
            hipLaunchKernelGGL((add_residual_norm_quant_kernel<NC, false, true>), dim3(num_tokens), dim3(TPB),        \
                               hidden * 2, (hipStream_t)stream, out, (_Float16*)hidden_io, (const _Float16*)delta,    \
                               (const _Float16*)weight, (__half*)input_sum, (__half*)scaling, epsilon, hidden);       \
        else if (hidden == NC * TPB * 8)                                                                              \
            hipLaunchKernelGGL((add_residual_norm_quant_kernel<NC, true>), dim3(num_tokens), dim3(TPB), 0,            \
                               (hipStream_t)stream, out, (_Float16*)hidden_io, (const _Float16*)delta,                \
                               (const _Float16*)weight, (__half*)input_sum, (__half*)scaling, epsilon, hidden);       \
        else                                                                                                          \
            hipLaunchKernelGGL((add_residual_norm_quant_kernel<NC, false>), dim3(num_tokens), dim3(TPB), 0,           \
                               (hipStream_t)stream, out, (_Float16*)hidden_io, (const _Float16*)delta,                \
                               (const _Float16*)weight, (__half*)input_sum, (__half*)scaling, epsilon, hidden);       \
    } while (0)
    switch (nc) {
        case 1: QS_N(1); break;
        case 2: QS_N(2); break;
        case 3: case 4: QS_N(4); break;
        default: QS_N(8); break;
    }
#undef QS_N
    return qs_launch_status("add_residual_rms_norm_general");
}

extern "C" int qs_add_residual_rms_norm_general_planes(int8_t* out, void* hidden_io, const int32_t* planes, int k_slices,
                                                       int64_t plane_stride, const void* wscales, const void* w_szs,
                                                       const void* ascales, const void* a_ssums, const void* weight,
                                                       void* input_sum, void* scaling, float epsilon, int num_tokens,
                                                       int hidden, qs_stream_t stream) {
    QS_REQUIRE(out && hidden_io && planes && wscales && ascales && weight && scaling,
               "add_residual_rms_norm_general_planes: null pointer");
    QS_REQUIRE((w_szs == nullptr) == (a_ssums == nullptr),
               "add_residual_rms_norm_general_planes: w_szs and a_ssums come together (per-channel) or not at all (per-group)");
    QS_REQUIRE(k_slices == 1 || k_slices == 2 || k_slices == 4, "add_residual_rms_norm_general_planes: k_slices=%d not in {1, 2, 4}",
               k_slices);
    QS_REQUIRE(hidden > 0 && hidden % 8 == 0 && plane_stride >= (int64_t)num_tokens * hidden && plane_stride % 4 == 0,
               "add_residual_rms_norm_general_planes: hidden=%d (multiple of 8) / plane stride %lld", hidden, (long long)plane_stride);
    QS_REQUIRE(!((reinterpret_cast<uintptr_t>(planes) | reinterpret_cast<uintptr_t>(wscales) | reinterpret_cast<uintptr_t>(w_szs)) & 15),
               "add_residual_rms_norm_general_planes: planes / wscales / w_szs must be 16-byte aligned");
    if (num_tokens <= 0) return QS_OK;
    QS_REQUIRE(hidden <= 2 * TPB * 8, "add_residual_rms_norm_general_planes: hidden=%d larger than %d is not supported", hidden,
               2 * TPB * 8);
    const int nc = (hidden + TPB * 8 - 1) / (TPB * 8);
    const bool refsum = g_row_refsum && input_sum;
#define QS_P(NC, KS, MODE)                                                                                                \
    hipLaunchKernelGGL((add_residual_norm_quant_planes_kernel<NC, KS, MODE, FULLV, REFV>), dim3(num_tokens), dim3(TPB),    \
                       REFV ? hidden * 2 : 0, (hipStream_t)stream, out, (_Float16*)hidden_io, planes, (size_t)plane_stride, \
                       (const _Float16*)wscales, (const _Float16*)w_szs, (const __half*)ascales, (const __half*)a_ssums,  \
                       (const _Float16*)weight, (__half*)input_sum, (__half*)scaling, epsilon, hidden, g_epi_fma)
#define QS_PF(NC, KS, MODE)                                       \
    do {                                                          \
        if (refsum) {                                             \
            constexpr bool FULLV = false, REFV = true;            \
            QS_P(NC, KS, MODE);                                   \
        } else if (hidden == NC * TPB * 8) {                      \
            constexpr bool REFV = false;                          \
            constexpr bool FULLV = true;                          \
            QS_P(NC, KS, MODE);                                   \
        } else {                                                  \
            constexpr bool FULLV = false, REFV = false;           \
            QS_P(NC, KS, MODE);                                   \
        }                                                         \
    } while (0)
#define QS_PK(NC, MODE)                         \
    do {                                        \
        if (k_slices == 1) QS_PF(NC, 1, MODE);  \
        else if (k_slices == 2) QS_PF(NC, 2, MODE); \
        else QS_PF(NC, 4, MODE);                \
    } while (0)
    if (w_szs) {
        if (nc == 1) QS_PK(1, 0);
        else QS_PK(2, 0);
    } else {
        if (nc == 1) QS_PK(1, 1);
        else QS_PK(2, 1);
    }
#undef QS_PK
#undef QS_PF
#undef QS_P
    return qs_launch_status("add_residual_rms_norm_general_planes");
}

extern "C" int qs_silu_and_mul_quant(int8_t* out, const void* input, void* input_sum, void* scale, int num_tokens,
                                     int d, qs_stream_t stream) {
    QS_REQUIRE(out && input && scale, "silu_and_mul_quant: null pointer");
    QS_REQUIRE(d > 0 && d % 8 == 0, "silu_and_mul_quant: d=%d must be a positive multiple of 8", d);
    if (num_tokens <= 0) return QS_OK;
    QS_REQUIRE(d <= 32768, "silu_and_mul_quant: d=%d larger than 32768 is not supported", d);
#define QS_S(NC, NT)                                                                                                \
    hipLaunchKernelGGL((silu_mul_quant_kernel<NC, NT>), dim3(num_tokens), dim3(NT), 0, (hipStream_t)stream, out,       \
                       (const _Float16*)input, (__half*)input_sum, (__half*)scale, d)
    if (d > WIDE_ROW) {                                  // same geometry rule as qs_invoke_quant
        if (d <= 8192) QS_S(1, 1024);
        else if (d <= 16384) QS_S(2, 1024);
        else QS_S(4, 1024);
    } else {
        const int nc = (d + TPB * 8 - 1) / (TPB * 8);
        if (nc == 1) QS_S(1, TPB);
        else QS_S(2, TPB);
    }
#undef QS_S
    return qs_launch_status("silu_and_mul_quant");
}
