// Round 5: issue rate of the VALU instructions the per-group level-2 dequant is (or could be) made of, and the two dequant
// sequences head to head.  One workgroup of 256 threads per CU x WPS workgroups (1 or 2 waves per SIMD), 16 independent chains
// per lane so that latency never binds.  Reported: ns per wave-instruction per SIMD relative to v_add_u32.
//   dequant A (library, common.h unpack_lo/hi<1>):  u = raw & 0x0F0F0F0F ; vadd4(u * s, zb)               (32-bit multiply)
//   dequant B (candidate): nibbles in 16-bit lanes, v_pk_mad_u16 (product + zero byte in one instruction, carries die in the
//              lane), two lanesets merged by v_perm_b32 - equal to A whenever no byte product exceeds 255 (s <= 17)
// build: hipcc -O3 --offload-arch=gfx950 scripts/microbench_valu.hip -o /tmp/mb_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32;

__device__ __forceinline__ u32 vadd4(u32 a, u32 b) { return ((a & 0x7f7f7f7fu) + (b & 0x7f7f7f7fu)) ^ ((a ^ b) & 0x80808080u); }

template <int OP>
__device__ __forceinline__ u32 op(u32 x, u32 s, u32 z) {
    u32 r;
    if (OP == 0) asm volatile("v_add_u32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(s));
    if (OP == 1) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(s));
    if (OP == 2) asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(x), "v"(s));
    if (OP == 3) asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(s), "v"(z));
    if (OP == 4) asm volatile("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(s), "v"(z));
    if (OP == 5) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(s), "v"(z));
    if (OP == 6) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(s), "v"(z));
    if (OP == 7) asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(s), "v"(z));
    if (OP == 8) asm volatile("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(s));
    if (OP == 9) asm volatile("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(r) : "v"(x), "v"(s));
    return r;
}

template <int OP>
__global__ __launch_bounds__(256) void k_rate(u32* out, int iters) {
    u32 x[16];
    const u32 t = threadIdx.x + blockIdx.x * 256;
    for (int i = 0; i < 16; ++i) x[i] = t * 0x9E3779B9u + i;
    const u32 s = (t & 15) + 1, z = t | 0x01020304u;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = op<OP>(x[i], s, z);
    u32 a = 0;
    for (int i = 0; i < 16; ++i) a ^= x[i];
    out[t] = a;
}

// dequant of one raw dword (8 nibbles) into the two operand dwords
template <int V>
__device__ __forceinline__ void dq(u32 raw, u32 s, u32 zb, u32 s16, u32 s16h, u32 zb16, u32 zb16h, u32& lo, u32& hi) {
    if (V == 0) {
        lo = vadd4((raw & 0x0F0F0F0Fu) * s, zb);
        hi = vadd4(((raw >> 4) & 0x0F0F0F0Fu) * s, zb);
    } else {
        const u32 t = raw >> 8;
        u32 e, o, e2, o2;
        // low nibbles: lanes hold n in byte 0 -> result in byte 0 of each 16-bit lane
        asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(e) : "v"(raw & 0x000F000Fu), "v"(s16), "v"(zb16));
        asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(o) : "v"(t & 0x000F000Fu), "v"(s16), "v"(zb16));
        // high nibbles: n << 4 times s << 4 -> result in byte 1 of each lane
        asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(e2) : "v"(raw & 0x00F000F0u), "v"(s16h), "v"(zb16h));
        asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(o2) : "v"(t & 0x00F000F0u), "v"(s16h), "v"(zb16h));
        // v_perm_b32 D = bytes of {S0 (4..7), S1 (0..3)}: lo = [e.b0, o.b0, e.b2, o.b2], hi = [e2.b1, o2.b1, e2.b3, o2.b3]
        lo = __builtin_amdgcn_perm(o, e, 0x06020400u);
        hi = __builtin_amdgcn_perm(o2, e2, 0x07030501u);
    }
}

template <int V>
__global__ __launch_bounds__(256) void k_dq(u32* out, int iters, int smax) {
    const u32 t = threadIdx.x + blockIdx.x * 256;
    u32 raw[8];
    for (int i = 0; i < 8; ++i) raw[i] = (t + 1) * 0x9E3779B9u + i * 0x85EBCA6Bu;
    u32 acc = 0;
    for (int it = 0; it < iters; ++it) {
        const u32 s = ((t * 7 + it) % smax) + 1, z = (t * 13 + it * 5) & 0xFF;
        const u32 zb = z * 0x01010101u, s16 = s * 0x00010001u, s16h = s16 << 4, zb16 = z * 0x00010001u, zb16h = zb16 << 8;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                u32 lo, hi;
                dq<V>(raw[i], s, zb, s16, s16h, zb16, zb16h, lo, hi);
                acc += lo ^ (hi * 3);
                raw[i] = raw[i] * 5 + 1;
                asm volatile("" : "+v"(raw[i]));
            }
    }
    out[t] = acc;
}

template <typename F>
static float timeit(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    u32* out;
    hipMalloc(&out, 4 * 256 * 2048);
    const char* names[] = {"v_add_u32", "v_mul_lo_u32", "v_mul_u32_u24", "v_mad_u32_u24", "v_pk_mad_u16", "v_perm_b32",
                           "v_and_or_b32", "v_bfi_b32", "v_pk_mul_lo_u16", "v_lshl_add_u32"};
    const int iters = 4000;
    for (int wps = 1; wps <= 2; ++wps) {
        const int blocks = 256 * wps;
        float base = 0;
        printf("--- %d wave(s) per SIMD (%d workgroups x 256 threads), %d x 64 instructions per wave\n", wps, blocks, iters);
#define RATE(OP)                                                                                      \
    {                                                                                                 \
        float ms = timeit([&] { k_rate<OP><<<blocks, 256>>>(out, iters); });                          \
        if (OP == 0) base = ms;                                                                       \
        printf("%-16s %8.3f ms   %5.2f ns per wave-instruction and SIMD   %4.2fx v_add_u32\n", names[OP], ms, \
               ms * 1e6 / (iters * 64.0 * wps), ms / base);                                            \
    }
        RATE(0) RATE(1) RATE(2) RATE(3) RATE(4) RATE(5) RATE(6) RATE(7) RATE(8) RATE(9)
    }
    // the two dequant sequences: equal results for s <= 17?
    u32 *oa, *ob;
    hipMalloc(&oa, 4 * 256 * 512);
    hipMalloc(&ob, 4 * 256 * 512);
    for (int smax : {17, 255}) {
        k_dq<0><<<512, 256>>>(oa, 64, smax);
        k_dq<1><<<512, 256>>>(ob, 64, smax);
        static u32 ha[256 * 512], hb[256 * 512];
        hipMemcpy(ha, oa, sizeof(ha), hipMemcpyDeviceToHost);
        hipMemcpy(hb, ob, sizeof(hb), hipMemcpyDeviceToHost);
        int diff = 0;
        for (int i = 0; i < 256 * 512; ++i) diff += ha[i] != hb[i];
        printf("dequant A vs B, scales 1..%d: %d of %d lanes differ\n", smax, diff, 256 * 512);
    }
    for (int wps = 1; wps <= 2; ++wps) {
        const int blocks = 256 * wps, it2 = 20000;
        float a = timeit([&] { k_dq<0><<<blocks, 256>>>(out, it2, 17); });
        float b = timeit([&] { k_dq<1><<<blocks, 256>>>(out, it2, 17); });
        printf("%d wave(s) per SIMD: dequant A %.3f ms, B %.3f ms per %d x 32 raw dwords (incl. 3 bookkeeping VALU per dword): A/B = %.2f\n",
               wps, a, b, it2, a / b);
    }
    return 0;
}
