// Round 5: what does a ROW phase cost inside a launch whose other workgroups prefetch weights - and does THINNING the loaders help?
// (the round-3 "row kernel as the head of the GEMM" experiment lost 17.3 vs 12.9 us; its row work took 6.5 us inside
// the launch against 3 us alone.  Was that the row workgroups' requests queueing behind the OTHER workgroups' weight bursts in the
// fabric - then thinning a CU's own loader cannot help - or could a throttled prefetch have kept the row phase short?)
//
// One launch, 256 workgroups x 512 threads (the ring GEMM's shape), 64 KiB of LDS ring each:
//   workgroups 0 .. 63   ROW work of one token row (add + norm + quant sized: 3 x 8 KB in, three block reductions, 4 KB int8 out
//                        write-through + 8 KB fp16 out, drain, arrive on a counter); s_memtime from entry to arrival
//   workgroups 64 .. 255 LOADERS: stream STAGES x 16 KiB of "weights" by LDS-DMA (global_load_lds_dwordx4 nt, as the ring GEMM does);
//                        at most PRE stages may be requested before the counter shows 64 (in the real GEMM nothing can be
//                        CONSUMED before the activations exist), the rest only after it:
//       mode 0  burst: the PRE stages at once at kernel entry (the round-3 head experiment)
//       mode 1  thinned: one 16 KiB fill outstanding at a time until the counter is seen
//       mode 2  none: no request before the counter is seen (what a separate launch does, without its boundary)
//       mode 3  no rows at all: the loaders do not wait (the stream alone)
//   out: per workgroup {cycles entry -> arrival / counter seen, cycles entry -> done}
// Host: every mode as ONE launch, and the two-launch baseline (row kernel with 64 workgroups, then mode 3 with 192), hipGraph-timed.
// build: hipcc -O3 --offload-arch=gfx950 scripts/microbench_rowhead.hip -o /tmp/mb_rowhead ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef unsigned int u32;
typedef u32 v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int HID = 4096, ROWS = 64, STAGES = 16, PRE = 4, SLOTS = 4;

__device__ __forceinline__ void dma16_nt(const void* src, u32 lds) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(src), "s"(lds) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(512, 1) void fused(const uint8_t* __restrict__ W, const __half* __restrict__ x, const __half* __restrict__ res,
                                                const __half* __restrict__ gamma, int8_t* __restrict__ q, __half* __restrict__ hout,
                                                unsigned* __restrict__ counter, unsigned gen, unsigned long long* __restrict__ out,
                                                size_t wstride, int nrows, const uint8_t* __restrict__ act) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ float red[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long t1 = 0;
    if (b < nrows) {
        // ---- row work -------------------------------------------------------------------------------------------------
        const size_t base = (size_t)b * HID + tid * 8;
        const v4u a = *reinterpret_cast<const v4u*>(x + base);
        const v4u r = *reinterpret_cast<const v4u*>(res + base);
        const v4u g = *reinterpret_cast<const v4u*>(gamma + tid * 8);
        float v[8], s = 0.f;
        const __half* ah = reinterpret_cast<const __half*>(&a);
        const __half* rh = reinterpret_cast<const __half*>(&r);
        const __half* gh = reinterpret_cast<const __half*>(&g);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v[i] = __half2float(ah[i]) + __half2float(rh[i]);
            s += v[i];
        }
        auto block_sum = [&](float t) {
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) t += __shfl_xor(t, o, 64);
            __syncthreads();
            if (lane == 0) red[wave] = t;
            __syncthreads();
            float z = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) z += red[w];
            return z;
        };
        const float mean = block_sum(s) / HID;
        float vs = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) vs += (v[i] - mean) * (v[i] - mean);
        const float rstd = rsqrtf(block_sum(vs) / HID + 1e-5f);
        float mx = 1e-6f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v[i] = (v[i] - mean) * rstd * __half2float(gh[i]);
            mx = fmaxf(mx, fabsf(v[i]));
        }
        const float amax = -block_sum(-mx) * 0.f + mx;     // (a third reduction round; the value itself is irrelevant here)
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            lo |= ((unsigned)(int)rintf(v[i] * (127.f / amax)) & 0xFFu) << (8 * i);
            hi |= ((unsigned)(int)rintf(v[4 + i] * (127.f / amax)) & 0xFFu) << (8 * i);
        }
        // the int8 row write-through (the consumers read it from other CUs), the fp16 row plain
        uint2* qd = reinterpret_cast<uint2*>(q + (size_t)b * HID + tid * 8);
        const uint2 qv = make_uint2(lo, hi);
        asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(qd), "v"(qv) : "memory");
        *reinterpret_cast<v4u*>(hout + base) = a;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t1 = __builtin_amdgcn_s_memtime();
    } else {
        // ---- loader ----------------------------------------------------------------------------------------------------
        const u32 lds0 = (u32)(size_t)(lptr_t)smem;
        const uint8_t* src = W + (size_t)(b - nrows) * wstride + (size_t)wave * 1024 + lane * 16;
        auto issue = [&](int st) {                                   // 16 KiB = 2 x 8 waves x 1 KiB
            dma16_nt(src + (size_t)st * 16384, lds0 + (st % SLOTS) * 16384 + wave * 1024);
            dma16_nt(src + (size_t)st * 16384 + 8192, lds0 + (st % SLOTS) * 16384 + 8192 + wave * 1024);
        };
        // ONE poller per workgroup (wave 0, lane 0, relaxed agent-scope load + s_sleep; 1 536 pollers on one word were measured
        // first: the arrivals themselves then queue behind the polls and the counter is seen 5 us late); the other waves wait at a
        // barrier.  Mode 1: every wave issues its share of a fill, the workgroup drains it, wave 0 looks at the counter, repeat.
        __shared__ int s_seen;
        auto seen_wg = [&]() {                                       // all waves call; true once the counter shows this launch's rows
            if (tid == 0) s_seen = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gen;
            __syncthreads();
            const bool r = s_seen != 0;
            __syncthreads();
            return r;
        };
        int st = 0;
        if (MODE == 0) {
            for (; st < PRE; ++st) issue(st);
            for (int i = 0; i < (1 << 18) && !seen_wg(); ++i) __builtin_amdgcn_s_sleep(8);
        } else if (MODE == 1) {
            bool ok = false;
            for (int i = 0; i < (1 << 18) && !ok; ++i) {
                if (st < PRE) {
                    issue(st++);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else {
                    __builtin_amdgcn_s_sleep(8);
                }
                ok = seen_wg();
            }
        } else if (MODE == 2) {
            for (int i = 0; i < (1 << 18) && !seen_wg(); ++i) __builtin_amdgcn_s_sleep(8);
        }
        t1 = __builtin_amdgcn_s_memtime();
        // the ACTIVATIONS of the first ring fill (what the rows produced: 64 KiB of an L2-resident 256 KiB matrix shared by all
        // workgroups, cache-bypassing sc0 sc1): the real GEMM cannot consume a weight stage before they are there.  In the
        // two-launch form (MODE 3) they are requested with the first weight stages and land long before them.
        {
            const uint8_t* ap = act + (size_t)wave * 8192 + lane * 16;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off sc0 sc1" ::"v"(ap + i * 1024),
                             "s"(lds0 + SLOTS * 16384 + wave * 8192 + i * 1024) : "memory");
            if (MODE == 3) {                                   // separate launch: weights requested together with them
                for (; st < PRE; ++st) issue(st);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (first consumption needs stage 0 AND its activations)
            __syncthreads();
        }
        // the stream proper: up to SLOTS - 1 stages beyond the oldest in flight (a consumed slot is refilled at once)
        for (; st < STAGES; ++st) {
            issue(st);
            if (st >= SLOTS - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (SLOTS - 2)) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (tid == 0) {
        out[2 * b] = t1 - t0;
        out[2 * b + 1] = t2 - t0;
    }
}

int main() {
    const size_t wstride = (size_t)STAGES * 16384;                 // 256 KiB per loader
    const int NSET = 24;                                           // weight sets rotated per launch (nothing cache-resident)
    uint8_t* W;
    __half *x, *res, *gamma, *hout;
    int8_t* q;
    unsigned* counter;
    unsigned long long* out;
    hipMalloc(&W, wstride * 256 * NSET);
    hipMemset(W, 0x5a, wstride * 256 * NSET);
    hipMalloc(&x, ROWS * HID * 2);
    hipMalloc(&res, ROWS * HID * 2);
    hipMalloc(&gamma, HID * 2);
    hipMalloc(&hout, ROWS * HID * 2);
    hipMalloc(&q, ROWS * HID);
    hipMemset(x, 0x11, ROWS * HID * 2);
    hipMemset(res, 0x12, ROWS * HID * 2);
    hipMemset(gamma, 0x3c, HID * 2);
    hipMalloc(&counter, 4);
    hipMemset(counter, 0, 4);
    hipMalloc(&out, 256 * 2 * 8);
    uint8_t* act;
    hipMalloc(&act, 262144);
    hipMemset(act, 0x21, 262144);
    hipStream_t st;
    hipStreamCreate(&st);
    const int smem = SLOTS * 16384 + 65536;
    hipFuncSetAttribute(reinterpret_cast<const void*>(fused<0>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipFuncSetAttribute(reinterpret_cast<const void*>(fused<1>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipFuncSetAttribute(reinterpret_cast<const void*>(fused<2>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipFuncSetAttribute(reinterpret_cast<const void*>(fused<3>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    unsigned gen = 0;
    std::vector<unsigned long long> h(512);
    auto stats = [&](const char* what, int nrows, float us) {
        hipMemcpy(h.data(), out, 512 * 8, hipMemcpyDeviceToHost);
        std::vector<double> rowt, seent, endt;
        for (int b = 0; b < 256; ++b) {
            if (b < nrows) rowt.push_back((double)h[2 * b]);
            else seent.push_back((double)h[2 * b]);
            endt.push_back((double)h[2 * b + 1]);
        }
        auto med = [](std::vector<double> v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        auto mx = [](const std::vector<double>& v) { double m = 0; for (double x : v) m = std::max(m, x); return m; };
        printf("%-44s %7.2f us per launch | row work: median %6.0f max %6.0f cyc | loaders see the rows at: median %6.0f cyc | done: max %6.0f cyc (s_memtime ticks)\n",
               what, us, med(rowt), mx(rowt), med(seent), mx(endt));
    };
    for (int pass = 0; pass < 2; ++pass) {
        for (int mode = 0; mode < 5; ++mode) {
            const int reps = 24;
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            float ms = 0;
            hipDeviceSynchronize();
            hipEventRecord(e0, st);
            for (int r = 0; r < reps; ++r) {
                const uint8_t* Wp = W + (size_t)(r % NSET) * wstride * 256;
                if (mode < 3) {
                    gen += ROWS;
                    if (mode == 0) fused<0><<<256, 512, smem, st>>>(Wp, x, res, gamma, q, hout, counter, gen, out, wstride, ROWS, act);
                    if (mode == 1) fused<1><<<256, 512, smem, st>>>(Wp, x, res, gamma, q, hout, counter, gen, out, wstride, ROWS, act);
                    if (mode == 2) fused<2><<<256, 512, smem, st>>>(Wp, x, res, gamma, q, hout, counter, gen, out, wstride, ROWS, act);
                } else if (mode == 3) {                              // two launches: rows (64 workgroups), then the stream (192)
                    gen += ROWS;
                    fused<3><<<ROWS, 512, smem, st>>>(Wp, x, res, gamma, q, hout, counter, gen, out, wstride, ROWS, act);
                    fused<3><<<192, 512, smem, st>>>(Wp, x, res, gamma, q, hout, counter, gen, out + 128, wstride, 0, act);
                } else {                                             // the stream alone
                    fused<3><<<192, 512, smem, st>>>(Wp, x, res, gamma, q, hout, counter, gen, out + 128, wstride, 0, act);
                }
            }
            hipEventRecord(e1, st);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            const char* names[5] = {"one launch, loaders BURST 4 stages first", "one launch, loaders THINNED (1 fill in flight)",
                                    "one launch, loaders wait (no prefetch)", "two launches: rows | stream", "the stream alone (192 workgroups)"};
            if (pass == 1) stats(names[mode], mode < 4 ? ROWS : 0, ms * 1e3f / reps);
        }
    }
    return 0;
}
