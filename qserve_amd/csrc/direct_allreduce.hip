// direct_allreduce.hip -- latency-oriented fp16 sum all-reduce for the tensor-parallel decode step:
// every rank reads its slice of the payload straight out of the peers' buffers and writes the reduced slice straight
// into theirs, over xGMI's point-to-point links (or, with several ranks on one device, through device memory).
//
// Why: the row-parallel linears (o_proj, down_proj) are followed by a sum over the ranks of a [batch, hidden] fp16
// tensor - 0.5 MiB per 64 sequences at hidden 4096, 2 per layer.  That is far below the size where a ring's bandwidth
// matters; what counts is the number of dependent hops.  Here: two flag exchanges and one round of remote loads /
// remote stores (a "two-shot" reduce-scatter + all-gather in ONE kernel), launched on the caller's stream like any
// other kernel - so the whole tensor-parallel step can be captured in one hipGraph, with no collective library in the
// captured region.  The reference has no counterpart (its tensor parallelism is inert, llama_w4a8_unpad.py:433-434,513-514).
//
// Memory: one UNCACHED device allocation per rank (hipDeviceMallocUncached: neither this GPU's L2s nor a peer's keep
// lines of it, so plain visibility rules apply: a store is visible once it is acknowledged), exported with
// hipIpcGetMemHandle and mapped by every peer:
//     [ flags: sigA[world][MAXG] | sigB[world][MAXG]  (u32 epochs) ][ IN: payload ][ OUT: payload ]
// Protocol of one call, per workgroup b of G (all ranks launch the same G):
//     epoch += 1 (device-side counter: graph replays advance it)
//     A  write epoch into sigA[my rank][b] of EVERY peer ("my IN is complete": it was written by earlier kernels of this
//        stream); wait until every peer's epoch shows up in my sigA[peer][b]
//     R  for my slice of the payload: sum the peers' IN in fp32, in rank order, round once to fp16, store the result into
//        the OUT of every rank
//     B  wait for the stores' acknowledgement (vmcnt 0), write epoch into sigB[my rank][b] of every peer; wait until
//        every peer's epoch shows up in my sigB[peer][b] - the kernel does not retire before every slice of my OUT is there
// Slices are reduced by exactly one rank each, so all ranks end with identical bits.  Every wait is bounded: after
// a few seconds a workgroup gives up, raises the error word (qs_comm_error) and leaves - a crashed peer cannot hang the GPU.
// UNMEASURED on multi-GPU hardware (the development boxes have one GPU): functional tests run two ranks on one device.
#include "common.h"
#include <cstring>

namespace {
constexpr int MAXR = 16;     // ranks
constexpr int MAXG = 64;     // workgroups per call (flag slots per rank)
constexpr size_t FLAG_BYTES = 16384;
static_assert(2 * MAXR * MAXG * sizeof(unsigned) <= FLAG_BYTES, "flag area");

struct PeerTable {
    uint8_t* base[MAXR];
};

struct StateTable {            // per-rank device-side state (one entry used per launch, all of them by the group launch)
    unsigned* epoch[MAXR];
    unsigned* err[MAXR];
};

struct Comm {
    int rank, world, device, grid;
    size_t payload;            // bytes of IN (= bytes of OUT)
    uint8_t* base;             // this rank's region
    PeerTable peers;           // every rank's region as mapped HERE (peers.base[rank] == base)
    bool ipc_opened[MAXR];
    bool connected;
    unsigned* epoch;           // device, [MAXG]
    unsigned* err;             // device, 1 word
};

__device__ __forceinline__ void st_sys(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned ld_sys(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ v4u ld16_sys(const uint8_t* p) {
    v4u v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st16_sys(uint8_t* p, v4u v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// bounded wait for `*p` to reach epoch `ep` (wrap-safe); false on timeout
__device__ __forceinline__ bool wait_epoch(const unsigned* p, unsigned ep) {
    for (int it = 0; it < (1 << 22); ++it) {
        if ((int)(ld_sys(p) - ep) >= 0) return true;
        __builtin_amdgcn_s_sleep(16);
    }
    return false;
}

// rank_arg >= 0: this launch is rank `rank_arg`; rank_arg < 0: the launch carries ALL ranks of a same-process group, rank =
// blockIdx.y (tests: one dispatch, so that the ranks are certainly resident together)
__global__ __launch_bounds__(256) void direct_allreduce_f16_kernel(PeerTable t, StateTable st, int rank_arg, int world,
                                                                    size_t payload, long chunks_per_rank) {
    const int b = blockIdx.x, G = gridDim.x, tid = threadIdx.x;
    const int rank = rank_arg >= 0 ? rank_arg : (int)blockIdx.y;
    unsigned* const epoch = st.epoch[rank];
    unsigned* const err = st.err[rank];
    __shared__ unsigned s_ep;
    __shared__ int s_ok;
    if (tid == 0) {
        s_ep = epoch[b] + 1;
        epoch[b] = s_ep;
        s_ok = 1;
    }
    __syncthreads();
    const unsigned ep = s_ep;
    unsigned* const my_flags = reinterpret_cast<unsigned*>(t.base[rank]);
    // ---- A: inputs complete everywhere -------------------------------------------------------------------------
    if (tid < world) {
        st_sys(reinterpret_cast<unsigned*>(t.base[tid]) + rank * MAXG + b, ep);
        if (!wait_epoch(my_flags + tid * MAXG + b, ep)) s_ok = 0;
    }
    __syncthreads();
    if (!s_ok) {
        if (tid == 0) atomicExch(err, 1u);
        return;
    }
    // ---- R: reduce this rank's slice, publish it to every rank -------------------------------------------------------
    const size_t in_off = FLAG_BYTES, out_off = FLAG_BYTES + payload;
    for (long c = (long)b * 256 + tid; c < chunks_per_rank; c += (long)G * 256) {
        const size_t off = ((size_t)rank * chunks_per_rank + c) * 16;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        v4u v[MAXR];
#pragma unroll
        for (int r = 0; r < MAXR; ++r)
            if (r < world) v[r] = ld16_sys(t.base[r] + in_off + off);                      // all requests first
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < MAXR; ++r)                                                     // rank order: same bits on every rank
            if (r < world) {
                asm volatile("" : "+v"(v[r]));
                const h8 x = __builtin_bit_cast(h8, v[r]);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += (float)x[j];
            }
        h8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (_Float16)acc[j];
        const v4u ov = __builtin_bit_cast(v4u, o);
#pragma unroll
        for (int r = 0; r < MAXR; ++r)
            if (r < world) st16_sys(t.base[r] + out_off + off, ov);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every remote store acknowledged
    __syncthreads();
    // ---- B: outputs complete everywhere ------------------------------------------------------------------------
    if (tid < world) {
        st_sys(reinterpret_cast<unsigned*>(t.base[tid]) + (MAXR + rank) * MAXG + b, ep);
        if (!wait_epoch(my_flags + (MAXR + tid) * MAXG + b, ep)) s_ok = 0;
    }
    __syncthreads();
    if (!s_ok && tid == 0) atomicExch(err, 1u);
}

bool hip_ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    qs_set_error("direct all-reduce: %s: %s", what, hipGetErrorString(e));
    (void)hipGetLastError();
    return false;
}
}  // namespace

extern "C" int qs_comm_create(int rank, int world, int64_t payload_bytes, void** comm_out, void* ipc_handle64) {
    QS_REQUIRE(comm_out && ipc_handle64, "comm_create: null pointer");
    QS_REQUIRE(world >= 1 && world <= MAXR && rank >= 0 && rank < world, "comm_create: rank %d of %d (at most %d ranks)", rank,
               world, MAXR);
    QS_REQUIRE(payload_bytes > 0 && payload_bytes % (16 * (int64_t)world) == 0,
               "comm_create: payload of %lld bytes is not a multiple of 16 x world", (long long)payload_bytes);
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    Comm* c = new Comm();
    c->rank = rank, c->world = world, c->payload = (size_t)payload_bytes, c->connected = false, c->grid = 32;
    for (int r = 0; r < MAXR; ++r) c->peers.base[r] = nullptr, c->ipc_opened[r] = false;
    const size_t total = FLAG_BYTES + 2 * c->payload;
    void* p = nullptr;
    if (!hip_ok(hipGetDevice(&c->device), "hipGetDevice") ||
        !hip_ok(hipExtMallocWithFlags(&p, total, hipDeviceMallocUncached), "hipExtMallocWithFlags(uncached)") ||
        !hip_ok(hipMemset(p, 0, total), "hipMemset") || !hip_ok(hipMalloc(&c->epoch, (MAXG + 1) * sizeof(unsigned)), "hipMalloc") ||
        !hip_ok(hipMemset(c->epoch, 0, (MAXG + 1) * sizeof(unsigned)), "hipMemset") ||
        !hip_ok(hipDeviceSynchronize(), "hipDeviceSynchronize")) {
        if (p) (void)hipFree(p);
        delete c;
        return QS_EINVAL;
    }
    c->base = static_cast<uint8_t*>(p);
    c->err = c->epoch + MAXG;
    c->peers.base[rank] = c->base;
    hipIpcMemHandle_t h;
    if (!hip_ok(hipIpcGetMemHandle(&h, p), "hipIpcGetMemHandle")) {
        (void)hipFree(p);
        (void)hipFree(c->epoch);
        delete c;
        return QS_EINVAL;
    }
    std::memcpy(ipc_handle64, &h, 64);
    *comm_out = c;
    return QS_OK;
}

// handles: world x 64 bytes in rank order (this rank's own entry is ignored).  Peers on other devices get peer access.
extern "C" int qs_comm_connect(void* comm, const void* handles) {
    Comm* c = static_cast<Comm*>(comm);
    QS_REQUIRE(c && handles, "comm_connect: null pointer");
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank || c->peers.base[r]) continue;
        hipIpcMemHandle_t h;
        std::memcpy(&h, static_cast<const uint8_t*>(handles) + 64 * r, 64);
        void* p = nullptr;
        if (!hip_ok(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle")) return QS_EINVAL;
        c->peers.base[r] = static_cast<uint8_t*>(p);
        c->ipc_opened[r] = true;
    }
    c->connected = true;
    return QS_OK;
}

// same-process ranks (tests; several communicators of one process): the peers' regions by address, no IPC
extern "C" int qs_comm_connect_local(void* comm, void* const* peer_comms) {
    Comm* c = static_cast<Comm*>(comm);
    QS_REQUIRE(c && peer_comms, "comm_connect_local: null pointer");
    for (int r = 0; r < c->world; ++r) {
        const Comm* o = static_cast<const Comm*>(peer_comms[r]);
        QS_REQUIRE(o && o->rank == r && o->world == c->world && o->payload == c->payload,
                   "comm_connect_local: communicator %d does not match", r);
        c->peers.base[r] = o->base;
    }
    c->connected = true;
    return QS_OK;
}

extern "C" void* qs_comm_input(void* comm) { return comm ? static_cast<Comm*>(comm)->base + FLAG_BYTES : nullptr; }
extern "C" void* qs_comm_output(void* comm) {
    Comm* c = static_cast<Comm*>(comm);
    return c ? c->base + FLAG_BYTES + c->payload : nullptr;
}

// In: this rank's addend in qs_comm_input (numel fp16, written by earlier work on `stream`).  Out: the sum over all ranks
// in qs_comm_output once the kernel has retired.  numel * 2 <= payload, numel % (8 * world) == 0.
extern "C" int qs_comm_all_reduce_f16(void* comm, int64_t numel, qs_stream_t stream) {
    Comm* c = static_cast<Comm*>(comm);
    QS_REQUIRE(c && c->connected, "comm_all_reduce: communicator not connected");
    QS_REQUIRE(numel > 0 && numel % (8 * (int64_t)c->world) == 0 && (size_t)numel * 2 <= c->payload,
               "comm_all_reduce: numel=%lld (must be a positive multiple of 8 x world = %d, at most %zu)", (long long)numel,
               8 * c->world, c->payload / 2);
    const long chunks = (long)(numel / 8 / c->world);
    StateTable st = {};
    st.epoch[c->rank] = c->epoch;
    st.err[c->rank] = c->err;
    hipLaunchKernelGGL(direct_allreduce_f16_kernel, dim3(c->grid), dim3(256), 0, (hipStream_t)stream, c->peers, st, c->rank,
                       c->world, c->payload, chunks);
    return qs_launch_status("direct all-reduce");
}

// Same-process group (qs_comm_connect_local): the calls of ALL ranks as one dispatch.  For tests - separate launches of
// one process are only resident together if they happen to sit on different hardware queues.
extern "C" int qs_comm_all_reduce_f16_group(void* const* comms, int world, int64_t numel, qs_stream_t stream) {
    QS_REQUIRE(comms && world >= 1 && world <= MAXR, "comm_all_reduce_group: bad arguments");
    const Comm* c0 = static_cast<const Comm*>(comms[0]);
    QS_REQUIRE(c0 && c0->connected && c0->world == world, "comm_all_reduce_group: communicator 0 does not match");
    QS_REQUIRE(numel > 0 && numel % (8 * (int64_t)world) == 0 && (size_t)numel * 2 <= c0->payload,
               "comm_all_reduce_group: numel=%lld", (long long)numel);
    StateTable st = {};
    for (int r = 0; r < world; ++r) {
        const Comm* c = static_cast<const Comm*>(comms[r]);
        QS_REQUIRE(c && c->connected && c->rank == r && c->world == world && c->payload == c0->payload &&
                       c->peers.base[0] == c0->peers.base[0],
                   "comm_all_reduce_group: communicator %d does not belong to the group", r);
        st.epoch[r] = c->epoch;
        st.err[r] = c->err;
    }
    hipLaunchKernelGGL(direct_allreduce_f16_kernel, dim3(c0->grid, world), dim3(256), 0, (hipStream_t)stream, c0->peers, st, -1,
                       world, c0->payload, (long)(numel / 8 / world));
    return qs_launch_status("direct all-reduce (group)");
}

// synchronises the device; 1 if any call timed out waiting for a peer since the last query
extern "C" int qs_comm_error(void* comm) {
    Comm* c = static_cast<Comm*>(comm);
    QS_REQUIRE(c, "comm_error: null pointer");
    unsigned e = 0;
    if (!hip_ok(hipMemcpy(&e, c->err, sizeof(e), hipMemcpyDeviceToHost), "hipMemcpy")) return -1;
    if (e) (void)hipMemset(c->err, 0, sizeof(unsigned));
    return (int)e;
}

extern "C" int qs_comm_destroy(void* comm) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c) return QS_OK;
    (void)hipDeviceSynchronize();
    for (int r = 0; r < c->world; ++r)
        if (c->ipc_opened[r]) (void)hipIpcCloseMemHandle(c->peers.base[r]);
    (void)hipFree(c->base);
    (void)hipFree(c->epoch);
    delete c;
    return QS_OK;
}
