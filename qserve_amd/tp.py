"""Tensor-parallel sharding of the packed W4A8 weights + the one collective the path needs.

The reference is single-GPU (tp_size = 1 everywhere, llama_w4a8_unpad.py:513-514); this is the Megatron-style
extension: column-parallel qkv_proj / gate_up_proj (split N), row-parallel o_proj / down_proj (split K) followed by
ONE fp16 sum all-reduce (RCCL over xGMI: torch.distributed backend "nccl" on ROCm).

Sharding rules for the reference's packed layout (w4a8_linear.py:196-277):
  * qweight bytes are [N/32][K/32][512]: a split of N at a multiple of 32 rows is a plain row slice of the
    [N, K/2] view; a split of K must slice the 4-D tile view on its K/32 axis (a column slice of the 2-D view would
    be WRONG) and, for g128 weights, at multiples of 4 tiles so that groups stay intact;
  * s1_scales / s1_szeros [N]: slice for column-parallel, replicate for row-parallel;
  * s2_scales / s2_zeros [K/128, N] (per-32 permuted along N): slice dim 1 at multiples of 32 (column-parallel),
    slice dim 0 (row-parallel).
Works on torch tensors on any device (pure indexing, no kernels).
"""
import torch


def shard_column_parallel(qweight, vecs_n=(), mats_gn=(), rank=0, world=1):
    """Split the OUTPUT dim N of ONE separately packed projection (q_proj, gate_proj, ...).  qweight int8 [N, K/2];
    vecs_n: per-channel [N] tensors (s1_scales, s1_szeros, bias); mats_gn: [K/128, N] tensors (s2_scales, s2_zeros).
    Returns (qweight_r, [vecs...], [mats...]).

    NOT for an already fused `qkv_proj` / `gate_up_proj` tensor: a contiguous slice of [q; k; v] would hand rank 0 only
    query heads.  Fused tensors go through `shard_fused_column_parallel` (or are sharded per projection before the
    fusion, `qserve_amd.loader.fuse_column_parallel`)."""
    N = qweight.shape[0]
    assert N % (32 * world) == 0, "column-parallel split must fall on 32-row tiles"
    n0, n1 = rank * N // world, (rank + 1) * N // world
    return (qweight[n0:n1].contiguous(), [v[n0:n1].contiguous() for v in vecs_n],
            [m[:, n0:n1].contiguous() for m in mats_gn])


def shard_fused_column_parallel(qweight, sizes, vecs_n=(), mats_gn=(), rank=0, world=1, replicas=None):
    """Column-parallel shard of a FUSED projection whose rows are the concatenation of blocks of `sizes` rows
    ([q; k; v] -> sizes = [H*128, Hkv*128, Hkv*128]; [gate; up] -> [inter, inter]): every block is split over the ranks
    separately (block i over world // replicas[i] shards; replicas > 1 = KV heads shared by several ranks) and the
    rank's pieces are concatenated again, so each rank gets its own q heads AND its kv heads."""
    assert sum(sizes) == qweight.shape[0]
    replicas = replicas or [1] * len(sizes)
    idx, off = [], 0
    for n, rep in zip(sizes, replicas):
        shards = world // rep
        assert n % (32 * shards) == 0, "column-parallel split must fall on 32-row tiles"
        per, sid = n // shards, rank // rep
        idx.append((off + sid * per, off + (sid + 1) * per))
        off += n
    cat = lambda t, dim: torch.cat([t[a:b] if dim == 0 else t[:, a:b] for a, b in idx], dim=dim).contiguous()  # noqa: E731
    return cat(qweight, 0), [cat(v, 0) for v in vecs_n], [cat(m, 1) for m in mats_gn]


def shard_row_parallel(qweight, mats_gn=(), rank=0, world=1, group_size=-1):
    """Split the INPUT dim K.  Slices the tile view [N/32, K/32, 512] on axis 1.  Per-channel vectors are
    replicated (not returned).  mats_gn ([K/128, N]) are sliced on dim 0."""
    N, K2 = qweight.shape
    K = K2 * 2
    kt = K // 32
    unit = 4 if group_size == 128 else 1          # keep 128-wide groups intact
    assert kt % (world * unit) == 0, "row-parallel split must fall on whole 32-k tiles (128-k groups for g128)"
    t0, t1 = rank * kt // world, (rank + 1) * kt // world
    tiles = qweight.reshape(N // 32, kt, 512)[:, t0:t1].contiguous()
    qw = tiles.reshape(N, (t1 - t0) * 16)
    mats = []
    for m in mats_gn:
        g = m.shape[0]
        mats.append(m[rank * g // world:(rank + 1) * g // world].contiguous())
    return qw, mats


def all_reduce_sum_(t, group=None):
    """In-place fp16 sum all-reduce of the row-parallel partial outputs (2 per layer).  No-op without a process
    group or at world size 1."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


class _DevicePtr:
    """A raw device allocation as a __cuda_array_interface__ object (torch.as_tensor wraps it without copying)."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 3,
                                         "strides": None}
        self._owner = owner


class DirectAllReduce:
    """The library's direct-access all-reduce (include/qserve_amd.h `qs_comm_*`, csrc/direct_allreduce.hip) for one rank.

    Usage: the row-parallel GEMM writes its partial output into `input((B, hidden))`, `all_reduce(numel)` launches the
    kernel on the current stream, the sum over the ranks is in `output((B, hidden))`.  Opt-in (DecodeEngine
    `direct_allreduce=`, bench.py `--direct-allreduce`): not measured on multi-GPU hardware; `all_reduce_sum_` over
    torch.distributed stays the default.  `numel` must be a multiple of 8 * world."""

    def __init__(self, max_numel, group=None, device=None, _local=None):
        import ctypes as C

        import torch
        import torch.distributed as dist

        from ._lib import check, lib
        self._lib, self._check = lib, check
        if _local is None:
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        else:
            self.rank, self.world = _local
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        unit = 8 * self.world
        self.max_numel = (int(max_numel) + unit - 1) // unit * unit
        comm, handle = C.c_void_p(), C.create_string_buffer(64)
        err = None
        with torch.cuda.device(self.device):
            try:
                # warm-up: a one-rank communicator runs the kernel once, so that this process has loaded the code object
                # before the first real call - whose waits for the peers are bounded to a few seconds
                w, wh = C.c_void_p(), C.create_string_buffer(64)
                check(lib.qs_comm_create(0, 1, 16, C.byref(w), wh), "comm_create (warm-up)")
                check(lib.qs_comm_connect(w, bytes(wh.raw)), "comm_connect (warm-up)")
                check(lib.qs_comm_all_reduce_f16(w, 8, None), "comm_all_reduce (warm-up)")
                lib.qs_comm_error(w)
                lib.qs_comm_destroy(w)
                check(lib.qs_comm_create(self.rank, self.world, self.max_numel * 2, C.byref(comm), handle), "comm_create")
            except RuntimeError as e:              # keep the collective calls below matched on every rank
                if _local is not None:
                    raise
                err = str(e)
        self._comm = comm if err is None else None
        self.handle = bytes(handle.raw)
        if _local is None:
            handles = [None] * self.world
            dist.all_gather_object(handles, None if err else self.handle, group=group)
            if err is None and all(h is not None for h in handles):
                with torch.cuda.device(self.device):
                    try:
                        check(lib.qs_comm_connect(comm, b"".join(handles)), "comm_connect")
                    except RuntimeError as e:
                        err = str(e)
            elif err is None:
                err = "another rank could not create its communicator"
            oks = [None] * self.world
            dist.all_gather_object(oks, err, group=group)
            if any(o is not None for o in oks):
                self.close()
                raise RuntimeError("direct all-reduce unavailable: " + "; ".join(f"rank {r}: {o}" for r, o in enumerate(oks) if o))

    @classmethod
    def local_group(cls, world, max_numel, device=None):
        """`world` communicators of ONE process, connected by address (tests: ranks driven on separate streams)."""
        import ctypes as C
        comms = [cls(max_numel, device=device, _local=(r, world)) for r in range(world)]
        arr = (C.c_void_p * world)(*[c._comm for c in comms])
        for c in comms:
            c._check(c._lib.qs_comm_connect_local(c._comm, arr), "comm_connect_local")
        return comms

    def _view(self, ptr, shape):
        import torch
        n = 1
        for d in shape:
            n *= d
        assert n <= self.max_numel
        return torch.as_tensor(_DevicePtr(ptr, shape, "<f2", self), device=self.device)

    def input(self, shape):
        return self._view(self._lib.qs_comm_input(self._comm), shape)

    def output(self, shape):
        return self._view(self._lib.qs_comm_output(self._comm), shape)

    def all_reduce(self, numel):
        from .backend._util import stream
        self._check(self._lib.qs_comm_all_reduce_f16(self._comm, int(numel), stream()), "comm_all_reduce")

    @staticmethod
    def all_reduce_group(comms, numel):
        """All ranks of a `local_group` in ONE dispatch on the current stream (tests)."""
        import ctypes as C

        from .backend._util import stream
        arr = (C.c_void_p * len(comms))(*[c._comm for c in comms])
        comms[0]._check(comms[0]._lib.qs_comm_all_reduce_f16_group(arr, len(comms), int(numel), stream()), "comm_all_reduce_group")

    def error(self):
        """Synchronises; True if a call gave up waiting for a peer since the last query."""
        return bool(self._lib.qs_comm_error(self._comm))

    def close(self):
        if self._comm:
            self._lib.qs_comm_destroy(self._comm)
            self._comm = None


class RowParallelLinear:
    """y = all_reduce( gemm(x_shard) ) (+ bias once, after the reduce).

    `gemm(x_q, ascales, a_ssums, out)` is a closure over this rank's weight shard, e.g.
        lambda x, sa, ss, out: qgemm_w4a8_per_chn.gemm_forward_cuda(x, qweight_r, s1_scales, sa, s1_szeros, ss, out)
    (the signature `W4A8OF16LinearDynamicInputScale.forward` has, w4a8_linear.py:105-134).  Every rank quantises its
    own activation slice, so ascales / a_ssums are per shard (the zero-point term is linear in K, partial corrections
    add up: SURVEY 8e)."""

    def __init__(self, gemm, bias=None, group=None):
        self.gemm, self.bias, self.group = gemm, bias, group

    def __call__(self, x_q, ascales, a_ssums, out):
        self.gemm(x_q, ascales, a_ssums, out)
        all_reduce_sum_(out, self.group)
        if self.bias is not None:
            out += self.bias
        return out
