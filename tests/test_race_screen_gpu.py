"""Race screen for the hand-scheduled GEMM pipelines (counted vmcnt waits + raw barriers: a misplaced wait shows up as a
rare wrong tile that comes and goes with timing, not as a failing unit test).  Every shape runs REPS times on the same
inputs while a second stream thrashes HBM / L2 in bursts (so that DMA landing times move around); the first result is
checked against an exact reference built on the device (integer matmul + the epilogue in separate fp32 torch ops), every
later one must be bit-identical to the first.  QS_RACE_REPS=200 turns it into a longer screen (scripts/README.md)."""
import os

import pytest
import torch

from _helpers import int_matmul_torch, per_group_problem_torch, unpack_qweight_torch

pytestmark = pytest.mark.gpu
REPS = int(os.environ.get("QS_RACE_REPS", "12"))

# (M, N, K): prompt shapes (tiled kernel, several tiles per workgroup, ragged last tile), config 1, decode shapes (ring
# kernel: two / four units, K slices), a mid-size batch
SHAPES = [(8192, 6144, 4096), (4096, 4096, 4096), (5000, 28672, 4096), (8192, 4096, 14336), (64, 6144, 4096),
          (64, 4096, 4096), (64, 28672, 4096), (64, 4096, 14336), (128, 28672, 4096), (300, 4096, 4096)]


def thrash(stream, bufs, n):
    with torch.cuda.stream(stream):
        for i in range(n):
            bufs[(i + 1) % len(bufs)].copy_(bufs[i % len(bufs)])


# (shape, mode, forced variant): every shape under the dispatcher's own choice (-1: per-group 256-token tiles already take the
# four-wave tile, gemm_w4a8_wide.hip; g128 gate_up at 128 tokens the 128-token ring workgroups) + the four-wave tile FORCED (3003) on
# the per-channel prompt shapes, where the dispatcher prefers the eight-wave tile
CASES = [(M, N, K, mode, -1) for mode in ("per_channel", "per_group") for (M, N, K) in SHAPES] + \
        [(M, N, K, "per_channel", 3003) for (M, N, K) in SHAPES if M >= 1024]


@pytest.fixture
def gemm_variant(request):
    from qserve_amd import _lib
    _lib.lib.qs_set_gemm_variant(request.param)
    yield request.param
    _lib.lib.qs_set_gemm_variant(-1)


@pytest.mark.parametrize("M,N,K,mode,gemm_variant", CASES, indirect=["gemm_variant"],
                         ids=[f"{m}-{M}x{N}x{K}-{'dispatcher' if v < 0 else 'four_wave_tile'}" for (M, N, K, m, v) in CASES])
def test_repeated_runs_are_identical_and_exact(gpu, M, N, K, mode, gemm_variant):
    import qserve_backend.qgemm_w4a8_per_chn as opc
    import qserve_backend.qgemm_w4a8_per_group as opg
    from qserve_amd import fused as fz
    g = torch.Generator(device=gpu).manual_seed(M + N + K)
    sa = (torch.rand((M,), device=gpu, generator=g) * 0.02 + 0.005).half()
    if mode == "per_channel":
        W = torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=gpu, generator=g)
        A = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=gpu, generator=g)
        ws = (torch.rand((N,), device=gpu, generator=g) * 0.004 + 0.001).half()
        wz = (torch.randint(0, 16, (N,), device=gpu, generator=g).half() * ws).half()
        ss = (sa.float() * A.float().sum(1)).half()
        acc = int_matmul_torch(A, unpack_qweight_torch(W)).float()
        t = acc * ws.float()[None, :]                 # epi_per_chn: ((acc * ws) * sa) - (wz * ss), no contraction
        t = t * sa.float()[:, None]
        want = (t - wz.float()[None, :] * ss.float()[:, None]).half()
        run = lambda o: opc.gemm_forward_cuda(A, W, ws, sa, wz, ss, o)
        run_act = lambda o: fz.gemm_silu_and_mul_per_chn(A, W, ws, sa, wz, ss, o, None)
    else:
        pr = per_group_problem_torch(M, N, K, gpu, seed=M + N + K)
        A, W, ws = pr["A"], pr["qweight"], pr["wscales"]
        sa = pr["ascales"]
        acc = int_matmul_torch(A, pr["w8"]).float()
        want = (acc * (ws.float()[None, :] * sa.float()[:, None])).half()     # epi_per_group: acc * (ws * sa)
        run = lambda o: opg.gemm_forward_cuda(A, W, pr["s2_zeros"], pr["s2_scales"], ws, sa, o)
        run_act = lambda o: fz.gemm_silu_and_mul_per_group(A, W, pr["s2_zeros"], pr["s2_scales"], ws, sa, o, None)
    del acc
    side = torch.cuda.Stream(device=gpu)
    bufs = [torch.empty((64 << 20,), dtype=torch.uint8, device=gpu) for _ in range(3)]
    first = torch.empty((M, N), dtype=torch.float16, device=gpu)
    run(first)
    assert torch.equal(first.view(torch.int16), want.view(torch.int16)), "first run differs from the exact reference"
    out = torch.empty_like(first)
    for r in range(REPS):
        if r % 3 != 2:
            thrash(side, bufs, 2 + r % 4)
        out.fill_(float("nan"))
        run(out)
        assert torch.equal(out.view(torch.int16), first.view(torch.int16)), f"run {r} differs"
    if N % 128 == 0 and (K >= 1024 or M > 64):        # the silu * mul epilogue (ring without K slices / tiled kernels)
        act0 = torch.empty((M, N // 2), dtype=torch.float16, device=gpu)
        try:
            run_act(act0)
        except RuntimeError as e:                     # shapes served by a kernel family without the epilogue need scratch
            assert "scratch" in str(e)
            torch.cuda.synchronize()
            return
        act = torch.empty_like(act0)
        for r in range(REPS):
            if r % 3 != 2:
                thrash(side, bufs, 2 + r % 4)
            act.fill_(float("nan"))
            run_act(act)
            assert torch.equal(act.view(torch.int16), act0.view(torch.int16)), f"silu * mul run {r} differs"
    torch.cuda.synchronize()


@pytest.mark.parametrize("B,H,Hkv,L,int4", [(64, 32, 8, 1033, True), (8, 32, 8, 4000, True), (64, 32, 8, 1033, False),
                                             (8, 32, 8, 7680, False), (3, 8, 2, 130, True)])
def test_decode_attention_repeats_are_identical(gpu, B, H, Hkv, L, int4):
    """The decode attention launch (LDS-DMA page pipeline with counted waits, service-wave flag, split-KV partials, the
    per-sequence arrival ticket of the fused quantiser): same inputs, REPS launches under a thrashing side stream - fp16
    output, int8 row, scale, sum and every cache byte identical to the first launch (whose correctness is
    tests/test_attention_gpu.py's subject)."""
    from test_attention_gpu import ROPE, DevPools
    from qserve_amd import fused
    g = torch.Generator(device=gpu).manual_seed(B + H + L)
    mb = (L + 63) // 64 + 1
    dhb = 64 if int4 else 128
    nblocks = B * mb
    pools = DevPools(nblocks, Hkv, int4, gpu, fill=0)
    nd = Hkv * 64 * dhb
    for p in (pools.k, pools.v):
        p[:, :nd] = torch.randint(0, 256, (nblocks, nd), dtype=torch.uint8, device=gpu, generator=g)
        p[:, nd:].view(torch.float16).copy_((torch.rand((nblocks, (pools.pb - nd) // 2), device=gpu, generator=g) * 0.5 + 0.05).half())
    tables = torch.stack([torch.randperm(nblocks, generator=torch.Generator().manual_seed(1)).reshape(B, mb),
                          torch.randperm(nblocks, generator=torch.Generator().manual_seed(2)).reshape(B, mb)], dim=1).numpy()
    ptrs = pools.pointers(tables)
    new = torch.randn((B, (H + 2 * Hkv) * 128), generator=g, device=gpu, dtype=torch.float16)
    q, k, v = new.split([H * 128, Hkv * 128, Hkv * 128], dim=-1)
    q, k, v = q.reshape(B, H, 128), k.reshape(B, Hkv, 128), v.reshape(B, Hkv, 128)
    lens = torch.randint(max(1, L - 200), L + 1, (B,), generator=torch.Generator().manual_seed(3)).to(torch.int32).to(gpu)
    lens[0] = L
    side = torch.cuda.Stream(device=gpu)
    bufs = [torch.empty((64 << 20,), dtype=torch.uint8, device=gpu) for _ in range(3)]
    first = None
    for r in range(REPS + 1):
        if r % 3 != 2:
            thrash(side, bufs, 2 + r % 4)
        qq = torch.full((B, H * 128), 55, dtype=torch.int8, device=gpu)
        sc = torch.full((B,), 5.0, dtype=torch.float16, device=gpu)
        sm = torch.full((B,), 7.0, dtype=torch.float16, device=gpu)
        out = fused.single_query_attention_quant(q, k, v, ptrs, lens, qq, sc, 8192, 64, Hkv * dhb, L, 128, ROPE, True, int4,
                                                 True, quant_sum=sm)
        cur = (out.view(torch.int16).clone(), qq, sc.view(torch.int16).clone(), sm.view(torch.int16).clone(),
               pools.k.clone(), pools.v.clone())
        if first is None:
            first = cur
        else:
            for a, b, what in zip(cur, first, ("output", "int8 row", "scale", "sum", "k pages", "v pages")):
                assert torch.equal(a, b), f"launch {r}: {what} differs"
    torch.cuda.synchronize()
