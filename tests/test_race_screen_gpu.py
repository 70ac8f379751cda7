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


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("mode", ["per_channel", "per_group"])
def test_repeated_runs_are_identical_and_exact(gpu, M, N, K, mode):
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
