"""GPU parity of the decode GEMMs' row-op tails (gemm_w4a8_ring.hip ring_tail; qserve_amd.fused.gemm_add_norm_quant_* and
gemm_silu_and_mul_quant_*): one launch must produce, BIT FOR BIT, what the launches it stands for produce -
    GEMM -> residual add -> rms_norm_general(_fuse_sum)   (llama_w4a8_unpad.py:346-351, 358-360, 337)
    gate_up GEMM -> silu_and_mul -> invoke_quant(_fuse_sum)   (llama_w4a8_unpad.py:69-93)
for every ring geometry (forced), the K-sliced seam, ragged token counts, fewer workgroups than rows, with / without the
row sum, per-channel and g128, repeated launches over CHANGING data (a stale read of a previous launch's rows would
differ) under a thrashing side stream, inside a hipGraph, and for whole decode steps.  The separate launches themselves
are checked against the oracle in tests/test_gemm_gpu.py / tests/test_fused_gpu.py."""
import os

import pytest
import torch

from _helpers import per_group_problem_torch

pytestmark = pytest.mark.gpu
REPS = int(os.environ.get("QS_RACE_REPS", "12"))
EPS = 1e-5


def lib():
    from qserve_amd._lib import lib as L
    return L


def problem(gpu, M, N, K, mode, seed):
    g = torch.Generator(device=gpu).manual_seed(seed)
    if mode == "per_channel":
        A = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=gpu, generator=g)
        W = torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=gpu, generator=g)
        ws = (torch.rand((N,), device=gpu, generator=g) * 0.004 + 0.001).half()
        sa = (torch.rand((M,), device=gpu, generator=g) * 0.02 + 0.005).half()
        wz = (torch.randint(0, 16, (N,), device=gpu, generator=g).half() * ws).half()
        ss = (sa.float() * A.float().sum(1)).half()
        rest = (ws, sa, wz, ss)
    else:
        pr = per_group_problem_torch(M, N, K, gpu, seed=seed)
        A, W = pr["A"], pr["qweight"]
        rest = (pr["s2_zeros"], pr["s2_scales"], pr["wscales"], pr["ascales"])
    return A, W, rest


def run_add_norm(gpu, M, N, K, mode, with_sum, variant=-1, expect_tail=True, reps=2, thrash=None, seed=0):
    import qserve_backend.qgemm_w4a8_per_chn as opc
    import qserve_backend.qgemm_w4a8_per_group as opg
    from qserve_amd import fused as fz
    L = lib()
    g = torch.Generator(device=gpu).manual_seed(seed + 99)
    gamma = (torch.rand((N,), device=gpu, generator=g) + 0.5).half()
    gemm = opc.gemm_forward_cuda if mode == "per_channel" else opg.gemm_forward_cuda
    fused = fz.gemm_add_norm_quant_per_chn if mode == "per_channel" else fz.gemm_add_norm_quant_per_group
    n_tail = 0
    for rep in range(reps):                       # new data every launch: nothing may survive from the previous one
        A, W, rest = problem(gpu, M, N, K, mode, seed + 17 * rep + M + N + K)
        h0 = (torch.randn((M, N), device=gpu, generator=g) * 2).half()
        # the launches it stands for (default dispatcher)
        L.qs_set_gemm_variant(-1)
        out_ref = torch.empty((M, N), dtype=torch.float16, device=gpu)
        gemm(A, W, *rest, out_ref)
        h_ref = h0.clone()
        q_ref = torch.empty((M, N), dtype=torch.int8, device=gpu)
        sc_ref = torch.full((M,), -1, dtype=torch.float16, device=gpu)
        sm_ref = torch.full((M,), -1, dtype=torch.float16, device=gpu)
        fz.add_residual_rms_norm_general(q_ref, h_ref, out_ref, gamma, sc_ref, EPS, sm_ref if with_sum else None)
        # one call
        out = torch.full((M, N), float("nan"), dtype=torch.float16, device=gpu)
        h = h0.clone()
        q = torch.full((M, N), 77, dtype=torch.int8, device=gpu)
        sc = torch.full((M,), -1, dtype=torch.float16, device=gpu)
        sm = torch.full((M,), -1, dtype=torch.float16, device=gpu)
        if thrash is not None and rep % 3 != 2:
            thrash(2 + rep % 4)
        before = L.qs_debug_tail_launches()
        L.qs_set_gemm_variant(variant)
        try:
            fused(A, W, *rest, out, h, gamma, q, sc, EPS, sm if with_sum else None)
        finally:
            L.qs_set_gemm_variant(-1)
        n_tail += L.qs_debug_tail_launches() - before
        torch.cuda.synchronize()
        assert torch.equal(out.view(torch.int16), out_ref.view(torch.int16)), f"rep {rep}: GEMM output differs"
        assert torch.equal(h.view(torch.int16), h_ref.view(torch.int16)), f"rep {rep}: residual stream differs"
        assert torch.equal(q, q_ref), f"rep {rep}: int8 row differs"
        assert torch.equal(sc.view(torch.int16), sc_ref.view(torch.int16)), f"rep {rep}: scale differs"
        assert torch.equal(sm.view(torch.int16), sm_ref.view(torch.int16)), f"rep {rep}: row sum differs (or touched unasked)"
    assert not fz.fused_tail_gave_up()
    if expect_tail is not None:
        assert (n_tail == reps) == expect_tail, f"{n_tail} of {reps} launches took the tail, expected_tail={expect_tail}"


def run_silu_quant(gpu, M, N, K, mode, with_sum, variant=-1, expect_tail=True, reps=2, thrash=None, seed=0):
    import qserve_backend.fused_kernels as fk
    from qserve_amd import fused as fz
    L = lib()
    pair = fz.gemm_silu_and_mul_per_chn if mode == "per_channel" else fz.gemm_silu_and_mul_per_group
    fused = fz.gemm_silu_and_mul_quant_per_chn if mode == "per_channel" else fz.gemm_silu_and_mul_quant_per_group
    n_tail = 0
    for rep in range(reps):
        A, W, rest = problem(gpu, M, N, K, mode, seed + 31 * rep + M + N + K)
        tmp = torch.empty((M, N), dtype=torch.float16, device=gpu)
        L.qs_set_gemm_variant(-1)
        act_ref = torch.empty((M, N // 2), dtype=torch.float16, device=gpu)
        pair(A, W, *rest, act_ref, tmp)
        q_ref = torch.empty((M, N // 2), dtype=torch.int8, device=gpu)
        sc_ref = torch.full((M,), -1, dtype=torch.float16, device=gpu)
        sm_ref = torch.full((M,), -1, dtype=torch.float16, device=gpu)
        if with_sum:
            fk.invoke_quant_fuse_sum(q_ref, act_ref, sm_ref, sc_ref)
        else:
            fk.invoke_quant(q_ref, act_ref, sc_ref)
        act = torch.full((M, N // 2), float("nan"), dtype=torch.float16, device=gpu)
        q = torch.full((M, N // 2), 77, dtype=torch.int8, device=gpu)
        sc = torch.full((M,), -1, dtype=torch.float16, device=gpu)
        sm = torch.full((M,), -1, dtype=torch.float16, device=gpu)
        if thrash is not None and rep % 3 != 2:
            thrash(2 + rep % 4)
        before = L.qs_debug_tail_launches()
        L.qs_set_gemm_variant(variant)
        try:
            fused(A, W, *rest, act, q, sc, tmp, sm if with_sum else None)
        finally:
            L.qs_set_gemm_variant(-1)
        n_tail += L.qs_debug_tail_launches() - before
        torch.cuda.synchronize()
        assert torch.equal(act.view(torch.int16), act_ref.view(torch.int16)), f"rep {rep}: silu * mul output differs"
        assert torch.equal(q, q_ref), f"rep {rep}: int8 row differs"
        assert torch.equal(sc.view(torch.int16), sc_ref.view(torch.int16)), f"rep {rep}: scale differs"
        assert torch.equal(sm.view(torch.int16), sm_ref.view(torch.int16)), f"rep {rep}: row sum differs (or touched unasked)"
    assert not fz.fused_tail_gave_up()
    if expect_tail is not None:
        assert (n_tail == reps) == expect_tail, f"{n_tail} of {reps} launches took the tail, expected_tail={expect_tail}"


# ---- the decode step's own shapes (BASELINE configs 2 and 3), dispatcher's choice ------------------------------------
@pytest.mark.parametrize("mode", ["per_channel", "per_group"])
@pytest.mark.parametrize("with_sum", [True, False])
@pytest.mark.parametrize("M,N,K", [(64, 4096, 4096), (64, 4096, 14336), (16, 4096, 4096), (33, 4096, 14336), (1, 4096, 4096)])
def test_o_and_down_with_tail_equal_the_launch_pair(gpu, M, N, K, with_sum, mode):
    run_add_norm(gpu, M, N, K, mode, with_sum, seed=1)


@pytest.mark.parametrize("mode", ["per_channel", "per_group"])
@pytest.mark.parametrize("with_sum", [True, False])
@pytest.mark.parametrize("M,N,K", [(64, 28672, 4096), (16, 28672, 4096), (50, 28672, 4096)])
def test_gate_up_with_tail_equals_the_launches(gpu, M, N, K, with_sum, mode):
    run_silu_quant(gpu, M, N, K, mode, with_sum, seed=2)


# ---- every ring geometry, forced (4100 + 100 (k_slices - 1) + 10 m_tiles + units), incl. fewer workgroups than rows -----
GEOS = [(1, 1, 1), (1, 2, 1), (1, 4, 1), (1, 2, 2), (1, 4, 2), (1, 4, 4), (2, 2, 1), (2, 4, 2), (4, 1, 1), (2, 1, 1)]


@pytest.mark.parametrize("mode", ["per_channel", "per_group"])
@pytest.mark.parametrize("ks,mt,wn", GEOS)
def test_every_ring_geometry_with_add_norm_tail(gpu, ks, mt, wn, mode):
    variant = 4100 + 100 * (ks - 1) + 10 * mt + wn
    # N = 2048: the one-chunk row layout, grids of 8 .. 128 workgroups; M = 70: two ragged token blocks for mt <= 2
    for (M, N, K) in [(70, 2048, 4096), (64, 4096, 4096)]:
        grid = (N // (64 * wn)) * ((M + 16 * mt - 1) // (16 * mt)) * ks
        run_add_norm(gpu, M, N, K, mode, True, variant=variant, expect_tail=grid <= 256, seed=ks * 100 + mt * 10 + wn)


@pytest.mark.parametrize("mode", ["per_channel", "per_group"])
@pytest.mark.parametrize("mt,wn", [(1, 1), (2, 1), (4, 1), (2, 2), (4, 2), (4, 4)])
def test_every_ring_geometry_with_quant_tail(gpu, mt, wn, mode):
    variant = 4100 + 10 * mt + wn
    # act rows of 2048 (256-thread layout, one chunk), 4096 (two chunks), 7168 (1024-thread layout)
    for (M, N, K) in [(40, 4096, 1024), (64, 8192, 2048), (64, 14336, 1024)]:
        grid = (N // (64 * wn)) * ((M + 16 * mt - 1) // (16 * mt))
        run_silu_quant(gpu, M, N, K, mode, True, variant=variant, expect_tail=grid <= 256, seed=mt * 10 + wn)


def test_shapes_beyond_one_round_fall_back_to_separate_launches(gpu):
    """More workgroups than compute units (the owners' wait would not be safe), rows beyond the tail's layouts, the tiled
    and the older kernels: the same entry points issue the row kernel themselves - identical results."""
    run_add_norm(gpu, 256, 4096, 4096, "per_channel", True, expect_tail=None, seed=5)       # whatever the dispatcher takes
    run_add_norm(gpu, 64, 8192, 8192, "per_channel", True, expect_tail=False, seed=6)       # hidden 8192 > 4096
    run_add_norm(gpu, 2048, 4096, 4096, "per_group", False, expect_tail=False, seed=7)      # tiled kernel
    run_add_norm(gpu, 64, 4096, 640, "per_channel", True, expect_tail=False, seed=8)        # K % 512 != 0: split-K kernel
    run_silu_quant(gpu, 128, 28672, 4096, "per_channel", True, expect_tail=None, seed=9)    # four-unit workgroups: 224
    run_silu_quant(gpu, 300, 8192, 4096, "per_group", True, expect_tail=None, seed=10)
    run_silu_quant(gpu, 1024, 28672, 4096, "per_channel", False, expect_tail=False, seed=11)  # tiled kernel


# ---- race screen: changing data, thrashing side stream, many launches --------------------------------------------------
@pytest.mark.parametrize("mode", ["per_channel", "per_group"])
def test_tail_race_screen(gpu, mode):
    side = torch.cuda.Stream(device=gpu)
    bufs = [torch.empty((64 << 20,), dtype=torch.uint8, device=gpu) for _ in range(3)]

    def thrash(n):
        with torch.cuda.stream(side):
            for i in range(n):
                bufs[(i + 1) % 3].copy_(bufs[i % 3])
    run_add_norm(gpu, 64, 4096, 4096, mode, True, reps=REPS, thrash=thrash, seed=11)
    run_add_norm(gpu, 64, 4096, 14336, mode, True, reps=REPS, thrash=thrash, seed=12)
    run_silu_quant(gpu, 64, 28672, 4096, mode, True, reps=REPS, thrash=thrash, seed=13)
    torch.cuda.synchronize()


def test_tails_replay_from_a_hipgraph(gpu):
    """The arrival words reset themselves: a captured launch replays with fresh inputs and keeps matching the pair."""
    import qserve_backend.qgemm_w4a8_per_chn as opc
    from qserve_amd import fused as fz
    M, N, K = 64, 4096, 4096
    A, W, rest = problem(gpu, M, N, K, "per_channel", 3)
    gamma = torch.ones((N,), dtype=torch.float16, device=gpu)
    h = torch.zeros((M, N), dtype=torch.float16, device=gpu)
    out = torch.empty((M, N), dtype=torch.float16, device=gpu)
    q = torch.empty((M, N), dtype=torch.int8, device=gpu)
    sc = torch.empty((M,), dtype=torch.float16, device=gpu)
    sm = torch.empty((M,), dtype=torch.float16, device=gpu)
    fz.gemm_add_norm_quant_per_chn(A, W, *rest, out, h, gamma, q, sc, EPS, sm)      # eager first (scratch allocation)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(3):                       # three dependent launches per replay: h accumulates
            fz.gemm_add_norm_quant_per_chn(A, W, *rest, out, h, gamma, q, sc, EPS, sm)
    for rep in range(4):
        A.copy_(torch.randint(-127, 128, (M, K), dtype=torch.int8, device=gpu))
        h0 = (torch.randn((M, N), device=gpu) * 2).half()
        h.copy_(h0)
        g.replay()
        torch.cuda.synchronize()
        out_ref = torch.empty_like(out)
        opc.gemm_forward_cuda(A, W, *rest, out_ref)
        h_ref, q_ref, sc_ref, sm_ref = h0.clone(), torch.empty_like(q), torch.empty_like(sc), torch.empty_like(sm)
        for _ in range(3):
            fz.add_residual_rms_norm_general(q_ref, h_ref, out_ref, gamma, sc_ref, EPS, sm_ref)
        torch.cuda.synchronize()
        assert torch.equal(h.view(torch.int16), h_ref.view(torch.int16)) and torch.equal(q, q_ref)
        assert torch.equal(sc.view(torch.int16), sc_ref.view(torch.int16)) and torch.equal(sm.view(torch.int16), sm_ref.view(torch.int16))
    assert not fz.fused_tail_gave_up()


@pytest.mark.parametrize("gs", [-1, 128])
def test_decode_steps_with_tails_equal_pairs_equal_op_by_op(gpu, gs):
    """Whole decode steps of a 3-layer Llama-3-8B-shaped model at the benchmark's batch: tails == fused pairs == the
    reference's op-by-op sequence (hidden state, final norm, sampled tokens), eager and replayed from one hipGraph."""
    from qserve_amd.decode import LLAMA3_8B, DecodeEngine
    L = lib()
    cfg = dict(LLAMA3_8B, layers=3, vocab=4096)
    outs = []
    for fuse, tails, graph in ((False, False, False), (True, False, False), (True, True, False), (True, True, True)):
        eng = DecodeEngine(cfg, batch=64, prompt_len=100, max_new=8, group_size=gs, device="cuda:0", seed=5,
                           fuse_pairs=fuse, fuse_tails=tails)
        eng.prefill_cache(100)
        before = L.qs_debug_tail_launches()
        toks = []
        if graph:
            eng.capture()                      # runs one warm-up step eagerly
            toks.append(eng.tokens.clone())
            for _ in range(3):
                eng.run()
                toks.append(eng.tokens.clone())
        else:
            for _ in range(4):
                eng.step()
                toks.append(eng.tokens.clone())
        torch.cuda.synchronize()
        took = L.qs_debug_tail_launches() - before
        if tails:
            assert took > 0, "the engine did not take a single row-op tail"
        else:
            assert took == 0
        outs.append((eng.hidden.clone(), eng.final.clone(), torch.stack(toks)))
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert torch.equal(a, b)
    from qserve_amd import fused as fz
    assert not fz.fused_tail_gave_up()
