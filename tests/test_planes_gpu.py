"""K-slice planes (include/qserve_amd.h, round 4): a row-parallel W4A8 GEMM leaves int32 partial sums per K slice and the add +
norm + quant launch behind it sums them and applies the GEMM's epilogue.  Everything that leaves the pair - the residual stream,
the int8 row, the fp16 scale and row sum - must be BIT-IDENTICAL to the ordinary pair (GEMM with its own epilogue, then
add_residual_rms_norm_general), for both epilogues, for every geometry the planes launch can take and for the planner's own choice;
and the planes themselves must sum to the GEMM's exact int32 accumulators."""
import numpy as np
import pytest
import torch

from oracle import synth, w4a8
from qserve_amd import _lib, fused

pytestmark = pytest.mark.gpu


def _problem(M, N, K, per_group, gpu):
    if per_group:
        pr = synth.per_group_problem(M, N, K, seed=M * 3 + N + K, valid=True)
        acc = w4a8.gemm_per_group_acc(pr["A"], pr["qweight"], pr["s2_zeros"], pr["s2_scales"])
    else:
        pr = synth.per_channel_problem(M, N, K, seed=M + N + K)
        acc = w4a8.gemm_per_chn_acc(pr["A"], pr["qweight"])
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(gpu) for k, v in pr.items() if isinstance(v, np.ndarray)}
    return pr, t, acc


def _pair(t, per_group, hidden, gamma, M, N):
    import qserve_backend.qgemm_w4a8_per_chn as gc
    import qserve_backend.qgemm_w4a8_per_group as gg
    delta = torch.empty((M, N), dtype=torch.float16, device=hidden.device)
    if per_group:
        gg.gemm_forward_cuda(t["A"], t["qweight"], t["s2_zeros"], t["s2_scales"], t["wscales"], t["ascales"], delta)
    else:
        gc.gemm_forward_cuda(t["A"], t["qweight"], t["wscales"], t["ascales"], t["w_szs"], t["a_ssums"], delta)
    h = hidden.clone()
    q = torch.empty((M, N), dtype=torch.int8, device=hidden.device)
    sc = torch.empty((M,), dtype=torch.float16, device=hidden.device)
    sm = torch.full((M,), 3.0, dtype=torch.float16, device=hidden.device)
    fused.add_residual_rms_norm_general(q, h, delta, gamma, sc, 1e-5, None if per_group else sm)
    return h, q, sc, sm


@pytest.mark.parametrize("per_group", [False, True], ids=["per_channel", "g128"])
@pytest.mark.parametrize("M,N,K,variant", [
    (64, 4096, 14336, -1),      # Llama-3-8B down_proj, the planner's choice
    (64, 4096, 4096, -1),       # o_proj
    (33, 4096, 14336, -1), (16, 4096, 14336, -1), (1, 2048, 4096, -1), (64, 4096, 14336, 4600 + 100 + 20 + 1),   # <2,1> x 2 slices
    (64, 4096, 14336, 4600 + 300 + 20 + 2), (64, 4096, 14336, 4600 + 300 + 40 + 1), (64, 2048, 8192, 4600 + 10 + 1),
    (48, 4096, 2048, 4600 + 100 + 40 + 2), (128, 4096, 4096, 4600 + 100 + 80 + 2), (100, 2048, 8192, 4600 + 300 + 80 + 2),
])
def test_planes_pair_is_bit_identical_to_the_ordinary_pair(gpu, M, N, K, variant, per_group):
    pr, t, acc = _problem(M, N, K, per_group, gpu)
    g = torch.Generator(device=gpu).manual_seed(M + N)
    hidden = (torch.randn((M, N), device=gpu, generator=g) * 0.7).half()
    gamma = (torch.rand((N,), device=gpu, generator=g) + 0.5).half()
    h1, q1, sc1, sm1 = _pair(t, per_group, hidden, gamma, M, N)
    try:
        _lib.lib.qs_set_gemm_variant(variant)
        fused._PLANES_PLAN["nocache"] = True            # (the plan depends on the A/B switch this test turns)
        ks = fused.gemm_planes_plan(M, N, K, per_group)
        assert ks in (1, 2, 4)
        if variant >= 4600:
            assert ks == (variant - 4600) // 100 + 1
        planes = torch.full((ks, M, N), 12345, dtype=torch.int32, device=gpu)
        if per_group:
            fused.gemm_planes(t["A"], t["qweight"], planes, t["s2_zeros"], t["s2_scales"])
        else:
            fused.gemm_planes(t["A"], t["qweight"], planes)
    finally:
        _lib.lib.qs_set_gemm_variant(-1)
        fused._PLANES_PLAN.clear()
    assert np.array_equal(planes.sum(dim=0, dtype=torch.int64).cpu().numpy(), acc.astype(np.int64)), "planes do not sum to the accumulators"
    h2 = hidden.clone()
    q2 = torch.empty((M, N), dtype=torch.int8, device=gpu)
    # the GEMM's activation scale / sum live where the row kernel writes its own (as in the decode engine)
    sc2, sm2 = t["ascales"].clone(), (torch.full((M,), 3.0, dtype=torch.float16, device=gpu) if per_group else t["a_ssums"].clone())
    if per_group:
        fused.add_residual_rms_norm_general_planes(q2, h2, planes, t["wscales"], sc2, gamma, sc2, 1e-5)
    else:
        fused.add_residual_rms_norm_general_planes(q2, h2, planes, t["wscales"], sc2, gamma, sc2, 1e-5, w_szs=t["w_szs"],
                                                   a_ssums=sm2, input_sum=sm2)
    torch.cuda.synchronize()
    assert torch.equal(h2.view(torch.int16), h1.view(torch.int16)), "residual stream differs"
    assert torch.equal(sc2.view(torch.int16), sc1.view(torch.int16)), "scale differs"
    assert torch.equal(q2, q1), "int8 row differs"
    assert torch.equal(sm2.view(torch.int16), sm1.view(torch.int16)), "row sum differs (or was touched without being asked for)"


def test_planes_plan_and_argument_checks(gpu):
    assert fused.gemm_planes_plan(64, 4096, 14336) in (1, 2, 4)
    assert fused.gemm_planes_plan(64, 4096, 512) == 0            # short K: no planes launch, the caller runs the pair
    assert fused.gemm_planes_plan(2048, 4096, 4096) == 0         # prompt-sized M
    pr, t, _ = _problem(16, 4096, 4096, False, gpu)
    with pytest.raises(RuntimeError):
        fused.gemm_planes(t["A"], t["qweight"], torch.empty((3, 16, 4096), dtype=torch.int32, device=gpu))
    planes = torch.zeros((2, 16, 4096), dtype=torch.int32, device=gpu)
    h = torch.zeros((16, 4096), dtype=torch.float16, device=gpu)
    q = torch.empty((16, 4096), dtype=torch.int8, device=gpu)
    sc = torch.ones((16,), dtype=torch.float16, device=gpu)
    with pytest.raises(RuntimeError):      # w_szs without a_ssums
        fused.add_residual_rms_norm_general_planes(q, h, planes, t["wscales"], sc, t["wscales"], sc, 1e-5, w_szs=t["w_szs"])
