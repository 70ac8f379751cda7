"""GPU parity of the activation-side kernels (they define the GEMM's A / ascales / a_ssums)."""
import numpy as np
import pytest
import torch

from _helpers import dev, ulp_diff_f16
from oracle import fused

pytestmark = pytest.mark.gpu


def _check_int8(q_gpu, q_ref, pre):
    """int8 must match exactly except where the pre-rounding fp32 value is within 2e-3 of a tie (the rounding then
    depends on fp32 summation order of the row statistics, which differs between any two implementations)."""
    q = q_gpu.cpu().numpy().astype(np.int32)
    frac = np.abs(pre - np.floor(pre) - 0.5)
    safe = frac > 2e-3
    assert np.array_equal(q[safe], q_ref.astype(np.int32)[safe])
    assert np.abs(q - q_ref.astype(np.int32)).max() <= 1


@pytest.mark.parametrize("T,H", [(1, 64), (5, 4096), (64, 4096), (3, 14336), (2, 8192)])
def test_invoke_quant_fuse_sum(gpu, T, H):
    import qserve_backend.fused_kernels as op
    x = (np.random.default_rng(T + H).standard_normal((T, H)) * 2).astype(np.float16)
    q_ref, s_ref, sum_ref, pre = fused.quant_per_token(x, with_sum=True)
    out = torch.empty((T, H), dtype=torch.int8, device=gpu)
    sc = torch.empty((T,), dtype=torch.float16, device=gpu)
    sm = torch.empty((T,), dtype=torch.float16, device=gpu)
    op.invoke_quant_fuse_sum(out, dev(x), sm, sc)
    assert np.array_equal(sc.cpu().numpy().view(np.uint16), s_ref.view(np.uint16))       # amax is order independent
    assert np.array_equal(out.cpu().numpy(), q_ref)                                        # so the int8 are exact
    assert ulp_diff_f16(sm.cpu().numpy(), sum_ref).max() <= 2 or np.allclose(sm.cpu().numpy().astype(np.float32), sum_ref.astype(np.float32), atol=0.05)
    out2 = torch.empty_like(out)
    sc2 = torch.empty_like(sc)
    op.invoke_quant(out2, dev(x), sc2)
    assert torch.equal(out2, out) and torch.equal(sc2, sc)


@pytest.mark.parametrize("T,H", [(1, 64), (4, 4096), (64, 4096), (2, 8192)])
def test_rms_norm_general_fuse_sum(gpu, T, H):
    import qserve_backend.layernorm_ops as op
    r = np.random.default_rng(T * 7 + H)
    x = (r.standard_normal((T, H)) * 1.5 + 0.3).astype(np.float16)
    g = r.uniform(0.5, 1.5, H).astype(np.float16)
    q_ref, s_ref, sum_ref, pre = fused.rms_norm_general(x, g, 1e-5, with_sum=True)
    out = torch.empty((T, H), dtype=torch.int8, device=gpu)
    sc = torch.empty((T,), dtype=torch.float16, device=gpu)
    sm = torch.empty((T,), dtype=torch.float16, device=gpu)
    op.rms_norm_general_fuse_sum(out, dev(x), dev(g), sm, sc, 1e-5, True)
    assert ulp_diff_f16(sc.cpu().numpy(), s_ref).max() <= 1
    _check_int8(out, q_ref, pre)
    # a_ssum: the reference accumulates per-thread partials in fp16 (1024-thread partition); any other partition
    # differs by fp16 rounding noise of the partials
    assert np.allclose(sm.cpu().numpy().astype(np.float32), sum_ref.astype(np.float32), atol=0.25 + 2e-3 * H / 64)
    out2 = torch.empty_like(out)
    sc2 = torch.empty_like(sc)
    op.rms_norm_general(out2, dev(x), dev(g), sc2, 1e-5, True)
    assert torch.equal(out2, out) and torch.equal(sc2, sc)


def test_rms_norm_and_silu(gpu):
    import qserve_backend.activation_ops as act
    import qserve_backend.layernorm_ops as ln
    r = np.random.default_rng(8)
    x = r.standard_normal((9, 4096)).astype(np.float16)
    w = r.uniform(0.5, 1.5, 4096).astype(np.float16)
    out = torch.empty((9, 4096), dtype=torch.float16, device=gpu)
    ln.rms_norm(out, dev(x), dev(w), 1e-5)
    # half(x*rstd) can flip by one ulp with the fp32 summation order of the variance, the product by one more
    assert ulp_diff_f16(out.cpu().numpy(), fused.rms_norm(x, w, 1e-5)).max() <= 2
    y = (r.standard_normal((6, 2 * 14336)) * 2).astype(np.float16)
    o2 = torch.empty((6, 14336), dtype=torch.float16, device=gpu)
    act.silu_and_mul(o2, dev(y))
    # expf differs by an fp32 ulp between libraries: half(silu) can flip by one ulp, the product by one more
    assert ulp_diff_f16(o2.cpu().numpy(), fused.silu_and_mul(y)).max() <= 2
