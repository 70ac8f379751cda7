"""GPU parity: W4A8 GEMMs through the C ABI (via the qserve_backend mirror) vs the CPU oracle.
Bar: INT32 accumulators bit-exact; fp16 output bit-exact vs the oracle's un-contracted fp32 epilogue
(<= 1 fp16 ulp is what the reference itself guarantees)."""
import numpy as np
import pytest
import torch

from _helpers import (dev, int_matmul_torch, pack_qweight_torch, per_group_problem_torch, ulp_diff_f16,
                      unpack_qweight_torch)
from oracle import synth, w4a8

pytestmark = pytest.mark.gpu

SHAPES = [(1, 64, 128), (7, 64, 256), (16, 128, 128), (17, 192, 384), (33, 64, 1024), (48, 128, 512),
          (64, 256, 512), (65, 64, 256), (100, 128, 640), (128, 192, 256), (200, 64, 384)]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_per_channel_vs_oracle(gpu, M, N, K):
    import qserve_backend.qgemm_w4a8_per_chn as op
    pr = synth.per_channel_problem(M, N, K, seed=M + N + K)
    acc_ref, out_ref = w4a8.gemm_per_chn(pr["A"], pr["qweight"], pr["wscales"], pr["ascales"], pr["w_szs"], pr["a_ssums"])
    A, W = dev(pr["A"]), dev(pr["qweight"])
    acc = torch.full((M, N), -7, dtype=torch.int32, device=gpu)
    op.gemm_forward_acc(A, W, acc)
    assert np.array_equal(acc.cpu().numpy(), acc_ref), "int32 accumulator mismatch"
    out = torch.full((M, N), float("nan"), dtype=torch.float16, device=gpu)
    op.gemm_forward_cuda(A, W, dev(pr["wscales"]), dev(pr["ascales"]), dev(pr["w_szs"]), dev(pr["a_ssums"]), out)
    d = ulp_diff_f16(out.cpu().numpy(), out_ref)
    assert d.max() == 0, f"fp16 output differs by up to {d.max()} ulp"


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("valid", [True, False])
def test_per_group_vs_oracle(gpu, M, N, K, valid):
    import qserve_backend.qgemm_w4a8_per_group as op
    pr = synth.per_group_problem(M, N, K, seed=M * 3 + N + K, valid=valid)
    acc_ref, out_ref = w4a8.gemm_per_group(pr["A"], pr["qweight"], pr["s2_zeros"], pr["s2_scales"], pr["wscales"], pr["ascales"])
    A, W, Z, S = dev(pr["A"]), dev(pr["qweight"]), dev(pr["s2_zeros"]), dev(pr["s2_scales"])
    acc = torch.full((M, N), -7, dtype=torch.int32, device=gpu)
    op.gemm_forward_acc(A, W, Z, S, acc)
    assert np.array_equal(acc.cpu().numpy(), acc_ref), "int32 accumulator (dequantised int8 path) mismatch"
    out = torch.full((M, N), float("nan"), dtype=torch.float16, device=gpu)
    op.gemm_forward_cuda(A, W, Z, S, dev(pr["wscales"]), dev(pr["ascales"]), out)
    assert ulp_diff_f16(out.cpu().numpy(), out_ref).max() == 0


TILED = [(256, 256, 256), (300, 512, 384), (513, 256, 1024), (1000, 768, 512)]   # prefill-sized: LDS-tiled kernel


@pytest.fixture(params=[3001, 3002, 3003], ids=["tile256", "tile128", "wide256"])
def tiled_variant(request):
    """The dispatcher only picks the tiled kernel for chip-filling shapes; force it for oracle-sized ones."""
    from qserve_amd import _lib
    _lib.lib.qs_set_gemm_variant(request.param)
    yield request.param
    _lib.lib.qs_set_gemm_variant(-1)


@pytest.mark.parametrize("M,N,K", TILED)
def test_tiled_per_channel_vs_oracle(gpu, tiled_variant, M, N, K):
    import qserve_backend.qgemm_w4a8_per_chn as op
    pr = synth.per_channel_problem(M, N, K, seed=M + N + K)
    acc_ref, out_ref = w4a8.gemm_per_chn(pr["A"], pr["qweight"], pr["wscales"], pr["ascales"], pr["w_szs"], pr["a_ssums"])
    A, W = dev(pr["A"]), dev(pr["qweight"])
    acc = torch.full((M + 3, N), -7, dtype=torch.int32, device=gpu)
    op.gemm_forward_acc(A, W, acc[:M])
    assert np.array_equal(acc[:M].cpu().numpy(), acc_ref)
    assert torch.all(acc[M:] == -7), "rows beyond M were written"
    out = torch.full((M, N), float("nan"), dtype=torch.float16, device=gpu)
    op.gemm_forward_cuda(A, W, dev(pr["wscales"]), dev(pr["ascales"]), dev(pr["w_szs"]), dev(pr["a_ssums"]), out)
    assert ulp_diff_f16(out.cpu().numpy(), out_ref).max() == 0


@pytest.mark.parametrize("M,N,K", TILED)
@pytest.mark.parametrize("valid", [True, False])
def test_tiled_per_group_vs_oracle(gpu, tiled_variant, M, N, K, valid):
    import qserve_backend.qgemm_w4a8_per_group as op
    pr = synth.per_group_problem(M, N, K, seed=M * 3 + N + K, valid=valid)
    acc_ref, out_ref = w4a8.gemm_per_group(pr["A"], pr["qweight"], pr["s2_zeros"], pr["s2_scales"], pr["wscales"], pr["ascales"])
    A, W, Z, S = dev(pr["A"]), dev(pr["qweight"]), dev(pr["s2_zeros"]), dev(pr["s2_scales"])
    acc = torch.full((M, N), -7, dtype=torch.int32, device=gpu)
    op.gemm_forward_acc(A, W, Z, S, acc)
    assert np.array_equal(acc.cpu().numpy(), acc_ref)
    out = torch.full((M, N), float("nan"), dtype=torch.float16, device=gpu)
    op.gemm_forward_cuda(A, W, Z, S, dev(pr["wscales"]), dev(pr["ascales"]), out)
    assert ulp_diff_f16(out.cpu().numpy(), out_ref).max() == 0


@pytest.mark.parametrize("kernel", [3001, 3003], ids=["tile256", "wide256"])
@pytest.mark.parametrize("mode", ["per_channel", "per_group"])
@pytest.mark.parametrize("M,N,K", [(1000, 768, 512), (1281, 1024, 256), (700, 2560, 1152)])
def test_tiled_workgroup_walks_several_tiles(gpu, mode, M, N, K, kernel):
    """Variant 3220: three workgroups walk all the (256-token) tiles - the next tile's pipeline fill is issued before the
    epilogue of the current one (what one workgroup per CU does at prompt sizes).  Same bits as one workgroup per tile
    (3210) and as the oracle."""
    import qserve_backend.qgemm_w4a8_per_chn as opc
    import qserve_backend.qgemm_w4a8_per_group as opg
    from qserve_amd import _lib
    if mode == "per_channel":
        pr = synth.per_channel_problem(M, N, K, seed=M + K)
        _, out_ref = w4a8.gemm_per_chn(pr["A"], pr["qweight"], pr["wscales"], pr["ascales"], pr["w_szs"], pr["a_ssums"])
        args = [dev(pr[k]) for k in ("A", "qweight", "wscales", "ascales", "w_szs", "a_ssums")]
        fn = opc.gemm_forward_cuda
    else:
        pr = synth.per_group_problem(M, N, K, seed=M + K)
        _, out_ref = w4a8.gemm_per_group(pr["A"], pr["qweight"], pr["s2_zeros"], pr["s2_scales"], pr["wscales"], pr["ascales"])
        args = [dev(pr[k]) for k in ("A", "qweight", "s2_zeros", "s2_scales", "wscales", "ascales")]
        fn = opg.gemm_forward_cuda
    outs = []
    try:
        _lib.lib.qs_set_gemm_variant(kernel)          # the eight-wave tile / the four-wave tile (gemm_w4a8_wide.hip)
        for v in (3220, 3210):
            _lib.lib.qs_set_gemm_variant(v)
            out = torch.full((M + 2, N), float("nan"), dtype=torch.float16, device=gpu)
            for _ in range(2):                         # twice: nothing is left behind in LDS or in the workspace
                fn(*args, out[:M])
            assert torch.isnan(out[M:]).all(), "rows beyond M were written"
            outs.append(out[:M].cpu().numpy())
    finally:
        _lib.lib.qs_set_gemm_variant(3200)
        _lib.lib.qs_set_gemm_variant(-1)
    assert ulp_diff_f16(outs[0], out_ref).max() == 0
    assert np.array_equal(outs[0].view(np.uint16), outs[1].view(np.uint16))


# gate_up GEMM + silu_and_mul in one launch (qserve_amd.fused.gemm_silu_and_mul_*): (variant, M, N, K); N stacks
# [gate | up].  -1 = the dispatcher's own choice; 41xx ring geometries; 42xx K-sliced (no activation epilogue: two
# launches through tmp); 3001 / 3002 tiled; 3220 tiled with three workgroups walking the tiles
GATE_UP = [(4111, 16, 128, 1024), (4121, 23, 256, 1024), (4141, 50, 256, 2048), (4122, 32, 256, 512),
           (4142, 64, 512, 1024), (4144, 100, 512, 1024), (4144, 128, 1024, 2048), (4222, 20, 256, 1024),
           (3001, 300, 512, 384), (3002, 513, 1024, 512), (3220, 1000, 1536, 512), (-1, 64, 1792, 1024),
           (-1, 7, 256, 128), (3003, 300, 512, 384), (3003, 513, 1024, 512), (3223, 1000, 1536, 512),
           (4182, 128, 256, 1024), (4182, 100, 512, 2048), (4182, 200, 512, 512)]


@pytest.mark.parametrize("variant,M,N,K", GATE_UP)
@pytest.mark.parametrize("mode", ["per_channel", "per_group"])
def test_gate_up_silu_mul_vs_oracle_and_op_pair(gpu, variant, M, N, K, mode):
    """One call == plain GEMM entry followed by activation_ops.silu_and_mul, bit for bit; against the oracle within the
    two fp16 ulps the exponential's library leaves (tests/test_fused_gpu.py::test_rms_norm_and_silu)."""
    import qserve_backend.activation_ops as act
    import qserve_backend.qgemm_w4a8_per_chn as opc
    import qserve_backend.qgemm_w4a8_per_group as opg
    from oracle import fused as ofused
    from qserve_amd import _lib, fused as fz
    if mode == "per_channel":
        pr = synth.per_channel_problem(M, N, K, seed=M + N)
        _, y_ref = w4a8.gemm_per_chn(pr["A"], pr["qweight"], pr["wscales"], pr["ascales"], pr["w_szs"], pr["a_ssums"])
        args = [dev(pr[k]) for k in ("A", "qweight", "wscales", "ascales", "w_szs", "a_ssums")]
        plain, fusedop = opc.gemm_forward_cuda, fz.gemm_silu_and_mul_per_chn
    else:
        pr = synth.per_group_problem(M, N, K, seed=M + N)
        _, y_ref = w4a8.gemm_per_group(pr["A"], pr["qweight"], pr["s2_zeros"], pr["s2_scales"], pr["wscales"], pr["ascales"])
        args = [dev(pr[k]) for k in ("A", "qweight", "s2_zeros", "s2_scales", "wscales", "ascales")]
        plain, fusedop = opg.gemm_forward_cuda, fz.gemm_silu_and_mul_per_group
    y = torch.empty((M, N), dtype=torch.float16, device=gpu)
    pair = torch.empty((M, N // 2), dtype=torch.float16, device=gpu)
    plain(*args, y)
    act.silu_and_mul(pair, y)
    outs = []
    try:
        if variant == 3220:
            _lib.lib.qs_set_gemm_variant(3001)
        if variant == 3223:                            # the four-wave tile with three workgroups walking the tiles
            _lib.lib.qs_set_gemm_variant(3003)
            variant = 3220
        _lib.lib.qs_set_gemm_variant(variant)
        for two_launches in (0, 1):
            _lib.lib.qs_set_gemm_variant(3300 + two_launches)
            out = torch.full((M + 1, N // 2), float("nan"), dtype=torch.float16, device=gpu)
            tmp = torch.empty((M, N), dtype=torch.float16, device=gpu)
            fusedop(*args, out[:M], tmp)
            assert torch.isnan(out[M:]).all(), "rows beyond M were written"
            outs.append(out[:M].cpu().numpy())
        if variant // 100 != 42 and not (variant == -1 and K < 1024):    # the one-launch form needs no scratch
            _lib.lib.qs_set_gemm_variant(3300)
            out = torch.full((M, N // 2), float("nan"), dtype=torch.float16, device=gpu)
            fusedop(*args, out, None)
            outs.append(out.cpu().numpy())
    finally:
        _lib.lib.qs_set_gemm_variant(3300)
        _lib.lib.qs_set_gemm_variant(3200)
        _lib.lib.qs_set_gemm_variant(-1)
    want = pair.cpu().numpy()
    for o in outs:
        assert np.array_equal(o.view(np.uint16), want.view(np.uint16))
    assert ulp_diff_f16(want, ofused.silu_and_mul(y_ref)).max() <= 2


def test_gate_up_silu_mul_model_shapes_equal_op_pair(gpu):
    """Llama-3-8B gate_up at decode (ring kernel, two and four units) and prompt (tiled kernel, many tiles per workgroup)
    sizes, per-channel and g128: the one-launch form against the op pair on the device."""
    import qserve_backend.activation_ops as act
    import qserve_backend.qgemm_w4a8_per_chn as opc
    import qserve_backend.qgemm_w4a8_per_group as opg
    from qserve_amd import fused as fz
    N, K = 28672, 4096
    g = torch.Generator(device=gpu).manual_seed(11)
    W = torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=gpu, generator=g)
    ws = (torch.rand((N,), device=gpu, generator=g) * 0.004 + 0.001).half()
    wz = (torch.randint(0, 16, (N,), device=gpu, generator=g).half() * ws).half()
    s2 = torch.randint(1, 9, (K // 128, N), dtype=torch.int8, device=gpu, generator=g)
    z2 = (-(torch.randint(0, 16, (K // 128, N), device=gpu, generator=g).to(torch.int16) * s2.to(torch.int16))).to(torch.int8)
    for M in (64, 128, 3000):
        A = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=gpu, generator=g)
        sa = (torch.rand((M,), device=gpu, generator=g) * 0.02 + 0.005).half()
        ss = (sa.float() * A.float().sum(1)).half()
        y = torch.empty((M, N), dtype=torch.float16, device=gpu)
        pair = torch.empty((M, N // 2), dtype=torch.float16, device=gpu)
        out = torch.empty((M, N // 2), dtype=torch.float16, device=gpu)
        for per_group in (False, True):
            if per_group:
                opg.gemm_forward_cuda(A, W, z2, s2, ws, sa, y)
            else:
                opc.gemm_forward_cuda(A, W, ws, sa, wz, ss, y)
            act.silu_and_mul(pair, y)
            out.fill_(float("nan"))
            if per_group:
                fz.gemm_silu_and_mul_per_group(A, W, z2, s2, ws, sa, out, None)
            else:
                fz.gemm_silu_and_mul_per_chn(A, W, ws, sa, wz, ss, out, None)
            assert torch.equal(out.view(torch.int16), pair.view(torch.int16)), (M, per_group)


def test_tiled_kernel_equals_decode_kernel(gpu):
    """Same problem through both code paths (variant 3000 = tiled kernel off, 3001 = forced with the 128-token tile)."""
    import qserve_backend.qgemm_w4a8_per_group as op
    from qserve_amd import _lib
    pr = synth.per_group_problem(96, 512, 640, seed=5)
    A, W, Z, S = dev(pr["A"]), dev(pr["qweight"]), dev(pr["s2_zeros"]), dev(pr["s2_scales"])
    outs = []
    try:
        for v in (3000, 3001, 3002, 3003):
            _lib.lib.qs_set_gemm_variant(v)
            out = torch.full((96, 512), float("nan"), dtype=torch.float16, device=gpu)
            op.gemm_forward_cuda(A, W, Z, S, dev(pr["wscales"]), dev(pr["ascales"]), out)
            outs.append(out.cpu().numpy())
    finally:
        _lib.lib.qs_set_gemm_variant(-1)
    assert np.array_equal(outs[0].view(np.uint16), outs[1].view(np.uint16))
    assert np.array_equal(outs[0].view(np.uint16), outs[2].view(np.uint16))
    assert np.array_equal(outs[0].view(np.uint16), outs[3].view(np.uint16))


# decode ring kernel: (variant = 4100 + 100*(ksplit-1) + 10*m_tiles + units, M, N, K); K/64/ksplit divisible by
# 16/units, ragged M, M split
RING = [(4111, 16, 64, 1024), (4111, 40, 512, 2048), (4121, 23, 64, 1024), (4121, 64, 512, 1024),
        (4141, 50, 64, 2048), (4141, 100, 512, 1024), (4122, 32, 128, 512), (4122, 20, 256, 1024),
        (4142, 64, 128, 512), (4142, 37, 384, 1536),
        # K slices (4100 + 100*(ksplit-1) + ...): partial tiles meet in the sentinel-filled workspace, the last slice finishes
        (4211, 16, 64, 2048), (4221, 23, 64, 2048), (4241, 50, 128, 4096), (4222, 20, 256, 1024),
        (4422, 32, 256, 2048), (4442, 64, 128, 4096), (4411, 9, 192, 4096),
        # odd numbers of stages per K-group (K/64 not a multiple of 2*groups), down to a single stage
        (4122, 30, 128, 768), (4111, 16, 64, 1536), (4142, 64, 128, 256), (4121, 20, 64, 512), (4222, 33, 256, 1536),
        (4122, 64, 256, 11008),
        # four-unit workgroups (two K-groups): where 64 / 128 tokens meet many channels
        (4144, 64, 256, 1024), (4144, 128, 512, 4096), (4144, 100, 256, 256), (4244, 64, 512, 2048), (4144, 37, 256, 11008),
        # 128-token workgroups (round 5: 8 m-tiles per wave, two-pass reduction, ring depth 3-4), un-split and K-sliced
        (4182, 128, 256, 1024), (4182, 100, 128, 2048), (4182, 65, 384, 768), (4182, 97, 256, 256),
        (4282, 128, 128, 2048), (4482, 120, 256, 4096), (4182, 256, 128, 1536)]


@pytest.mark.parametrize("variant,M,N,K", RING)
@pytest.mark.parametrize("mode", ["per_channel", "per_group", "per_group_any_bytes"])
def test_ring_kernel_vs_oracle(gpu, variant, M, N, K, mode):
    from qserve_amd import _lib
    _lib.lib.qs_set_gemm_variant(variant)
    try:
        if mode == "per_channel":
            import qserve_backend.qgemm_w4a8_per_chn as op
            pr = synth.per_channel_problem(M, N, K, seed=M + N + K)
            acc_ref, out_ref = w4a8.gemm_per_chn(pr["A"], pr["qweight"], pr["wscales"], pr["ascales"], pr["w_szs"],
                                                 pr["a_ssums"])
            A, W = dev(pr["A"]), dev(pr["qweight"])
            acc = torch.full((M + 2, N), -7, dtype=torch.int32, device=gpu)
            op.gemm_forward_acc(A, W, acc[:M])
            out = torch.full((M + 2, N), float("nan"), dtype=torch.float16, device=gpu)
            op.gemm_forward_cuda(A, W, dev(pr["wscales"]), dev(pr["ascales"]), dev(pr["w_szs"]), dev(pr["a_ssums"]),
                                 out[:M])
        else:
            import qserve_backend.qgemm_w4a8_per_group as op
            pr = synth.per_group_problem(M, N, K, seed=M * 3 + N + K, valid=mode == "per_group")
            acc_ref, out_ref = w4a8.gemm_per_group(pr["A"], pr["qweight"], pr["s2_zeros"], pr["s2_scales"],
                                                   pr["wscales"], pr["ascales"])
            A, W, Z, S = dev(pr["A"]), dev(pr["qweight"]), dev(pr["s2_zeros"]), dev(pr["s2_scales"])
            acc = torch.full((M + 2, N), -7, dtype=torch.int32, device=gpu)
            op.gemm_forward_acc(A, W, Z, S, acc[:M])
            out = torch.full((M + 2, N), float("nan"), dtype=torch.float16, device=gpu)
            op.gemm_forward_cuda(A, W, Z, S, dev(pr["wscales"]), dev(pr["ascales"]), out[:M])
    finally:
        _lib.lib.qs_set_gemm_variant(-1)
    assert np.array_equal(acc[:M].cpu().numpy(), acc_ref), "int32 accumulators"
    assert torch.all(acc[M:] == -7) and torch.all(torch.isnan(out[M:])), "rows beyond M were written"
    assert ulp_diff_f16(out[:M].cpu().numpy(), out_ref).max() == 0


# K slices ACROSS the XCDs (round 6, gemm_w4a8_ring.hip ring_coords): the launcher takes the mapping for two-slice seams and for
# the planes form (2 / 4 slices) when the channel blocks divide by 8 / ksplit; ring flag 4096 keeps the mapping of rounds 3-5.
# Same exact results from both (the seam's "the finishing slice is dispatched last" survives: inside 8 consecutive workgroups the
# slice index grows); the four-slice seam cases below run the old mapping either way and stay as a control.
KXCD = [(4221, 40, 512, 2048), (4421, 64, 1024, 4096), (4222, 100, 1024, 2048), (4241, 120, 256, 4096), (4422, 64, 512, 2048),
        (4221, 64, 4096, 14336)]


@pytest.mark.parametrize("variant,M,N,K", KXCD)
@pytest.mark.parametrize("mode", ["per_channel", "per_group_any_bytes"])
def test_k_slices_across_xcds_and_on_one_xcd_are_exact(gpu, variant, M, N, K, mode):
    from qserve_amd import _lib
    g = torch.Generator(device=gpu).manual_seed(variant + M)
    outs = []
    try:
        for flags in (0, 4096):
            _lib.lib.qs_set_gemm_variant(5000 + flags)
            _lib.lib.qs_set_gemm_variant(variant)
            if mode == "per_channel":
                import qserve_backend.qgemm_w4a8_per_chn as op
                A = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=gpu, generator=g) if not outs else A
                W = torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=gpu, generator=g) if not outs else W
                acc = torch.full((M, N), -7, dtype=torch.int32, device=gpu)
                for rep in range(2):                      # (twice: the second launch finds the sentinels restored)
                    op.gemm_forward_acc(A, W, acc)
                ref = int_matmul_torch(A, unpack_qweight_torch(W))
            else:
                import qserve_backend.qgemm_w4a8_per_group as op
                if not outs:
                    A = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=gpu, generator=g)
                    W = torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=gpu, generator=g)
                    Z = torch.randint(-128, 128, (K // 128, N), dtype=torch.int8, device=gpu, generator=g)
                    S = torch.randint(-128, 128, (K // 128, N), dtype=torch.int8, device=gpu, generator=g)
                acc = torch.full((M, N), -7, dtype=torch.int32, device=gpu)
                for rep in range(2):
                    op.gemm_forward_acc(A, W, Z, S, acc)
                ref = None
            torch.cuda.synchronize()
            outs.append(acc.clone())
            if ref is not None:
                assert torch.equal(acc.to(torch.int64), ref), f"flags {flags}: int32 accumulators differ from the integer matmul"
    finally:
        _lib.lib.qs_set_gemm_variant(5000)
        _lib.lib.qs_set_gemm_variant(-1)
    assert torch.equal(outs[0], outs[1]), "the two workgroup -> (channel block, K slice) mappings disagree"


def test_rows_beyond_M_untouched_and_empty_batch(gpu):
    import qserve_backend.qgemm_w4a8_per_chn as op
    pr = synth.per_channel_problem(5, 64, 128, seed=9)
    buf = torch.full((8, 64), 123.0, dtype=torch.float16, device=gpu)
    op.gemm_forward_cuda(dev(pr["A"]), dev(pr["qweight"]), dev(pr["wscales"]), dev(pr["ascales"]), dev(pr["w_szs"]),
                         dev(pr["a_ssums"]), buf[:5])
    assert torch.all(buf[5:] == 123.0)
    op.gemm_forward_cuda(dev(pr["A"])[:0], dev(pr["qweight"]), dev(pr["wscales"]), dev(pr["ascales"])[:0],
                         dev(pr["w_szs"]), dev(pr["a_ssums"])[:0], buf[:0])   # M = 0: no-op


def test_bad_arguments_raise(gpu):
    import qserve_backend.qgemm_w4a8_per_chn as op
    pr = synth.per_channel_problem(4, 64, 128, seed=1)
    out = torch.empty((4, 64), dtype=torch.float16, device=gpu)
    with pytest.raises(RuntimeError):     # dtype mismatch (reference: data_ptr<int8_t>() throws)
        op.gemm_forward_cuda(dev(pr["A"]).to(torch.int32), dev(pr["qweight"]), dev(pr["wscales"]), dev(pr["ascales"]),
                             dev(pr["w_szs"]), dev(pr["a_ssums"]), out)
    with pytest.raises(RuntimeError):     # N not a multiple of 64
        op.gemm_forward_cuda(dev(pr["A"]), dev(pr["qweight"]), dev(pr["wscales"]), dev(pr["ascales"]),
                             dev(pr["w_szs"]), dev(pr["a_ssums"]), out[:, :32].contiguous())


LLAMA = [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)]


@pytest.mark.parametrize("N,K", LLAMA)
@pytest.mark.parametrize("M", [64, 128])
def test_full_size_llama_shapes_exact_on_device(gpu, M, N, K):
    """BASELINE.json sizes: the oracle is too slow, so check against an independent exact integer matmul on the
    GPU (torch fp32 slices of the unpacked weights) + a checksum of checksums on the CPU."""
    import qserve_backend.qgemm_w4a8_per_chn as op
    g = torch.Generator(device=gpu).manual_seed(N + K + M)
    W = torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=gpu, generator=g)
    A = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=gpu, generator=g)
    acc = torch.empty((M, N), dtype=torch.int32, device=gpu)
    op.gemm_forward_acc(A, W, acc)
    Q = unpack_qweight_torch(W)
    ref = int_matmul_torch(A, Q)
    assert torch.equal(acc.to(torch.int64), ref)
    # linearity: acc(A, W) summed over tokens == acc(sum-free check) column checksums
    col = (A.to(torch.int64).sum(0)[None, :] * Q.to(torch.int64)).sum(1)
    assert torch.equal(acc.to(torch.int64).sum(0), col)


# per-rank shard shapes of the tensor-parallel path (qserve_amd/tp.py): Llama-3-8B at TP=2/4/8 and configs[3]
# (Qwen1.5-72B TP=8): column-parallel qkv / gate_up (N split), row-parallel o / down (K split)
TP_SHARDS = [(3072, 4096), (4096, 2048), (14336, 4096), (4096, 7168),        # llama3-8b tp2
             (1536, 4096), (4096, 1024), (7168, 4096), (4096, 3584),         # tp4
             (768, 4096), (4096, 512), (3584, 4096), (4096, 1792),           # tp8
             (3072, 8192), (8192, 1024), (6144, 8192), (8192, 3072)]         # qwen1.5-72b tp8


@pytest.mark.parametrize("N,K", TP_SHARDS)
@pytest.mark.parametrize("M", [64, 37, 512])
def test_tp_shard_shapes_exact(gpu, M, N, K):
    """Every per-rank GEMM shape the N>1 bench issues: INT32 accumulator exact vs a device integer matmul, fp16 output
    bit-exact vs the oracle epilogue applied to that (verified) accumulator."""
    import qserve_backend.qgemm_w4a8_per_chn as op
    g = torch.Generator(device=gpu).manual_seed(N * 3 + K + M)
    W = torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=gpu, generator=g)
    A = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=gpu, generator=g)
    acc = torch.empty((M, N), dtype=torch.int32, device=gpu)
    op.gemm_forward_acc(A, W, acc)
    ref = int_matmul_torch(A, unpack_qweight_torch(W))
    assert torch.equal(acc.to(torch.int64), ref)
    r = np.random.default_rng(N + K)
    ws = r.uniform(0.001, 0.01, N).astype(np.float16)
    wz = r.uniform(-0.05, 0.05, N).astype(np.float16)
    sa = r.uniform(0.005, 0.05, M).astype(np.float16)
    ss = r.uniform(-20, 20, M).astype(np.float16)
    out = torch.full((M, N), float("nan"), dtype=torch.float16, device=gpu)
    op.gemm_forward_cuda(A, W, dev(ws), dev(sa), dev(wz), dev(ss), out)
    out_ref = w4a8.epilogue_per_chn(acc.cpu().numpy(), ws, sa, wz, ss)
    assert ulp_diff_f16(out.cpu().numpy(), out_ref).max() == 0


# one layer of every other dense model the reference lists (qserve README model zoo): hidden / heads / kv heads / inter
OTHER_MODELS = {"llama2-7b": (4096, 32, 32, 11008), "llama2-13b": (5120, 40, 40, 13824), "yi-34b": (7168, 56, 8, 20480),
                "llama2-70b": (8192, 64, 8, 28672), "qwen1.5-72b": (8192, 64, 64, 24576), "mistral-7b": (4096, 32, 8, 14336)}
OTHER_GEMMS = sorted({(n, k) for hid, H, Hkv, inter in OTHER_MODELS.values()
                      for n, k in (((H + 2 * Hkv) * 128, hid), (hid, H * 128), (2 * inter, hid), (hid, inter))})


@pytest.mark.parametrize("N,K", OTHER_GEMMS)
@pytest.mark.parametrize("M,per_group", [(6, False), (64, False), (6, True)])
def test_other_model_shapes_exact(gpu, M, N, K, per_group):
    """Every GEMM shape of one layer of the other dense models (K = 11 008, 13 824, 20 480, 24 576, 28 672; N up to
    57 344) through the dispatcher's own choice at decode batch sizes: INT32 accumulators exact against an independent
    integer matmul, fp16 output bit-exact against the oracle epilogue."""
    if per_group:
        if K % 128:
            pytest.skip("g128 needs K % 128 == 0")
        import qserve_backend.qgemm_w4a8_per_group as op
        pr = per_group_problem_torch(M, N, K, gpu, seed=N + K)
        acc = torch.full((M, N), -7, dtype=torch.int32, device=gpu)
        op.gemm_forward_acc(pr["A"], pr["qweight"], pr["s2_zeros"], pr["s2_scales"], acc)
        ref = int_matmul_torch(pr["A"], pr["w8"])
        assert torch.equal(acc.to(torch.int64), ref)
        out = torch.full((M, N), float("nan"), dtype=torch.float16, device=gpu)
        op.gemm_forward_cuda(pr["A"], pr["qweight"], pr["s2_zeros"], pr["s2_scales"], pr["wscales"], pr["ascales"], out)
        out_ref = w4a8.epilogue_per_group(ref.cpu().numpy().astype(np.int32), pr["wscales"].cpu().numpy(),
                                          pr["ascales"].cpu().numpy())
        assert ulp_diff_f16(out.cpu().numpy(), out_ref).max() == 0
        return
    import qserve_backend.qgemm_w4a8_per_chn as op
    g = torch.Generator(device=gpu).manual_seed(N * 5 + K + M)
    W = torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=gpu, generator=g)
    A = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=gpu, generator=g)
    acc = torch.empty((M, N), dtype=torch.int32, device=gpu)
    op.gemm_forward_acc(A, W, acc)
    ref = int_matmul_torch(A, unpack_qweight_torch(W))
    assert torch.equal(acc.to(torch.int64), ref)
    r = np.random.default_rng(N + K)
    ws = r.uniform(0.001, 0.01, N).astype(np.float16)
    wz = r.uniform(-0.05, 0.05, N).astype(np.float16)
    sa = r.uniform(0.005, 0.05, M).astype(np.float16)
    ss = r.uniform(-20, 20, M).astype(np.float16)
    out = torch.full((M, N), float("nan"), dtype=torch.float16, device=gpu)
    op.gemm_forward_cuda(A, W, dev(ws), dev(sa), dev(wz), dev(ss), out)
    out_ref = w4a8.epilogue_per_chn(acc.cpu().numpy(), ws, sa, wz, ss)
    assert ulp_diff_f16(out.cpu().numpy(), out_ref).max() == 0


@pytest.mark.parametrize("M,N,K", [(64, 49152, 1024), (37, 57344, 2048), (16, 65536, 1024)])
def test_very_wide_n(gpu, M, N, K):
    """N >= 49152 (single-GPU 70 B-class gate_up; more than one round of workgroups): per-channel exact vs a device
    integer matmul + oracle epilogue, per-group vs the oracle."""
    import qserve_backend.qgemm_w4a8_per_chn as op
    import qserve_backend.qgemm_w4a8_per_group as opg
    g = torch.Generator(device=gpu).manual_seed(N + K + M)
    W = torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=gpu, generator=g)
    A = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=gpu, generator=g)
    acc = torch.full((M, N), -7, dtype=torch.int32, device=gpu)
    op.gemm_forward_acc(A, W, acc)
    assert torch.equal(acc.to(torch.int64), int_matmul_torch(A, unpack_qweight_torch(W)))
    r = np.random.default_rng(N + K)
    ws = r.uniform(0.001, 0.01, N).astype(np.float16)
    wz = r.uniform(-0.05, 0.05, N).astype(np.float16)
    sa = r.uniform(0.005, 0.05, M).astype(np.float16)
    ss = r.uniform(-20, 20, M).astype(np.float16)
    out = torch.full((M, N), float("nan"), dtype=torch.float16, device=gpu)
    op.gemm_forward_cuda(A, W, dev(ws), dev(sa), dev(wz), dev(ss), out)
    assert ulp_diff_f16(out.cpu().numpy(), w4a8.epilogue_per_chn(acc.cpu().numpy(), ws, sa, wz, ss)).max() == 0
    if M <= 16:      # the numpy per-group oracle at this width takes a few seconds
        pr = synth.per_group_problem(M, N, K, seed=5)
        acc_ref, out_ref = w4a8.gemm_per_group(pr["A"], pr["qweight"], pr["s2_zeros"], pr["s2_scales"], pr["wscales"],
                                               pr["ascales"])
        Ag, Wg, Z, S = dev(pr["A"]), dev(pr["qweight"]), dev(pr["s2_zeros"]), dev(pr["s2_scales"])
        accg = torch.full((M, N), -7, dtype=torch.int32, device=gpu)
        opg.gemm_forward_acc(Ag, Wg, Z, S, accg)
        assert np.array_equal(accg.cpu().numpy(), acc_ref)
        outg = torch.full((M, N), float("nan"), dtype=torch.float16, device=gpu)
        opg.gemm_forward_cuda(Ag, Wg, Z, S, dev(pr["wscales"]), dev(pr["ascales"]), outg)
        assert ulp_diff_f16(outg.cpu().numpy(), out_ref).max() == 0


def test_config1_4096_cubed_per_channel(gpu):
    """configs[0] of BASELINE.json (the reference's CPU-runnable case) on the GPU, exact."""
    import qserve_backend.qgemm_w4a8_per_chn as op
    g = torch.Generator(device=gpu).manual_seed(0)
    W = torch.randint(-128, 128, (4096, 2048), dtype=torch.int8, device=gpu, generator=g)
    A = torch.randint(-127, 128, (4096, 4096), dtype=torch.int8, device=gpu, generator=g)
    acc = torch.empty((4096, 4096), dtype=torch.int32, device=gpu)
    op.gemm_forward_acc(A, W, acc)
    ref = int_matmul_torch(A, unpack_qweight_torch(W))
    assert torch.equal(acc.to(torch.int64), ref)


@pytest.mark.parametrize("M", [128, 2048])     # decode kernel / LDS-tiled kernel (dispatcher's own choice)
def test_per_group_full_size_exact_on_device(gpu, M):
    import qserve_backend.qgemm_w4a8_per_group as op
    N, K = 4096, 4096
    pr = synth.per_group_problem(M, N, K, seed=11)
    A, W, Z, S = dev(pr["A"]), dev(pr["qweight"]), dev(pr["s2_zeros"]), dev(pr["s2_scales"])
    acc = torch.empty((M, N), dtype=torch.int32, device=gpu)
    op.gemm_forward_acc(A, W, Z, S, acc)
    w8 = (pr["q"].astype(np.int16) - np.repeat(pr["z"], 128, 1)) * np.repeat(pr["s2"], 128, 1)
    ref = int_matmul_torch(A, dev(w8.astype(np.int8)))
    assert torch.equal(acc.to(torch.int64), ref)


def test_torch_problem_generator_matches_oracle_packer(gpu):
    """The device-side generator used for the full-size cases packs exactly like the (reference-pinned) oracle packer."""
    pr = per_group_problem_torch(8, 64, 256, gpu, seed=3)
    q = unpack_qweight_torch(pr["qweight"]).cpu().numpy()
    assert np.array_equal(w4a8.pack_qweight(q), pr["qweight"].cpu().numpy())
    assert np.array_equal(pack_qweight_torch(torch.from_numpy(q)).numpy(), pr["qweight"].cpu().numpy())
    w8 = w4a8.dequant_per_group_w8(pr["qweight"].cpu().numpy(), pr["s2_zeros"].cpu().numpy(), pr["s2_scales"].cpu().numpy())
    assert np.array_equal(w8, pr["w8"].cpu().numpy())


@pytest.mark.parametrize("M,N,K", [(128, 28672, 4096), (128, 4096, 14336), (128, 6144, 4096), (128, 4096, 4096)])
def test_config3_per_group_full_shapes(gpu, M, N, K):
    """BASELINE.json configs[2] (Llama-3-8B g128, bs=128): every GEMM of the layer at its real size through the
    dispatcher's own choice (gate_up: ring(4,2,2); down: K-sliced ring) - int32 accumulators exact against an independent
    integer matmul of the level-2 de-quantised weights, fp16 output bit-exact against the oracle epilogue."""
    import qserve_backend.qgemm_w4a8_per_group as op
    pr = per_group_problem_torch(M, N, K, gpu, seed=N + K)
    acc = torch.full((M, N), -7, dtype=torch.int32, device=gpu)
    op.gemm_forward_acc(pr["A"], pr["qweight"], pr["s2_zeros"], pr["s2_scales"], acc)
    ref = int_matmul_torch(pr["A"], pr["w8"])
    assert torch.equal(acc.to(torch.int64), ref)
    out = torch.full((M, N), float("nan"), dtype=torch.float16, device=gpu)
    op.gemm_forward_cuda(pr["A"], pr["qweight"], pr["s2_zeros"], pr["s2_scales"], pr["wscales"], pr["ascales"], out)
    out_ref = w4a8.epilogue_per_group(ref.cpu().numpy().astype(np.int32), pr["wscales"].cpu().numpy(),
                                      pr["ascales"].cpu().numpy())
    assert ulp_diff_f16(out.cpu().numpy(), out_ref).max() == 0


@pytest.mark.parametrize("M", [64, 128])
@pytest.mark.parametrize("N,K", [(28672, 4096), (4096, 14336), (6144, 4096), (4096, 4096)])
def test_config2_per_channel_full_shapes_fp16(gpu, M, N, K):
    """BASELINE.json configs[1] GEMM shapes at decode batch: accumulators exact AND the fp16 epilogue bit-exact at full size
    (the small-shape tests pin the epilogue, the large-shape ones so far only the accumulators)."""
    import qserve_backend.qgemm_w4a8_per_chn as op
    g = torch.Generator(device=gpu).manual_seed(N + K + M)
    W = torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=gpu, generator=g)
    A = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=gpu, generator=g)
    acc = torch.full((M, N), -7, dtype=torch.int32, device=gpu)
    op.gemm_forward_acc(A, W, acc)
    ref = int_matmul_torch(A, unpack_qweight_torch(W))
    assert torch.equal(acc.to(torch.int64), ref)
    r = np.random.default_rng(N + K)
    ws = r.uniform(0.002, 0.02, N).astype(np.float16)
    z = r.integers(0, 16, N)
    wz = (z.astype(np.float16) * ws).astype(np.float16)
    sa = r.uniform(0.005, 0.05, M).astype(np.float16)
    ss = (sa.astype(np.float32) * A.cpu().numpy().astype(np.int64).sum(1).astype(np.float32)).astype(np.float16)
    out = torch.full((M, N), float("nan"), dtype=torch.float16, device=gpu)
    op.gemm_forward_cuda(A, W, dev(ws), dev(sa), dev(wz), dev(ss), out)
    assert ulp_diff_f16(out.cpu().numpy(), w4a8.epilogue_per_chn(ref.cpu().numpy().astype(np.int32), ws, sa, wz, ss)).max() == 0


@pytest.mark.parametrize("N,K", [(28672, 4096), (4096, 14336), (6144, 4096), (4096, 4096)])
def test_epilogue_fma_envelope(gpu, N, K):
    """The reference's per-channel epilogue line (gemm_cuda.cu:586) may be contracted by nvcc (--fmad=true is its default);
    the reference cannot be compiled here, so WHICH rounding sequence its binary uses is unpinned.  This records, at the
    BASELINE configs[1] shapes, how many fp16 outputs differ between the un-contracted evaluation (what the HIP kernels
    compute, bit for bit) and the two legal contractions, and pins the envelope: one fp32 ulp of the larger product (an
    absolute step that is many fp16 ulps of the result where the products cancel)."""
    import json
    import os
    import qserve_backend.qgemm_w4a8_per_chn as op
    M = 64
    g = torch.Generator(device=gpu).manual_seed(N + K + 5)
    W = torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=gpu, generator=g)
    A = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=gpu, generator=g)
    acc = int_matmul_torch(A, unpack_qweight_torch(W)).cpu().numpy().astype(np.int32)
    r = np.random.default_rng(N + K + 5)
    ws = r.uniform(0.002, 0.02, N).astype(np.float16)
    wz = (r.integers(0, 16, N).astype(np.float16) * ws).astype(np.float16)
    sa = r.uniform(0.005, 0.05, M).astype(np.float16)
    ss = (sa.astype(np.float32) * A.cpu().numpy().astype(np.int64).sum(1).astype(np.float32)).astype(np.float16)
    out = torch.full((M, N), float("nan"), dtype=torch.float16, device=gpu)
    op.gemm_forward_cuda(A, W, dev(ws), dev(sa), dev(wz), dev(ss), out)
    o = out.cpu().numpy()
    plain = w4a8.epilogue_per_chn(acc, ws, sa, wz, ss)
    fma = w4a8.epilogue_per_chn(acc, ws, sa, wz, ss, fma=True)
    sub = w4a8.epilogue_per_chn(acc, ws, sa, wz, ss, fma="sub")
    assert ulp_diff_f16(o, plain).max() == 0, "the HIP epilogue is the un-contracted evaluation, bit for bit"
    d_fma, d_sub = ulp_diff_f16(plain, fma), ulp_diff_f16(plain, sub)
    # The envelope is ABSOLUTE, not "one fp16 ulp": a contraction removes one fp32 rounding of t * sa (or of w_sz * a_ssum), i.e.
    # it moves the result by at most one fp32 ulp of that product - but where the two products nearly cancel the result is small
    # and that same absolute step is many fp16 ulps OF THE RESULT (measured: up to ~20 at these shapes).
    t32 = (acc.astype(np.float32) * ws.astype(np.float32)[None, :]).astype(np.float32) * sa.astype(np.float32)[:, None]
    u32 = wz.astype(np.float32)[None, :] * ss.astype(np.float32)[:, None]
    bound = np.spacing(np.maximum(np.abs(t32), np.abs(u32)).astype(np.float32)).astype(np.float64) \
        + np.spacing(np.abs(plain).astype(np.float16)).astype(np.float64)            # + the fp16 output rounding
    for other in (fma, sub):
        assert (np.abs(plain.astype(np.float64) - other.astype(np.float64)) <= bound).all()
    a_fma = np.abs(plain.astype(np.float64) - fma.astype(np.float64))
    a_sub = np.abs(plain.astype(np.float64) - sub.astype(np.float64))
    rec = dict(shape=[M, N, K], outputs=int(o.size), hip_equals="no-contraction evaluation (every output)",
               differ_from_fmaf_t_sa_minus_u=int((d_fma != 0).sum()), differ_from_fmaf_minus_wz_ss_plus_t=int((d_sub != 0).sum()),
               max_fp16_ulps_of_the_result=int(max(d_fma.max(), d_sub.max())), max_abs_difference=float(max(a_fma.max(), a_sub.max())),
               max_abs_output=float(np.abs(plain.astype(np.float64)).max()),
               bound="one fp32 ulp of the larger product + one fp16 ulp of the result (asserted per element)")
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(path, exist_ok=True)
        fn = os.path.join(path, "round4_epilogue_fma_envelope.json")
        allrec = json.load(open(fn)) if os.path.exists(fn) else {}
        allrec[f"{M}x{N}x{K}"] = rec
        json.dump(allrec, open(fn, "w"), indent=1, sort_keys=True)
    except OSError:
        pass


# qs_set_gemm_epilogue(1): the fmaf form of the per-channel epilogue, in EVERY kernel family: (forced variant, M, N, K)
EPI_FMA = [(-1, 64, 28672, 4096), (-1, 64, 4096, 14336), (-1, 64, 6144, 4096), (-1, 64, 4096, 4096),     # config-2 shapes (ring)
           (4221, 40, 256, 2048),                     # ring, K-sliced seam
           (2000, 50, 192, 512), (2001, 300, 384, 512),   # split-K / LDS-pair kernels
           (3001, 300, 512, 384), (3002, 513, 1024, 512), (3003, 300, 512, 384)]   # tiled 256 / 128, four-wave tile


@pytest.mark.parametrize("variant,M,N,K", EPI_FMA)
def test_epilogue_fma_convention_selectable(gpu, variant, M, N, K):
    """The likely CUDA convention of gemm_cuda.cu:586 - fmaf(acc * wscale, ascale, -(w_sz * a_ssum)) - is selectable at run
    time (qs_set_gemm_epilogue) and then equals oracle.w4a8.epilogue_per_chn(fma=True) BIT FOR BIT, kernel family by kernel
    family; the default stays the un-contracted evaluation.  gate_up + silu * mul follows the GEMM's convention as well."""
    import qserve_backend.activation_ops as act
    import qserve_backend.qgemm_w4a8_per_chn as op
    from qserve_amd import _lib, fused as fz
    lib = _lib.lib
    g = torch.Generator(device=gpu).manual_seed(N + K + M)
    W = torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=gpu, generator=g)
    A = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=gpu, generator=g)
    acc = int_matmul_torch(A, unpack_qweight_torch(W)).cpu().numpy().astype(np.int32)
    r = np.random.default_rng(N + K + M)
    ws = r.uniform(0.002, 0.02, N).astype(np.float16)
    wz = (r.integers(0, 16, N).astype(np.float16) * ws).astype(np.float16)
    sa = r.uniform(0.005, 0.05, M).astype(np.float16)
    ss = (sa.astype(np.float32) * A.cpu().numpy().astype(np.int64).sum(1).astype(np.float32)).astype(np.float16)
    want = {0: w4a8.epilogue_per_chn(acc, ws, sa, wz, ss), 1: w4a8.epilogue_per_chn(acc, ws, sa, wz, ss, fma=True)}
    try:
        lib.qs_set_gemm_variant(variant)
        for conv in (1, 0, 1):
            assert lib.qs_set_gemm_epilogue(conv) == 0 and lib.qs_get_gemm_epilogue() == conv
            out = torch.full((M, N), float("nan"), dtype=torch.float16, device=gpu)
            op.gemm_forward_cuda(A, W, dev(ws), dev(sa), dev(wz), dev(ss), out)
            assert np.array_equal(out.cpu().numpy().view(np.uint16), want[conv].view(np.uint16)), (variant, conv)
            if N % 128 == 0 and variant in (-1, 3001, 3002, 3003):     # the activation epilogue: silu_and_mul of THAT fp16 result
                pair = torch.empty((M, N // 2), dtype=torch.float16, device=gpu)
                act.silu_and_mul(pair, out)
                one = torch.full((M, N // 2), float("nan"), dtype=torch.float16, device=gpu)
                tmp = torch.empty((M, N), dtype=torch.float16, device=gpu)
                fz.gemm_silu_and_mul_per_chn(A, W, dev(ws), dev(sa), dev(wz), dev(ss), one, tmp)
                assert torch.equal(one.view(torch.int16), pair.view(torch.int16)), (variant, conv)
        assert lib.qs_set_gemm_epilogue(2) == -1
    finally:
        lib.qs_set_gemm_epilogue(0)
        lib.qs_set_gemm_variant(-1)
    if variant == -1 and (want[0].view(np.uint16) != want[1].view(np.uint16)).sum() == 0:
        pytest.skip("the two conventions agree on every output of this problem")


def test_epilogue_fma_convention_through_the_planes_row_kernel(gpu):
    """K-slice planes: the row kernel that finishes the GEMM applies the SAME convention as the GEMM launch would have."""
    import qserve_backend.qgemm_w4a8_per_chn as op
    from qserve_amd import _lib, fused as fz
    lib = _lib.lib
    M, N, K = 64, 4096, 14336
    g = torch.Generator(device=gpu).manual_seed(77)
    W = torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=gpu, generator=g)
    A = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=gpu, generator=g)
    ws = (torch.rand((N,), device=gpu, generator=g) * 0.018 + 0.002).half()
    wz = (torch.randint(0, 16, (N,), device=gpu, generator=g).half() * ws).half()
    sa = (torch.rand((M,), device=gpu, generator=g) * 0.045 + 0.005).half()
    ss = (sa.float() * A.float().sum(1)).half()
    gamma = (torch.rand((N,), device=gpu, generator=g) + 0.5).half()
    hid0 = (torch.randn((M, N), device=gpu, generator=g) * 2).half()
    ks = fz.gemm_planes_plan(M, N, K, False)
    assert ks >= 1
    try:
        res = {}
        for conv in (0, 1):
            lib.qs_set_gemm_epilogue(conv)
            y = torch.empty((M, N), dtype=torch.float16, device=gpu)
            op.gemm_forward_cuda(A, W, ws, sa, wz, ss, y)
            h1, q1 = hid0.clone(), torch.empty((M, N), dtype=torch.int8, device=gpu)
            s1, m1 = torch.empty((M,), dtype=torch.float16, device=gpu), torch.empty((M,), dtype=torch.float16, device=gpu)
            fz.add_residual_rms_norm_general(q1, h1, y, gamma, s1, 1e-5, input_sum=m1)
            planes = torch.empty((ks, M, N), dtype=torch.int32, device=gpu)
            fz.gemm_planes(A, W, planes)
            h2, q2 = hid0.clone(), torch.empty((M, N), dtype=torch.int8, device=gpu)
            s2, m2 = torch.empty((M,), dtype=torch.float16, device=gpu), torch.empty((M,), dtype=torch.float16, device=gpu)
            fz.add_residual_rms_norm_general_planes(q2, h2, planes, ws, sa, gamma, s2, 1e-5, w_szs=wz, a_ssums=ss, input_sum=m2)
            assert torch.equal(h1.view(torch.int16), h2.view(torch.int16)) and torch.equal(q1, q2), conv
            assert torch.equal(s1.view(torch.int16), s2.view(torch.int16)) and torch.equal(m1.view(torch.int16), m2.view(torch.int16))
            res[conv] = y.clone()
        assert not torch.equal(res[0].view(torch.int16), res[1].view(torch.int16)), "the conventions should differ somewhere here"
    finally:
        lib.qs_set_gemm_epilogue(0)


def _float_reference(A, sa, Wdeq):
    """A_deq . W_deq^T in float64: what the quantised GEMM approximates (SURVEY 8d config 1 "plumbing" reference)."""
    return (A.astype(np.float64) * sa.astype(np.float64)[:, None]) @ Wdeq.T


@pytest.mark.parametrize("name", ["w4a8_pack_per_chn_64x128", "w4a8_pack_per_chn_128x512"])
def test_reference_packed_golden_per_channel_into_hip_gemm(gpu, name):
    """qweight / s1_scales / s1_szeros PRODUCED BY THE REFERENCE's from_linear (tests/golden/*.npz) go straight into the
    HIP GEMM; the expected accumulator is computed from the ORIGINAL nibbles q (never touching the oracle's unpacker)."""
    import os
    import qserve_backend.qgemm_w4a8_per_chn as op
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    N, K = d["q"].shape
    r = np.random.default_rng(N + K)
    M = 37
    A = r.integers(-127, 128, (M, K), dtype=np.int8)
    sa = r.uniform(0.005, 0.05, M).astype(np.float16)
    ss = (sa.astype(np.float32) * A.astype(np.int64).sum(1).astype(np.float32)).astype(np.float16)
    acc_ref = (A.astype(np.int64) @ d["q"].astype(np.int64).T).astype(np.int32)
    acc = torch.full((M, N), -7, dtype=torch.int32, device=gpu)
    op.gemm_forward_acc(dev(A), dev(d["qweight"]), acc)
    assert np.array_equal(acc.cpu().numpy(), acc_ref)
    out = torch.full((M, N), float("nan"), dtype=torch.float16, device=gpu)
    op.gemm_forward_cuda(dev(A), dev(d["qweight"]), dev(d["s1_scales"]), dev(sa), dev(d["s1_szeros"]), dev(ss), out)
    o = out.cpu().numpy()
    assert ulp_diff_f16(o, w4a8.epilogue_per_chn(acc_ref, d["s1_scales"], sa, d["s1_szeros"], ss)).max() == 0
    # and it is the GEMM it claims to be: A_deq . ((q - z) s1)^T up to fp16 rounding of ssum / szero / output
    Wdeq = (d["q"].astype(np.float64) - d["z"].astype(np.float64)[:, None]) * d["s1"].astype(np.float64)[:, None]
    ref = _float_reference(A, sa, Wdeq)
    assert np.abs(o.astype(np.float64) - ref).max() <= 2e-3 * np.abs(ref).max() + 0.05


@pytest.mark.parametrize("name", ["w4a8_pack_per_group_64x256", "w4a8_pack_per_group_128x384"])
def test_reference_packed_golden_per_group_into_hip_gemm(gpu, name):
    import os
    import qserve_backend.qgemm_w4a8_per_group as op
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    N, K = d["q"].shape
    r = np.random.default_rng(N + K)
    M = 21
    A = r.integers(-127, 128, (M, K), dtype=np.int8)
    sa = r.uniform(0.005, 0.05, M).astype(np.float16)
    w8 = (d["q"].astype(np.int64) - np.repeat(d["z"].astype(np.int64), 128, 1)) * np.repeat(d["s2"].astype(np.int64), 128, 1)
    acc_ref = (A.astype(np.int64) @ w8.T).astype(np.int32)
    acc = torch.full((M, N), -7, dtype=torch.int32, device=gpu)
    op.gemm_forward_acc(dev(A), dev(d["qweight"]), dev(d["s2_zeros"]), dev(d["s2_scales"]), acc)
    assert np.array_equal(acc.cpu().numpy(), acc_ref)
    out = torch.full((M, N), float("nan"), dtype=torch.float16, device=gpu)
    op.gemm_forward_cuda(dev(A), dev(d["qweight"]), dev(d["s2_zeros"]), dev(d["s2_scales"]), dev(d["s1_scales"]), dev(sa), out)
    o = out.cpu().numpy()
    assert ulp_diff_f16(o, w4a8.epilogue_per_group(acc_ref, d["s1_scales"], sa)).max() == 0
    ref = _float_reference(A, sa, w8.astype(np.float64) * d["s1"].astype(np.float64)[:, None])
    assert np.abs(o.astype(np.float64) - ref).max() <= 2e-3 * np.abs(ref).max() + 0.05


@pytest.mark.parametrize("kind", ["per_chn", "per_group"])
@pytest.mark.parametrize("M", [64, 2048])
def test_reference_pinned_full_size_checkpoint_into_hip_gemm(gpu, kind, M):
    """Llama-3-8B o_proj-sized tensors whose bytes are SHA-256-verified against the reference's from_linear
    (tests/test_oracle_golden.py::full_size_reference_pinned) through the decode (M=64) and tiled (M=2048) kernels."""
    from test_oracle_golden import full_size_reference_pinned
    i, p = full_size_reference_pinned(kind)
    N, K = i["q"].shape
    g = torch.Generator(device=gpu).manual_seed(M)
    A = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=gpu, generator=g)
    sa = (torch.rand((M,), device=gpu, generator=g) * 0.045 + 0.005).half()
    acc = torch.full((M, N), -7, dtype=torch.int32, device=gpu)
    out = torch.full((M, N), float("nan"), dtype=torch.float16, device=gpu)
    if kind == "per_chn":
        import qserve_backend.qgemm_w4a8_per_chn as op
        ref = int_matmul_torch(A, dev(i["q"].astype(np.int8)))
        op.gemm_forward_acc(A, dev(p["qweight"]), acc)
        assert torch.equal(acc.to(torch.int64), ref)
        ss = (sa.float() * A.sum(dim=1, dtype=torch.int64).float()).half()
        op.gemm_forward_cuda(A, dev(p["qweight"]), dev(p["s1_scales"]), sa, dev(p["s1_szeros"]), ss, out)
        exp = w4a8.epilogue_per_chn(ref.cpu().numpy().astype(np.int32), p["s1_scales"], sa.cpu().numpy(), p["s1_szeros"],
                                    ss.cpu().numpy())
    else:
        import qserve_backend.qgemm_w4a8_per_group as op
        w8 = (i["q"].astype(np.int16) - np.repeat(i["z"].astype(np.int16), 128, 1)) * np.repeat(i["s2"].astype(np.int16), 128, 1)
        ref = int_matmul_torch(A, dev(w8.astype(np.int8)))
        op.gemm_forward_acc(A, dev(p["qweight"]), dev(p["s2_zeros"]), dev(p["s2_scales"]), acc)
        assert torch.equal(acc.to(torch.int64), ref)
        op.gemm_forward_cuda(A, dev(p["qweight"]), dev(p["s2_zeros"]), dev(p["s2_scales"]), dev(p["s1_scales"]), sa, out)
        exp = w4a8.epilogue_per_group(ref.cpu().numpy().astype(np.int32), p["s1_scales"], sa.cpu().numpy())
    assert ulp_diff_f16(out.cpu().numpy(), exp).max() == 0


def test_w8a8_module(gpu):
    import qserve_backend.qgemm_w8a8 as op
    r = np.random.default_rng(3)
    M, N, K = 37, 96, 256
    A = r.integers(-127, 128, (M, K), dtype=np.int8)
    W = r.integers(-127, 128, (N, K), dtype=np.int8)
    ws = r.uniform(0.002, 0.02, N).astype(np.float16)
    sa = r.uniform(0.005, 0.05, M).astype(np.float16)
    _, ref = w4a8.gemm_w8a8(A, W, ws, sa)
    out = torch.empty((M, N), dtype=torch.float16, device=gpu)
    op.w8a8_gemm_forward_cuda(dev(A), dev(W), dev(ws), dev(sa), out)
    assert ulp_diff_f16(out.cpu().numpy(), ref).max() == 0
