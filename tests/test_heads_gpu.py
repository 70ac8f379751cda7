"""Row-op HEAD of the ring GEMM (qs_add_norm_quant_w4a8_gemm / qserve_amd.fused.add_norm_quant_gemm, round 5): residual add +
norm + quant runs as the first workgroups of the GEMM launch it feeds.  Bit-identical to the two launches in EVERY output (int8
row, scale, sum, residual stream, GEMM result), launch after launch and under hipGraph replay; whole decode steps equal with the
heads on and off; its waits are bounded."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HID = 4096


def _weights(gpu, N, K, per_group, g):
    w = dict(qweight=torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=gpu, generator=g),
             wscales=(torch.rand((N,), device=gpu, generator=g) * 0.004 + 0.001).half())
    if per_group:
        w["scales_i8"] = torch.randint(1, 9, (K // 128, N), dtype=torch.int8, device=gpu, generator=g)
        zz = torch.randint(0, 16, (K // 128, N), device=gpu, generator=g).to(torch.int16)
        w["zeros"] = (-(zz * w["scales_i8"].to(torch.int16))).to(torch.int8)
    else:
        w["w_szs"] = (torch.randint(0, 16, (N,), device=gpu, generator=g).half() * w["wscales"]).half()
    return w


def _case(gpu, M, N, per_group, kind, silu, seed):
    """inputs of one (row op -> GEMM) pair; kind: 'delta' or ('planes', K_prev) - planes of a previous GEMM [M, HID] x K_prev"""
    from qserve_amd import fused as fz
    g = torch.Generator(device=gpu).manual_seed(seed)
    c = dict(M=M, N=N, per_group=per_group, silu=silu)
    c["hidden"] = (torch.randn((M, HID), device=gpu, generator=g) * 0.7).half()
    c["gamma"] = (torch.rand((HID,), device=gpu, generator=g) + 0.5).half()
    c["w"] = _weights(gpu, N, HID, per_group, g)
    if kind == "delta":
        c["delta"] = (torch.randn((M, HID), device=gpu, generator=g) * 0.5).half()
    else:
        Kp = kind[1]
        prev = _weights(gpu, HID, Kp, per_group, g)
        A = torch.randint(-127, 128, (M, Kp), dtype=torch.int8, device=gpu, generator=g)
        ks = fz.gemm_planes_plan(M, HID, Kp, per_group)
        assert ks in (2, 4), (M, Kp, ks)
        planes = torch.empty((ks, M, HID), dtype=torch.int32, device=gpu)
        if per_group:
            fz.gemm_planes(A, prev["qweight"], planes, prev["zeros"], prev["scales_i8"])
        else:
            fz.gemm_planes(A, prev["qweight"], planes)
        sa = (torch.rand((M,), device=gpu, generator=g) * 0.02 + 0.005).half()
        ss = (sa.float() * A.float().sum(1)).half()
        c.update(planes=planes, prev=prev, p_sa=sa, p_ss=ss)
    return c


def _run(c, gpu, one_call):
    """-> (q, scale, sum, hidden, out) of the pair, through the one-call entry or through the two ops"""
    import qserve_backend.qgemm_w4a8_per_chn as opc
    import qserve_backend.qgemm_w4a8_per_group as opg
    from qserve_amd import fused as fz
    M, N, pg, silu, w = c["M"], c["N"], c["per_group"], c["silu"], c["w"]
    h = c["hidden"].clone()
    q = torch.full((M, HID), 77, dtype=torch.int8, device=gpu)
    # the scale / sum buffers the row op writes ARE the ones the previous GEMM's input was quantised with (the engine aliases them)
    sc = c["p_sa"].clone() if "planes" in c else torch.full((M,), float("nan"), dtype=torch.float16, device=gpu)
    sm = c["p_ss"].clone() if "planes" in c else torch.full((M,), float("nan"), dtype=torch.float16, device=gpu)
    out = torch.full((M, N // 2 if silu else N), float("nan"), dtype=torch.float16, device=gpu)
    tmp = torch.empty((M, N), dtype=torch.float16, device=gpu)
    if one_call:
        kw = dict(input_sum=None if pg else sm, silu_mul=silu, tmp=tmp)
        if "planes" in c:
            kw.update(planes=c["planes"], p_wscales=c["prev"]["wscales"], p_ascales=sc,
                      p_w_szs=None if pg else c["prev"]["w_szs"], p_a_ssums=None if pg else sm)
        else:
            kw.update(delta=c["delta"])
        if pg:
            kw.update(zeros=w["zeros"], scales_i8=w["scales_i8"])
        else:
            kw.update(w_szs=w["w_szs"])
        fz.add_norm_quant_gemm(q, h, c["gamma"], sc, 1e-5, w["qweight"], w["wscales"], out, **kw)
    else:
        if "planes" in c:
            fz.add_residual_rms_norm_general_planes(q, h, c["planes"], c["prev"]["wscales"], sc, c["gamma"], sc, 1e-5,
                                                    w_szs=None if pg else c["prev"]["w_szs"], a_ssums=None if pg else sm,
                                                    input_sum=None if pg else sm)
        else:
            fz.add_residual_rms_norm_general(q, h, c["delta"], c["gamma"], sc, 1e-5, input_sum=None if pg else sm)
        if silu:
            if pg:
                fz.gemm_silu_and_mul_per_group(q, w["qweight"], w["zeros"], w["scales_i8"], w["wscales"], sc, out, tmp)
            else:
                fz.gemm_silu_and_mul_per_chn(q, w["qweight"], w["wscales"], sc, w["w_szs"], sm, out, tmp)
        elif pg:
            opg.gemm_forward_cuda(q, w["qweight"], w["zeros"], w["scales_i8"], w["wscales"], sc, out)
        else:
            opc.gemm_forward_cuda(q, w["qweight"], w["wscales"], sc, w["w_szs"], sm, out)
    torch.cuda.synchronize()
    return q, sc, (None if pg else sm), h, out


def _same(a, b):
    for x, y in zip(a, b):
        if x is None:
            continue
        assert torch.equal(x.view(torch.int16) if x.dtype == torch.float16 else x, y.view(torch.int16) if y.dtype == torch.float16 else y)


CASES = [(64, 6144, "delta", False), (64, 28672, "delta", True), (64, 6144, ("planes", 14336), False), (64, 28672, ("planes", 4096), True),
         (33, 6144, ("planes", 14336), False), (63, 28672, "delta", True), (1, 6144, "delta", False), (40, 6144, ("planes", 4096), False)]


@pytest.mark.parametrize("per_group", [False, True], ids=["per_channel", "g128"])
@pytest.mark.parametrize("M,N,kind,silu", CASES)
def test_head_launch_equals_the_two_launches(gpu, M, N, kind, silu, per_group):
    from qserve_amd import _lib
    lib = _lib.lib
    assert lib.qs_device_reset() == 0
    c = _case(gpu, M, N, per_group, kind, silu, seed=M + N)
    want = _run(c, gpu, one_call=False)
    n0 = lib.qs_debug_head_launch_count()
    for _ in range(4):                                   # launch after launch: the epoch moves on, nothing stale is mistaken
        _same(_run(c, gpu, one_call=True), want)
    taken = lib.qs_debug_head_launch_count() - n0
    assert taken in (0, 4)
    if M > 32 and N in (6144, 28672):
        assert taken == 4, "these shapes have a head instantiation: the one-launch form must be the one that ran"
    try:
        lib.qs_set_gemm_variant(4005)                    # the entry's own two-launch form
        _same(_run(c, gpu, one_call=True), want)
        assert lib.qs_debug_head_launch_count() - n0 == taken
    finally:
        lib.qs_set_gemm_variant(-1)
    assert _lib.device_status() == 0


@pytest.mark.parametrize("per_group", [False, True], ids=["per_channel", "g128"])
def test_head_launches_replay_from_a_hipgraph(gpu, per_group):
    """qkv-shaped and gate_up-shaped head launches chained in one hipGraph (what the decode step does), replayed: every replay
    reproduces the eager result (the epoch protocol needs no host-side state between replays)."""
    from qserve_amd import _lib
    from qserve_amd import fused as fz
    c1 = _case(gpu, 64, 6144, per_group, ("planes", 14336), False, seed=5)
    c2 = _case(gpu, 64, 28672, per_group, "delta", True, seed=6)
    want1, want2 = _run(c1, gpu, False), _run(c2, gpu, False)
    _run(c1, gpu, True)                                   # eager first: scratch allocated outside the capture
    bufs = []

    def issue(c):
        M, N, pg, silu, w = c["M"], c["N"], c["per_group"], c["silu"], c["w"]
        h = c["hidden"].clone()
        q = torch.empty((M, HID), dtype=torch.int8, device=gpu)
        sc = c["p_sa"].clone() if "planes" in c else torch.empty((M,), dtype=torch.float16, device=gpu)
        sm = c["p_ss"].clone() if "planes" in c else torch.empty((M,), dtype=torch.float16, device=gpu)
        out = torch.empty((M, N // 2 if silu else N), dtype=torch.float16, device=gpu)
        tmp = torch.empty((M, N), dtype=torch.float16, device=gpu)
        src = dict(h=c["hidden"], sc=c.get("p_sa"), sm=c.get("p_ss"))
        bufs.append((c, h, q, sc, sm, out, src))

        def go():
            h.copy_(src["h"])
            if src["sc"] is not None:
                sc.copy_(src["sc"])
                sm.copy_(src["sm"])
            kw = dict(input_sum=None if pg else sm, silu_mul=silu, tmp=tmp)
            if "planes" in c:
                kw.update(planes=c["planes"], p_wscales=c["prev"]["wscales"], p_ascales=sc,
                          p_w_szs=None if pg else c["prev"]["w_szs"], p_a_ssums=None if pg else sm)
            else:
                kw.update(delta=c["delta"])
            kw.update(dict(zeros=w["zeros"], scales_i8=w["scales_i8"]) if pg else dict(w_szs=w["w_szs"]))
            fz.add_norm_quant_gemm(q, h, c["gamma"], sc, 1e-5, w["qweight"], w["wscales"], out, **kw)
        return go

    g1, g2 = issue(c1), issue(c2)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        g1(), g2()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            for _ in range(3):
                g1()
                g2()
    for _ in range(5):
        graph.replay()
        torch.cuda.synchronize()
        for (c, h, q, sc, sm, out, _), want in zip(bufs, (want1, want2)):
            _same((q, sc, None if c["per_group"] else sm, h, out), want)
    assert _lib.device_status() == 0


def test_decode_steps_with_heads_equal_steps_without(gpu):
    """Llama-3-8B widths, 3 layers, bs = 64: the fused engine with the row-op heads == without them == op by op, token for token,
    eager and from the hipGraph."""
    from qserve_amd import _lib
    from qserve_amd.decode import LLAMA3_8B, DecodeEngine
    cfg = dict(LLAMA3_8B, layers=3, vocab=4096)
    for gs in (-1, 128):
        outs = []
        n0 = _lib.lib.qs_debug_head_launch_count()
        for fuse, heads in ((False, False), (True, False), (True, True)):
            eng = DecodeEngine(cfg, batch=64, prompt_len=200, max_new=12, group_size=gs, device="cuda:0", seed=3, fuse_pairs=fuse,
                               heads=heads)
            assert eng.heads == heads
            eng.prefill_cache(200)
            toks = []
            for _ in range(2):
                eng.step()
                toks.append(eng.tokens.clone())
            eng.capture()
            for _ in range(4):
                eng.run()
                toks.append(eng.tokens.clone())
            torch.cuda.synchronize()
            eng.check()
            outs.append((eng.hidden.clone(), eng.final.clone(), torch.stack(toks)))
        assert _lib.lib.qs_debug_head_launch_count() > n0, "no head launch was taken at the production widths"
        for other in outs[1:]:
            for a, b in zip(outs[0], other):
                assert torch.equal(a, b)


def test_head_wait_is_bounded(gpu):
    from qserve_amd import _lib
    lib = _lib.lib
    c = _case(gpu, 64, 6144, False, "delta", False, seed=9)
    want = _run(c, gpu, one_call=False)
    try:
        assert lib.qs_device_reset() == 0
        _same(_run(c, gpu, one_call=True), want)
        assert lib.qs_debug_inject_fault(4) == 0         # row 0's flag is never written: the GEMM workgroups give up
        bad = _run(c, gpu, one_call=True)
        assert _lib.device_status() & 4
        assert lib.qs_device_reset() == 0 and _lib.device_status() == 0
        for _ in range(3):
            _same(_run(c, gpu, one_call=True), want)
        assert _lib.device_status() == 0
        del bad
    finally:
        lib.qs_debug_inject_fault(0)
        lib.qs_device_reset()
