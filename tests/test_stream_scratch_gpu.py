"""Per-stream scratch (include/qserve_amd.h, qs_stream_scratch_bind): by default every stream of a device shares ONE set of the
library's scratch areas (K-slice slabs, split-KV partials, hand-over rows, argmax keys), so scratch-using launches must not
overlap across streams; a stream that was bound owns a set.  Here two streams - one bound, one on the shared set - run the
scratch-using launches of the decode step CONCURRENTLY on different inputs, unsynchronised, launch after launch: every result must
equal the one the same call produces alone."""
import numpy as np
import pytest
import torch

from _helpers import dev
from oracle import synth, w4a8

pytestmark = pytest.mark.gpu


def _bind(stream):
    import os
    from qserve_amd._lib import lib
    if os.environ.get("QS_TEST_NO_BIND"):      # demonstration only: the concurrent tests below then share one scratch set
        return 0
    return lib.qs_stream_scratch_bind(stream.cuda_stream)


def _unbind(stream):
    from qserve_amd._lib import lib
    return lib.qs_stream_scratch_unbind(stream.cuda_stream)


def test_bind_is_idempotent_bounded_and_refused_while_capturing(gpu):
    from qserve_amd._lib import lib
    streams = [torch.cuda.Stream() for _ in range(8)]
    try:
        for s in streams[:7]:
            assert _bind(s) == 0
        assert _bind(streams[3]) == 0                            # idempotent
        assert _bind(streams[7]) != 0 and b"slots" in lib.qs_last_error()
        assert _unbind(streams[2]) == 0 and _bind(streams[7]) == 0   # a freed slot is handed on
        cap = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
            assert lib.qs_stream_scratch_bind(cap.cuda_stream) != 0
            torch.zeros(4, device=gpu)
        assert b"capturing" in lib.qs_last_error()
    finally:
        for s in streams:
            _unbind(s)


def _gemm_problem(seed, M, N, K):
    pr = synth.per_channel_problem(M, N, K, seed=seed)
    _, ref = w4a8.gemm_per_chn(pr["A"], pr["qweight"], pr["wscales"], pr["ascales"], pr["w_szs"], pr["a_ssums"])
    return [dev(pr[k]) for k in ("A", "qweight", "wscales", "ascales", "w_szs", "a_ssums")], ref


def _replay_together(graphs, streams, times=5):
    """both graphs in flight at once, several times over (eager launches from Python are too far apart to overlap)"""
    torch.cuda.synchronize()
    for _ in range(times):
        for g, s in zip(graphs, streams):
            with torch.cuda.stream(s):
                g.replay()
    torch.cuda.synchronize()


@pytest.mark.parametrize("variant,M,N,K", [(4221, 40, 256, 2048), (-1, 64, 4096, 14336)], ids=["16-workgroup launches", "down_proj"])
def test_k_sliced_gemms_on_two_streams_at_once(gpu, variant, M, N, K):
    """K slices meet in the slot's slabs.  A bound stream and a stream on the shared set each replay a hipGraph of 40 K-sliced
    launches (captured on that stream: the capture finds the bound slot's areas, allocated by the bind) at the same time, on
    different weights and activations - every output bit-exact against the oracle.  (With QS_TEST_NO_BIND=1 the small case
    fails: 16-workgroup launches of two streams do run side by side and both seams use tile 0's slab.)"""
    import qserve_backend.qgemm_w4a8_per_chn as op
    from qserve_amd import _lib
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    assert _bind(s1) == 0
    _lib.lib.qs_set_gemm_variant(variant)
    try:
        probs = [_gemm_problem(21, M, N, K), _gemm_problem(22, M, N, K)]
        outs = [[torch.empty((M, N), dtype=torch.float16, device=gpu) for _ in range(40)] for _ in range(2)]
        graphs = []
        for j, s in enumerate((s0, s1)):
            with torch.cuda.stream(s):
                op.gemm_forward_cuda(*probs[j][0], outs[j][0])         # eager first (the shared set is allocated lazily)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for i in range(40):
                    op.gemm_forward_cuda(*probs[j][0], outs[j][i])
            graphs.append(g)
        _replay_together(graphs, (s0, s1))
        for j in range(2):
            for i in range(40):
                assert np.array_equal(outs[j][i].cpu().numpy().view(np.uint16), probs[j][1].view(np.uint16)), (j, i)
        from qserve_amd._lib import device_status
        assert device_status() == 0
    finally:
        _lib.lib.qs_set_gemm_variant(-1)
        _unbind(s1)


def test_decode_engines_on_two_streams_at_once(gpu):
    """Two tiny decode engines (attention + quant hand-over through the slot's exchange rows, split argmax, the GEMM families of the
    tiny widths), one per stream, stepping concurrently: each reproduces the tokens it generates alone."""
    from qserve_amd.decode import TINY, DecodeEngine

    def engine(seed, batch):
        e = DecodeEngine(TINY, batch=batch, prompt_len=200, max_new=12, group_size=-1, device="cuda:0", seed=seed, fuse_pairs=True)
        e.prefill_cache(200)
        return e

    alone = []
    for seed, batch in ((5, 3), (6, 4)):
        e = engine(seed, batch)
        toks = []
        for _ in range(8):
            e.step()
            toks.append(e.tokens.clone())
        e.check()
        alone.append(toks)
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    assert _bind(s1) == 0
    try:
        e0, e1 = engine(5, 3), engine(6, 4)
        torch.cuda.synchronize()
        got = [[], []]
        for i in range(8):
            with torch.cuda.stream(s0):
                e0.step()
                got[0].append(e0.tokens.clone())
            with torch.cuda.stream(s1):
                e1.step()
                got[1].append(e1.tokens.clone())
        torch.cuda.synchronize()
        for j in range(2):
            for i in range(8):
                assert torch.equal(got[j][i], alone[j][i]), (j, i)
        e0.check()
    finally:
        _unbind(s1)


@pytest.mark.parametrize("int4", [True, False], ids=["kv4", "kv8"])
def test_split_kv_attention_on_two_streams_at_once(gpu, int4):
    """Forced 3-way split-KV (partials travel through the slot's workspace to the merge launch): two streams, different
    sequences, a hipGraph of 30 launches each replayed at the same time - every output equals the launch run alone, bit for bit."""
    import qserve_backend.fused_attention as fa
    from oracle import kvattn
    from qserve_amd import _lib
    from test_attention_gpu import ROPE, DevPools
    B, H, Hkv = 4, 8, 2
    spt = Hkv * (64 if int4 else 128)
    cases = []
    for seed in (31, 32):
        pr = synth.attention_problem(B, H, Hkv, [700, 650, 512, 690], seed=seed)
        pool = DevPools(pr["nblocks"], Hkv, int4, gpu)
        ptrs = pool.pointers(pr["tables"])
        seq = (pr["lengths"] - 1).astype(np.int32)
        hist = np.concatenate(pr["hist"])
        cu = np.concatenate([[0], np.cumsum(seq)]).astype(np.int32)
        pad = fa.compute_padding_offsets(dev(cu), int(seq.max()), hist.shape[0])
        fa.apply_bias_rope_update_kv_cache(dev(hist), dev(seq), pad, ptrs, H, Hkv, int(seq.max()), 64, spt, 128, ROPE, 8192, True,
                                           int4, True)
        buf = dev(np.concatenate([pr["q"].reshape(B, -1), pr["k"].reshape(B, -1), pr["v"].reshape(B, -1)], axis=1))
        q, k, v = buf.split([H * 128, Hkv * 128, Hkv * 128], dim=-1)
        args = (q.reshape(B, H, 128), k.reshape(B, Hkv, 128), v.reshape(B, Hkv, 128), ptrs, dev(pr["lengths"]), None, 8192, 64,
                spt, int(pr["lengths"].max()), 128, ROPE, True, int4, True)
        cases.append((args, pool))
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    assert _bind(s1) == 0
    _lib.lib.qs_set_attention_variant(103)
    try:
        alone = [fa.single_query_attention(*c[0]).clone() for c in cases]
        torch.cuda.synchronize()
        outs, graphs = [[], []], []
        for j, s in enumerate((s0, s1)):
            with torch.cuda.stream(s):
                fa.single_query_attention(*cases[j][0])
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for i in range(30):
                    outs[j].append(fa.single_query_attention(*cases[j][0]))
            graphs.append(g)
        _replay_together(graphs, (s0, s1))
        for j in range(2):
            for i in range(30):
                assert torch.equal(outs[j][i], alone[j]), (j, i)
    finally:
        _lib.lib.qs_set_attention_variant(0)
        _unbind(s1)
