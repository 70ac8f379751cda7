"""Tensor-parallel sharding of the packed weights + the all-reduce of row-parallel partials,
exercised with world_size = 2 over gloo on CPU.  The per-rank GEMM is the oracle here (tests may use it);
on the GPU the same shards feed qserve_backend.*.gemm_forward_cuda."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import synth, w4a8
from qserve_amd import tp

WORLD = 2


def _worker(rank, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        # ---------------- per-channel: column-parallel then row-parallel ----------------
        pc = synth.per_channel_problem(9, 128, 512, seed=5)
        acc_full, out_full = w4a8.gemm_per_chn(pc["A"], pc["qweight"], pc["wscales"], pc["ascales"], pc["w_szs"], pc["a_ssums"])
        qw = torch.from_numpy(pc["qweight"])
        qw_c, (ws_c, wz_c), _ = tp.shard_column_parallel(qw, [torch.from_numpy(pc["wscales"]), torch.from_numpy(pc["w_szs"])],
                                                         (), rank, WORLD)
        acc_c, out_c = w4a8.gemm_per_chn(pc["A"], qw_c.numpy(), ws_c.numpy(), pc["ascales"], wz_c.numpy(), pc["a_ssums"])
        n0, n1 = rank * 64, (rank + 1) * 64
        assert np.array_equal(acc_c, acc_full[:, n0:n1]) and np.array_equal(out_c.view(np.uint16), out_full[:, n0:n1].view(np.uint16))

        qw_r, _ = tp.shard_row_parallel(qw, (), rank, WORLD)
        k0, k1 = rank * 256, (rank + 1) * 256
        acc_r = w4a8.gemm_per_chn_acc(pc["A"][:, k0:k1], qw_r.numpy())
        t = torch.from_numpy(acc_r.astype(np.int64))
        tp.all_reduce_sum_(t)                                   # the one collective of the path
        assert np.array_equal(t.numpy(), acc_full.astype(np.int64)), "row-parallel partial accumulators do not add up"
        # the reference's generic [:, start:end] slice (weight_utils.py:192-220) would be wrong for packed weights
        naive = qw[:, rank * 128:(rank + 1) * 128].contiguous().numpy()
        assert not np.array_equal(w4a8.gemm_per_chn_acc(pc["A"][:, k0:k1], naive), acc_r)
        # fp16 path: each rank uses its own activation scale / partial sum; the fp32 partial outputs add up to the
        # un-sharded result up to fp16 rounding of the per-shard sums
        ssum_r = (pc["ascales"].astype(np.float32) * pc["A"][:, k0:k1].astype(np.int64).sum(1).astype(np.float32)).astype(np.float16)
        part = w4a8.epilogue_per_chn(acc_r, pc["wscales"], pc["ascales"], pc["w_szs"], ssum_r).astype(np.float32)
        tt = torch.from_numpy(part)
        rp = tp.RowParallelLinear(lambda x, a, s, out: None)     # gemm already done above
        rp(None, None, None, tt)
        assert np.allclose(tt.numpy(), out_full.astype(np.float32), rtol=2e-3, atol=0.1)

        # ---------------- per-group: row-parallel keeps 128-groups intact ----------------
        pg = synth.per_group_problem(5, 64, 512, seed=6)
        acc_g = w4a8.gemm_per_group_acc(pg["A"], pg["qweight"], pg["s2_zeros"], pg["s2_scales"])
        qw_g, (z_g, s_g) = tp.shard_row_parallel(torch.from_numpy(pg["qweight"]),
                                                 (torch.from_numpy(pg["s2_zeros"]), torch.from_numpy(pg["s2_scales"])),
                                                 rank, WORLD, group_size=128)
        acc_gr = w4a8.gemm_per_group_acc(pg["A"][:, k0:k1], qw_g.numpy(), z_g.numpy(), s_g.numpy())
        t = torch.from_numpy(acc_gr.astype(np.int64))
        tp.all_reduce_sum_(t)
        assert np.array_equal(t.numpy(), acc_g.astype(np.int64))
        # column-parallel per-group: s2 tensors slice on dim 1 at multiples of 32
        qw_gc, _, (z_gc, s_gc) = tp.shard_column_parallel(torch.from_numpy(pg["qweight"]), (),
                                                          (torch.from_numpy(pg["s2_zeros"]), torch.from_numpy(pg["s2_scales"])),
                                                          rank, WORLD)
        acc_gc = w4a8.gemm_per_group_acc(pg["A"], qw_gc.numpy(), z_gc.numpy(), s_gc.numpy())
        assert np.array_equal(acc_gc, acc_g[:, rank * 32:(rank + 1) * 32])
        ret[rank] = "ok"
    except Exception as e:  # noqa: BLE001
        ret[rank] = f"{type(e).__name__}: {e}"
    finally:
        dist.destroy_process_group()


def test_tp2_sharding_and_allreduce_gloo():
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(port, ret), nprocs=WORLD, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}, dict(ret)


def test_shard_shapes_llama3_tp8():
    # Llama-3-8B, TP=8: column shards of qkv / gate_up, row shards of o / down stay on tile boundaries
    q_proj = torch.zeros((4096, 2048), dtype=torch.int8)
    assert tp.shard_column_parallel(q_proj, (), (), 3, 8)[0].shape == (512, 2048)
    # the fused [q; k; v] tensor: every rank must get 4 query heads + 1 k head + 1 v head, not a contiguous slice
    qkv = torch.arange(6144, dtype=torch.int32)[:, None].expand(6144, 4).contiguous()
    rows = tp.shard_fused_column_parallel(qkv, [4096, 1024, 1024], (), (), 3, 8)[0][:, 0]
    assert rows.tolist() == list(range(1536, 2048)) + list(range(4096 + 384, 4096 + 512)) + list(range(5120 + 384, 5120 + 512))
    # fewer KV heads than ranks: KV heads replicated over world / Hkv ranks (llama_w4a8_unpad.py:118-127)
    rows = tp.shard_fused_column_parallel(qkv, [4096, 1024, 1024], (), (), 5, 16, [1, 2, 2])[0][:, 0]
    assert rows.tolist() == list(range(5 * 256, 6 * 256)) + list(range(4096 + 2 * 128, 4096 + 3 * 128)) + \
        list(range(5120 + 2 * 128, 5120 + 3 * 128))
    down = torch.zeros((4096, 7168), dtype=torch.int8)
    qw, _ = tp.shard_row_parallel(down, (), 7, 8, group_size=128)
    assert qw.shape == (4096, 896)      # K/8 = 1792 -> 896 bytes per row, 14 groups of 128
    with pytest.raises(AssertionError):
        tp.shard_row_parallel(torch.zeros((64, 16 * 3), dtype=torch.int8), (), 0, 2)   # 3 tiles do not split in 2
