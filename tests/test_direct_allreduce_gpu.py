"""The library's direct-access all-reduce (csrc/direct_allreduce.hip, qserve_amd.tp.DirectAllReduce) on a single-GPU
box: several ranks share cuda:0 - in one process (communicators connected by address, one stream per rank) and in two
processes (regions exchanged through HIP IPC, handles gathered over gloo).  The arithmetic contract is exact: every
element is the fp32 sum of the ranks' fp16 addends in rank order, rounded once to fp16, identical on all ranks.
Functional only: the xGMI path itself needs a multi-GPU node (unmeasured)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def expected(parts):
    acc = np.zeros(parts[0].shape, np.float32)
    for p in parts:                                    # rank order, fp32, one rounding
        acc = acc + p.astype(np.float32)
    return acc.astype(np.float16)


@pytest.mark.parametrize("world,numel", [(2, 64 * 4096), (4, 8 * 4096), (8, 64 * 8192), (3, 24 * 8), (8, 64), (2, 16)])
def test_local_ranks_exact_and_identical(gpu, world, numel):
    """All ranks of one process in ONE dispatch (separate launches of a process are resident together only when they
    happen to sit on different hardware queues; one process per rank - the deployment form - is the test below)."""
    from qserve_amd import tp
    comms = tp.DirectAllReduce.local_group(world, numel, device=gpu)
    r = np.random.default_rng(world * 100 + numel)
    try:
        for rnd in range(3):                           # epochs advance on the device: consecutive calls reuse the flags
            parts = [(r.standard_normal(numel) * (1 + rnd)).astype(np.float16) for _ in range(world)]
            for c, p in zip(comms, parts):
                c.input((numel,)).copy_(torch.from_numpy(p))
            tp.DirectAllReduce.all_reduce_group(comms, numel)
            torch.cuda.synchronize()
            want = expected(parts)
            for c in comms:
                assert not c.error()
                assert np.array_equal(c.output((numel,)).cpu().numpy(), want)
    finally:
        for c in comms:
            c.close()


def test_replayed_from_a_graph(gpu):
    """The collective is an ordinary kernel: captured once, replayed with new inputs (the epoch lives on the device)."""
    from qserve_amd import tp
    world, numel = 4, 16 * 4096
    comms = tp.DirectAllReduce.local_group(world, numel, device=gpu)
    r = np.random.default_rng(5)
    try:
        s = torch.cuda.Stream(device=gpu)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            tp.DirectAllReduce.all_reduce_group(comms, numel)
        for rnd in range(3):
            parts = [r.standard_normal(numel).astype(np.float16) for _ in range(world)]
            for c, p in zip(comms, parts):
                c.input((numel,)).copy_(torch.from_numpy(p))
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            for c in comms:
                assert not c.error()
                assert np.array_equal(c.output((numel,)).cpu().numpy(), expected(parts))
    finally:
        for c in comms:
            c.close()


def test_missing_peer_times_out_instead_of_hanging(gpu):
    from qserve_amd import tp
    comms = tp.DirectAllReduce.local_group(2, 4096, device=gpu)
    try:
        comms[0].all_reduce(4096)                      # rank 1 never arrives
        assert comms[0].error()                        # synchronises: the kernel gave up after its bounded wait
        assert not comms[0].error()                    # the flag is cleared by the query
    finally:
        for c in comms:
            c.close()


def test_rejects_bad_sizes(gpu):
    from qserve_amd import tp
    comms = tp.DirectAllReduce.local_group(2, 4096, device=gpu)
    try:
        with pytest.raises(RuntimeError):
            comms[0].all_reduce(4096 + 8)              # larger than the payload
        with pytest.raises(RuntimeError):
            comms[0].all_reduce(24)                    # not a multiple of 8 x world
    finally:
        for c in comms:
            c.close()


WORKER = r"""
import os, sys, json
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from qserve_amd import tp
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
numel = 32 * 4096
comm = tp.DirectAllReduce(numel, device="cuda:0")     # IPC handles gathered over the process group
ok = True
for rnd in range(4):
    parts = [np.random.default_rng(1000 * rnd + r).standard_normal(numel).astype(np.float16) for r in range(world)]
    comm.input((numel,)).copy_(torch.from_numpy(parts[rank]))
    torch.cuda.synchronize()
    dist.barrier()
    comm.all_reduce(numel)
    acc = np.zeros(numel, np.float32)
    for p in parts:
        acc = acc + p.astype(np.float32)
    ok = ok and not comm.error() and np.array_equal(comm.output((numel,)).cpu().numpy(), acc.astype(np.float16))
dist.barrier()
comm.close()
if world != 2:
    os.write(1, (json.dumps({"rank": rank, "ok": bool(ok), "engine_ok": True, "graph_rel": 0.0}) + "\n").encode())
    dist.destroy_process_group()
    sys.exit(0)
# tensor-parallel engine: the row-parallel partials summed by the direct all-reduce (whole step in ONE hipGraph) against
# the same engine over torch.distributed (gloo; with two ranks fp16(a + b) is the same single rounding)
from qserve_amd import decode as D
cfg = dict(D.LLAMA3_8B, layers=2)
B = 8
comm = tp.DirectAllReduce(B * cfg["hidden"], device="cuda:0")
h0 = torch.randn((B, cfg["hidden"]), generator=torch.Generator().manual_seed(1)).half().cuda()
finals = []
for ar in (None, comm):
    e = D.DecodeEngine(cfg, B, 80, 8, device="cuda:0", seed=3, tp_rank=rank, tp_world=world, with_lm_head=False,
                       direct_allreduce=ar)
    e.prefill_cache(80)
    outs = []
    e.hidden.copy_(h0)
    torch.cuda.synchronize()
    dist.barrier()                                     # the ranks enter the step together (the kernel's waits are bounded)
    e.step()
    outs.append(e.final.clone())
    if ar is not None:
        torch.cuda.synchronize()
        dist.barrier()
        g = e.capture()                                # one graph: the collectives are ordinary kernels
        assert e.pieces is None and g is not None
        e.hidden.copy_(h0)
        e.lengths.fill_(81)
        torch.cuda.synchronize()
        dist.barrier()
        e.run()
        torch.cuda.synchronize()
        outs.append(e.final.clone())
    finals.append(outs)
    dist.barrier()
eng_ok = (not comm.error()) and torch.equal(finals[0][0], finals[1][0]) and torch.isfinite(finals[1][0].float()).all().item()
# the captured step started from the same hidden state and cache length: same result as the eager one, except for the
# cache positions the warm-up steps appended (context 81 is re-written with the same token) - compare loosely
rel = ((finals[1][1].float() - finals[1][0].float()).norm() / finals[1][0].float().norm()).item()
dist.barrier()
comm.close()
os.write(1, (json.dumps({"rank": rank, "ok": bool(ok), "engine_ok": bool(eng_ok), "graph_rel": rel}) + "\n").encode())   # one write: lines of different ranks cannot interleave
dist.destroy_process_group()
"""


@pytest.mark.parametrize("nproc", [2, 4])
def test_processes_over_ipc_and_tp_engine(gpu, tmp_path, nproc):
    """One process per rank, all on cuda:0, regions mapped through HIP IPC.  Two ranks: also the tensor-parallel engine
    (direct all-reduce, one hipGraph) against the same engine over torch.distributed."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(29545 + nproc), str(script), ROOT]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    import re
    outs = [json.loads(m) for m in re.findall(r"\{[^{}]*\}", r.stdout)]
    assert len(outs) == nproc and all(o["ok"] for o in outs), r.stdout[-2000:]
    assert all(o["engine_ok"] for o in outs), r.stdout[-2000:]          # direct all-reduce engine == gloo engine, bit for bit
    assert all(o["graph_rel"] < 1e-2 for o in outs), r.stdout[-2000:]     # and the single-graph replay computes the same step
