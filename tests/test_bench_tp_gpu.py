"""The N > 1 flow of bench.py (process group, tensor-parallel engine, all-reduces, max-over-ranks timing, one JSON line)
executed for real on a single-GPU box: two ranks share cuda:0 and talk gloo (QS_DIST_BACKEND / QS_DIST_DEVICE, see
bench.py).  Functional only - the numbers mean nothing."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, port):
    env = dict(os.environ, QS_DIST_BACKEND="gloo", QS_DIST_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--model", "tiny",
           "--batch", "4", "--prompt-len", "96", "--max-new", "48", "--steps", "4", "--warmup", "2", "--no-prefill"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0]), r.stderr


def test_bench_two_ranks_eager(gpu):
    out, _ = _run(["--no-graph", "--collective", "rccl"], 29541)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["global_batch"] == 8
    assert out["config"]["parallelism"] == "tp2" and out["value"] > 0 and out["config"]["hipgraph"] is False


def test_bench_two_ranks_piecewise_graphs(gpu):
    """Default N > 1 mode: one hipGraph per segment between the all-reduces, the collectives issued eagerly - works with
    any backend (here gloo, which could never be captured)."""
    out, err = _run(["--collective", "rccl"], 29542)
    assert out["n_gpus"] == 2 and out["value"] > 0
    assert str(out["config"]["hipgraph"]).startswith("piecewise"), err[-2000:]
    assert out["config"]["step_collectives"].startswith("torch.distributed")


def test_bench_two_ranks_direct_allreduce_single_graph(gpu):
    """`--direct-allreduce`: the library's own collective (an ordinary kernel) - the whole tensor-parallel step is ONE
    hipGraph; the line carries the collective's latency next to the process group's."""
    out, err = _run(["--direct-allreduce"], 29543)
    assert out["n_gpus"] == 2 and out["value"] > 0
    assert out["config"]["hipgraph"] is True, err[-2000:]
    assert out["config"]["direct_all_reduce_us"] > 0 and out["config"]["direct_all_reduce_timeouts"] is False


def test_bench_two_ranks_default_picks_a_checked_collective(gpu):
    """Default (`--collective auto`): the direct all-reduce is used only after its start-up self-check passed on this
    machine; either way the line says which collective ran and why."""
    out, err = _run([], 29544)
    assert out["n_gpus"] == 2 and out["value"] > 0
    note = out["config"]["step_collectives"]
    assert note.startswith("library direct-access all-reduce") or note.startswith("torch.distributed ("), note
    if note.startswith("library"):
        assert out["config"]["hipgraph"] is True and out["config"]["direct_all_reduce_timeouts"] is False


def test_bench_two_ranks_full_graph_request_over_an_uncapturable_backend(gpu):
    """`--tp-full-graph` over a host-staged backend (gloo): capturing it could only fail, and a failed capture leaves the
    HIP context unusable - bench.py refuses up front, says so and runs the default piecewise scheme."""
    out, err = _run(["--collective", "rccl", "--tp-full-graph"], 29545)
    assert out["n_gpus"] == 2 and out["value"] > 0
    assert str(out["config"]["hipgraph"]).startswith("piecewise")
    assert "--tp-full-graph ignored" in err
