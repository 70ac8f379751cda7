"""One rank's share of the tensor-parallel decode step on ONE GPU (no process group: the two all-reduces per layer are
no-ops), i.e. the compute side of `bench.py --gpus N`:  weak (batch 64*N) and strong (batch 64) scaling, with the
per-kernel timing of the shard shapes.  The xGMI all-reduce time is what a real N-GPU run adds on top.

    python scripts/bench_tp_shard.py [--tp 2 4 8] [--layers 32]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from qserve_amd import decode as D  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tp", type=int, nargs="+", default=[2, 4, 8])
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--modes", nargs="+", default=["weak", "strong"])
    ap.add_argument("--model", default="llama3-8b", choices=["llama3-8b", "llama2-70b", "qwen1.5-72b"])
    a = ap.parse_args()
    torch.cuda.set_device(0)
    base = {"llama3-8b": D.LLAMA3_8B, "llama2-70b": D.LLAMA2_70B, "qwen1.5-72b": D.QWEN15_72B}[a.model]
    cfg = dict(base, layers=min(a.layers, base["layers"]))
    for tp in a.tp:
        for mode in a.modes:
            B = 64 * tp if mode == "weak" else 64
            eng = D.DecodeEngine(cfg, B, 1024, 512, device="cuda:0", tp_rank=0, tp_world=tp)
            eng.prefill_cache(1024)
            eng.capture()
            for _ in range(4):
                eng.run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                eng.run()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / a.steps * 1e3
            ks = bench.kernel_bench(eng, None, torch, [1025 + 4 + a.steps])
            print(json.dumps({"tp": tp, "mode": mode, "global_batch": B, "ms_per_step_compute_only": round(ms, 3),
                              "tokens_per_s_if_comm_free": round(B / ms * 1e3, 1),
                              "kernels": {k["kernel"]: round(k["us"], 2) for k in ks}}), flush=True)
            del eng
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
