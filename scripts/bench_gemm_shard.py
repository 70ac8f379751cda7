"""Dispatcher vs forced kernel geometries on the tensor-parallel shard shapes (and any M,N,K given on the command line).

    python scripts/bench_gemm_shard.py                 # Llama-3-8B shard shapes at tp 2/4/8, weak (M = 64 tp) + strong (M = 64)
    python scripts/bench_gemm_shard.py 512,3584,4096   # one shape
Variants: -1 dispatcher; 41xy ring (x m-tiles, y units); 3001 / 3002 tiled 256 / 128 tokens; 3000 tiled off; 2001 pair; 4000 ring off.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
import qserve_backend.qgemm_w4a8_per_chn as op  # noqa: E402
import qserve_backend.qgemm_w4a8_per_group as opg  # noqa: E402
from qserve_amd import _lib  # noqa: E402

VARIANTS = [-1, 4141, 4142, 4121, 4122, 4111, 3002, 3001, 2001, 4000]
if os.environ.get("VARIANTS"):       # e.g. VARIANTS=-1,4001,4422 (4100 + 100*(ksplit-1) + 10*mt + wn; 4001 = no K slices)
    VARIANTS = [int(x) for x in os.environ["VARIANTS"].split(",")]


def shard_shapes():
    out = []
    for tp in (2, 4, 8):
        for M in (64 * tp, 64):
            out += [(M, 6144 // tp, 4096), (M, 4096, 4096 // tp), (M, 28672 // tp, 4096), (M, 4096, 14336 // tp)]
    return out


def main():
    torch.cuda.set_device(0)
    if os.environ.get("RING_FLAGS"):     # sticky A/B switches of the ring kernel (qs_set_gemm_variant(5000 + bits), gemm_w4a8_ring.hip)
        _lib.lib.qs_set_gemm_variant(5000 + int(os.environ["RING_FLAGS"]))
    shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or shard_shapes()
    g = torch.Generator(device="cuda").manual_seed(0)
    for M, N, K in shapes:
        nl = int(os.environ["NLW"]) if os.environ.get("NLW") else max(2, min(32, int(600e6 // (N * K // 2))))
        Ws = [torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device="cuda", generator=g) for _ in range(nl)]
        A = torch.randint(-127, 128, (M, K), dtype=torch.int8, device="cuda", generator=g)
        ws = torch.rand(N, device="cuda").half() * 0.01
        wz = torch.rand(N, device="cuda").half() * 0.01
        sa = torch.rand(M, device="cuda").half() * 0.01
        ss = torch.rand(M, device="cuda").half()
        out = torch.empty((M, N), dtype=torch.float16, device="cuda")
        group = os.environ.get("MODE", "chn") == "group"       # MODE=group: per-group (g128) kernels
        if group:
            Z = torch.randint(0, 16, (K // 128, N), dtype=torch.int8, device="cuda", generator=g)
            S = torch.randint(1, 8, (K // 128, N), dtype=torch.int8, device="cuda", generator=g)
            Z = (-Z * S).to(torch.int8)

        def run(i):
            if group:
                opg.gemm_forward_cuda(A, Ws[i % nl], Z, S, ws, sa, out)
            else:
                op.gemm_forward_cuda(A, Ws[i % nl], ws, sa, wz, ss, out)
        res = {}
        for v in VARIANTS:
            _lib.lib.qs_set_gemm_variant(v)
            try:
                res[v] = bench.time_kernel(run, 2 * nl, torch)
            except RuntimeError:
                res[v] = None
                torch.cuda.synchronize()
            finally:
                _lib.lib.qs_set_gemm_variant(-1)
        best = min((t, v) for v, t in res.items() if t is not None and (v != -1 or len(res) == 1))
        print(f"M={M:4d} N={N:5d} K={K:5d}  " + "  ".join(
            f"{v}:{'   n/a' if t is None else f'{t:6.2f}'}" for v, t in res.items()) +
            f"   best {best[1]}" + (f" ({res[-1] / best[0]:.2f}x of dispatcher)" if -1 in res else ""), flush=True)
        del Ws


if __name__ == "__main__":
    main()
