#!/usr/bin/env python3
"""Combine the two PMC passes (scripts/gpu_pmc.sh -> gpurun_out/pmc_TAG_{FETCH,WRITE}_SIZE.json) into
profiles/TAG_pmc_traffic.json: HBM bytes per launch of the decode step's dominant kernels.

    python scripts/pmc_traffic.py r04

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE is in KB and tallies 128-B requests of wide coalesced
reads as 64 B -> read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE (KB) is used as is (it matches the GEMM outputs' M*N*2).
The workload (scripts/pmc_workload.py) launches qkv / gate_up / down GEMMs and the decode attention 64 times each.
"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
fe = json.load(open(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_FETCH_SIZE.json")))["kernels"]
wr = json.load(open(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_WRITE_SIZE.json")))["kernels"]
NAMES = {   # bench.py kernel label -> symbol prefix in the counter files
    "w4a8_gemm[gate_up+silu*mul M=64 N=28672 K=4096]": "w4a8_gemm_ring<4, 2, 0, 2, false",
    "w4a8_gemm[qkv M=64 N=6144 K=4096]": "w4a8_gemm_ring<2, 1, 0, 0, false",
    "w4a8_gemm[down M=64 N=4096 K=14336]": "w4a8_gemm_ring<2, 1, 0, 0, true",     # 2 K slices x 2 token blocks (round 4)
    "w4a8_gemm[o M=64 N=4096 K=4096]": "w4a8_gemm_ring<1, 1, 0, 0, false",
    "w4a8_gemm[down as K-slice planes M=64 N=4096 K=14336]": "w4a8_gemm_ring<2, 2, 0, 3, true",   # what the fused step launches
    "add_residual_norm_quant over 4 planes [64 x 4096]": "add_residual_norm_quant_planes_kernel",
    "decode_attention[B=64 H=32 Hkv=8 L=1033]": "decode_attention_mfma_kernel",
}
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, --kernel-trace only), "
                 "scripts/gpu_pmc.sh + scripts/pmc_workload.py, 64 launches per kernel rotating over the 32 layers' "
                 "weights / KV pools; combined by scripts/pmc_traffic.py",
       "correction": "read bytes = 2 * FETCH_SIZE * 1024 (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE (KB) as is",
       "kernels": {}}
for label, sym in NAMES.items():
    kf = [k for k in fe if sym in k]
    kw = [k for k in wr if sym in k]
    if not kf or not kw:
        continue
    f, w = fe[kf[0]], wr[kw[0]]
    rd, wb = int(2 * f["mean"] * 1024), int(w["mean"] * 1024)
    out["kernels"][label] = {"kernel_symbol": sym, "launches": f["launches"], "fetch_size_kb": round(f["mean"], 1),
                             "write_size_kb": round(w["mean"], 1), "hbm_read_bytes": rd, "hbm_write_bytes": wb,
                             "hbm_bytes": rd + wb}
dst = os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json")
json.dump(out, open(dst, "w"), indent=1)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    shutil.copy(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_{c}.json"), os.path.join(ROOT, "profiles", f"{tag}_pmc_{c}.json"))
print(json.dumps({k: v["hbm_bytes"] for k, v in out["kernels"].items()}, indent=1))
