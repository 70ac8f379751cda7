#!/usr/bin/env python3
"""step_timeline.py kernel_trace.csv: the decode step as the GPU ran it (rocprofv3 --kernel-trace of bench.py): per kernel of the
hipGraph-replayed step its duration IN the step and the gap to the next kernel, averaged over the layers of the last steps."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::|_GLOBAL__N_1", "", n)
    m = re.search(r"(w4a8_gemm_\w+<[^>]*>|decode_attention_mfma\w*|add_residual_norm_quant_planes_kernel|add_residual_norm_quant_kernel|quant_kernel|Cijk|argmax_rows\w*|rms_norm_kernel|general_norm_quant_kernel|vectorized_gather|residual_add)", n)
    return m.group(1) if m else n[:40]
names = [short(r["Kernel_Name"]) for r in rows]
st = [int(r["Start_Timestamp"]) for r in rows]
en = [int(r["End_Timestamp"]) for r in rows]
# steps end with the lm_head GEMM (Cijk) followed by argmax: take the last `nsteps` complete steps
heads = [i for i, n in enumerate(names) if n == "Cijk"]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sel = heads[-nsteps - 1:]
dur = collections.defaultdict(list); gap = collections.defaultdict(list); order = []
step_times = []
for a, b in zip(sel[:-1], sel[1:]):
    step_times.append((st[b] - st[a]) / 1e3)
    for i in range(a, b):
        key = names[i]
        if key not in order: order.append(key)
        dur[key].append((en[i] - st[i]) / 1e3)
        gap[key].append((st[i + 1] - en[i]) / 1e3)
print(f"steps (lm_head start to lm_head start): {' '.join(f'{t:.1f}' for t in step_times)} us")
tot_d = tot_g = 0
for k in order:
    n = len(dur[k]) / len(step_times)
    d = sum(dur[k]) / len(dur[k]); g = sum(gap[k]) / len(gap[k])
    tot_d += sum(dur[k]) / len(step_times); tot_g += sum(gap[k]) / len(step_times)
    print(f"{k:48s} x{n:5.1f} per step   duration {d:7.2f} us   gap to next {g:6.2f} us")
print(f"per step: kernels {tot_d:.1f} us + gaps {tot_g:.1f} us")
