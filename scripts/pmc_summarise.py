#!/usr/bin/env python3
"""Average a rocprofv3 counter-collection CSV per kernel: {kernel: {launches, mean counter value}}."""
import csv, json, sys
path, counter = sys.argv[1], sys.argv[2]
acc = {}
with open(path, newline="") as f:
    rd = csv.DictReader(f)
    for row in rd:
        if row.get("Counter_Name") != counter:
            continue
        name = row["Kernel_Name"]
        short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:100]
        a = acc.setdefault(short, [0, 0.0])
        a[0] += 1
        a[1] += float(row["Counter_Value"])
out = {k: {"launches": n, "mean": s / n} for k, (n, s) in acc.items() if n >= 8}
print(json.dumps({"counter": counter, "unit_note": "raw rocprofv3 value (FETCH_SIZE/WRITE_SIZE are in KiB-like units of "
                  "the tool; see DESIGN.md for the gfx950 correction)", "kernels": out}, indent=1))
