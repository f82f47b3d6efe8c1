#!/usr/bin/env python
"""Condense rocprofv3 outputs (kernel stats CSV + PMC CSVs) into a small markdown/JSON summary."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    name = name.replace("hgs::", "").replace("void ", "")
    return name[:70]


print(f"# rocprofv3 summary ({out})\n")
stats = glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)
summary = {"kernels": {}, "pmc": {}}
if stats:
    print("## kernel-trace --stats\n")
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---|---|---|---|")
    with open(stats[0]) as f:
        for row in csv.DictReader(f):
            name = short(row.get("Name", ""))
            calls = int(row.get("Calls", 0))
            tot = float(row.get("TotalDurationNs", 0)) / 1e6
            avg = float(row.get("AverageNs", 0)) / 1e3
            pct = row.get("Percentage", "")
            summary["kernels"][name] = {"calls": calls, "total_ms": tot, "avg_us": avg}
            print(f"| `{name}` | {calls} | {tot:.3f} | {avg:.2f} | {pct} |")
    print()
else:
    print("no kernel stats found\n")

agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for d in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
    for path in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = short(row.get("Kernel_Name", ""))
                c = row.get("Counter_Name", "")
                v = float(row.get("Counter_Value", 0) or 0)
                a = agg[k][c]
                a[0] += v
                a[1] += 1
if agg:
    print("## PMC counters (mean per dispatch)\n")
    for k in sorted(agg, key=lambda k: -sum(v[0] for v in agg[k].values())):
        if not any(s in k for s in ("col_", "row_kernel", "cgemm", "sep_", "c_n2f", "c_f2n")):
            continue
        print(f"### `{k}`\n")
        print("| counter | mean / dispatch | dispatches |")
        print("|---|---|---|")
        summary["pmc"][k] = {}
        for c in sorted(agg[k]):
            tot, n = agg[k][c]
            summary["pmc"][k][c] = tot / max(n, 1)
            print(f"| {c} | {tot / max(n, 1):.4g} | {n} |")
        print()
with open(os.path.join(out, "summary.json"), "w") as f:
    json.dump(summary, f, indent=1)
