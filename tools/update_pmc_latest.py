#!/usr/bin/env python
"""Refresh profiles/pmc_latest.json (HBM traffic per launch of the fused column kernel, cfg 2) from a
tools/profile.sh run:  python tools/update_pmc_latest.py gpurun_out/prof_<tag>  profiles/<round>/<name>.json

FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled per the gfx950 half-count correction of
MI355X_MICROARCH.md (HBM / rocprofv3 section); WRITE_SIZE matches the designed store volume
uncorrected (checked against the row kernel, whose stores are exactly GH + phase)."""
import json
import os
import sys

src = sys.argv[1]
kept = sys.argv[2] if len(sys.argv) > 2 else os.path.join(src, "summary.json")
s = json.load(open(os.path.join(src, "summary.json")))
out = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), {kept}", "kernels": {}}
for name, c in s["pmc"].items():
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    f, w = 2.0 * c["FETCH_SIZE"] * 1024, c["WRITE_SIZE"] * 1024
    out["kernels"][name] = {"fetch_bytes_per_launch": f, "write_bytes_per_launch": w, "bytes_per_launch": f + w}
    if name.startswith("col_tile_kernel") or name.startswith("col_fused_kernel"):
        out["kernel"] = name
        out["col_fused_bytes_per_launch"] = f + w
out["workload"] = "cfg2 fp32, 1 hologram"
out["note"] = "FETCH_SIZE x2 (gfx950 half-count correction), WRITE_SIZE as reported"
json.dump(out, open(os.path.join(os.path.dirname(__file__), "..", "profiles", "pmc_latest.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
