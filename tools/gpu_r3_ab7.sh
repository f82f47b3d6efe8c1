#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
HGS_ROW_SHIFT=1 bash tools/profile.sh r03_shift1 > /dev/null 2>&1
HGS_ROW_SHIFT=0 bash tools/profile.sh r03_shift0 > /dev/null 2>&1
for t in shift1 shift0; do echo "== $t"; python - <<PY
import json
d=json.load(open("gpurun_out/prof_r03_$t/summary.json"))
for k,v in d["kernels"].items():
    if "row_kernel<float, 4096, 2" in k: print(k, v)
for k,v in d["pmc"].items():
    if "row_kernel<float, 4096, 2" in k:
        print({c: float("%.4g"%x) for c,x in v.items()})
PY
done
