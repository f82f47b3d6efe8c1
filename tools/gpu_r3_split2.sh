#!/bin/bash
# single-pass MRAF: HGS_MRAF_SPLIT = 1 (row kernel sums the weight-norm partials), 2 (two small launches), 0 (two passes)
mkdir -p gpurun_out; export TMPDIR=/tmp
show='
import sys,json
d=json.loads(sys.stdin.read()); r=d["roofline"]; e=d.get("engine_default_path") or {}
k=r.get("kernels_us") or {}
print(sys.argv[1], d["metric"][:14], round(d["value"],1), "col_us", round(r["launch_us"],1), "row_us", round(r.get("row_launch_us") or 0,1), "default", round(e.get("value",0),1), "col", round(e.get("col_kernel_us",0),1), "row", round(e.get("row_kernel_us",0),1))'
for x in 1 2 0; do
for args in "--workload cfg5mraf --steps 20 --warmup 3" "--workload cfg5mraf --method WGS-Kim --steps 20 --warmup 3"; do
HGS_MRAF_SPLIT=$x timeout 600 python bench.py $args --cpu-iters 0 --pmc 0 2>/dev/null | tail -1 | python -c "$show" "split=$x"
done; done
timeout 1500 python -m pytest tests -m gpu -q -x -k "mraf or cfg5 or tile_rounded or single_pass" 2>&1 | tail -3
