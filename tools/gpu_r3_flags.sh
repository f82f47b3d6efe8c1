#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
show='
import sys,json
d=json.loads(sys.stdin.read()); r=d["roofline"]; e=d.get("engine_default_path") or {}
print(sys.argv[1], d["metric"][:14], round(d["value"],1), "col_us", round(r["launch_us"],1), "default", round(e.get("value",0),1), "col", round(e.get("col_kernel_us",0),1), "row", round(e.get("row_kernel_us",0),1))'
for args in "--workload cfg5mraf --steps 20 --warmup 3" "--workload cfg5mraf --method WGS-Kim --steps 20 --warmup 3"; do
timeout 600 python bench.py $args --cpu-iters 0 --pmc 0 2>/dev/null | tail -1 | python -c "$show" "$args"
done
timeout 1500 python -m pytest tests -m gpu -q -x -k "mraf or cfg5 or tile_rounded or single_pass or batch_with or statistics or sparse" 2>&1 | tail -3
