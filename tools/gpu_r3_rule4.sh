#!/bin/bash
# single-pass MRAF with the rule compiled in (col_tile_kernel RULE 4; HGS_TILE_RULE=0 = generic RULE 3)
mkdir -p gpurun_out; export TMPDIR=/tmp
show='
import sys,json
d=json.loads(sys.stdin.read()); r=d["roofline"]; e=d.get("engine_default_path") or {}
print(sys.argv[1], d["metric"][:14], round(d["value"],1), "col_us", round(r["launch_us"],1), "default", round(e.get("value",0),1), "col", round(e.get("col_kernel_us",0),1), "row", round(e.get("row_kernel_us",0),1))'
for x in 1 0 1 0; do
for args in "--workload cfg5mraf --steps 20 --warmup 3"; do
HGS_TILE_RULE=$x timeout 600 python bench.py $args --cpu-iters 0 --pmc 0 2>/dev/null | tail -1 | python -c "$show" "rule=$x"
done; done
HGS_TILE_RULE=1 timeout 600 python bench.py --workload cfg5mraf --method WGS-Kim --steps 20 --warmup 3 --cpu-iters 0 --pmc 0 2>/dev/null | tail -1 | python -c "$show" "kim rule=1"
HGS_TILE_RULE=0 timeout 600 python bench.py --workload cfg5mraf --method WGS-Kim --steps 20 --warmup 3 --cpu-iters 0 --pmc 0 2>/dev/null | tail -1 | python -c "$show" "kim rule=0"
timeout 1500 python -m pytest tests -m gpu -q -x -k "mraf or cfg5 or tile_rounded or single_pass or batch_with" 2>&1 | tail -3
