#!/bin/bash
# 8192-point transform: parity of everything that runs it, then the 8192 workloads
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -k "8192 or cfg5 or tile_rounded or single_pass or every_kind or general or sweep" 2>&1 | tail -4
show='
import sys,json
d=json.loads(sys.stdin.read()); r=d["roofline"]; e=d.get("engine_default_path") or {}
print(sys.argv[1], d["config"]["workload"][:22], d["metric"][:14], round(d["value"],1), "col_us", round(r["launch_us"],1), "frac", round(r["frac"],3), "default", round(e.get("value",0),1), "col", round(e.get("col_kernel_us",0),1), "row", round(e.get("row_kernel_us",0),1))'
for args in "--workload cfg5pad" "--workload cfg5mraf --steps 20 --warmup 3" "--workload cfg5mraf --method GS --steps 20 --warmup 3" "--workload cfg5mraf --dtype f64 --steps 20 --warmup 3" "--workload cfg4grid"; do
timeout 600 python bench.py $args --cpu-iters 0 --pmc 0 2>/dev/null | tail -1 | python -c "$show" "8k"
done
