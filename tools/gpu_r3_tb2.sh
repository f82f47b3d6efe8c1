#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
show='
import sys,json
d=json.loads(sys.stdin.read()); r=d["roofline"]; e=d.get("engine_default_path") or {}
print(sys.argv[1], round(d["value"],1), "col_us", round(r["launch_us"],1), "row_us", r.get("row_launch_us"), "default", round(e.get("value",0),1), "col", round(e.get("col_kernel_us",0),1), "row", round(e.get("row_kernel_us",0),1))'
for rb in 16384 576 384; do
HGS_ROW_BLOCKS=$rb timeout 600 python bench.py --workload cfg5pad --cpu-iters 0 --pmc 0 2>/dev/null | tail -1 | python -c "$show" "row_blocks=$rb cfg5pad"
done
for cb in 768 256; do
HGS_COL_BLOCKS=$cb timeout 600 python bench.py --workload cfg5mraf --dtype f64 --steps 20 --warmup 3 --cpu-iters 0 --pmc 0 2>/dev/null | tail -1 | python -c "$show" "col_blocks=$cb cfg5 f64"
done
