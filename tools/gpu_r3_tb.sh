#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
show='
import sys,json
d=json.loads(sys.stdin.read()); r=d["roofline"]; e=d.get("engine_default_path") or {}
print(sys.argv[1], round(d["value"],1), "col_us", round(r["launch_us"],1), "default", round(e.get("value",0),1), "col", round(e.get("col_kernel_us",0),1))'
for tb in 512 256 768; do
for args in "--workload cfg5pad" "--workload cfg5mraf --steps 20 --warmup 3" "--workload cfg4grid"; do
HGS_TILE_BLOCKS=$tb timeout 600 python bench.py $args --cpu-iters 0 --pmc 0 2>/dev/null | tail -1 | python -c "$show" "tile_blocks=$tb $args"
done; done
