#!/bin/bash
# tile-rounded column lists on the tile-resident kernel (HGS_TILE_LIST, read by hgs_create): the engine-default path of cfg 5
mkdir -p gpurun_out; export TMPDIR=/tmp
show='
import sys,json
d=json.loads(sys.stdin.read()); r=d["roofline"]; e=d.get("engine_default_path") or {}
print(sys.argv[1], d["config"]["workload"][:22], d["metric"][:14], round(d["value"],1), "col_us", round(r["launch_us"],1), "frac", round(r["frac"],3), "default_path", e)'
for x in 1 0; do
for args in "--workload cfg5mraf --steps 20 --warmup 3" "--workload cfg5mraf --method GS --steps 20 --warmup 3" "--workload cfg5mraf --method WGS-Kim --steps 20 --warmup 3"; do
HGS_TILE_LIST=$x timeout 600 python bench.py $args --cpu-iters 0 2>/dev/null | tail -1 | python -c "$show" "tile_list=$x"
done; done
timeout 600 python bench.py --cpu-iters 0 2>/dev/null | tail -1 | python -c "$show" headline
timeout 600 python bench.py --workload cfg2dense --cpu-iters 0 2>/dev/null | tail -1 | python -c "$show" cfg2dense
timeout 2000 python -m pytest tests -m gpu -q -x 2>&1 | tail -5
