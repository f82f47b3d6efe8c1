#!/bin/bash
# 8192-point tile kernel with the next tile staged global -> LDS: parity subset, then the 8192 workloads
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_full_configs.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "cfg5 or 8192 or cfg4_grid or grid_companion" 2>&1 | tail -3
for wl in cfg5pad cfg5mraf cfg4grid; do
timeout 400 python bench.py --workload $wl --steps 40 --warmup 5 --cpu-iters 0 --pmc 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; e=d.get('engine_default_path') or {}; print('$wl', round(d['value'],1), 'col_us', round(r['launch_us'],1), 'frac', round(r['frac'],3), 'row', round(r['row_kernel']['launch_us'],1), 'default', round(e.get('value',0)))"
done
