#!/bin/bash
# grid-size sweeps of the two hot kernels (developer overrides read once at hgs_create)
run() { env "$@" timeout 300 python bench.py --steps 200 --warmup 20 --cpu-iters 0 --pmc 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; e=d.get('engine_default_path') or {}
        print('%-40s it/s %8.0f  col_us %6.1f row_us %6.1f | default it/s %8.0f col %5.1f row %5.1f'%('$*',d['value'],r['launch_us'],r['row_kernel']['launch_us'],e.get('value',0),e.get('col_kernel_us') or 0,e.get('row_kernel_us') or 0))
"; }
for tb in 256 384 512 768 1024; do run HGS_TILE_BLOCKS=$tb; done
for rb in 576 768 1024 1152 2304; do run HGS_ROW_BLOCKS=$rb; done
run HGS_ROW_XCD=0
