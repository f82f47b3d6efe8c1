#!/bin/bash
mkdir -p gpurun_out
run() { echo "== $@"; timeout 600 python bench.py --cpu-iters 0 "$@" 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('  it/s %.0f  ms/step %.4f  col_us %.1f row_us %.1f'%(d['value'],d['ms_per_step'],r['launch_us'],r['row_kernel_us']))
    elif 'Error' in l or 'error' in l: print(l.strip())
"; }
run --steps 100 --warmup 10 --batch 8
run --steps 100 --warmup 10 --batch 2
run --steps 50 --warmup 5 --workload cfg5pad
run --steps 100 --warmup 10 --workload small
run --steps 100 --warmup 10 --method WGS-Kim
run --steps 100 --warmup 10 --method GS
