#!/bin/bash
# A/B of engine knobs on the 2048^2 workload (dense kernels); usage: gpu_ab_hd.sh "VAR=1" ...
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout 300 python bench.py --workload hd --steps 200 --warmup 10 --cpu-iters 0 --no-extra-pass 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('  it/s %.0f  col_us %.1f row_us %.1f'%(d['value'],r['launch_us'],r['row_kernel_us']))
"
done
