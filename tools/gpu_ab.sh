#!/bin/bash
# A/B bench of engine variants selected by environment variables; usage: gpu_ab.sh "VAR=1 VAR2=x" "..."
mkdir -p gpurun_out
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout 300 python bench.py --steps 100 --warmup 10 --cpu-iters 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('  it/s %.0f  ms/step %.4f  col_us %.1f row_us %.1f'%(d['value'],d['ms_per_step'],r['launch_us'],r['row_kernel_us']))
"
done 2>&1 | tee -a gpurun_out/ab.log
