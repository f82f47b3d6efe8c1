#!/bin/bash
# quick bench sweep over grid-size knobs (no CPU baseline)
mkdir -p gpurun_out
for cfg in "1024 768" "2048 768" "2048 4096" "1024 4096" "4096 2048"; do
  set -- $cfg
  echo "ROW_BLOCKS=$1 COL_BLOCKS=$2" 
  HGS_ROW_BLOCKS=$1 HGS_COL_BLOCKS=$2 timeout 300 python bench.py --steps 100 --warmup 10 --cpu-iters 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('  it/s %.0f  ms/step %.4f  col_us %.1f row_us %.1f'%(d['value'],d['ms_per_step'],r['launch_us'],r['row_kernel_us']))
"
done 2>&1 | tee gpurun_out/sweep.log
