#!/bin/bash
# single-pass MRAF (HGS_MRAF_SPLIT, read by hgs_create): A/B on cfg 5, then the MRAF parity tests
mkdir -p gpurun_out; export TMPDIR=/tmp
for x in 1 0; do
for args in "--workload cfg5mraf --steps 20 --warmup 3" "--workload cfg5mraf --method WGS-Kim --steps 20 --warmup 3"; do
HGS_MRAF_SPLIT=$x timeout 600 python bench.py $args --cpu-iters 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('split=$x', d['config']['workload'][:24], d['metric'][:14], round(d['value'],1), 'ms/it', round(d['ms_per_step'],4), 'col_us', round(r['launch_us'],1), 'frac', round(r['frac'],3), 'traffic', r.get('traffic'), 'model', r.get('bytes_per_launch'))"
done; done
timeout 1500 python -m pytest tests -m gpu -q -x -k "mraf or cfg5" 2>&1 | tail -5
