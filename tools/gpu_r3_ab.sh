#!/bin/bash
# round-3 kernel check: headline, cfg3 and hd with PMC traffic (row kernel traffic / model), quick parity subset
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_full_configs.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "cfg2 or cfg3 or sparse or transform or single_step" > gpurun_out/ab_tests.log 2>&1; tail -3 gpurun_out/ab_tests.log
for wl in cfg2 cfg3 hd cfg5pad; do
  timeout 600 python bench.py --workload $wl --steps 100 --warmup 10 --cpu-iters 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; e=d.get('engine_default_path') or {}; rk=r['row_kernel']
        print('%-8s it/s %8.0f col_us %6.1f (traffic/model %.2f) row_us %6.1f (traffic/model %s) | default it/s %8.0f col %5.1f row %5.1f'%('$wl',d['value'],r['launch_us'],(r['traffic'] or 0)/r['bytes_per_launch'],rk['launch_us'],('%.2f'%(rk['traffic']/rk['bytes_per_launch'])) if rk.get('traffic') else 'n/a',e.get('value',0),e.get('col_kernel_us') or 0,e.get('row_kernel_us') or 0))
"
done 2>&1 | tee gpurun_out/r3_ab.log
HGS_ROW_BLOCKS=16384 timeout 600 python bench.py --workload cfg3 --steps 100 --warmup 10 --cpu-iters 0 --pmc 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('cfg3 HGS_ROW_BLOCKS=16384: it/s %.0f col %.1f row %.1f'%(d['value'],r['launch_us'],r['row_kernel']['launch_us']))
" | tee -a gpurun_out/r3_ab.log
