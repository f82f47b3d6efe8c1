#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
WORKLOADS="cfg2 cfg5pad hd" bash tools/gpu_ab.sh main nolicm nolicm4 2>&1 | tail -12
python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "double or f64 or float64 or cfg5 or cfg3 or general_shape or state_persists" > gpurun_out/ab2_tests.log 2>&1; tail -4 gpurun_out/ab2_tests.log
for dt in f64 f32; do for m in WGS-Leonardo GS; do
  timeout 600 python bench.py --workload cfg5mraf --dtype $dt --method $m --steps 20 --warmup 3 --cpu-iters 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; e=d.get('engine_default_path') or {}; rk=r['row_kernel']
        print('cfg5mraf $dt $m it/s %8.0f col_us %6.1f frac %.3f (traffic/model %.2f) row_us %6.1f | default it/s %8.0f'%(d['value'],r['launch_us'],r['frac'],(r['traffic'] or 0)/r['bytes_per_launch'],rk['launch_us'],e.get('value',0)))
"
done; done 2>&1 | tee gpurun_out/r3_ab2.log
python tools/e2e_timing.py gpurun_out/e2e_timing.json > gpurun_out/e2e.log 2>&1; python - <<'PY'
import json; d=json.load(open("gpurun_out/e2e_timing.json")); print(json.dumps(d["cold_call_breakdown"], indent=0)); print({k:round(v,2) for k,v in d["dense_kernels"].items()}); print({k:round(v,2) for k,v in d["engine_default"].items()})
PY
