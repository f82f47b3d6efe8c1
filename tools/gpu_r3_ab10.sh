#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
for r in 1 0 1 0; do
HGS_TILE_RULE=$r WORKLOADS="cfg2 hd" bash tools/gpu_ab.sh main 2>&1 | tail -2 | sed "s/^/rule=$r /"
done
for r in 1 0 1 0; do
for wl in cfg1 small; do
HGS_TILE_RULE=$r timeout 300 python bench.py --workload $wl --steps 400 --warmup 20 --cpu-iters 0 --pmc 0 --no-roofline-pass 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('rule=$r $wl it/s %.0f'%d['value'])"
done
HGS_TILE_RULE=$r timeout 300 python bench.py --workload refbench --cpu-iters 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('rule=$r refbench it/s %.0f'%d['value'], {k:round(v) for k,v in d['methods'].items()})"
done
