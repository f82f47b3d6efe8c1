#!/bin/bash
# XCD-aware mapping of the dense per-column kernel: A/B (HGS_COL_XMAP=0/1, read by hgs_create) with measured traffic
mkdir -p gpurun_out; export TMPDIR=/tmp
for x in 1 0; do
for args in "--workload hd --steps 100 --warmup 10" "--workload cfg5mraf --dtype f64 --steps 20 --warmup 3" "--workload cfg5mraf --dtype f64 --method GS --steps 20 --warmup 3"; do
HGS_COL_XMAP=$x timeout 600 python bench.py $args --cpu-iters 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; e=d.get('engine_default_path') or {}
print('xmap=$x', d['config']['workload'][:24], d['metric'][:12], round(d['value'],1), 'col_us', round(r['launch_us'],1), 'frac', round(r['frac'],3), 'traffic/model', round(r['traffic']/r['bytes_per_launch'],3) if r.get('traffic') else None)"
done; done
