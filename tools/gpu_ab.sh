#!/bin/bash
# A/B bench of engine builds: tools/gpu_ab.sh name1 name2 ...  (libhgs_<name>.so from tools/build_variant.sh; "main" = libhgs.so)
mkdir -p gpurun_out
for v in "$@"; do
  lib=slmsuite_amd/libhgs_$v.so; [ "$v" = main ] && lib=slmsuite_amd/libhgs.so
  for wl in ${WORKLOADS:-cfg2}; do
  HGS_LIB=$PWD/$lib timeout 300 python bench.py --steps 200 --warmup 20 --cpu-iters 0 --pmc 0 --workload $wl $BENCH_ARGS 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; e=d.get('engine_default_path') or {}
        print('%-10s %-9s it/s %8.0f  col_us %6.1f row_us %6.1f | default it/s %8.0f col %5.1f row %5.1f'%('$v','$wl',d['value'],r['launch_us'],r['row_kernel']['launch_us'],e.get('value',0),e.get('col_kernel_us') or 0,e.get('row_kernel_us') or 0))
"
  done
done 2>&1 | tee -a gpurun_out/ab.log
