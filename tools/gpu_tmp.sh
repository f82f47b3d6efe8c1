mkdir -p gpurun_out; export TMPDIR=/tmp
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$1 it/s %8.0f  col_us %6.2f row_us %6.2f frac %.3f iter %.3f'%(d['value'],r['launch_us'],r['row_kernel']['launch_us'],r['frac'],r.get('frac_iteration') or 0))
"; }
python -m pytest tests/test_full_configs.py tests/test_dispatch.py tests/test_gpu_round5.py -m gpu -q --tb=short -p no:cacheprovider -k "cfg2 or cfg3 or dispatch or batch or dense or callback" 2>&1 | tail -5
for v in nopark main nopark main; do lib=slmsuite_amd/libhgs_$v.so; [ $v = main ] && lib=slmsuite_amd/libhgs.so
  HGS_LIB=$PWD/$lib python bench.py --steps 200 --warmup 20 --cpu-iters 0 --pmc 0 2>/dev/null | line "cfg2 $v"
  HGS_LIB=$PWD/$lib python bench.py --workload cfg3 --steps 100 --warmup 10 --streams 1 --cpu-iters 0 --pmc 0 2>/dev/null | line "cfg3 $v"
  HGS_LIB=$PWD/$lib python bench.py --workload cfg2dense --steps 100 --warmup 10 --cpu-iters 0 --pmc 0 2>/dev/null | line "cfg2dense $v"
done 2>&1 | tee gpurun_out/j_ab_tile2_park.log
