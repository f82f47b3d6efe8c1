mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "kim or Kim or fixed or fuzz or dispatch" 2>&1 | tail -4
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; e=d.get('engine_default_path') or {}
        print('$1 it/s %8.0f col_us %6.2f row_us %6.2f | default it/s %8.0f col %5.2f row %5.2f'%(d['value'],r['launch_us'],r['row_kernel']['launch_us'],e.get('value',0),e.get('col_kernel_us') or 0,e.get('row_kernel_us') or 0))
"; }
for v in nopf main nopf main; do lib=slmsuite_amd/libhgs_$v.so; [ $v = main ] && lib=slmsuite_amd/libhgs.so
  HGS_LIB=$PWD/$lib python bench.py --method WGS-Kim --steps 200 --warmup 40 --cpu-iters 0 --pmc 0 2>/dev/null | line "cfg2 Kim $v"
  HGS_LIB=$PWD/$lib python bench.py --workload hd --method WGS-Kim --steps 200 --warmup 40 --cpu-iters 0 --pmc 0 2>/dev/null | line "hd Kim $v"
  HGS_LIB=$PWD/$lib python bench.py --workload cfg1 --method WGS-Kim --steps 400 --warmup 40 --cpu-iters 0 --pmc 0 2>/dev/null | line "cfg1 Kim $v"
done 2>&1 | tee gpurun_out/m_ab_pf_ahead.log
