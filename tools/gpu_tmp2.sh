cd /root/repo
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline') or {}; e=d.get('engine_default_path') or {}
        print('   %-22s it/s %8.0f | default it/s %8.0f col %5.1f row %5.1f'%(d['config']['workload'][:22],d['value'],e.get('value',0),e.get('col_kernel_us') or 0,e.get('row_kernel_us') or 0))"; }
for rep in 1 2; do
for lib in libhgs_nosload.so libhgs.so; do
  echo "== $lib"
  b() { HGS_LIB=$PWD/slmsuite_amd/$lib timeout 300 python bench.py --cpu-iters 0 --pmc 0 "$@" 2>/dev/null | line; }
  b --steps 200 --warmup 20
  b --workload cfg3 --steps 100 --warmup 10
  b --workload hd --steps 100 --warmup 10
  b --workload cfg5pad --steps 50 --warmup 5
  b --workload refbench
  b --workload cfg2 --method WGS-Kim --steps 200 --warmup 20
done; done
