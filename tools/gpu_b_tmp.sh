mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_gpu_round5.py tests/test_compressed.py -m gpu -q --tb=short -p no:cacheprovider -k "float64_row or monomial or float64_column" > gpurun_out/e_pytest.log 2>&1; tail -12 gpurun_out/e_pytest.log
python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "float64 or f64 or cfg5 or double or multiplane" > gpurun_out/e_pytest64.log 2>&1; tail -5 gpurun_out/e_pytest64.log
for sh in 0 1 0 1; do HGS_ROW_SHIFT64=$sh python bench.py --workload cfg5mraf --steps 20 --warmup 3 --dtype f64 --cpu-iters 0 --pmc 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; e=d.get('engine_default_path') or {}
        print('rowshift64=$sh it/s %8.1f col_us %7.1f row_us %6.1f frac %.3f | default it/s %8.1f'%(d['value'],r['launch_us'],(r.get('row_kernel') or {}).get('launch_us',0),r['frac'],e.get('value',0)))
"; done 2>&1 | tee gpurun_out/e_ab_f64_row.log
