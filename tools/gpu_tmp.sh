mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_full_configs.py tests/test_dispatch.py tests/test_distributed_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "cfg2 or cfg3 or dispatch or batch or dense or distributed" 2>&1 | tail -4
rm -rf gpurun_out/prof_r05_cfg3 gpurun_out/prof_r05_cfg3_2streams
prof() { n=$1; shift; bash tools/profile.sh r05_$n "$@" > gpurun_out/prof_$n.log 2>&1; }
prof cfg3 --workload cfg3 --streams 1; prof cfg3_2streams --workload cfg3
for i in 1 2; do python bench.py --workload cfg3 --steps 100 --warmup 10 2>/dev/null | grep '^{' > gpurun_out/l_cfg3_$i.json; python -c "
import json; d=json.load(open('gpurun_out/l_cfg3_$i.json')); r=d['roofline']; print('cfg3', d['value'], r['launch_us'], r['row_kernel']['launch_us'], r['frac'], r.get('traffic'))"; done
python bench.py --steps 200 --warmup 20 --cpu-iters 0 --pmc 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('cfg2', d['value'], r['launch_us'], r['row_kernel']['launch_us'], r['frac'])"
sed -n 5,8p gpurun_out/prof_r05_cfg3/summary.md
