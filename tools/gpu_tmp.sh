mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/h_pytest.log 2>&1; grep -E "^FAILED|passed|failed" gpurun_out/h_pytest.log | tail -20
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$1 it/s %8.0f  col_us %6.2f row_us %6.2f frac %.3f iter %.3f traffic %s %s'%(d['value'],r['launch_us'],r['row_kernel']['launch_us'],r['frac'],r['frac_iteration'],r.get('traffic'),r['kernel'][:50]))
"; }
python bench.py --steps 200 --warmup 20 --cpu-iters 0 2>/dev/null | line "default"
for blk in 704 736 800 832 896; do HGS_TILE2_BLOCKS=$blk python bench.py --steps 200 --warmup 20 --cpu-iters 0 --pmc 0 2>/dev/null | line "tile2 blocks=$blk"; done 2>&1 | tee gpurun_out/h_tile2_blocks.log
