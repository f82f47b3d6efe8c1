mkdir -p gpurun_out; export TMPDIR=/tmp
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$1 it/s %8.0f  col_us %6.2f row_us %6.2f frac %.3f'%(d['value'],r['launch_us'],r['row_kernel']['launch_us'],r['frac']))
"; }
for v in 0 1 0 1; do HGS_ROW_RPF=$v python bench.py --steps 200 --warmup 20 --cpu-iters 0 --pmc 0 2>/dev/null | line "cfg2 rpf=$v"; done 2>&1 | tee gpurun_out/k_ab_rpf_single.log
for blk in 640 704 832 896 1024; do HGS_ROW_RPF=1 HGS_ROW_RPF_BLOCKS=$blk python bench.py --steps 200 --warmup 20 --cpu-iters 0 --pmc 0 2>/dev/null | line "cfg2 rpf blocks=$blk"; done 2>&1 | tee -a gpurun_out/k_ab_rpf_single.log
for kb in 48 34 48 34; do HGS_ROW_DENSE_LDS_KB=$kb python bench.py --workload cfg3 --steps 100 --warmup 10 --streams 1 --cpu-iters 0 --pmc 0 2>/dev/null | line "cfg3 row lds_kb=$kb"; done 2>&1 | tee gpurun_out/k_ab_row4.log
for kb in 48 34; do HGS_ROW_PREF=0 HGS_ROW_DENSE_LDS_KB=$kb python bench.py --steps 200 --warmup 20 --cpu-iters 0 --pmc 0 2>/dev/null | line "cfg2 nopref row lds_kb=$kb"; done 2>&1 | tee -a gpurun_out/k_ab_row4.log
