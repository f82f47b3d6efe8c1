#!/bin/bash
# Round-6 GPU batches: tools/gpu_r6.sh STEP [-- STEP ...]   (the steps of tools/gpu_r5.sh plus:)
#   base                  bench lines of the workloads this round works on -> gpurun_out/r6_base.jsonl
mkdir -p gpurun_out
export TMPDIR=/tmp
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; e=d.get('engine_default_path') or {}
        print('%-70s it/s %8.0f  col_us %6.1f (frac %.3f) row_us %6.1f | default it/s %8.0f'%(d['config']['workload'][:70],d['value'],r['launch_us'],r['frac'],(r.get('row_kernel') or {}).get('launch_us',0),e.get('value',0)))
"; }
run6() {
  case "$1" in
    base) OUT=gpurun_out/r6_base.jsonl; : > $OUT
      b() { timeout 600 python bench.py --cpu-iters 0 --pmc 0 "$@" 2>gpurun_out/r6_base.err | grep '^{' | tee -a $OUT | line; }
      b --steps 20 --warmup 5
      b --steps 200 --warmup 20
      b --workload cfg2dense --steps 100 --warmup 12
      b --workload cfg2dense --steps 100 --warmup 12 --method WGS-Kim
      b --workload cfg5mraf --steps 40 --warmup 5
      b --workload cfg5mraf --steps 20 --warmup 3 --dtype f64
      b --workload cfg5pad --steps 50 --warmup 5
      b --workload hd --steps 100 --warmup 10
      b --workload cfg3 --steps 100 --warmup 10 ;;
    mraf) OUT=gpurun_out/r6_mraf.jsonl; : > $OUT        # single-inverse MRAF (col_presum_kernel + RULE 5) against the split form
      b() { timeout 600 python bench.py --cpu-iters 0 --pmc ${PMC:-0} "$@" 2>gpurun_out/r6_mraf.err | grep '^{' | tee -a $OUT | line; }
      for v in 0 1 0 1; do echo "HGS_MRAF_PRESUM=$v"; HGS_MRAF_PRESUM=$v b --workload cfg5mraf --steps 40 --warmup 5; done
      HGS_MRAF_PRESUM=1 b --workload cfg5mraf --steps 40 --warmup 5 --method WGS-Kim ;;
    presum) OUT=gpurun_out/r6_presum.jsonl; : > $OUT     # the pre-pass' forms: tile-register form (1 workgroup per CU at 8192) against the lean one
      b() { timeout 600 python bench.py --cpu-iters 0 --pmc 0 --no-extra-pass "$@" 2>gpurun_out/r6_presum.err | grep '^{' | tee -a $OUT | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print('   it/s %7.0f col %6.1f row %5.1f presum %5.1f us'%(d['value'],r['launch_us'],r['row_launch_us'],(r.get('presum_launch') or {}).get('launch_us',0)))"; }
      for v in "HGS_PRESUM_LEAN=0" "HGS_PRESUM_LEAN=1" "HGS_PRESUM_LEAN=1 HGS_PRESUM_BLOCKS=256" "HGS_PRESUM_LEAN=1 HGS_PRESUM_BLOCKS=768" "HGS_PRESUM_LEAN=0 HGS_PRESUM_BLOCKS=512" "HGS_PRESUM_LEAN=0" "HGS_PRESUM_LEAN=1"; do
        echo "$v"; env $v bash -c "$(declare -f b); b --workload cfg5mraf --steps 40 --warmup 5"; done ;;
    final)    # the evidence of the round on the final tree: gpurun_out/ -> profiles/r06/ (copied by hand)
      HGS_TEST_BUDGET_S=0 timeout 900 python -m pytest tests/test_fuzz_parity.py -m gpu -q -k "large" -p no:cacheprovider 2>&1 | tail -5 > gpurun_out/r6_slow_cases.log
      timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_bench_cfg2_driver_protocol.json 2> gpurun_out/r6_bench_driver.err
      timeout 600 python bench.py > gpurun_out/r6_bench_cfg2_n1.json 2> gpurun_out/r6_bench_n1.err
      bash tools/gpu_r4.sh profiles r06
      bash tools/gpu_r4.sh configs
      python tools/call_overhead_probe.py > gpurun_out/r6_call_overhead.json 2> gpurun_out/r6_call_overhead.err
      python tools/e2e_timing.py gpurun_out/r6_e2e.json > /dev/null 2> gpurun_out/r6_e2e.err
      : > gpurun_out/r6_first_use.jsonl
      for m in "" dense "" dense; do python tools/first_use_probe.py $m 2>/dev/null >> gpurun_out/r6_first_use.jsonl; done
      python tools/amp_array_probe.py > gpurun_out/r6_amp_array.json 2>/dev/null
      bash tools/cfg3_steadiness.sh 10 > /dev/null 2>&1
      tools/microbench/pow_rule64 > gpurun_out/r6_pow_rule64.log 2>&1
      for v in 0 1; do HGS_MRAF_PRESUM=$v timeout 300 python bench.py --cpu-iters 0 --pmc 0 --workload cfg5mraf --steps 40 --warmup 5 2>/dev/null | grep '^{' >> gpurun_out/r6_ab_mraf_presum.jsonl
                       HGS_MRAF_PRESUM=$v timeout 300 python bench.py --cpu-iters 0 --pmc 0 --workload cfg5mraf --dtype f64 --steps 20 --warmup 3 2>/dev/null | grep '^{' >> gpurun_out/r6_ab_mraf_presum.jsonl; done
      ls gpurun_out | head -80 ;;
    final2)   # the lines that later kernel changes of the round touched, re-taken on the final tree
      timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_bench_cfg2_driver_protocol.json 2> gpurun_out/r6_bench_driver.err
      timeout 600 python bench.py > gpurun_out/r6_bench_cfg2_n1.json 2> gpurun_out/r6_bench_n1.err
      bash tools/gpu_r4.sh configs
      prof() { n=$1; shift; bash tools/profile.sh r06_$n "$@" > gpurun_out/prof_$n.log 2>&1; }
      prof cfg5mraf --workload cfg5mraf; prof cfg5mraf_gs --workload cfg5mraf --method GS; prof cfg5mraf_f64 --workload cfg5mraf --dtype f64; prof cfg2kim --method WGS-Kim
      prof cfg2 ; prof refbench --workload refbench
      python tools/e2e_timing.py gpurun_out/r6_e2e.json > /dev/null 2> gpurun_out/r6_e2e.err
      ls gpurun_out | wc -l ;;
    final3)   # ... and once more after the 8192-row tile kernel's waits moved (cfg 5 / cfg5pad lines and profiles, every bench line)
      bash tools/gpu_r4.sh configs
      prof() { n=$1; shift; bash tools/profile.sh r06_$n "$@" > gpurun_out/prof_$n.log 2>&1; }
      prof cfg5pad --workload cfg5pad; prof cfg5mraf --workload cfg5mraf; prof cfg5mraf_gs --workload cfg5mraf --method GS
      for m in WGS-Leonardo GS; do timeout 300 python bench.py --workload cfg5mraf --method $m --sparse-columns 1 --steps 60 --warmup 8 --cpu-iters 0 --pmc 0 2>/dev/null; done > gpurun_out/r6_cfg5_engine_default.jsonl
      timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_bench_cfg2_driver_protocol.json 2> gpurun_out/r6_bench_driver.err
      ls gpurun_out | wc -l ;;
    t) shift; timeout ${TMO:-1200} python -m pytest "$@" -m gpu -q -x -p no:cacheprovider 2>&1 | tail -${TAIL:-25} ;;
    *) bash tools/gpu_r5.sh "$@" ;;
  esac
}
args=()
for a in "$@"; do
  if [ "$a" == "--" ]; then run6 "${args[@]}"; args=(); else args+=("$a"); fi
done
[ ${#args[@]} -gt 0 ] && run6 "${args[@]}"
exit 0
