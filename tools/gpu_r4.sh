#!/bin/bash
# Round-4 GPU batches (one parameterised script instead of one file per experiment): tools/gpu_r4.sh STEP [STEP ...]
#   tests [pytest args]   the GPU suite (or a selection) -> gpurun_out/r4_pytest_<tag>.log
#   bench NAME args...    one bench.py line -> gpurun_out/r4_bench_NAME.json
#   pow                   accuracy of the weight rule's power (tools/microbench/pow_rule)
#   e2e                   tools/e2e_timing.py -> gpurun_out/r4_e2e.json
# Steps are separated by "--".
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${R4_TAG:-run}
run_step() {
  case "$1" in
    tests) shift; rm -f gpurun_out/parity_report.jsonl; python -m pytest "$@" -m gpu -q 2>&1 | tail -${R4_TAIL:-80} > gpurun_out/r4_pytest_$TAG.log; cp gpurun_out/parity_report.jsonl gpurun_out/r4_parity_$TAG.jsonl 2>/dev/null ;;
    bench) shift; name=$1; shift; timeout 900 python bench.py "$@" > gpurun_out/r4_bench_$name.json 2> gpurun_out/r4_bench_$name.err ;;
    pow) tools/microbench/pow_rule > gpurun_out/r4_pow_rule.log 2>&1 ;;
    e2e) python tools/e2e_timing.py gpurun_out/r4_e2e.json > /dev/null 2> gpurun_out/r4_e2e.err ;;
    sh) shift; bash -c "$*" ;;
    *) echo "unknown step $1" >&2 ;;
  esac
}
args=()
for a in "$@"; do
  if [ "$a" == "--" ]; then run_step "${args[@]}"; args=(); else args+=("$a"); fi
done
[ ${#args[@]} -gt 0 ] && run_step "${args[@]}"
exit 0
