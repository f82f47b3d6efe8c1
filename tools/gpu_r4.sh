#!/bin/bash
# Round-4 GPU batches (one parameterised script instead of one file per experiment): tools/gpu_r4.sh STEP [STEP ...]
#   tests [pytest args]   the GPU suite (or a selection) -> gpurun_out/r4_pytest_<tag>.log
#   bench NAME args...    one bench.py line -> gpurun_out/r4_bench_NAME.json
#   pow                   accuracy of the weight rule's power (tools/microbench/pow_rule)
#   e2e                   tools/e2e_timing.py -> gpurun_out/r4_e2e.json
#   profiles TAG          rocprofv3 --kernel-trace --stats + PMC summaries of every configuration (tools/profile.sh)
#                         -> gpurun_out/prof_<TAG>_<config>/summary.md
#   configs               the bench lines of the other BASELINE configurations (tools/gpu_configs.sh) -> gpurun_out/configs.jsonl
#   sweep                 tools/cfg5_sweep.py -> gpurun_out/cfg5_sweep.json
#   sh CMD...             any shell command
# Steps are separated by "--".  A/B builds: tools/build_variant.sh + tools/gpu_ab.sh; grid sweeps: tools/gpu_sweep.sh.
# (Rounds 2 and 3 kept one script per experiment, tools/gpu_r3_*.sh: see git history; their results are DESIGN.md Appendix A.)
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${R4_TAG:-run}
run_step() {
  case "$1" in
    tests) shift; rm -f gpurun_out/parity_report.jsonl; python -m pytest "$@" -m gpu -q 2>&1 | tail -${R4_TAIL:-80} > gpurun_out/r4_pytest_$TAG.log; cp gpurun_out/parity_report.jsonl gpurun_out/r4_parity_$TAG.jsonl 2>/dev/null ;;
    bench) shift; name=$1; shift; timeout 900 python bench.py "$@" > gpurun_out/r4_bench_$name.json 2> gpurun_out/r4_bench_$name.err ;;
    pow) tools/microbench/pow_rule > gpurun_out/r4_pow_rule.log 2>&1 ;;
    e2e) python tools/e2e_timing.py gpurun_out/r4_e2e.json > /dev/null 2> gpurun_out/r4_e2e.err ;;
    profiles) t=${2:-r04}
      prof() { n=$1; shift; bash tools/profile.sh ${t}_$n "$@" > gpurun_out/prof_$n.log 2>&1; }
      prof cfg2; prof cfg3 --workload cfg3 --streams 1; prof cfg3_2streams --workload cfg3; prof cfg5pad --workload cfg5pad; prof cfg5mraf --workload cfg5mraf
      prof cfg5mraf_f64 --workload cfg5mraf --dtype f64; prof hd --workload hd; prof cfg2dense --workload cfg2dense
      prof cfg4 --workload cfg4 --steps 20; prof cfg4zern --workload cfg4zern --steps 20; prof cfg1 --workload cfg1 --steps 200 ;;
    configs) bash tools/gpu_configs.sh > gpurun_out/configs.log 2>&1
      timeout 900 python bench.py --workload refbench 2>/dev/null | grep '^{' >> gpurun_out/configs.jsonl ;;
    sweep) timeout 1500 python tools/cfg5_sweep.py gpurun_out/cfg5_sweep.json > gpurun_out/cfg5_sweep.log 2>&1 ;;
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
