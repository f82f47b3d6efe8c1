#!/bin/bash
# Round-5 GPU batches: tools/gpu_r5.sh STEP [-- STEP ...]   (the steps of tools/gpu_r4.sh plus:)
#   trace8k               the 8192-row column kernels' launch times and s_memtime timelines -> gpurun_out/r5_trace8k.log
#   ab NAME...            tools/gpu_ab.sh over engine builds (WORKLOADS / BENCH_ARGS from the environment)
mkdir -p gpurun_out
export TMPDIR=/tmp
run5() {
  case "$1" in
    trace8k) for b in pad6 pad4 pad3 pad3_exp mraf mraf_exp f64 f64_exp; do echo "-- $b"; tools/microbench/trace8k_$b; tools/microbench/trace8k_$b | tail -1; done > gpurun_out/r5_trace8k.log 2>&1
             for b in pad6 pad4 pad3 mraf f64; do echo "==== $b"; tools/microbench/trace8k_${b}_t; done > gpurun_out/r5_trace8k_timeline.log 2>&1 ;;
    ab) shift; bash tools/gpu_ab.sh "$@" ;;
    *) bash tools/gpu_r4.sh "$@" ;;
  esac
}
args=()
for a in "$@"; do
  if [ "$a" == "--" ]; then run5 "${args[@]}"; args=(); else args+=("$a"); fi
done
[ ${#args[@]} -gt 0 ] && run5 "${args[@]}"
exit 0
