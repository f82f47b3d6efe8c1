#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "cfg2 or cfg5 or single_step or sparse_column or trajectory or spot_hologram or mraf" > gpurun_out/ab5_tests.log 2>&1; tail -3 gpurun_out/ab5_tests.log
for r in 1 0 1 0; do
HGS_TILE_RULE=$r WORKLOADS="cfg2" bash tools/gpu_ab.sh main 2>&1 | tail -1 | sed "s/^/rule=$r /"
done
for r in 1 0; do
HGS_TILE_RULE=$r WORKLOADS="cfg2dense cfg5pad" bash tools/gpu_ab.sh main 2>&1 | tail -2 | sed "s/^/rule=$r /"
HGS_TILE_RULE=$r WORKLOADS="cfg2dense" BENCH_ARGS="--method GS" bash tools/gpu_ab.sh main 2>&1 | tail -1 | sed "s/^/rule=$r GS /"
done
