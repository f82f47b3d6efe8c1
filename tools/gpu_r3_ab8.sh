#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for r in 1 0 1 0; do
HGS_ROW_SHIFT=$r WORKLOADS="cfg2" bash tools/gpu_ab.sh main 2>&1 | tail -1 | sed "s/^/shift=$r /"
done
for r in 1 0; do
HGS_ROW_SHIFT=$r WORKLOADS="cfg3 cfg5pad" bash tools/gpu_ab.sh main 2>&1 | tail -2 | sed "s/^/shift=$r /"
done
