#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for l in 0 49152 0 49152 ; do
HGS_ROW_LDS_MIN=$l WORKLOADS="cfg2" bash tools/gpu_ab.sh main 2>&1 | tail -1 | sed "s/^/ldsmin=$l /"
done
HGS_ROW_LDS_MIN=0 WORKLOADS="cfg3" bash tools/gpu_ab.sh main 2>&1 | tail -1 | sed "s/^/ldsmin=0 /"
