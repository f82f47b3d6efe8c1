#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in v1 v1w v1 v1w; do WORKLOADS="cfg2" bash tools/gpu_ab.sh $v 2>&1 | tail -1; done
