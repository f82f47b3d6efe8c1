#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
HGS_LIB=$PWD/slmsuite_amd/libhgs_split.so python -m pytest tests/test_gpu_round3.py -m gpu -q -p no:cacheprovider -k prefetching 2>&1 | tail -1
for v in main split main split; do WORKLOADS="cfg2" bash tools/gpu_ab.sh $v 2>&1 | tail -1; done
