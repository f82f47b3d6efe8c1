#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
HGS_ROW_PREF=2 python -m pytest tests/test_full_configs.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "cfg2 or trajectory or single_step" 2>&1 | tail -2
for p in 2 1 0 2 1; do HGS_ROW_PREF=$p WORKLOADS="cfg2" bash tools/gpu_ab.sh main 2>&1 | tail -1 | sed "s/^/pref=$p /"; done
for nb in 576 640 768 896 1024; do HGS_ROW_PREF=2 HGS_ROW_PREF_BLOCKS=$nb WORKLOADS="cfg2" bash tools/gpu_ab.sh main 2>&1 | tail -1 | sed "s/^/pref=2 blocks=$nb /"; done
