#!/bin/bash
# prefetching row kernel: parity subset, then A/B against the one-row-per-workgroup launch (HGS_ROW_PREF=0)
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_full_configs.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "cfg2 or trajectory or single_step" 2>&1 | tail -3
for p in 1 0 1 0; do
HGS_ROW_PREF=$p WORKLOADS="cfg2" bash tools/gpu_ab.sh main 2>&1 | tail -1 | sed "s/^/pref=$p /"
done
for nb in 256 384 512 768; do
HGS_ROW_PREF_BLOCKS=$nb WORKLOADS="cfg2" bash tools/gpu_ab.sh main 2>&1 | tail -1 | sed "s/^/pref blocks=$nb /"
done
