#!/bin/bash
# cross-compile the ablation variants of the two hot kernels (run tools/microbench/run_ablate.sh on the GPU box)
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize"
build() { name=$1; shift; hipcc $F "$@" ablate.hip -o ablate_$name & }
build base
build wt -DHGS_ABL_WT=1
build gh -DHGS_ABL_GH=1
build wtgh -DHGS_ABL_WT=1 -DHGS_ABL_GH=1
wait
build nofft -DHGS_ABL_XCHG=1 -DHGS_ABL_BFLY=1
build noxchg -DHGS_ABL_XCHG=1
build nobfly -DHGS_ABL_BFLY=1
build empty -DHGS_ABL_WT=1 -DHGS_ABL_GH=1 -DHGS_ABL_XCHG=1 -DHGS_ABL_BFLY=1
wait
ls -la ablate_*
