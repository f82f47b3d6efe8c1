#!/bin/bash
# builds tools/microbench/trace8k_{pad6,pad4,pad3,mraf,f64}[_t] (plain = stand-alone launch time, _t = traced)
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-unused-value"
TR="-DHGS_TRACE=1 -DHGS_TRACE_OFF=147456 -DHGS_TRACE_SKIP=${SKIP:-80}"
hipcc $F -DWHICH=0 -DNRT=6 trace8k.hip -o trace8k_pad6 &
hipcc $F -DWHICH=0 -DNRT=4 trace8k.hip -o trace8k_pad4 &
hipcc $F -DWHICH=0 -DNRT=3 trace8k.hip -o trace8k_pad3 &
hipcc $F -DWHICH=1 trace8k.hip -o trace8k_mraf &
hipcc $F -DWHICH=2 -mllvm -disable-machine-licm trace8k.hip -o trace8k_f64 &
# A/B in one GPU call: the round-5 experiments that were not adopted (no wait ahead of the staged tile, L2 prefetches)
EXP="-DHGS_TILE_STAGE_WAIT=0 -DHGS_F64_WT_PREFETCH=1 -DHGS_SPLIT_L2_PREFETCH=1"
hipcc $F $EXP -DWHICH=0 -DNRT=3 trace8k.hip -o trace8k_pad3_exp &
hipcc $F $EXP -DWHICH=1 trace8k.hip -o trace8k_mraf_exp &
hipcc $F $EXP -DWHICH=2 -mllvm -disable-machine-licm trace8k.hip -o trace8k_f64_exp &
wait
hipcc $F $TR -DWHICH=0 -DNRT=6 trace8k.hip -o trace8k_pad6_t &
hipcc $F $TR -DWHICH=0 -DNRT=4 trace8k.hip -o trace8k_pad4_t &
hipcc $F $TR -DWHICH=0 -DNRT=3 trace8k.hip -o trace8k_pad3_t &
hipcc $F $TR -DWHICH=1 trace8k.hip -o trace8k_mraf_t &
hipcc $F $TR -DWHICH=2 -mllvm -disable-machine-licm trace8k.hip -o trace8k_f64_t &
hipcc $F -DWHICH=3 -mllvm -disable-machine-licm trace8k.hip -o trace8k_f64main &
hipcc $F -DWHICH=4 -mllvm -disable-machine-licm trace8k.hip -o trace8k_f64pre &
hipcc $F $TR -DWHICH=3 -mllvm -disable-machine-licm trace8k.hip -o trace8k_f64main_t &
hipcc $F $TR -DWHICH=4 -mllvm -disable-machine-licm trace8k.hip -o trace8k_f64pre_t &
wait
ls -la trace8k_*
