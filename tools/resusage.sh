#!/bin/bash
# usage: tools/resusage.sh file.hip  -> one line per kernel: name VGPRs AGPRs scratch occupancy spilled VGPRs
# (the *_f64 translation units are built with -mllvm -disable-machine-licm, as in the Makefile; more flags through EXTRA=...)
F=""
case "$1" in *_f64.hip) F="-mllvm -disable-machine-licm" ;; esac
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC $F $EXTRA -c "$1" -o /tmp/_res_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 | \
 awk '/Function Name:/{name=$(NF-1)} / VGPRs:/{v=$(NF-1)} /AGPRs:/{a=$(NF-1)} /ScratchSize/{s=$(NF-1)} /Occupancy/{o=$(NF-1)} /VGPRs Spill/{sp=$(NF-1)} /LDS Size/{print name, "vgpr="v, "agpr="a, "scratch="s, "occ="o, "vspill="sp}' | c++filt | sed 's/hgs:://g'
rm -f /tmp/_res_$$.o
