#!/bin/bash
# usage: tools/resusage.sh file.hip  -> one line per kernel: name VGPRs AGPRs scratch occupancy LDS
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -c "$1" -o /tmp/_res.o -Rpass-analysis=kernel-resource-usage 2>&1 | \
 awk '/Function Name:/{name=$(NF-1)} / VGPRs:/{v=$(NF-1)} /AGPRs:/{a=$(NF-1)} /ScratchSize/{s=$(NF-1)} /Occupancy/{o=$(NF-1)} /VGPRs Spill/{sp=$(NF-1)} /LDS Size/{print name, "vgpr="v, "agpr="a, "scratch="s, "occ="o, "vspill="sp}' | c++filt | sed 's/hgs:://g'
