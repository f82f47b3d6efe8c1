#!/bin/bash
# A/B build of the hot kernels: tools/build_variant.sh NAME "-DHGS_X=1 ..."  ->  slmsuite_amd/libhgs_NAME.so
# (only the fp32 row / fused-column translation units are recompiled -- or those named in TUS="launch_tile_rule_f32 ..." --
#  everything else is linked from the main build)
# use it with  HGS_LIB=slmsuite_amd/libhgs_NAME.so python bench.py ...
set -e
NAME=$1; shift
cd "$(dirname "$0")/../slmsuite_amd/csrc"
mkdir -p ../../build/ab_$NAME
F="--offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -Wno-unused-value -Wno-unused-function $@"
for tu in ${TUS:-launch_row_f32 launch_fused_f32}; do
  hipcc $F -c $tu.hip -o ../../build/ab_$NAME/$tu.o &
done
wait
OBJS=""
for o in $(grep '^OBJ' Makefile | sed 's/OBJ *= *//; s/\.o//g'); do
  if [ -f ../../build/ab_$NAME/$o.o ]; then OBJS="$OBJS ../../build/ab_$NAME/$o.o"; else OBJS="$OBJS $o.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhgs_$NAME.so $OBJS
ls -la ../libhgs_$NAME.so
