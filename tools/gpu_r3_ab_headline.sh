#!/bin/bash
# headline A/B: HGS_LIB variants, alternating, three rounds
mkdir -p gpurun_out; export TMPDIR=/tmp
show='
import sys,json
d=json.loads(sys.stdin.read()); r=d["roofline"]
print(sys.argv[1], round(d["value"],1), "col_us", round(r["launch_us"],2), "frac", round(r["frac"],3))'
for rep in 1 2 3; do
for v in "$@"; do
  lib=slmsuite_amd/libhgs.so; [ "$v" != main ] && lib=slmsuite_amd/libhgs_$v.so
  HGS_LIB=$lib timeout 600 python bench.py --cpu-iters 0 --pmc 0 --no-extra-pass 2>/dev/null | tail -1 | python -c "$show" "$v"
done; done
