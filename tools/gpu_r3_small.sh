#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; ROOTDIR=$(pwd)
for wl in cfg1 small; do
cd /tmp; rocprofv3 --output-format csv --kernel-trace --stats -d $ROOTDIR/gpurun_out/prof_small_$wl -o s -- python $ROOTDIR/bench.py --workload $wl --steps 400 --warmup 20 --cpu-iters 0 --pmc 0 --no-roofline-pass --reps 3 > $ROOTDIR/gpurun_out/prof_small_$wl.log 2>&1; cd $ROOTDIR
grep '^{' gpurun_out/prof_small_$wl.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$wl it/s', round(d['value']), 'ms_per_step', d['ms_per_step'], 'event', d['event_ms_per_step'])"
python - <<PY
import csv,glob
for f in glob.glob("gpurun_out/prof_small_$wl/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:5]: print(r["Name"][:60], r["Calls"], "avg us %.2f"%(float(r["AverageNs"])/1e3))
PY
done
