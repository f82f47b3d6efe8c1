#!/bin/bash
# Ten back-to-back `bench.py --workload cfg3` lines (the unit every GPU of the 8-GPU run executes) with the clocks sampled
# before each: how steady is the per-GPU number a scaling curve would be built on?  -> gpurun_out/cfg3_steadiness.jsonl / .txt
mkdir -p gpurun_out
OUT=gpurun_out/cfg3_steadiness
: > $OUT.jsonl; : > $OUT.txt
for i in $(seq 1 ${1:-10}); do
  clk=$(rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | head -3 | sed 's/.*: *//' | tr '\n' ' ')
  tmp=$(rocm-smi --showtemp --showpower 2>/dev/null | grep -E "junction|Socket Power|Average Graphics" | head -2 | sed 's/.*: *//' | tr '\n' ' ')
  timeout 300 python bench.py --workload cfg3 --steps 100 --warmup 10 --cpu-iters 0 --pmc 0 --no-extra-pass ${CFG3_ARGS} 2>/dev/null | grep '^{' | tee -a $OUT.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print('run $i  it/s %8.0f  min %8.0f max %8.0f  col_us %6.1f row_us %6.1f | clocks: $clk | $tmp'%(d['value'], d['steps']/d['ms_per_step_max']*1e3*d['config']['holograms_per_gpu'], d['steps']/d['ms_per_step_min']*1e3*d['config']['holograms_per_gpu'], r['launch_us'], r['row_launch_us']))" | tee -a $OUT.txt
done
python - <<'PY' | tee -a gpurun_out/cfg3_steadiness.txt
import json
v=[json.loads(l)['value'] for l in open('gpurun_out/cfg3_steadiness.jsonl')]
v.sort(); med=v[len(v)//2]
print('median %.0f it/s, min %.0f (%.1f %%), max %.0f (+%.1f %%) over %d runs'%(med, v[0], (v[0]/med-1)*100, v[-1], (v[-1]/med-1)*100, len(v)))
PY
