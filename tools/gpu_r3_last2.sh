#!/bin/bash
# 8192-row workloads after the one-round tile grid: tests, profiles, lines
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "8192 or cfg5 or tile_rounded or single_pass" 2>&1 | tail -2
bash tools/profile.sh r03_cfg5pad --workload cfg5pad > gpurun_out/prof_cfg5pad.log 2>&1
bash tools/profile.sh r03_cfg5mraf --workload cfg5mraf > gpurun_out/prof_cfg5mraf.log 2>&1
OUT=gpurun_out/lines8k.jsonl; : > $OUT
run() { timeout 900 python bench.py "$@" 2>gpurun_out/configs.err | grep '^{' >> $OUT || { echo "FAILED: $@"; tail -5 gpurun_out/configs.err; }; }
run --workload cfg4grid --steps 40 --warmup 12
run --workload cfg5mraf --steps 40 --warmup 5 --cpu-iters 0
run --workload cfg5mraf --steps 40 --warmup 5 --method GS --cpu-iters 0
run --workload cfg5pad --steps 50 --warmup 5
python - <<'PY'
import json
for l in open("gpurun_out/lines8k.jsonl"):
    d = json.loads(l); r = d.get("roofline") or {}; e = d.get("engine_default_path") or {}
    print(d["config"]["workload"][:40], round(d["value"], 1), "col", round(r.get("launch_us", 0), 1), "frac", round(r.get("frac", 0), 3), "traffic", r.get("traffic"), "default", e.get("value"))
PY
