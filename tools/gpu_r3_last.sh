#!/bin/bash
# last visit of the round: cfg 5 profile + lines of the final build, build() + smoke()
mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/profile.sh r03_cfg5mraf --workload cfg5mraf > gpurun_out/prof_cfg5mraf.log 2>&1
OUT=gpurun_out/cfg5_lines.jsonl; : > $OUT
run() { timeout 900 python bench.py "$@" 2>gpurun_out/configs.err | grep '^{' >> $OUT || { echo "FAILED: $@"; tail -5 gpurun_out/configs.err; }; }
run --workload cfg5mraf --steps 40 --warmup 5 --cpu-iters 0
run --workload cfg5mraf --steps 40 --warmup 5 --method GS --cpu-iters 0
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
python - <<'PY'
import json
for l in open("gpurun_out/cfg5_lines.jsonl"):
    d = json.loads(l); r = d.get("roofline") or {}; e = d.get("engine_default_path") or {}
    print(d["config"]["workload"][:40], round(d["value"], 1), "frac", round(r.get("frac", 0), 3), "traffic", r.get("traffic"), "default", e.get("value"))
PY
