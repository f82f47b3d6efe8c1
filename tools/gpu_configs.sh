#!/bin/bash
# bench.py lines of the other BASELINE configurations -> gpurun_out/configs.jsonl
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/configs.jsonl
: > $OUT
run() { echo "== $@" >&2; timeout 900 python bench.py "$@" 2>gpurun_out/configs.err | grep '^{' >> $OUT || { echo "FAILED: $@"; tail -5 gpurun_out/configs.err; }; }
run --workload cfg1 --steps 400 --warmup 20
run --workload cfg3 --steps 100 --warmup 10
run --workload cfg2dense --steps 100 --warmup 12
run --workload cfg2dense --steps 100 --warmup 12 --method GS
run --workload cfg2dense --steps 100 --warmup 12 --method WGS-Kim
run --workload cfg4 --steps 30 --warmup 12
run --workload cfg4d3 --steps 30 --warmup 12 --cpu-iters 0
run --workload cfg4zern --steps 50 --warmup 12
run --workload cfg4grid --steps 40 --warmup 12
run --workload cfg5mraf --steps 40 --warmup 5
run --workload cfg5mraf --steps 40 --warmup 5 --method GS --cpu-iters 0
run --workload cfg5mraf --steps 20 --warmup 3 --dtype f64 --cpu-iters 0
run --workload cfg5mraf --steps 20 --warmup 3 --dtype f64 --method GS --cpu-iters 0
run --workload cfg5pad --steps 50 --warmup 5
run --workload hd --steps 100 --warmup 10 --cpu-iters 0
python - <<'PY'
import json
for l in open("gpurun_out/configs.jsonl"):
    d = json.loads(l); r = d.get("roofline") or {}
    print("%-90s %10.1f it/s  frac %.3f  %s" % (d["config"]["workload"][:90], d["value"], r.get("frac", float("nan")),
          ("traffic %.1f MB vs model %.1f MB" % (r["traffic"] / 1e6, r["bytes_per_launch"] / 1e6)) if r.get("traffic") else ""))
PY
