mkdir -p gpurun_out; export TMPDIR=/tmp
python - > gpurun_out/b_pair_parity.log 2>&1 <<'PY'
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from conftest import dispatch_of, phase_rel_l2, rel_l2
from slmsuite_amd import _lib as L, synth
from slmsuite_amd.holography.algorithms import SpotHologram
out = {}
for slm in [(1152, 1920), (1000, 1000), (1400, 1920)]:
    for pair in ("1", "0"):
        os.environ["HGS_PAIR"] = pair
        h = SpotHologram.make_rectangular_array((4096, 4096), (32, 32), (64, 64), basis="knm", slm_shape=slm, phase=synth.seed_phase(2, slm), engine_options={L.OPT_SPARSE_COLUMNS: 0})
        h.optimize("WGS-Leonardo", maxiter=6, verbose=False)
        d = dispatch_of(h)
        out[pair] = (h.phase.copy(), h.weights.copy(), h.amp_ff.copy())
        print(slm, "pair", pair, [(r["name"], r["count"]) for r in d.records if "col_" in r["name"]])
        h._release_engine()
    a, b = out["1"], out["0"]
    print(slm, "phase", phase_rel_l2(a[0], b[0]), "weights", rel_l2(a[1], b[1]), "amp_ff", rel_l2(a[2], b[2]))
PY
cat gpurun_out/b_pair_parity.log
for p in 0 1 0 1; do HGS_PAIR=$p python bench.py --steps 200 --warmup 20 --cpu-iters 0 --pmc 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('pair=$p it/s %8.0f  col_us %6.2f row_us %6.2f frac %.3f iter %.3f %s'%(d['value'],r['launch_us'],r['row_kernel']['launch_us'],r['frac'],r['frac_iteration'],r['kernel'][:60]))
"; done 2>&1 | tee gpurun_out/b_ab_pair.log
for blk in 256 384 768 1024; do HGS_PAIR_BLOCKS=$blk python bench.py --steps 200 --warmup 20 --cpu-iters 0 --pmc 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('pair blocks=$blk it/s %8.0f  col_us %6.2f row_us %6.2f'%(d['value'],r['launch_us'],r['row_kernel']['launch_us']))
"; done 2>&1 | tee -a gpurun_out/b_ab_pair.log
