#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_full_configs.py -m gpu -q -p no:cacheprovider -k "cfg5_precision or cfg2" > gpurun_out/ab4_tests.log 2>&1; tail -4 gpurun_out/ab4_tests.log
HGS_TRACE_INIT=1 python - <<'PY' 2>&1 | tail -30
import sys, time
sys.path.insert(0, ".")
import numpy as np
from slmsuite_amd.engine import Engine
from slmsuite_amd import _lib as L
e0 = Engine((256, 256), (64, 64)); e0.close()
for i in range(2):
    t = time.perf_counter(); e = Engine((4096, 4096), (1152, 1920), n_spots=1024); print("Engine() %.2f ms" % (1e3 * (time.perf_counter() - t)))
    p = np.zeros((1152, 1920), np.float32)
    t = time.perf_counter(); e.set(L.PHASE, p); print("phase upload %.2f ms" % (1e3 * (time.perf_counter() - t)))
    t = time.perf_counter(); e.close(); print("close %.2f ms" % (1e3 * (time.perf_counter() - t)))
PY
bash tools/profile.sh r03_cfg5pad --workload cfg5pad > gpurun_out/prof_cfg5pad.log 2>&1; grep -A12 "col_tile_kernel<float, 8192" gpurun_out/prof_r03_cfg5pad/summary.md | head -60
