"""What the HIP-event pair of hgs_iterate_timed costs a short timed region: cfg 2, 20-step regions bracketed as bench.py brackets
them (engine sync + torch.cuda.synchronize on both sides), with the events (time_iterations) and without (optimize)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from slmsuite_amd import _lib as L  # noqa: E402
from slmsuite_amd import synth  # noqa: E402
from slmsuite_amd.batch import HologramBatch  # noqa: E402
from slmsuite_amd.holography.algorithms import SpotHologram  # noqa: E402

shape, slm = (4096, 4096), (1152, 1920)
host = SpotHologram.make_rectangular_array(shape, (32, 32), (64, 64), basis="knm", slm_shape=slm, phase=synth.seed_phase(2, slm))
hb = HologramBatch(shape, slm, np.asarray(host.target)[None], np.asarray(host.phase)[None], dtype=np.float32)
hb.set_option(L.OPT_SPARSE_COLUMNS, 0)
hb.optimize("WGS-Leonardo", maxiter=30)
hb.sync()


def sync_all():
    hb.sync()
    torch.cuda.synchronize()


out = {}
for K in (20, 200):
    for name, fn in (("events", lambda: hb.time_iterations("WGS-Leonardo", K)), ("plain", lambda: hb.optimize("WGS-Leonardo", maxiter=K))):
        ws = []
        for _ in range(60):
            sync_all()
            t0 = time.perf_counter()
            fn()
            sync_all()
            ws.append(time.perf_counter() - t0)
        ws.sort()
        out[f"K{K}_{name}_us_per_step"] = ws[len(ws) // 2] * 1e6 / K
        out[f"K{K}_{name}_it_s"] = K / ws[len(ws) // 2]
# the bracket itself: engine sync + torch sync (bench.py) against torch sync alone (a device-wide wait covers the engine's stream)
for name, sync in (("both_syncs", sync_all), ("torch_sync_only", torch.cuda.synchronize)):
    ws = []
    for _ in range(60):
        sync()
        t0 = time.perf_counter()
        hb.optimize("WGS-Leonardo", maxiter=20)
        sync()
        ws.append(time.perf_counter() - t0)
    ws.sort()
    out[f"K20_plain_{name}_us_per_step"] = ws[len(ws) // 2] * 1e6 / 20
print(json.dumps(out))
hb.close()
