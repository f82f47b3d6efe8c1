#!/usr/bin/env python
"""cfg 2 geometry with a DENSE random image target (no zero pixel: nothing can be skipped): it/s per method."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from slmsuite_amd import synth                       # noqa: E402
from slmsuite_amd.batch import HologramBatch         # noqa: E402

shape, slm = (4096, 4096), (1152, 1920)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
t = synth.random_target(11, shape, 0.2, 1.0)
t /= np.sqrt(np.sum(t.astype(np.float64) ** 2))
for method in ("GS", "WGS-Leonardo", "WGS-Kim", "WGS-Wu"):
    hb = HologramBatch(shape, slm, t.astype(np.float32), synth.seed_phase(3, slm)[None])
    hb.time_iterations(method, 12)
    hb.engine.profile_enable(True)
    ms = hb.time_iterations(method, K)
    p = hb.engine.profile_read()
    print(f"dense image target {method:13s} {K / (ms * 1e-3):8.0f} it/s   column {p['col_fused']['ms'] * 1e3 / p['col_fused']['launches']:6.1f} us"
          f"   row {p['row']['ms'] * 1e3 / p['row']['launches']:6.1f} us")
    hb.close()
