#!/usr/bin/env python
"""BASELINE config 5: Hologram with MRAF (NaN noise region) on an 8192^2 pad of 1152 x 1920; it/s of optimize()."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from slmsuite_amd import synth                                        # noqa: E402
from slmsuite_amd.holography.algorithms import Hologram               # noqa: E402

n = 8192
slm = (1152, 1920)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dt = np.float64 if (len(sys.argv) > 2 and sys.argv[2] == "f64") else np.float32
t = np.zeros((n, n), dtype=dt)
a, b = (n - 3072) // 2, (n + 3072) // 2
t[a:b, a:b] = np.nan
a, b = (n - 2048) // 2, (n + 2048) // 2
t[a:b, a:b] = synth.random_target(5, (b - a, b - a), 0.2, 1.0, dtype=dt)
for method in ("GS", "WGS-Leonardo"):
    h = Hologram(t, phase=synth.seed_phase(5, slm, dtype=dt), slm_shape=slm, dtype=dt)
    h.optimize(method, maxiter=2, verbose=False, mraf_factor=0.5)
    h._get_engine().sync()
    t0 = time.perf_counter()
    h.optimize(method, maxiter=K, verbose=False, mraf_factor=0.5)
    h._get_engine().sync()
    d = time.perf_counter() - t0
    print(f"cfg5 MRAF {np.dtype(dt).name} {method:14s} {K / d:8.1f} it/s  ({d / K * 1e3:7.3f} ms/it incl. populate and read-back)")
