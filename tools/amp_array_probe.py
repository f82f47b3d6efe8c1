"""
What an array-valued source amplitude (a Gaussian beam: the usual laboratory case; BASELINE's configurations all use the uniform
scalar) costs the row launch: cfg 2 geometry, dense kernels and engine default, scalar against array amplitude, with and without a
propagation kernel.  python tools/amp_array_probe.py
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slmsuite_amd import _lib as L              # noqa: E402
from slmsuite_amd import synth                  # noqa: E402
from slmsuite_amd.holography.algorithms import SpotHologram   # noqa: E402

shape, slm = (4096, 4096), (1152, 1920)
yy, xx = np.mgrid[:slm[0], :slm[1]]
gauss = np.exp(-(((yy - slm[0] / 2) / 500.0) ** 2 + ((xx - slm[1] / 2) / 800.0) ** 2)).astype(np.float32)
kern = (1e-5 * ((yy - slm[0] / 2) ** 2 + (xx - slm[1] / 2) ** 2)).astype(np.float32)
out = {}
for dense in (1, 0):
    for name, kw in (("scalar", {}), ("array_amp", dict(amp=gauss)), ("array_amp_and_kernel", dict(amp=gauss, propagation_kernel=kern))):
        h = SpotHologram.make_rectangular_array(shape, (32, 32), (64, 64), basis="knm", slm_shape=slm, phase=synth.seed_phase(2, slm),
                                                engine_options={L.OPT_SPARSE_COLUMNS: 0} if dense else {}, **kw)
        h.optimize("WGS-Leonardo", maxiter=20, verbose=False)
        e = h._get_engine()
        best = 1e9
        for _ in range(5):
            e.sync()
            t = time.perf_counter()
            h.optimize("WGS-Leonardo", maxiter=200, verbose=False)
            e.sync()
            best = min(best, time.perf_counter() - t)
        out[("dense " if dense else "default ") + name] = best / 200 * 1e6
        h._release_engine()
print(json.dumps({k: round(v, 2) for k, v in out.items()}))
