"""
Is the distance between the device-resident callback loop and the host-driven loop on a pixel-wise WGS image target
rounding growth or a semantic difference?  The same case (tests/test_gpu_round5.py, "image WGS-Kim") in float32 and
float64, with and without a callback, fused against stepwise, after 1 .. 6 bodies.

    python tools/diag_callback.py            (GPU)
"""
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import torch  # noqa: E402,F401
from conftest import force_stepwise, phase_rel_l2, rel_l2  # noqa: E402
from slmsuite_amd import synth  # noqa: E402
from slmsuite_amd.holography.algorithms import Hologram  # noqa: E402

shape, slm = (256, 256), (72, 120)


def make(dtype):
    return Hologram(synth.random_target(3, shape, 0.2, 1.0).astype(dtype), phase=synth.seed_phase(3, slm).astype(dtype), slm_shape=slm, dtype=dtype)


for dtype in (np.float32, np.float64):
    for n_it in (1, 2, 3, 4, 5, 6):
        row = []
        for cb in (None, lambda h: False):
            a, b = make(dtype), force_stepwise(make(dtype))
            for h in (a, b):
                h.optimize("WGS-Kim", maxiter=n_it, verbose=False, callback=cb, fix_phase_iteration=2)
            row.append((phase_rel_l2(a.phase, b.phase), rel_l2(a.amp_ff, b.amp_ff)))
        # fused with callback against fused without
        a, b = make(dtype), make(dtype)
        a.optimize("WGS-Kim", maxiter=n_it, verbose=False, callback=lambda h: False, fix_phase_iteration=2)
        b.optimize("WGS-Kim", maxiter=n_it, verbose=False, fix_phase_iteration=2)
        print(f"{np.dtype(dtype).name} bodies={n_it}: fused vs stepwise, no callback: phase {row[0][0]:.2e} amp_ff {row[0][1]:.2e} | "
              f"with callback: phase {row[1][0]:.2e} amp_ff {row[1][1]:.2e} | fused callback vs fused plain: phase {phase_rel_l2(a.phase, b.phase):.2e}")
