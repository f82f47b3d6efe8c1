#!/usr/bin/env python
"""Per-iteration, per-spot deviation engine vs CPU oracle on cfg 2 for one seed (GPU box; oracle = checker)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import hgs_oracle as orc
from slmsuite_amd import synth, _lib as L
from slmsuite_amd.holography.algorithms import SpotHologram

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
shape, slm = (4096, 4096), (1152, 1920)
o = orc.OracleSpotHologram(shape, orc.rectangular_array(shape, (32, 32), (64, 64)), slm_shape=slm, phase=synth.seed_phase(seed, slm))
h = SpotHologram.make_rectangular_array(shape, (32, 32), (64, 64), basis="knm", slm_shape=slm, phase=synth.seed_phase(seed, slm))
ky, kx = h.spot_knm_rounded[1], h.spot_knm_rounded[0]
for it in range(1, iters + 1):
    o.optimize("WGS-Leonardo", maxiter=1, populate=True)
    h.optimize("WGS-Leonardo", maxiter=1, verbose=False)
    a, b = h.amp_ff[ky, kx].astype(float), o.amp_ff[ky, kx].astype(float)
    d = a - b
    rel = np.linalg.norm(d) / np.linalg.norm(b)
    worst = np.argsort(-np.abs(d))[:4]
    wa, wb = h.weights[ky, kx].astype(float), o.weights[ky, kx].astype(float)
    pe = np.sqrt(np.mean(np.abs(np.exp(1j * h.phase.astype(float)) - np.exp(1j * o.phase.astype(float))) ** 2))
    nf_small = None
    print(f"it {it}: spot amp rel {rel:.2e}  max|d|/mean {np.abs(d).max() / b.mean():.2e} at spots {worst.tolist()} "
          f"(d/mean {[float('%.1e' % (d[i] / b.mean())) for i in worst]})  weights rel {np.linalg.norm(wa - wb) / np.linalg.norm(wb):.2e} "
          f"phase phasor rms {pe:.2e}  amp spread {b.std() / b.mean():.3f}")
