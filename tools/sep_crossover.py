"""Compressed transforms at SLM size 1152 x 1920: matrix-core (separable) form vs the direct run kernels vs the per-pixel
kernels, iterations/s of WGS-Kim for a range of spot counts (where should HGS_OPT_SEPARABLE_MIN_SPOTS sit?).  GPU."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slmsuite_amd import _lib as L, synth
from slmsuite_amd.hardware import SimpleFourierSLM, SimpleSLM
from slmsuite_amd.holography.algorithms import CompressedSpotHologram

SLM = (1152, 1920)
fs = SimpleFourierSLM(SimpleSLM(SLM, pitch_um=(8, 8), wav_um=0.78))
for D in (2, 3):
    for N in (16, 32, 64, 128, 256, 512, 1024, 2048, 4096):
        v = synth.uniform01(4, (D, N), 9) * 2 - 1
        v[:2] *= 0.02
        if D == 3:
            v[2] *= 1e-6
        rec = dict(D=D, N=N)
        for name, opts in (("mfma", {L.OPT_SEPARABLE: 1, L.OPT_SEPARABLE_MIN_SPOTS: 1}),
                           ("run", {L.OPT_SEPARABLE: 0, L.OPT_RUN_KERNELS: 1}),
                           ("pixel", {L.OPT_SEPARABLE: 0, L.OPT_RUN_KERNELS: 0})):
            if name == "pixel" and N > 1024:
                continue
            h = CompressedSpotHologram(v, basis="kxy", cameraslm=fs, engine_options=opts)
            h.reset_phase(synth.seed_phase(4, SLM))
            h.optimize("WGS-Kim", maxiter=3, verbose=False)
            e = h._get_engine()
            st = h._make_step()
            n = 20 if N <= 1024 else 6
            ms = min(e.iterate_timed(st, n) for _ in range(3))
            rec[name + "_ms_per_it"] = ms / n
            h._release_engine()
        print(json.dumps(rec), flush=True)
