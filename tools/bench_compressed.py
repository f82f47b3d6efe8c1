#!/usr/bin/env python
"""
BASELINE config 4: CompressedSpotHologram, N = 1e4 spots, SLM 1152 x 1920, D = 2 (tilt) or 3
(tilt + focus), WGS-Kim.  The path is VALU/transcendental bound: report kernel evaluations per
second (2*N*S per iteration) against the FP32 vector peak, counting 14 flop per evaluation for
D = 2.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slmsuite_amd import synth                                         # noqa: E402
from slmsuite_amd.hardware import SimpleFourierSLM, SimpleSLM          # noqa: E402
from slmsuite_amd.holography.algorithms import CompressedSpotHologram   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--spots", type=int, default=10000)
ap.add_argument("--dim", type=int, default=2)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--slm", type=int, nargs=2, default=(1152, 1920))
args = ap.parse_args()

slm = SimpleSLM(tuple(args.slm), pitch_um=(8, 8), wav_um=0.78)
v = synth.uniform01(4, (args.dim, args.spots), 9) * 2 - 1
v[:2] *= 0.02
if args.dim == 3:
    v[2] *= 1e-6
h = CompressedSpotHologram(v, basis="kxy", cameraslm=SimpleFourierSLM(slm))
h.reset_phase(synth.seed_phase(4, tuple(args.slm)))
h.optimize("WGS-Kim", maxiter=2, verbose=False)          # warm-up
h._engine.sync()
t0 = time.perf_counter()
h.optimize("WGS-Kim", maxiter=args.iters, verbose=False)
h._engine.sync()
dt = time.perf_counter() - t0
per_iter = dt / (args.iters + 0.5)                        # + the trailing forward of _populate_results
S = args.slm[0] * args.slm[1]
evals = 2.0 * args.spots * S / per_iter
flop = (2 * args.dim + 8 + 2)
mode = "direct kernels" if os.environ.get("HGS_C_SEPARABLE", "1") == "0" else "separable form on the matrix cores (D <= 3)"
print(json.dumps({"metric": "CompressedSpotHologram WGS-Kim iterations/s", "value": 1 / per_iter, "path": mode,
                  "mfma_tflops": 2 * 8.0 * args.spots * S / per_iter / 1e12,
                  "spots": args.spots, "slm": list(args.slm), "dim": args.dim, "ms_per_iter": per_iter * 1e3,
                  "kernel_evaluations_per_s": evals, "tflops_equiv": evals * flop / 1e12,
                  "frac_fp32_vector_peak": evals * flop / 157.3e12,
                  "uniformity_check": float(np.std(h.amp_ff) / np.mean(h.amp_ff))}))
