mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_compressed.py tests/test_dispatch.py tests/test_full_configs.py -m gpu -q --tb=short -p no:cacheprovider -k "compressed or monomial or cfg4 or zern" > gpurun_out/d_pytest.log 2>&1; tail -8 gpurun_out/d_pytest.log
python - > gpurun_out/d_wavefront_profile.log 2>&1 <<'PY'
import sys, json, os, time, cProfile, pstats
sys.path.insert(0, ".")
import torch
import numpy as np
from slmsuite_amd import synth
from slmsuite_amd.hardware import SimpleFourierSLM, SimpleSLM
from slmsuite_amd.holography.algorithms import CompressedSpotHologram
slm_shape = (1152, 1920)
fs = SimpleFourierSLM(SimpleSLM(slm_shape, pitch_um=(8, 8), wav_um=0.78))
basis = np.array([2, 1, 4, 3, 5, 7, 8, 6, 9, 12]); N = 16
z = np.zeros((len(basis), N)); z[:2] = 600 * (synth.uniform01(41, (2, N), 0) - 0.5); z[2:] = 1.0 * (synth.uniform01(42, (len(basis) - 2, N), 0) - 0.5)
h = CompressedSpotHologram(z.copy(), basis=basis, cameraslm=fs); h.reset_phase(synth.seed_phase(40, slm_shape))
h.optimize("GS", maxiter=3, verbose=False); _ = h.get_phase()
def rounds(n):
    for rnd in range(n):
        z[2 + rnd % 8, :] += 0.05
        h.spot_zernike = z.copy()
        h.optimize("GS", maxiter=3, verbose=False)
        _ = h.get_phase()
rounds(4)
t = time.perf_counter(); rounds(40); print("ms per round", 1e3 * (time.perf_counter() - t) / 40)
pr = cProfile.Profile(); pr.enable(); rounds(40); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
PY
head -70 gpurun_out/d_wavefront_profile.log | cut -c1-160
