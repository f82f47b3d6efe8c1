"""Wall time of optimize() as a user sees it (cfg 2, engine default path): construction, first call (engine creation +
uploads), repeated calls, a call with statistics, and what a call is made of."""
import sys
import time

sys.path.insert(0, ".")
from slmsuite_amd import synth
from slmsuite_amd.holography.algorithms import SpotHologram

SH, SLM = (4096, 4096), (1152, 1920)
t0 = time.perf_counter()
h = SpotHologram.make_rectangular_array(SH, (32, 32), (64, 64), basis="knm", slm_shape=SLM, phase=synth.seed_phase(2, SLM))
t1 = time.perf_counter()
print(f"construct {1e3 * (t1 - t0):.1f} ms")
for i in range(5):
    t = time.perf_counter()
    h.optimize("WGS-Leonardo", maxiter=50, verbose=False)
    ta = time.perf_counter()
    p = h.phase
    tb = time.perf_counter()
    print(f"optimize(50) call {i}: {1e3 * (ta - t):.2f} ms, reading .phase {1e3 * (tb - ta):.2f} ms")
t = time.perf_counter()
h.optimize("WGS-Leonardo", maxiter=50, verbose=False, stat_groups=["computational_spot"])
print(f"optimize(50, stat_groups=[computational_spot]): {1e3 * (time.perf_counter() - t):.2f} ms")
t = time.perf_counter()
w = h.weights
print(f"reading .weights (67 MB): {1e3 * (time.perf_counter() - t):.2f} ms")
