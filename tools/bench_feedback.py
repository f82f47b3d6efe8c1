#!/usr/bin/env python
"""cfg 2 with the secondary feedback modes (general path unless fused): iterations/s of optimize()."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from slmsuite_amd import synth                                        # noqa: E402
from slmsuite_amd.holography.algorithms import SpotHologram           # noqa: E402

shape, slm = (4096, 4096), (1152, 1920)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for method, fb in (("WGS-Leonardo", "computational"), ("WGS-Leonardo", "computational_spot"), ("WGS-Kim", "computational_spot"),
                   ("WGS-Leonardo", "external_spot")):
    h = SpotHologram.make_rectangular_array(shape, (32, 32), (64, 64), basis="knm", slm_shape=slm,
                                            phase=synth.seed_phase(2, slm))
    h.optimize(method, maxiter=2, verbose=False, feedback=fb)
    h._get_engine().sync()
    t0 = time.perf_counter()
    h.optimize(method, maxiter=K, verbose=False, feedback=fb)
    h._get_engine().sync()
    dt = time.perf_counter() - t0
    print(f"{method:14s} feedback={fb:20s} {K / dt:9.0f} it/s  ({dt / K * 1e6:8.1f} us/it incl. populate and read-back)")
