"""Single-pass MRAF against the two-pass form on 4096^2 / 8192^2 pads: wall time of resident optimize() calls (dense launches).
usage: python tools/mraf_split_probe.py   (HGS_MRAF_SPLIT is read by hgs_create, so each engine is created under its value)"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from slmsuite_amd import _lib as L, synth
from slmsuite_amd.holography.algorithms import Hologram


def target(n):
    t = np.zeros((n, n), dtype=np.float32)
    a, b = n // 2 - 3 * n // 16, n // 2 + 3 * n // 16
    t[a:b, a:b] = np.nan
    a, b = n // 2 - n // 8, n // 2 + n // 8
    t[a:b, a:b] = synth.random_target(5, (n // 4, n // 4), 0.2, 1.0)
    return t


# (the last two: SLM rows over more than six register slots -- no tile-resident kernel, the per-column kernel's single pass)
for n, slm in ((4096, (800, 1280)), (4096, (1152, 1920)), (8192, (1152, 1920)), (8192, (2600, 1920)), (4096, (2048, 1920)), (8192, (3600, 1920))):
    t = target(n)
    for sparse in (0, 1):
        res = {}
        for split in ("1", "0"):
            os.environ["HGS_MRAF_SPLIT"] = split
            h = Hologram(t, phase=synth.seed_phase(7, slm), slm_shape=slm, dtype=np.float32, engine_options={L.OPT_SPARSE_COLUMNS: sparse})
            h.optimize("WGS-Leonardo", maxiter=10, verbose=False, mraf_factor=0.5)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter()
                h.optimize("WGS-Leonardo", maxiter=40, verbose=False, mraf_factor=0.5)
                torch.cuda.synchronize()          # (the device loop returns before its launches have run)
                best = min(best, (time.perf_counter() - t0) / 40)
            res[split] = best * 1e6
            h._release_engine()
        print(f"{n}^2 slm {slm} {'column list' if sparse else 'dense'}: one pass {res['1']:.1f} us / iteration, two passes {res['0']:.1f} us", flush=True)
