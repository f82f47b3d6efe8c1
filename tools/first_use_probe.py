"""
Where the first use of the engine in a fresh process spends its time: hgs_create (HIP runtime and
context), the first engine calls (code objects of the translation units they launch from: loaded on first use), against the
same calls warm.  Run once per configuration in a FRESH process:  python tools/first_use_probe.py [dense]
"""
import json
import os
import sys
import time

t_start = time.perf_counter()
import numpy as np                                                   # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.perf_counter()
from slmsuite_amd import _lib as L                                   # noqa: E402
from slmsuite_amd import synth                                       # noqa: E402
from slmsuite_amd.holography.algorithms import SpotHologram          # noqa: E402
t_import = time.perf_counter() - t0

dense = len(sys.argv) > 1 and sys.argv[1] == "dense"
shape, slm = (4096, 4096), (1152, 1920)
out = {"mode": "dense kernels" if dense else "engine default", "import_ms": t_import * 1e3,
       "HIP_ENABLE_DEFERRED_LOADING": os.environ.get("HIP_ENABLE_DEFERRED_LOADING", "(unset)")}


def lap(name, f):
    t = time.perf_counter()
    r = f()
    out[name] = (time.perf_counter() - t) * 1e3
    return r


phase = synth.seed_phase(2, slm)
h = lap("construct_host_ms", lambda: SpotHologram.make_rectangular_array(shape, (32, 32), (64, 64), basis="knm", slm_shape=slm, phase=phase,
                                                                         engine_options={L.OPT_SPARSE_COLUMNS: 0} if dense else {}))
lap("load_library_ms", lambda: L.load())
e = lap("engine_create_and_uploads_ms", lambda: h._get_engine())
lap("sync_ms", lambda: e.sync())
lap("first_optimize_1_ms", lambda: (h.optimize("WGS-Leonardo", maxiter=1, verbose=False), e.sync()))
lap("second_optimize_1_ms", lambda: (h.optimize("WGS-Leonardo", maxiter=1, verbose=False), e.sync()))
lap("optimize_48_ms", lambda: (h.optimize("WGS-Leonardo", maxiter=48, verbose=False), e.sync()))
lap("first_read_phase_ms", lambda: h.phase)
lap("first_read_amp_ff_ms", lambda: h.amp_ff)
lap("warm_optimize_50_ms", lambda: (h.optimize("WGS-Leonardo", maxiter=50, verbose=False), e.sync()))
g = lap("second_hologram_construct_ms", lambda: SpotHologram.make_rectangular_array(shape, (32, 32), (64, 64), basis="knm", slm_shape=slm, phase=phase,
                                                                                     engine_options={L.OPT_SPARSE_COLUMNS: 0} if dense else {}))
lap("second_hologram_optimize_50_ms", lambda: (g.optimize("WGS-Leonardo", maxiter=50, verbose=False), g._engine.sync()))
out["total_first_hologram_ms"] = sum(out[k] for k in ("construct_host_ms", "load_library_ms", "engine_create_and_uploads_ms", "sync_ms",
                                                      "first_optimize_1_ms", "second_optimize_1_ms", "optimize_48_ms"))
out["process_ms"] = (time.perf_counter() - t_start) * 1e3
print(json.dumps(out))
