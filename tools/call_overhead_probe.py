"""
What one engine call costs beyond its loop bodies at the headline geometry (cfg 2, dense kernels): wall time between
synchronisation points for K = 1 ... 200 bodies, fitted as intercept + slope * K.  The driver's protocol times K = 20.

    python tools/call_overhead_probe.py [out.json]
"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402,F401
from slmsuite_amd import _lib as L  # noqa: E402
from slmsuite_amd import synth  # noqa: E402
from slmsuite_amd.batch import HologramBatch  # noqa: E402
from slmsuite_amd.holography.algorithms import SpotHologram  # noqa: E402

SH, SLM = (4096, 4096), (1152, 1920)


def main():
    host = SpotHologram.make_rectangular_array(SH, (32, 32), (64, 64), basis="knm", slm_shape=SLM, phase=synth.seed_phase(2, SLM))
    hb = HologramBatch(SH, SLM, host.target, synth.seed_phase(2, SLM)[None], spot_index=host.spot_knm_rounded, spot_amp=host.spot_amp)
    out = {}
    for sparse, name in ((0, "dense"), (1, "default")):
        hb.set_option(L.OPT_SPARSE_COLUMNS, sparse)
        hb.time_iterations("WGS-Leonardo", 20)
        ks = [1, 2, 5, 10, 20, 50, 100, 200]
        wall, ev = [], []
        for k in ks:
            ws, es = [], []
            for _ in range(15):
                hb.sync()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ms = hb.time_iterations("WGS-Leonardo", k)
                hb.sync()
                torch.cuda.synchronize()
                ws.append((time.perf_counter() - t0) * 1e6)
                es.append(ms * 1e3)
            wall.append(float(np.median(ws)))
            ev.append(float(np.median(es)))
        a = np.polyfit(ks, wall, 1)
        b = np.polyfit(ks, ev, 1)
        out[name] = {"K": ks, "wall_us": wall, "event_us": ev, "wall_fit_us": {"per_body": float(a[0]), "per_call": float(a[1])},
                     "event_fit_us": {"per_body": float(b[0]), "per_call": float(b[1])}}
    txt = json.dumps(out, indent=1)
    print(txt)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
