"""Row launch vs number of SLM rows (4096-wide, one-row workgroups): what the partial last round costs.
python tools/row_tail_probe.py  (GPU)"""
import json, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slmsuite_amd import synth, _lib as L
from slmsuite_amd.batch import HologramBatch
from slmsuite_amd.holography.algorithms import SpotHologram

out = []
import os
for sh in ([int(x) for x in os.environ['ROWS'].split(',')] if os.environ.get('ROWS') else (512, 768, 1024, 1152, 1280, 1536)):
    shape, slm = (4096, 4096), (sh, 1920)
    host = SpotHologram.make_rectangular_array(shape, (32, 32), (64, 64), basis="knm", slm_shape=slm,
                                               phase=synth.seed_phase(2, slm), dtype=np.float32)
    ph = np.stack([synth.seed_phase(2, slm, dtype=np.float32)])
    for sparse in ((0,) if os.environ.get('DENSE_ONLY') else (0, 1)):
        hb = HologramBatch(shape, slm, host.target, ph, dtype=np.float32, device=0, spot_index=host.spot_knm_rounded,
                           spot_amp=host.spot_amp)
        hb.engine.set_option(L.OPT_SPARSE_COLUMNS, sparse)
        hb.time_iterations("WGS-Leonardo", 20)
        hb.engine.profile_enable(True)
        ms = hb.time_iterations("WGS-Leonardo", 200)
        prof = hb.engine.profile_read()
        hb.engine.profile_enable(False)
        rec = dict(slm_rows=sh, sparse=sparse, its=200 / ms * 1e3)
        for k, v in prof.items():
            if v["launches"] >= 100:
                rec[k + "_us"] = v["ms"] / v["launches"] * 1e3
        out.append(rec)
        print(json.dumps(rec), flush=True)
        hb.close()
