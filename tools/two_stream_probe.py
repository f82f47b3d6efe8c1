"""
VERDICT r3 item 7(a): does a VALU-bound column launch of one half of a batch overlap the latency-bound row launch of the
other half when the halves run on two streams?  cfg 3 (eight holograms at 4096^2 per GPU): one engine with batch 8 against
two engines with batch 4 each (every engine owns a HIP stream; hgs_iterate only enqueues), dense kernels and engine default.

    python tools/two_stream_probe.py [out.json]
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
from slmsuite_amd.engine import make_step  # noqa: E402
from slmsuite_amd.batch import batch_flags  # noqa: E402
from slmsuite_amd.holography.algorithms import SpotHologram  # noqa: E402

SH, SLM = (4096, 4096), (1152, 1920)
K = 100


def make(n, sparse, first=0):
    host = SpotHologram.make_rectangular_array(SH, (32, 32), (64, 64), basis="knm", slm_shape=SLM, phase=synth.seed_phase(2, SLM))
    phases = np.stack([synth.seed_phase(100 + first + i, SLM) for i in range(n)])
    hb = HologramBatch(SH, SLM, host.target, phases, spot_index=host.spot_knm_rounded, spot_amp=host.spot_amp)
    hb.engine.set_option(L.OPT_SPARSE_COLUMNS, sparse)
    hb.optimize("WGS-Leonardo", 5)
    return hb


def run(batches, k, stagger=False):
    flags = batch_flags("WGS-Leonardo")
    steps = [make_step(flags, hb.iter, mraf_enabled=False) for hb in batches]
    for hb in batches:
        hb.engine.sync()
    t = time.perf_counter()
    for i, (hb, st) in enumerate(zip(batches, steps)):
        if stagger and i % 2 == 1:
            hb.engine.nearfield2farfield(False)      # about half an iteration of work ahead of the loop: anti-phase start
        hb.engine.iterate(st, k)
    for hb in batches:
        hb.engine.sync()
    return time.perf_counter() - t


def main():
    out = {}
    for sparse, name in ((0, "dense"), (1, "default")):
        one = [make(8, sparse)]
        run(one, 10)
        t1 = min(run(one, K) for _ in range(5))
        one[0].close()
        two = [make(4, sparse, 0), make(4, sparse, 4)]
        run(two, 10)
        t2 = min(run(two, K) for _ in range(5))
        t2s = min(run(two, K, stagger=True) for _ in range(5))
        # staggered: the second half starts half an iteration late (one extra row launch ahead) -- same streams
        for hb in two:
            hb.close()
        four = [make(2, sparse, 2 * i) for i in range(4)]
        run(four, 10)
        t4 = min(run(four, K) for _ in range(5))
        for hb in four:
            hb.close()
        out[name] = {"one_engine_batch8_its": 8 * K / t1, "two_engines_batch4_its": 8 * K / t2, "two_engines_staggered_its": 8 * K / t2s, "four_engines_batch2_its": 8 * K / t4}
    txt = json.dumps(out, indent=1)
    print(txt)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
