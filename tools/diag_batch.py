"""tests/test_fuzz_parity.py::test_random_batch_fp64 for one seed, hologram by hologram: python tools/diag_batch.py SEED"""
import os
import sys
import warnings

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401
import numpy as np  # noqa: E402
import test_fuzz_parity as f  # noqa: E402
from test_fuzz_parity import L, METHODS, Hologram, synth, phase_rel_l2, _sparse_blocks  # noqa: E402
from slmsuite_amd.batch import HologramBatch  # noqa: E402

seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
dt = np.float64
H, W = int(rng.choice([64, 128, 256, 200])), int(rng.choice([64, 128, 512, 90]))
slm = (int(rng.integers(8, H + 1)), int(rng.integers(8, W + 1)))
n = int(rng.integers(2, 7))
kinds = [str(rng.choice(["image", "blocks", "blocks", "mraf"])) for _ in range(n)]
targets = []
for i, kind in enumerate(kinds):
    if kind == "blocks":
        t = _sparse_blocks(seed + 10 * i, (H, W), dt)
    else:
        t = synth.random_target(seed + 10 * i, (H, W), 0.2, 1.0, dtype=dt)
        if kind == "mraf":
            t[: max(1, H // 5), :] = np.nan
    targets.append(t)
targets = np.stack(targets)
phases = np.stack([synth.seed_phase(seed + 100 + i, slm, dtype=dt) for i in range(n)])
m, kw = METHODS[int(rng.integers(len(METHODS)))]
kw = dict(kw)
if "mraf" in kinds:
    kw["mraf_factor"] = 0.5
streams = int(rng.integers(1, 4))
sparse = int(rng.integers(2))
bodies = int(rng.integers(2, 5))
second = rng.random() < 0.5
print((H, W), slm, kinds, m, kw, "streams", streams, "sparse", sparse, "bodies", bodies, "second call", second)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for total in ([bodies, bodies + 2] if second else [bodies]):
        hb = HologramBatch((H, W), slm, targets, phases, dtype=dt, streams=streams)
        hb.set_option(L.OPT_SPARSE_COLUMNS, sparse)
        hb.optimize(m, maxiter=bodies, **kw)
        if total > bodies:
            hb.optimize(m, maxiter=2, **kw)
        got = hb.phases()
        hb.close()
        errs = []
        for i in range(n):
            h = Hologram(targets[i].copy(), phase=phases[i].copy(), slm_shape=slm, dtype=dt, engine_options={L.OPT_SPARSE_COLUMNS: sparse})
            h.optimize(m, maxiter=total, verbose=False, **kw)
            errs.append(phase_rel_l2(got[i], h.phase))
            h._release_engine()
        print("bodies", total, "per hologram:", ["%.2e" % e for e in errs])

# conditioning of the worst hologram: the same run on two engine paths and on the oracle (one call of `total` bodies)
from conftest import force_stepwise  # noqa: E402
from oracle import hgs_oracle as orc  # noqa: E402
i = int(np.argmax(errs))
total = bodies + 2 if second else bodies
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    runs = {}
    for name in ("fused", "stepwise"):
        h = Hologram(targets[i].copy(), phase=phases[i].copy(), slm_shape=slm, dtype=dt, engine_options={L.OPT_SPARSE_COLUMNS: sparse})
        if name == "stepwise":
            force_stepwise(h)
        h.optimize(m, maxiter=total, verbose=False, **kw)
        runs[name] = np.array(h.phase)
        h._release_engine()
    o = orc.OracleHologram(targets[i].copy(), phase=phases[i].copy(), slm_shape=slm, dtype=dt)
    o.optimize(m, maxiter=total, **kw)
    runs["oracle"] = np.array(o.phase)
print("hologram", i, kinds[i], "fused vs stepwise %.2e, fused vs oracle %.2e, stepwise vs oracle %.2e, batch vs oracle %.2e" % (
    phase_rel_l2(runs["fused"], runs["stepwise"]), phase_rel_l2(runs["fused"], runs["oracle"]),
    phase_rel_l2(runs["stepwise"], runs["oracle"]), phase_rel_l2(got[i], runs["oracle"])))
