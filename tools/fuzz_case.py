"""Where a fuzz case's distance comes from: tools/fuzz_case.py SEED [bodies] -- per-pixel weight / farfield differences of the
engine (fp32) and of the fp32 oracle against the fp64 oracle (tests/test_fuzz_parity.py), largest contributors first."""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401  (its HIP runtime first)
import test_fuzz_parity as f  # noqa: E402

seed = int(sys.argv[1])
bodies = int(sys.argv[2]) if len(sys.argv) > 2 else 2
case = f.draw(seed, np.float32)
print(f.describe(case))
h, o32, _ = f.run(case, np.float32, bodies)
_, o64, _ = f.run(case, np.float64, bodies, engine=False)
for name in ("weights", "amp_ff"):
    t = np.nan_to_num(np.asarray(getattr(o64, name), dtype=np.float64))
    for who, x in (("engine", h), ("oracle32", o32)):
        d = np.nan_to_num(np.asarray(getattr(x, name), dtype=np.float64)) - t
        tot = np.sum(d * d)
        idx = np.argsort((d * d).ravel())[::-1][:6]
        print(f"{name:8s} {who:9s} rel L2 {np.sqrt(tot) / np.linalg.norm(t):.3e}; top pixels (share of the squared distance, value, truth, target):")
        for i in idx:
            r, c = np.unravel_index(i, d.shape)
            print(f"    ({r:4d},{c:4d}) {d[r, c] ** 2 / tot:6.3f}  {np.asarray(getattr(x, name))[r, c]: .6e}  {t[r, c]: .6e}  T={np.asarray(o64.target)[r, c]: .4e}  |F|32={np.asarray(o32.amp_ff)[r, c]: .3e}")
