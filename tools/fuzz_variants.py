"""One fp32 fuzz case under variations of its ingredients: which one makes the engine's distance from the float64 run larger
than the float32 oracle's?   python tools/fuzz_variants.py SEED"""
import copy
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401
import test_fuzz_parity as f  # noqa: E402

seed = int(sys.argv[1])
base = f.draw(seed, np.float32)
print(f.describe(base))
variants = {"as drawn": {}, "no kernel": {"kernel": False}, "no amp": {"amp": False}, "no kernel, no amp": {"kernel": False, "amp": False},
            "WGS-Leonardo": {"method": "WGS-Leonardo", "kw": {}}, "Kim, never fixed": {"kw": {"fix_phase_iteration": 9}},
            "dense columns": {"sparse": 0}, "GS": {"method": "GS", "kw": {}}}
for bodies in (1, 2, 3):
    for name, ch in variants.items():
        case = copy.deepcopy(base)
        case.update(ch)
        h, o32, _ = f.run(case, np.float32, bodies)
        _, o64, _ = f.run(case, np.float64, bodies, engine=False)
        eng, ref = f.errors(case, h, o64), f.errors(case, o32, o64)
        print(f"bodies {bodies} {name:18s} " + "  ".join(f"{k} {eng[k]:.2e} / {ref[k]:.2e} = {eng[k] / max(ref[k], 1e-30):4.1f}x" for k in eng), flush=True)
        h._release_engine()
