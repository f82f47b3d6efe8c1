"""One random walk of tests/test_fuzz_parity.py::test_random_operation_sequence_fp64 step by step: tools/fuzz_walk.py SEED [TOL]"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401  (its HIP runtime first)
import test_fuzz_parity as f  # noqa: E402

for seed in sys.argv[1].split(","):
    print("seed", seed)
    try:
        print(f.walk(int(seed), tol=float(sys.argv[2]) if len(sys.argv) > 2 else 1e-9, log=print))
    except AssertionError as exc:
        print("FAILED", str(exc)[:300])
