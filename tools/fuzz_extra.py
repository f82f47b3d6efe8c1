"""
More seeds of tests/test_fuzz_parity.py than the suite's time budget holds (other seed ranges, same bounds):
    python tools/fuzz_extra.py [+SHIFT] [large fp32 fp64 walk walk32 compressed batch]   ->  one line per case, FAILED lines for anything over its bound.
Test infrastructure (it runs the oracle through the test module), like tools/fuzz_walk.py.
"""
import os
import sys
import time
import traceback

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401  (its HIP runtime first)
import test_fuzz_parity as f  # noqa: E402

RANGES = {"large": (7100, 7130), "fp32": (6100, 6200), "fp64": (5100, 5200), "walk": (9100, 9140), "walk32": (9600, 9606),
          "compressed": (8100, 8140), "batch": (9900, 9930)}
FUNCS = {"large": f.test_random_large_case_fp32, "fp32": f.test_random_case_fp32, "fp64": f.test_random_case_fp64,
         "walk": f.test_random_operation_sequence_fp64, "walk32": f.test_random_operation_sequence_fp32_large,
         "compressed": f.test_random_compressed_case_fp64, "batch": f.test_random_batch_fp64}
argv = sys.argv[1:]
shift = 0
if argv and argv[0].startswith("+"):          # "+200": the same ranges, 200 seeds further on
    shift = int(argv.pop(0))
    RANGES = {k: (lo + shift, hi + shift) for k, (lo, hi) in RANGES.items()}
which = argv or list(RANGES)
bad = 0
for name in which:
    lo, hi = RANGES[name]
    fn = getattr(FUNCS[name], "__wrapped__", FUNCS[name])
    t0 = time.time()
    n_bad = 0
    for seed in range(lo, hi):
        try:
            fn(seed)
        except AssertionError as exc:
            n_bad += 1
            print(f"FAILED {name} [{seed}]: {str(exc)[:400]}", flush=True)
        except Exception:
            n_bad += 1
            print(f"ERROR {name} [{seed}]: {traceback.format_exc()[-600:]}", flush=True)
    bad += n_bad
    print(f"{name}: seeds {lo}..{hi - 1}: {hi - lo - n_bad} ok, {n_bad} over their bound, {time.time() - t0:.0f} s", flush=True)
print("total over their bound:", bad)
