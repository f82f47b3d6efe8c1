import json
import os
import sys

import numpy as np
import pytest

try:        # before any engine is loaded: torch's bundled HIP runtime has to open the GPU first (slmsuite_amd._lib.load),
    import torch  # noqa: F401  and a test module that imports torch later would find "No HIP GPUs are available"
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: the large randomised cases (seconds of CPU oracle each): run LAST, and only while the "
                                       "session is inside its time budget")


# ---- time budget of the GPU suite ------------------------------------------------------------------------------------------
# The driver gives `pytest -m gpu` 1,200 s and counts a killed run as untested.  Tests marked `slow` (the twenty large cases of
# tests/test_fuzz_parity.py: random geometries at 2048 .. 8192 points per axis, seconds of float64 CPU oracle each, 190 s together)
# are moved to the end of the session and each of them starts only while the session has used less than HGS_TEST_BUDGET_S
# seconds (default 330; 0 = no limit -- how the builder runs them): every fixture, configured-workload and kernel-level test
# always runs, the randomised large cases fill what is left and are reported as skipped with the reason beyond it.
_T0 = [None]
_BUDGET_SKIPS = []


def pytest_collection_modifyitems(config, items):
    items.sort(key=lambda it: 1 if it.get_closest_marker("slow") else 0)         # (stable: file order inside both groups)


def pytest_runtest_setup(item):
    import time
    if _T0[0] is None:
        _T0[0] = time.time()
    budget = float(os.environ.get("HGS_TEST_BUDGET_S", "330"))
    if budget > 0 and item.get_closest_marker("slow") and time.time() - _T0[0] > budget:
        _BUDGET_SKIPS.append(item.nodeid)
        pytest.skip(f"time budget: {time.time() - _T0[0]:.0f} s of the session used (HGS_TEST_BUDGET_S = {budget:g}); "
                    "`slow` cases run only inside it")


def pytest_terminal_summary(terminalreporter):
    if _BUDGET_SKIPS:
        terminalreporter.write_line(f"{len(_BUDGET_SKIPS)} slow case(s) not started, session over its time budget: "
                                    + ", ".join(n.split('::')[-1] for n in _BUDGET_SKIPS))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    return meta, {k: z[k] for k in z.files if k != "meta"}


def golden_names(prefix):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))


def rel_l2(a, b):
    cplx = np.iscomplexobj(a) or np.iscomplexobj(b)
    a = np.asarray(a, dtype=np.complex128 if cplx else np.float64)
    b = np.asarray(b, dtype=np.complex128 if cplx else np.float64)
    d = np.sqrt(np.nansum(np.abs(a - b) ** 2))
    n = np.sqrt(np.nansum(np.abs(b) ** 2))
    return float(d / n) if n > 0 else float(d)


def phase_rel_l2(a, b):
    """Distance between phase maps as unit phasors (immune to +-pi wrap)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.sqrt(np.mean(np.abs(np.exp(1j * a) - np.exp(1j * b)) ** 2)))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def report(name, **values):
    """Append measured parity errors to gpurun_out/parity_report.jsonl (cited in DESIGN.md)."""
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_report.jsonl"), "a") as f:
            f.write(json.dumps({"case": name, **{k: float(v) for k, v in values.items()}}) + "\n")
    except OSError:
        pass


# ---- dispatch record (hgs_dispatch_read): which kernel instance a policy actually reached ---------------------------------
class Dispatch:
    """
    The launches of one engine since the previous read.  ``count(family, flags=..., without=..., **args)`` sums the
    launches of the instances of ``family`` whose template arguments equal ``args`` (compared as text: N=4096,
    RULE=1, PREF="true") and whose run-time flags include ``flags`` and none of ``without``.
    """

    def __init__(self, records):
        self.records = records

    def count(self, family, flags=(), without=(), **args):
        n = 0
        for r in self.records:
            if r["kernel"] != family:
                continue
            if any(r["args"].get(k) != _targ(v) for k, v in args.items()):
                continue
            if not set(flags) <= r["flags"] or (set(without) & r["flags"]):
                continue
            n += r["count"]
        return n

    def families(self):
        return {r["kernel"] for r in self.records}

    def __repr__(self):
        return "\n".join(f"{r['name']} x{r['count']}" for r in self.records) or "(no launches recorded)"


def _targ(v):
    return {True: "true", False: "false"}.get(v, str(v)) if isinstance(v, bool) else str(v)


def force_stepwise(h):
    """The general (materialising) operators for this hologram's loops: HGS_OPT_FORCE_STEPWISE in its engine options, which
    also keeps optimize(callback=...) on the host-driven loop of three engine calls per iteration (rounds 1 - 4: every
    callback did; since round 5 a callback runs against the device-resident loop)."""
    from slmsuite_amd import _lib as L
    h.engine_options[L.OPT_FORCE_STEPWISE] = 1
    if getattr(h, "_engine", None) is not None:
        h._engine.set_option(L.OPT_FORCE_STEPWISE, 1)
    return h


def dispatch_of(h):
    """Dispatch record of a hologram's (or batch's, or bare) engine; reading clears it."""
    e = getattr(h, "_engine", None) or getattr(h, "engine", None) or h
    return Dispatch(e.dispatch_read())
