import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running")


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
