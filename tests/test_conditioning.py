"""
How ill-conditioned is a free-phase WGS trajectory?  The CPU oracle (bit-faithful to the reference
on these inputs) is run twice, the second time with the seed phase perturbed by about one fp32 ulp.
The deviation after 30 WGS-Leonardo bodies bounds what ANY fp32 implementation can reproduce; it is
the justification of the trajectory tolerances in tests/test_gpu_parity.py (and of SURVEY 7-5).
"""
import numpy as np

from oracle import hgs_oracle as orc
from slmsuite_amd import synth


def test_one_ulp_perturbation_is_amplified():
    shape, slm = (512, 512), (144, 240)
    vec = orc.rectangular_array(shape, (16, 16), (16, 16))

    def run(ph, method):
        h = orc.OracleSpotHologram(shape, vec, slm_shape=slm, phase=ph)
        h.optimize(method, maxiter=30)
        ky, kx = h.spot_knm_rounded[1], h.spot_knm_rounded[0]
        return h.amp_ff[ky, kx].astype(float)

    p0 = synth.seed_phase(2, slm)
    rng = np.random.default_rng(1)
    p1 = (p0.astype(np.float64) * (1 + 1e-7 * rng.standard_normal(p0.shape))).astype(np.float32)
    assert 0 < np.max(np.abs(p1 - p0)) < 2.5e-7 * np.pi * 2      # at most ~2 ulp of a phase in [-pi, pi)
    dev = {}
    for method in ("WGS-Leonardo", "WGS-Kim"):
        a0, a1 = run(p0.copy(), method), run(p1.copy(), method)
        dev[method] = float(np.linalg.norm(a1 - a0) / np.linalg.norm(a0))
    # a 1e-7 relative perturbation of the input comes back 10-100x larger at the spots
    assert 5e-7 < dev["WGS-Leonardo"] < 1e-3
    # the phase-fixing variant is the stable one
    assert dev["WGS-Kim"] < dev["WGS-Leonardo"] * 2 + 1e-6
