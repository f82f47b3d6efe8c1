"""
Deterministic sweep over geometry x method x precision x path (MI355X): every case runs a few loop bodies on
the engine and on the CPU oracle from the same seeds.  fp64 pins the logic tightly for every combination; fp32
is held to the per-body tolerances of the step tests (trajectory tolerances where a dense weight update makes
the loop chaotic).  Covers combinations the targeted tests do not: the Bluestein path with MRAF / zero_factor /
Kim, batches on general shapes, odd SLM offsets, 8192-wide rows, multiplane children of different kinds of shape.
"""
import warnings

import numpy as np
import pytest

from conftest import rel_l2, phase_rel_l2, report
from oracle import hgs_oracle as orc
from slmsuite_amd import _lib as L
from slmsuite_amd import synth
from slmsuite_amd.batch import HologramBatch
from slmsuite_amd.holography.algorithms import Hologram, MultiplaneHologram, SpotHologram

pytestmark = pytest.mark.gpu

GEOMETRIES = [
    ((64, 64), (64, 64)), ((128, 256), (37, 100)), ((512, 512), (129, 250)), ((1024, 64), (600, 33)),
    ((96, 120), (40, 57)), ((101, 75), (33, 51)), ((250, 130), (250, 130)), ((65, 200), (64, 199)),
]
METHODS = [("GS", {}), ("WGS-Leonardo", {}), ("WGS-Kim", {"fix_phase_iteration": 2}), ("WGS-Nogrette", {}),
           ("WGS-Wu", {}), ("WGS-tanh", {})]


def _mraf_target(seed, shape, dtype):
    t = synth.random_target(seed, shape, 0.2, 1.0, dtype=dtype)
    h, w = shape
    t[: h // 5, :] = np.nan                       # noise region
    t[:, : w // 6] = 0                            # zero region
    return t


@pytest.mark.parametrize("gi", range(len(GEOMETRIES)))
@pytest.mark.parametrize("mi", range(len(METHODS)))
def test_engine_follows_oracle_fp64(gi, mi):
    """Three bodies in float64, array amplitude + depth kernel + statistics on odd geometries: 1e-9."""
    shape, slm = GEOMETRIES[gi]
    method, kw = METHODS[mi]
    seed = 700 + 10 * gi + mi
    dt = np.float64
    target = synth.random_target(seed, shape, dtype=dt)
    amp = synth.gaussian_amp(slm, dtype=dt)
    kern = (0.3 * synth.seed_phase(seed + 1, slm)).astype(dt)
    args = dict(amp=amp, slm_shape=slm, dtype=dt, propagation_kernel=kern)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        h = Hologram(target.copy(), phase=synth.seed_phase(seed + 2, slm, dtype=dt), **args)
        o = orc.OracleHologram(target.copy(), phase=synth.seed_phase(seed + 2, slm, dtype=dt), **args)
        h.optimize(method, maxiter=3, verbose=False, stat_groups=["computational"], **kw)
    o.optimize(method, maxiter=3, stat_groups=["computational"], **kw)
    errs = dict(phase=phase_rel_l2(h.phase, o.phase), amp_ff=rel_l2(h.amp_ff, o.amp_ff), weights=rel_l2(h.weights, o.weights))
    report(f"sweep fp64 {shape} {slm} {method}", **errs)
    assert max(errs.values()) < 1e-9, errs
    assert h.stats["flags"]["fixed_phase"] == o.stats["flags"]["fixed_phase"]
    np.testing.assert_allclose(h.stats["stats"]["computational"]["efficiency"], o.stats["stats"]["computational"]["efficiency"], rtol=1e-9)


@pytest.mark.parametrize("gi", [1, 2, 4, 5])
@pytest.mark.parametrize("method,kw", [("GS", {}), ("WGS-Leonardo", {}), ("WGS-Kim", {"fix_phase_iteration": 2})])
@pytest.mark.parametrize("flags", [{}, {"mraf_factor": 0.5}, {"mraf_factor": 0.7, "zero_factor": 1.0}])
def test_mraf_on_every_kind_of_shape_fp64(gi, method, kw, flags):
    """NaN noise region + zero region (with and without zero_weights feedback), power-of-two and Bluestein shapes."""
    shape, slm = GEOMETRIES[gi]
    seed = 900 + gi
    dt = np.float64
    target = _mraf_target(seed, shape, dt)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        h = Hologram(target.copy(), phase=synth.seed_phase(seed, slm, dtype=dt), slm_shape=slm, dtype=dt)
        o = orc.OracleHologram(target.copy(), phase=synth.seed_phase(seed, slm, dtype=dt), slm_shape=slm, dtype=dt)
        h.optimize(method, maxiter=3, verbose=False, **kw, **flags)
        o.optimize(method, maxiter=3, **kw, **flags)
    errs = dict(phase=phase_rel_l2(h.phase, o.phase), amp_ff=rel_l2(h.amp_ff, o.amp_ff),
                weights=rel_l2(np.nan_to_num(h.weights), np.nan_to_num(o.weights)))
    report(f"sweep MRAF fp64 {shape} {method} {flags}", **errs)
    assert max(errs.values()) < 1e-9, errs


@pytest.mark.parametrize("gi", [0, 2, 3, 5, 6])
@pytest.mark.parametrize("method,kw", [("GS", {}), ("WGS-Kim", {"fix_phase_iteration": 2}), ("WGS-Wu", {})])
def test_engine_follows_oracle_fp32_one_body(gi, method, kw):
    """fp32: two bodies (one with a weight update) at the per-body tolerance of the fixture step tests."""
    shape, slm = GEOMETRIES[gi]
    seed = 1100 + gi
    target = synth.random_target(seed, shape)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        h = Hologram(target.copy(), phase=synth.seed_phase(seed, slm), slm_shape=slm)
        o = orc.OracleHologram(target.copy(), phase=synth.seed_phase(seed, slm), slm_shape=slm)
        h.optimize(method, maxiter=2, verbose=False, **kw)
    o.optimize(method, maxiter=2, **kw)
    errs = dict(phase=phase_rel_l2(h.phase, o.phase), amp_ff=rel_l2(h.amp_ff, o.amp_ff), weights=rel_l2(h.weights, o.weights))
    report(f"sweep fp32 {shape} {slm} {method}", **errs)
    assert errs["phase"] < 3e-5 and errs["amp_ff"] < 1e-5 and errs["weights"] < 1e-5, errs


@pytest.mark.parametrize("shape,slm", [((100, 150), (48, 80)), ((256, 128), (60, 90))])
def test_batch_on_general_and_fast_shapes(shape, slm):
    """Three holograms with different targets in one engine (batch = 3) against three single holograms."""
    dt = np.float64
    targets = np.stack([synth.random_target(1300 + i, shape, dtype=dt) for i in range(3)])
    phases = np.stack([synth.seed_phase(1310 + i, slm, dtype=dt) for i in range(3)])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hb = HologramBatch(shape, slm, targets, phases, dtype=dt)
        try:
            hb.optimize("WGS-Leonardo", maxiter=4)
            got = hb.phases()
        finally:
            hb.close()
        for i in range(3):
            h = Hologram(targets[i].copy(), phase=phases[i].copy(), slm_shape=slm, dtype=dt)
            h.optimize("WGS-Leonardo", maxiter=4, verbose=False)
            assert phase_rel_l2(got[i], h.phase) < 1e-10, (shape, i)


def test_spot_feedback_modes_bluestein_fp64():
    """SpotHologram on a 90 x 125 grid in float64: all three feedback modes x two methods against the oracle at 1e-9."""
    shape = slm = (90, 125)
    dt = np.float64
    vec = orc.rectangular_array(shape, (5, 6), (12, 14))
    for fb in ("computational", "computational_spot", "external_spot"):
        for method, kw in (("WGS-Leonardo", {}), ("WGS-Kim", {"fix_phase_iteration": 2})):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                h = SpotHologram(shape, vec, basis="knm", slm_shape=slm, phase=synth.seed_phase(51, slm, dtype=dt), dtype=dt)
                o = orc.OracleSpotHologram(shape, vec, slm_shape=slm, phase=synth.seed_phase(51, slm, dtype=dt), dtype=dt)
                h.external_spot_amp = o.external_spot_amp = h.spot_amp * (1 + 0.1 * np.cos(np.arange(len(h))))
                h.optimize(method, maxiter=4, verbose=False, feedback=fb, stat_groups=["computational_spot"], **kw)
            o.optimize(method, maxiter=4, feedback=fb, stat_groups=["computational_spot"], **kw)
            ky, kx = h.spot_knm_rounded[1], h.spot_knm_rounded[0]
            errs = dict(phase=phase_rel_l2(h.phase, o.phase), spot_amp=rel_l2(h.amp_ff[ky, kx], o.amp_ff[ky, kx]),
                        weights=rel_l2(h.weights[ky, kx], o.weights[ky, kx]))
            assert max(errs.values()) < 1e-9, (fb, method, errs)
            np.testing.assert_allclose(h.stats["stats"]["computational_spot"]["uniformity"],
                                       o.stats["stats"]["computational_spot"]["uniformity"], rtol=1e-8)


def test_multiplane_children_of_mixed_shape_kinds_fp64():
    """One child on a power-of-two grid, one on a Bluestein grid, sharing the phase of a 40 x 57 SLM."""
    slm = (40, 57)
    dt = np.float64
    shapes = [(128, 128), (96, 120)]
    kern = (0.2 * synth.seed_phase(1402, slm)).astype(dt)
    phase0 = synth.seed_phase(1403, slm, dtype=dt)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hs = [Hologram(synth.random_target(1400, shapes[0], dtype=dt), phase=phase0.copy(), slm_shape=slm, dtype=dt),
              Hologram(synth.random_target(1401, shapes[1], dtype=dt), phase=phase0.copy(), slm_shape=slm, dtype=dt,
                       propagation_kernel=kern.copy())]
        os_ = [orc.OracleHologram(synth.random_target(1400, shapes[0], dtype=dt), phase=phase0.copy(), slm_shape=slm, dtype=dt),
               orc.OracleHologram(synth.random_target(1401, shapes[1], dtype=dt), phase=phase0.copy(), slm_shape=slm, dtype=dt,
                                  propagation_kernel=kern.copy())]
        m = MultiplaneHologram(hs, weights=[1.0, 0.7])
        mo = orc.OracleMultiplaneHologram(os_, weights=[1.0, 0.7])
        m.optimize("WGS-Leonardo", maxiter=4, verbose=False)
    mo.optimize("WGS-Leonardo", maxiter=4)
    assert phase_rel_l2(m.phase, mo.phase) < 1e-9
    for h, o in zip(hs, os_):
        assert rel_l2(h.weights, o.weights) < 1e-9
