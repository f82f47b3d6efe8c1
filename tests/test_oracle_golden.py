"""
Pin the CPU oracle (oracle/hgs_oracle.py) against golden vectors recorded from the real reference
(tools/make_golden.py).  CPU only.  Tolerances: the oracle issues the same NumPy op sequence as the
reference, so f32 trajectories agree to a few ulp-level perturbations amplified by the loop.
"""
import numpy as np
import pytest

from conftest import golden_names, load_golden, rel_l2, phase_rel_l2
from golden_cases import hologram_inputs, spot_external_amp, spot_null_ctor
from oracle import hgs_oracle as orc


def run_with_snapshots(h, method, maxiter, **kw):
    snaps = {}
    it0 = h.iter

    def cb(hh):
        k = hh.iter - it0
        snaps[k] = dict(phase=hh.phase.copy(), weights=hh.weights.copy(),
                        phase_ff=None if hh.phase_ff is None else hh.phase_ff.copy(),
                        amp_ff=hh.amp_ff.copy(), fixed=bool(hh.flags.get("fixed_phase", False)))
        return False

    h.optimize(method, maxiter=maxiter, callback=cb, **kw)
    return snaps


def check_traj(meta, gold, h, snaps, tol):
    for k, v in gold.items():
        head, _, tail = k.rpartition("_")
        if not tail.isdigit():
            continue
        it = int(tail)
        if head == "phase":
            assert phase_rel_l2(snaps[it]["phase"], v) < tol, k
        elif head == "weights":
            assert rel_l2(snaps[it]["weights"], v) < tol, k
        elif head == "phaseff":
            assert phase_rel_l2(snaps[it]["phase_ff"], v) < tol, k
        elif head == "fixed":
            assert snaps[it]["fixed"] == bool(v), k
    assert phase_rel_l2(h.phase, gold["final_phase"]) < tol
    if "final_weights" in gold:
        assert rel_l2(h.weights, gold["final_weights"]) < tol
    if "final_ampff" in gold:
        assert rel_l2(h.amp_ff, gold["final_ampff"]) < tol
    assert [bool(x) for x in h.stats["flags"]["fixed_phase"]] == [bool(x) for x in gold["fixed_history"]]
    for k, v in gold.items():
        if k.startswith("stats_computational_") and not k.startswith("stats_computational_spot"):
            name = k[len("stats_computational_"):]
            np.testing.assert_allclose(h.stats["stats"]["computational"][name], v, rtol=2e-3, atol=1e-6)


@pytest.mark.parametrize("name", golden_names("holo_") + golden_names("mraf_"))
def test_oracle_matches_reference_hologram(name):
    meta, gold = load_golden(name)
    h = orc.OracleHologram(**hologram_inputs(meta))
    snaps = run_with_snapshots(h, meta["method"], meta["maxiter"],
                               stat_groups=["computational"], **meta["kwargs"])
    tol = 1e-9 if meta["dtype"] == "float64" else 2e-5
    check_traj(meta, gold, h, snaps, tol)
    if "final_zero_weights" in gold:
        assert rel_l2(h.zero_weights, gold["final_zero_weights"]) < tol


@pytest.mark.parametrize("name", golden_names("spot_"))
def test_oracle_matches_reference_spot(name):
    meta, gold = load_golden(name)
    shape, slm = tuple(meta["shape"]), tuple(meta["slm_shape"])
    from slmsuite_amd import synth
    vec = orc.rectangular_array(shape, tuple(meta["array_shape"]), tuple(meta["array_pitch"]))
    np.testing.assert_array_equal(vec, gold["spot_knm"])
    h = orc.OracleSpotHologram(shape, vec, slm_shape=slm, phase=synth.seed_phase(meta["seed"], slm))
    assert h.spot_integration_width_knm == meta["width"]
    np.testing.assert_array_equal(h.spot_knm_rounded, gold["spot_knm_rounded"])
    ky, kx = h.spot_knm_rounded[1], h.spot_knm_rounded[0]
    np.testing.assert_allclose(h.target[ky, kx], gold["target_spots"], rtol=1e-6)
    if meta["feedback"] == "external_spot":
        h.external_spot_amp = spot_external_amp(meta, h.spot_amp)
        np.testing.assert_allclose(h.external_spot_amp, gold["external_spot_amp"], rtol=1e-12)
    snaps = run_with_snapshots(h, meta["method"], meta["maxiter"], feedback=meta["feedback"],
                               stat_groups=meta["stat_groups"], **meta["kwargs"])
    tol = 2e-5
    for k, v in gold.items():
        if k.startswith("phase_") and k.split("_")[1].isdigit():
            assert phase_rel_l2(snaps[int(k.split("_")[1])]["phase"], v) < tol, k
        if k.startswith("weights_") and k.endswith("_spots"):
            assert rel_l2(snaps[int(k.split("_")[1])]["weights"][ky, kx], v) < tol, k
        if k.startswith("ampff_") and k.endswith("_spots"):
            assert rel_l2(snaps[int(k.split("_")[1])]["amp_ff"][ky, kx], v) < tol, k
    assert phase_rel_l2(h.phase, gold["final_phase"]) < tol
    assert rel_l2(h.amp_ff[::4, ::4], gold["final_ampff_sub"]) < tol
    assert rel_l2(h.weights[ky, kx], gold["final_weights_spots"]) < tol
    assert abs(float(np.sum(h.weights.astype(float))) - float(gold["final_weights_sum"])) < 1e-3
    assert [bool(x) for x in h.stats["flags"]["fixed_phase"]] == [bool(x) for x in gold["fixed_history"]]
    for grp in ("computational", "computational_spot"):
        for n in ("efficiency", "uniformity", "pkpk_err", "std_err"):
            np.testing.assert_allclose(h.stats["stats"][grp][n], gold[f"stats_{grp}_{n}"],
                                       rtol=2e-3, atol=1e-6)


def check_spot_traj(meta, gold, h, snaps, tol):
    ky, kx = h.spot_knm_rounded[1], h.spot_knm_rounded[0]
    for k, v in gold.items():
        if k.startswith("phase_") and k.split("_")[1].isdigit():
            assert phase_rel_l2(snaps[int(k.split("_")[1])]["phase"], v) < tol, k
        if k.startswith("weights_") and k.endswith("_spots"):
            assert rel_l2(snaps[int(k.split("_")[1])]["weights"][ky, kx], v) < tol, k
        if k.startswith("ampff_") and k.endswith("_spots"):
            assert rel_l2(snaps[int(k.split("_")[1])]["amp_ff"][ky, kx], v) < tol, k
        if k.startswith("fixed_") and k.split("_")[1].isdigit():
            assert snaps[int(k.split("_")[1])]["fixed"] == bool(v), k
    assert phase_rel_l2(h.phase, gold["final_phase"]) < tol
    assert rel_l2(h.amp_ff[::4, ::4], gold["final_ampff_sub"]) < tol
    assert rel_l2(h.weights[ky, kx], gold["final_weights_spots"]) < tol
    assert abs(float(np.sum(h.weights.astype(float))) - float(gold["final_weights_sum"])) < 1e-3
    assert [bool(x) for x in h.stats["flags"]["fixed_phase"]] == [bool(x) for x in gold["fixed_history"]]
    for grp in meta["stat_groups"]:
        for n in ("efficiency", "uniformity", "pkpk_err", "std_err"):
            np.testing.assert_allclose(h.stats["stats"][grp][n], gold[f"stats_{grp}_{n}"], rtol=2e-3, atol=1e-6)


def test_oracle_kim_fixed_by_efficiency_hologram():
    """fix_phase_efficiency (_hologram.py:1560-1569), dense Hologram: the history, every recorded state and the
    statistics that drive the decision."""
    meta, gold = load_golden("kimeff_hologram")
    h = orc.OracleHologram(**hologram_inputs(meta))
    snaps = run_with_snapshots(h, meta["method"], meta["maxiter"], stat_groups=meta["stat_groups"], **meta["kwargs"])
    assert [bool(x) for x in gold["fixed_history"]] == [False] * 5 + [True] * 4
    check_traj(meta, gold, h, snaps, 2e-5)


def test_oracle_kim_fixed_by_efficiency_spot():
    """Same gate on a SpotHologram deciding on the spot group's efficiency (crosses 0.575 at iteration 4)."""
    from slmsuite_amd import synth
    meta, gold = load_golden("kimeff_spot")
    shape, slm = tuple(meta["shape"]), tuple(meta["slm_shape"])
    vec = orc.rectangular_array(shape, tuple(meta["array_shape"]), tuple(meta["array_pitch"]))
    h = orc.OracleSpotHologram(shape, vec, slm_shape=slm, phase=synth.seed_phase(meta["seed"], slm))
    snaps = run_with_snapshots(h, meta["method"], meta["maxiter"], feedback=meta["feedback"],
                               stat_groups=meta["stat_groups"], **meta["kwargs"])
    assert [bool(x) for x in gold["fixed_history"]] == [False] * 5 + [True] * 4
    check_spot_traj(meta, gold, h, snaps, 2e-5)


@pytest.mark.parametrize("name", golden_names("spotnull_"))
def test_oracle_matches_reference_spot_null(name):
    """SpotHologram(null_vectors, null_radius, null_region, null_region_radius_frac): target raster (NaN background,
    zero disks incl. the edge-clipping rule) bit-identical, then spot feedback through the MRAF branch."""
    from slmsuite_amd import synth
    meta, gold = load_golden(name)
    shape, slm = tuple(meta["shape"]), tuple(meta["slm_shape"])
    vec = orc.rectangular_array(shape, tuple(meta["array_shape"]), tuple(meta["array_pitch"]))
    h = orc.OracleSpotHologram(shape, vec, slm_shape=slm, phase=synth.seed_phase(meta["seed"], slm),
                               **spot_null_ctor(meta, gold))
    assert h.null_radius_knm == int(gold["null_radius_knm"])
    np.testing.assert_array_equal(np.isnan(h.target), np.isnan(gold["target"]))
    np.testing.assert_array_equal(np.nan_to_num(h.target, nan=-1), np.nan_to_num(gold["target"], nan=-1))
    snaps = run_with_snapshots(h, meta["method"], meta["maxiter"], feedback=meta["feedback"],
                               stat_groups=meta["stat_groups"], **meta["kwargs"])
    check_spot_traj(meta, gold, h, snaps, 2e-5)


def test_helpers_match_reference():
    meta, gold = load_golden("helpers")
    for row, out in zip(gold["unpad_in"], gold["unpad_out"]):
        assert orc.unpad_slices((row[0], row[1]), (row[2], row[3])) == tuple(out)
    for row, out in zip(gold["padshape_in"], gold["padshape_out"]):
        assert orc.padded_shape((row[0], row[1]), int(row[2]), bool(row[3])) == tuple(out)
    for w in (1, 3, 5):
        np.testing.assert_allclose(orc.take_sum(gold["take_img"], gold["take_vec"], w), gold[f"take_w{w}"],
                                   rtol=1e-12)


def test_cfg1_summary():
    """BASELINE config 1: Hologram 512^2 random amplitude, GS x20 on the CPU path."""
    from slmsuite_amd import synth
    meta, gold = load_golden("cfg1_summary")
    shape = tuple(meta["shape"])
    h = orc.OracleHologram(synth.random_target(1, shape), phase=synth.seed_phase(1, shape), slm_shape=shape)
    h.optimize("GS", maxiter=20, stat_groups=["computational"])
    assert rel_l2(h.amp_ff[::4, ::4], gold["ampff_sub"]) < 1e-5
    assert phase_rel_l2(h.phase[::4, ::4], gold["phase_sub"]) < 1e-5
    np.testing.assert_allclose(h.stats["stats"]["computational"]["efficiency"], gold["efficiency"], rtol=1e-4)
