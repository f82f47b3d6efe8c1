"""
The hologram-side work of the reference's FourierSLM callers: ij-basis SpotHologram
through an analytic Fourier calibration, fourier_grid_project's set-up, and the repeated
re-optimisation with changing Zernike coefficients that wavefront_calibrate_zernike performs.
"""
import numpy as np
import pytest

from conftest import load_golden, rel_l2, phase_rel_l2, report
from oracle import hgs_oracle as orc
from slmsuite_amd import synth
from slmsuite_amd.hardware import SimpleCamera, SimpleFourierSLM, SimpleSLM
from slmsuite_amd.holography.algorithms import SpotHologram, CompressedSpotHologram


def make_fs(meta):
    slm = SimpleSLM(tuple(meta["slm_shape"]), pitch_um=(8, 8), wav_um=0.78)
    fs = SimpleFourierSLM(slm, SimpleCamera((256, 256), pitch_um=(4, 4)))
    fs.fourier_calibrate_analytic(np.array(meta["M"]), np.array(meta["b"]))
    return fs


def test_ij_basis_setup_matches_reference():
    meta, gold = load_golden("fourier_callers")
    fs = make_fs(meta)
    np.testing.assert_allclose(fs.slm.get_spot_radius_kxy(), gold["psf_kxy"], rtol=1e-12)
    h = SpotHologram((128, 128), gold["spot_ij"], basis="ij", cameraslm=fs,
                     phase=synth.seed_phase(meta["seed"], tuple(meta["slm_shape"])))
    np.testing.assert_allclose(h.spot_knm, gold["spot_knm"], rtol=1e-12)
    np.testing.assert_allclose(h.spot_kxy, gold["spot_kxy"], rtol=1e-12)
    np.testing.assert_array_equal(h.spot_knm_rounded, gold["spot_knm_rounded"])
    assert h.spot_integration_width_knm == int(gold["width"])
    assert h.spot_integration_width_ij == int(gold["width_ij"])
    np.testing.assert_allclose(fs.kxyslm_to_ijcam(h.spot_kxy), gold["spot_ij"], rtol=1e-12)
    with pytest.raises(ValueError, match="camera bounds"):
        SpotHologram((128, 128), np.array([[1.0, 100.0], [100.0, 120.0]]), basis="ij", cameraslm=fs)
    with pytest.raises(RuntimeError):
        SimpleFourierSLM(fs.slm).kxyslm_to_ijcam([0, 0])


@pytest.mark.gpu
def test_ij_basis_spot_hologram_matches_reference():
    meta, gold = load_golden("fourier_callers")
    fs = make_fs(meta)
    h = SpotHologram((128, 128), gold["spot_ij"], basis="ij", cameraslm=fs,
                     phase=synth.seed_phase(meta["seed"], tuple(meta["slm_shape"])))
    h.optimize("WGS-Kim", maxiter=6, verbose=False, feedback="computational_spot", fix_phase_iteration=3,
               stat_groups=["computational_spot"])
    ky, kx = h.spot_knm_rounded[1], h.spot_knm_rounded[0]
    errs = dict(phase=phase_rel_l2(h.phase, gold["final_phase"]), amp=rel_l2(h.amp_ff[ky, kx], gold["final_ampff_spots"]),
                weights=rel_l2(h.weights[ky, kx], gold["final_weights_spots"]))
    report("ij-basis SpotHologram WGS-Kim 6 it vs reference", **errs)
    assert errs["phase"] < 2e-5 and errs["amp"] < 1e-5 and errs["weights"] < 1e-5
    for n in ("efficiency", "uniformity", "pkpk_err", "std_err"):
        np.testing.assert_allclose(h.stats["stats"]["computational_spot"][n], gold[f"stats_{n}"], rtol=2e-3, atol=2e-6)


@pytest.mark.gpu
def test_fourier_grid_project_setup_matches_reference():
    meta, gold = load_golden("fourier_callers")
    fs = make_fs(meta)
    written = {}
    fs.slm.set_phase = lambda phase, settle=False: written.setdefault("phase", np.array(phase))
    g = fs.fourier_grid_project(array_shape=(4, 3), array_pitch=(3, 4), array_center=(2, -1), maxiter=2, verbose=False)
    assert tuple(g.shape) == tuple(int(x) for x in gold["grid_shape"])
    np.testing.assert_allclose(g.spot_knm, gold["grid_spot_knm"], rtol=1e-12)       # orientation check: 2 spots dropped
    np.testing.assert_allclose(g.spot_ij, gold["grid_spot_ij"], rtol=1e-12)
    assert [g.spot_integration_width_knm, g.spot_integration_width_ij] == [int(x) for x in gold["grid_width"]]
    np.testing.assert_array_equal(np.array(np.nonzero(g.target)), gold["grid_target_nonzero"])
    np.testing.assert_allclose(g.target[np.nonzero(g.target)], gold["grid_target_values"], rtol=1e-6)
    assert g.iter == 2 and written["phase"].shape == fs.slm.shape
    np.testing.assert_allclose(written["phase"], g.get_phase())
    with pytest.warns(UserWarning, match="Unexpected argument"):
        fs.fourier_grid_project(array_shape=2, array_pitch=4, maxiter=1, verbose=False, bogus=1)


@pytest.mark.gpu
def test_wavefront_calibration_reoptimisation_pattern():
    """
    wavefront_calibrate_zernike (cameraslms.py:1840-1930): a CompressedSpotHologram over a 10-term
    Zernike basis is re-optimised ("GS", 3 it) again and again while spot_zernike changes, with
    set_weights / computational_spot statistics in between.  Every round vs the oracle.
    """
    slm_shape = (48, 64)
    fs = SimpleFourierSLM(SimpleSLM(slm_shape, pitch_um=(8, 8), wav_um=0.78))
    basis = np.array([2, 1, 4, 3, 5, 7, 8, 6, 9, 12])
    N = 12
    z = np.zeros((len(basis), N))
    z[:2] = 30 * (synth.uniform01(41, (2, N), 0) - 0.5)          # tilts, Zernike radians
    z[2:] = 1.0 * (synth.uniform01(42, (len(basis) - 2, N), 0) - 0.5)
    phase0 = synth.seed_phase(40, slm_shape)
    h = CompressedSpotHologram(z.copy(), basis=basis, cameraslm=fs)
    h.reset_phase(phase0)
    o = orc.OracleCompressedSpotHologram(z.copy(), h._xg, h._yg, zernike_basis=basis, phase=phase0.copy())
    h.optimize("GS", maxiter=3, verbose=False, stat_groups=["computational_spot"])
    o.optimize("GS", maxiter=3, stat_groups=["computational_spot"])
    worst = phase_rel_l2(h.get_phase() - np.pi, o.phase)
    for rnd in range(3):
        z[2 + rnd, :] += 0.5                                      # perturb one aberration for all spots
        z[3, rnd] -= 0.7                                          # and one coefficient of one spot
        h.spot_zernike = z.copy()
        o.spot_zernike = z.copy()
        o._kernel = None
        if rnd == 1:
            w = (1 + 0.1 * synth.uniform01(43, (N,), 0)).astype(np.float32)
            h.set_weights(w / np.linalg.norm(w))
            o.weights = (w / np.linalg.norm(w)).astype(np.float32)
        h.optimize("GS", maxiter=3, verbose=False)
        o.optimize("GS", maxiter=3)
        e = phase_rel_l2(h.get_phase() - np.pi, o.phase)
        worst = max(worst, e, rel_l2(h.farfield, o.farfield))
    report("wavefront-calibration re-optimisation pattern (D=10, 4 rounds)", worst=worst)
    assert worst < 3e-5
