"""CompressedSpotHologram: host-side set-up (CPU) and GPU parity against the reference fixtures."""
import numpy as np
import pytest

from conftest import golden_names, load_golden, rel_l2, phase_rel_l2, report
from slmsuite_amd import _lib as L
from slmsuite_amd import synth
from slmsuite_amd.hardware import SimpleFourierSLM, SimpleSLM
from slmsuite_amd.holography import toolbox
from slmsuite_amd.holography.algorithms import CompressedSpotHologram

CASES = [n for n in golden_names("compressed_") if n != "compressed_helpers"]


def make_hologram(meta, gold):
    slm = SimpleSLM(tuple(meta["slm_shape"]), pitch_um=(8, 8), wav_um=0.78)
    fs = SimpleFourierSLM(slm)
    basis = meta["basis"]
    spot_amp = gold["spot_amp_in"] if "spot_amp_in" in gold else None
    h = CompressedSpotHologram(gold["spot_vectors"], basis=basis, spot_amp=spot_amp, cameraslm=fs)
    h.reset_phase(synth.seed_phase(meta["seed"], tuple(meta["slm_shape"])))
    return h


@pytest.mark.parametrize("name", CASES)
def test_compressed_host_setup_matches_reference(name):
    """unit conversion, default Zernike basis, pupil-scaled grids, target normalisation (CPU only)."""
    meta, gold = load_golden(name)
    h = make_hologram(meta, gold)
    np.testing.assert_array_equal(h.zernike_basis, gold["zernike_basis"])
    np.testing.assert_allclose(h.spot_zernike, gold["spot_zernike"], rtol=1e-12)
    np.testing.assert_allclose(h.spot_kxy, gold["spot_kxy"], rtol=1e-12, atol=1e-18)
    np.testing.assert_allclose(h._xg, gold["xg"], rtol=1e-12)
    np.testing.assert_allclose(h._yg, gold["yg"], rtol=1e-12)
    np.testing.assert_allclose(np.nan_to_num(h.target, nan=-1), np.nan_to_num(gold["target"], nan=-1), rtol=1e-6)
    assert h.shape == tuple(meta["slm_shape"]) and len(h) == meta["N"]
    with pytest.raises(NameError):
        h.get_padded_shape()
    terms, w = toolbox.zernike_monomial_weights(h.zernike_basis, h.spot_zernike)
    assert w.shape == (terms.shape[0], meta["N"])


def test_zernike_tables_match_reference():
    import json
    _, gold = load_golden("compressed_helpers")
    for j, coeffs in json.loads(str(gold["zernike_coeff_json"])).items():
        want = {tuple(int(x) for x in k.split(",")): v for k, v in coeffs.items()}
        assert toolbox.zernike_cartesian(int(j)) == want


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_compressed_matches_reference(name):
    """8 loop bodies (Kim fixing, Nogrette, 3-D, N > 256, 5-term basis, MRAF N-vector) vs the reference."""
    meta, gold = load_golden(name)
    h = make_hologram(meta, gold)
    snaps = {}

    def cb(hh):
        snaps[hh.iter] = (hh.farfield.copy(), hh.weights.copy(), hh.phase.copy())
        return False

    h.optimize(meta["method"], maxiter=meta["maxiter"], verbose=False, callback=cb, **meta["kwargs"])
    assert h.flags["feedback"] == meta["feedback"]
    worst = dict(ff=0.0, w=0.0, ph=0.0)
    for k, (ff, w, ph) in snaps.items():
        worst["ff"] = max(worst["ff"], rel_l2(ff, gold[f"ff_{k}"]))
        worst["w"] = max(worst["w"], rel_l2(np.nan_to_num(w), np.nan_to_num(gold[f"weights_{k}"])))
        if f"phase_{k}" in gold:
            worst["ph"] = max(worst["ph"], phase_rel_l2(ph, gold[f"phase_{k}"]))
    worst["ph"] = max(worst["ph"], phase_rel_l2(h.phase, gold["final_phase"]))
    report(f"compressed {name}", **worst)
    # the reference evaluates its kernels in complex64; 2e-5 on the spot amplitudes / weights,
    # 1e-4 on SLM-plane phase phasors after 8 bodies.  300 spots on a 48x64 SLM is over-determined and
    # ill-conditioned: the reference's own fp32 and fp64 runs differ by 2.6e-5 / 2.0e-5 / 5.8e-5 there.
    tol = 1e-4 if meta["N"] > 256 else 2e-5
    assert worst["ff"] < tol and worst["w"] < tol and worst["ph"] < 5 * tol
    assert rel_l2(h.amp_ff, gold["final_ampff"]) < tol
    assert [bool(x) for x in h.stats["flags"]["fixed_phase"]] == [bool(x) for x in gold["fixed_history"]]
    # fused call path (no callback) gives the same end state
    h2 = make_hologram(meta, gold)
    h2.optimize(meta["method"], maxiter=meta["maxiter"], verbose=False, **meta["kwargs"])
    assert phase_rel_l2(h2.phase, h.phase) < 1e-6


@pytest.mark.gpu
def test_compressed_double_precision_and_external_feedback():
    meta, gold = load_golden("compressed_2d50")
    slm = SimpleSLM(tuple(meta["slm_shape"]), pitch_um=(8, 8), wav_um=0.78)
    h = CompressedSpotHologram(gold["spot_vectors"], basis="kxy", cameraslm=SimpleFourierSLM(slm), dtype=np.float64)
    h.reset_phase(synth.seed_phase(meta["seed"], tuple(meta["slm_shape"]), dtype=np.float64))
    h.optimize("WGS-Kim", maxiter=8, verbose=False, fix_phase_iteration=4)
    assert phase_rel_l2(h.phase, gold["final_phase"]) < 1e-4        # fp64 engine vs fp32 reference
    h.external_spot_amp = np.ones(len(h)) * (1 + 0.1 * np.cos(np.arange(len(h))))
    h.optimize("WGS-Leonardo", maxiter=2, verbose=False, feedback="external_spot")
    assert np.all(np.isfinite(h.weights)) and abs(float(np.sum(h.weights.astype(float) ** 2)) - 1) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("D", [2, 3])
def test_separable_matrix_core_path_matches_direct_kernels(D):
    """
    Tilt (D = 2) and tilt + focus (D = 3) kernels factorise into Ex[n][x] * Ey[n][y]: both transforms run as
    complex GEMMs on the matrix cores.  Same hologram through the direct (regenerate-on-the-fly) kernels.
    Odd sizes exercise the tile edges; Kim fixing, MRAF-free WGS and a propagation kernel ride along.
    """
    slm_shape = (70, 93)
    fs = SimpleFourierSLM(SimpleSLM(slm_shape, pitch_um=(8, 8), wav_um=0.78))
    N = 137
    v = np.vstack([0.03 * (synth.uniform01(51, (N,), k) - 0.5) for k in range(2)])
    if D == 3:
        v = np.vstack((v, 4e-6 * (synth.uniform01(51, (N,), 2) - 0.5)))
    amp = 0.5 + synth.uniform01(52, (N,), 0)
    kern = (0.3 * synth.seed_phase(53, slm_shape)).astype(np.float32)

    def run(sep):
        h = CompressedSpotHologram(v, basis="kxy", spot_amp=amp, cameraslm=fs, propagation_kernel=kern,
                                   engine_options={L.OPT_SEPARABLE: int(sep)})
        h.reset_phase(synth.seed_phase(50, slm_shape))
        h.optimize("WGS-Kim", maxiter=6, verbose=False, fix_phase_iteration=3)
        return h

    a, b = run(True), run(False)
    errs = dict(ff=rel_l2(a.farfield, b.farfield), w=rel_l2(a.weights, b.weights), ph=phase_rel_l2(a.phase, b.phase))
    report(f"compressed separable (MFMA) vs direct kernels D={D}", **errs)
    assert errs["ff"] < 2e-5 and errs["w"] < 2e-5 and errs["ph"] < 5e-5
    assert a.stats["flags"]["fixed_phase"] == b.stats["flags"]["fixed_phase"]
