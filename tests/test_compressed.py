"""CompressedSpotHologram: host-side set-up (CPU) and GPU parity against the reference fixtures."""
import numpy as np
import pytest

from conftest import dispatch_of, golden_names, load_golden, rel_l2, phase_rel_l2, report
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
        d = dispatch_of(h)
        if sep:
            assert d.families() == {"cgemm_streamk"} and d.count("cgemm_streamk") == 12, d
        else:
            assert d.count("cgemm_streamk") == 0 and d.count("c_n2f_run", DEG=D - 1) == 6 and d.count("c_f2n_run", DEG=D - 1) == 6, d
        return h

    a, b = run(True), run(False)
    errs = dict(ff=rel_l2(a.farfield, b.farfield), w=rel_l2(a.weights, b.weights), ph=phase_rel_l2(a.phase, b.phase))
    report(f"compressed separable (MFMA) vs direct kernels D={D}", **errs)
    assert errs["ff"] < 2e-5 and errs["w"] < 2e-5 and errs["ph"] < 5e-5
    assert a.stats["flags"]["fixed_phase"] == b.stats["flags"]["fixed_phase"]


def _kernel_phase64(h, orc):
    """phi_n(p) in float64 for every spot and pixel, from the oracle's monomial tables."""
    terms, wts = orc.monomial_weights(h.zernike_basis, h.spot_zernike)
    x = np.asarray(h._xg, dtype=np.float64).ravel()
    y = np.asarray(h._yg, dtype=np.float64).ravel()
    phi = np.zeros((wts.shape[1], x.size))
    for m, (px, py) in enumerate(terms):
        if px < 0:       # vortex plate pseudo-term (-1, 0): positive charges only (phase.py:1783-1790)
            phi += np.where(wts[m] > 0, wts[m], 0.0)[:, None] * np.arctan2(y, x)[None, :]
            continue
        phi += wts[m][:, None] * (x ** int(px) * y ** int(py))[None, :]
    return phi


@pytest.mark.gpu
@pytest.mark.parametrize("basis", ["kxy2", "kxy3", "zern5"])
@pytest.mark.parametrize("slm_shape", [(70, 93), (64, 128)])
def test_run_kernels_against_float64_sums(basis, slm_shape):
    """
    The direct transforms of a regular pixel grid advance exp(i phi) along runs of 16 pixels by recurrence (c_n2f_run /
    c_f2n_run; degree 1: tilts, degree 2: focus and astigmatisms, the oblique one is what breaks the matrix-core form).
    Both directions against float64 direct summation over every spot and pixel, next to the per-pixel kernels
    (HGS_OPT_RUN_KERNELS = 0) on the same inputs: widths that are not a multiple of the run, more spots than one chunk
    of 64, array amplitude and a propagation kernel ride along.
    """
    from oracle import hgs_oracle as orc
    fs = SimpleFourierSLM(SimpleSLM(slm_shape, pitch_um=(8, 8), wav_um=0.78))
    N = 137
    v = np.vstack([0.03 * (synth.uniform01(61, (N,), k) - 0.5) for k in range(2)])
    b = "kxy"
    if basis != "kxy2":
        v = np.vstack((v, 4e-6 * (synth.uniform01(61, (N,), 2) - 0.5)))
    if basis == "zern5":
        z, _ = toolbox.convert_vector_zernike(v, "kxy", fs)
        v = np.vstack([z, np.pi * (2 * synth.uniform01(61, (2, N), 3) - 1)])
        b = np.array([2, 1, 4, 3, 5])
    amp = 0.5 + synth.uniform01(62, (N,), 0)
    kern = (0.3 * synth.seed_phase(63, slm_shape)).astype(np.float32)
    phase0 = synth.seed_phase(60, slm_shape)
    S = slm_shape[0] * slm_shape[1]

    errs = {}
    for run in (1, 0):
        h = CompressedSpotHologram(v, basis=b, spot_amp=amp, cameraslm=fs, propagation_kernel=kern,
                                   engine_options={L.OPT_SEPARABLE: 0, L.OPT_RUN_KERNELS: run})
        h.reset_phase(phase0)
        h.optimize("WGS-Leonardo", maxiter=2, verbose=False)          # weights away from the target
        d = dispatch_of(h)
        deg = 1 if basis == "kxy2" else 2
        if run:
            assert d.count("c_n2f_run", DEG=deg) == 2 and d.count("c_f2n_run", DEG=deg) == 2 and len(d.families()) == 2, d
        else:
            assert d.count("c_n2f_partial", R="float", DEG=deg) == 2 and d.count("c_f2n", R="float", DEG=deg) == 2 and len(d.families()) == 2, d
        h.phase = phase0
        e = h._get_engine()
        phi = _kernel_phase64(h, orc)
        amp_nf = np.full(S, float(h.amp)) if np.isscalar(h.amp) else np.asarray(h.amp, dtype=np.float64).ravel()
        e.nearfield2farfield()
        ff = e.get(L.FARFIELD)[0].astype(np.complex128)
        nf = amp_nf * np.exp(1j * (phase0.astype(np.float64).ravel() + kern.astype(np.float64).ravel()))
        ref = np.sum(nf[None, :] * np.exp(-1j * phi), axis=1)
        ref /= np.sqrt(np.sum(np.abs(ref) ** 2))
        err_ff = rel_l2(ff, ref)
        st = h._make_step()
        e.farfield_constraint(st)
        ffc = e.get(L.FARFIELD)[0].astype(np.complex128)
        e.farfield2nearfield()
        ph = e.get(L.PHASE)[0].astype(np.float64).ravel()
        nfb = np.sum(ffc[:, None] * np.exp(1j * phi), axis=0)
        err_ph = phase_rel_l2(ph, np.angle(nfb) - kern.astype(np.float64).ravel())
        errs[run] = (err_ff, err_ph)
        h._release_engine()
    report(f"compressed run kernels {basis} {slm_shape} vs float64 sums", run_ff=errs[1][0], run_phase=errs[1][1],
           per_pixel_ff=errs[0][0], per_pixel_phase=errs[0][1])
    assert errs[1][0] < 2e-6 and errs[1][1] < 1e-5
    assert errs[0][0] < 2e-5 and errs[0][1] < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("ansi", [[2, 1, 4, 3, 5, 7, 8, 6, 9, 12], [2, 1, 4, -1]])
def test_monomial_table_kernels_against_float64_sums(ansi, dtype, monkeypatch):
    """
    Bases beyond degree 2 (and the vortex pseudo-term) run the per-pixel kernels; with at most 16 monomials they form the
    monomial values of a lane's pixels once, ahead of the spot loop (c_n2f_partial / c_f2n DEG = 3, MonoTab), instead of
    walking the exponent list for every spot and pixel (DEG = 0, HGS_MONO_TAB=0 at hgs_create) -- the basis
    wavefront_calibrate_zernike re-optimises (cameraslms.py:1840-1930): 16 spots x 10 terms.  Both forms against float64
    direct sums and against each other (the same products in the same order: they may differ by contraction only).
    """
    from oracle import hgs_oracle as orc
    slm_shape = (70, 93)
    fs = SimpleFourierSLM(SimpleSLM(slm_shape, pitch_um=(8, 8), wav_um=0.78))
    N = 21
    z = np.zeros((len(ansi), N))
    z[:2] = 60 * (synth.uniform01(71, (2, N), 0) - 0.5)
    z[2:] = 1.5 * (synth.uniform01(72, (len(ansi) - 2, N), 0) - 0.5)          # (vortex row: charges of both signs, negative ones are ignored)
    amp = 0.5 + synth.uniform01(73, (N,), 0)
    phase0 = synth.seed_phase(70, slm_shape).astype(dtype)
    S = slm_shape[0] * slm_shape[1]
    out, errs = {}, {}
    for tab in ("1", "0"):
        monkeypatch.setenv("HGS_MONO_TAB", tab)
        h = CompressedSpotHologram(z.copy(), basis=np.array(ansi), spot_amp=amp, cameraslm=fs, dtype=dtype)
        h.reset_phase(phase0)
        h.optimize("WGS-Leonardo", maxiter=3, verbose=False)
        d = dispatch_of(h)
        deg = 3 if tab == "1" else 0
        assert d.count("c_n2f_partial", DEG=deg) >= 3 and d.count("c_f2n", DEG=deg) >= 3 and d.families() == {"c_n2f_partial", "c_f2n"}, d
        out[tab] = (h.phase.copy(), np.array(h.weights, copy=True), np.array(h.farfield, copy=True))
        h.phase = phase0
        e = h._get_engine()
        phi = _kernel_phase64(h, orc)
        e.nearfield2farfield()
        ff = e.get(L.FARFIELD)[0].astype(np.complex128)
        nf = float(h.amp) * np.exp(1j * phase0.astype(np.float64).ravel()) if np.isscalar(h.amp) else \
            np.asarray(h.amp, dtype=np.float64).ravel() * np.exp(1j * phase0.astype(np.float64).ravel())
        ref = np.sum(nf[None, :] * np.exp(-1j * phi), axis=1)
        ref /= np.sqrt(np.sum(np.abs(ref) ** 2))
        errs[tab] = rel_l2(ff, ref)
        h._release_engine()
    between = dict(phase=phase_rel_l2(out["1"][0], out["0"][0]), weights=rel_l2(out["1"][1], out["0"][1]), ff=rel_l2(out["1"][2], out["0"][2]))
    report(f"compressed monomial-table kernels {ansi} {np.dtype(dtype).name}", table_ff=errs["1"], list_ff=errs["0"], **between)
    tol = 3e-5 if dtype == np.float32 else 1e-10
    assert errs["1"] < tol and errs["0"] < tol, errs
    assert max(between.values()) < (2e-5 if dtype == np.float32 else 1e-11), between
