"""
GPU parity tests (run with ``-m gpu`` on an MI355X): the HIP engine, called through the C ABI
(ctypes), against (i) the CPU oracle on the same seeded inputs and (ii) the golden fixtures
recorded from the real reference.

Tolerances (fp32 unless noted), all stated as relative L2 norms:
  * one transform (forward or inverse):           <= 2e-6   (hand-written FFT vs numpy.fft)
  * one teacher-forced loop body vs the reference: phase phasors <= 5e-6, weights <= 3e-6
    (the reference's own fp32-vs-fp64 disagreement per step is 1.6e-6 / 4e-7)
  * trajectories: GS 20 it and spot arrays <= 1e-5 on amp_ff (north_star tolerance); dense
    pixel-wise WGS is chaotic and is checked per step + through statistics instead.
  * fp64: 1e-11 per step.
"""
import numpy as np
import pytest

from conftest import dispatch_of, golden_names, load_golden, rel_l2, phase_rel_l2, report, force_stepwise
from golden_cases import hologram_inputs, spot_external_amp
from oracle import hgs_oracle as orc
from slmsuite_amd import _lib as L
from slmsuite_amd import synth
from slmsuite_amd.engine import Engine
from slmsuite_amd.holography.algorithms import Hologram, SpotHologram

pytestmark = pytest.mark.gpu


def oracle_forward(shape, slm, phase, amp=None, kernel=None, dtype=np.float32):
    h = orc.OracleHologram(shape, amp=amp, phase=phase, slm_shape=slm, dtype=dtype, propagation_kernel=kernel)
    h.nearfield2farfield()
    return h


TRANSFORM_CASES = [
    ((64, 64), (64, 64), False), ((128, 128), (48, 80), True), ((256, 512), (72, 120), False),
    ((512, 256), (100, 37), True), ((1024, 1024), (300, 500), False), ((2048, 128), (129, 65), True),
    ((4096, 4096), (1152, 1920), False), ((8192, 1024), (1152, 600), False),
]


@pytest.mark.parametrize("shape,slm,fancy", TRANSFORM_CASES)
def test_forward_transform_matches_numpy_fft(shape, slm, fancy):
    """hgs_nearfield2farfield == fftshift(fft2(fftshift(pad(amp*exp(i*phase))), 'ortho'))."""
    phase = synth.seed_phase(11, slm)
    amp = synth.gaussian_amp(slm) if fancy else None
    kern = (0.3 * synth.seed_phase(12, slm)).astype(np.float32) if fancy else None
    ref = oracle_forward(shape, slm, phase, amp, kern)
    e = Engine(shape, slm)
    if amp is not None:
        e.set(L.AMP, ref.amp)
        e.set(L.PROP_KERNEL, kern)
    e.set(L.PHASE, phase)
    e.nearfield2farfield(store_phase_ff=True)
    ff = e.get(L.FARFIELD)[0]
    report(f"forward {shape} {slm}", farfield=rel_l2(ff, ref.farfield))
    assert rel_l2(ff, ref.farfield) < 2e-6
    assert rel_l2(e.get(L.AMP_FF)[0], ref.amp_ff) < 2e-6
    # phase_ff only where the amplitude is not tiny (atan2 is ill-conditioned at speckle zeros)
    mask = ref.amp_ff > 1e-3 * ref.amp_ff.max()
    pf = e.get(L.PHASE_FF)[0]
    assert phase_rel_l2(pf[mask], np.arctan2(ref.farfield.imag, ref.farfield.real)[mask]) < 2e-4
    e.close()


@pytest.mark.parametrize("shape,slm,fancy", TRANSFORM_CASES[:7])
def test_inverse_transform_round_trip(shape, slm, fancy):
    """ifft2(fft2(nearfield)) restores the nearfield, so the extracted phase is the seed phase."""
    phase = synth.seed_phase(21, slm)
    kern = (0.3 * synth.seed_phase(22, slm)).astype(np.float32) if fancy else None
    e = Engine(shape, slm)
    mask = np.ones(slm, dtype=bool)
    if fancy:
        amp = synth.gaussian_amp(slm)
        e.set(L.AMP, amp / np.linalg.norm(amp))
        e.set(L.PROP_KERNEL, kern)
        mask = amp > 1e-2 * amp.max()     # the phase of an un-illuminated pixel is not recoverable
    e.set(L.PHASE, phase)
    e.nearfield2farfield()
    e.farfield2nearfield()
    out = e.get(L.PHASE)[0]
    err = phase_rel_l2(out[mask], phase[mask])
    report(f"roundtrip {shape} {slm}", phase=err)
    assert err < 5e-6
    e.close()


def test_double_precision_transform():
    shape, slm = (256, 256), (100, 120)
    phase = synth.seed_phase(31, slm, dtype=np.float64)
    ref = oracle_forward(shape, slm, phase, dtype=np.float64)
    e = Engine(shape, slm, dtype=np.float64)
    e.set(L.PHASE, phase)
    e.nearfield2farfield()
    assert rel_l2(e.get(L.FARFIELD)[0], ref.farfield) < 1e-13
    e.farfield2nearfield()
    assert phase_rel_l2(e.get(L.PHASE)[0], phase) < 1e-12
    e.close()


# ---- teacher-forced single steps against the reference fixtures -------------------------------------
def forced_hologram(meta, gold, k):
    """Product Hologram put into the recorded state before loop body k."""
    h = Hologram(**hologram_inputs(meta))
    if k > 0:
        h.phase = gold[f"phase_{k}"].copy()
        if f"weights_{k}" in gold:
            h.weights = gold[f"weights_{k}"].copy()
        else:
            assert k == 1          # body 0 never updates weights (_hologram.py:1552): weights_1 == weights_0
        h.iter = k
        if f"phaseff_{k}" in gold:
            h.phase_ff = gold[f"phaseff_{k}"].copy()
        h.flags["fixed_phase"] = bool(gold[f"fixed_{k}"])
        h.stats["flags"]["fixed_phase"] = [bool(x) for x in gold["fixed_history"][:k]]
        h.stats["method"] = [meta["method"]] * k
    return h


def step_pairs(gold):
    """k such that the state before body k and the phase after it (phase_{k+1}) are both recorded."""
    have_p = {int(k.split("_")[1]) for k in gold if k.startswith("phase_") and k.split("_")[1].isdigit()} | {0}
    have_w = {int(k.split("_")[1]) for k in gold if k.startswith("weights_") and k.split("_")[1].isdigit()} | {0, 1}
    return [k for k in sorted(have_p) if (k + 1) in have_p and k in have_w]


@pytest.mark.parametrize("mode", ["fused", "fused+stats", "stepwise", "callback"])
@pytest.mark.parametrize("name", golden_names("holo_"))
def test_single_step_matches_reference(name, mode):
    meta, gold = load_golden(name)
    f64 = meta["dtype"] == "float64"
    tol_p, tol_w = (1e-11, 1e-11) if f64 else (5e-6, 3e-6)
    pairs = step_pairs(gold)
    assert pairs, "fixture has no consecutive snapshots"
    for k in pairs:
        h = forced_hologram(meta, gold, k)
        kw = dict(meta["kwargs"])
        if mode == "stepwise":      # the general (materialising) operators, three engine calls per iteration from the host
            force_stepwise(h).optimize(meta["method"], maxiter=1, verbose=False, stat_groups=["computational"],
                                       callback=lambda hh: False, **kw)
        elif mode == "callback":    # a callback against the device-resident loop: one fused call per iteration
            h.optimize(meta["method"], maxiter=1, verbose=False, stat_groups=["computational"],
                       callback=lambda hh: False, **kw)
        elif mode == "fused+stats":   # device-resident loop with the statistics accumulated in the pass
            h.optimize(meta["method"], maxiter=1, verbose=False, stat_groups=["computational"], **kw)
        else:
            h.optimize(meta["method"], maxiter=1, verbose=False, **kw)
        ep = phase_rel_l2(h.phase, gold[f"phase_{k + 1}"])
        ew = rel_l2(h.weights, gold[f"weights_{k + 1}"]) if f"weights_{k + 1}" in gold else 0.0
        report(f"step {name} {mode} k={k}", phase=ep, weights=ew)
        assert ep < tol_p, (name, mode, k)
        assert ew < tol_w, (name, mode, k)
        assert bool(h.flags["fixed_phase"]) == bool(gold[f"fixed_{k + 1}"]), (name, mode, k)


@pytest.mark.parametrize("name", golden_names("holo_") + golden_names("mraf_"))
def test_first_steps_from_seed(name):
    """From the seed state the first two loop bodies are still well conditioned: compare phase_1/2."""
    meta, gold = load_golden(name)
    if meta["dtype"] != "float32":
        pytest.skip("fp64 covered by the single-step test")
    h = Hologram(**hologram_inputs(meta))
    h.optimize(meta["method"], maxiter=1, verbose=False, **meta["kwargs"])
    assert phase_rel_l2(h.phase, gold["phase_1"]) < 5e-6
    h.optimize(meta["method"], maxiter=1, verbose=False, **meta["kwargs"])
    # two bodies from the seed: the reference's own fp32-vs-fp64 gap is 1.2e-4 for MRAF + WGS
    # (unbounded (F/T)^-0.8 at speckle zeros), 2e-6 for plain WGS, 1e-6 for GS
    wgs = "WGS" in meta["method"]
    tol2 = 1e-3 if (meta["kind"] == "mraf" and wgs) else (2e-5 if wgs or meta["kind"] == "mraf" else 5e-6)
    e2 = phase_rel_l2(h.phase, gold["phase_2"])
    report(f"seed2 {name}", phase=e2)
    assert e2 < tol2
    if "weights_2" in gold:
        assert rel_l2(h.weights, gold["weights_2"]) < (1e-3 if meta["kind"] == "mraf" and wgs else 1e-5)


@pytest.mark.parametrize("name", ["holo_GS_A_f32", "holo_GS_B_f32", "holo_WGSKim_A_f32"])
def test_trajectory_matches_reference(name):
    """Whole recorded trajectory (8 bodies + populate) incl. the Kim flag history and the statistics."""
    meta, gold = load_golden(name)
    h = Hologram(**hologram_inputs(meta))
    h.optimize(meta["method"], maxiter=meta["maxiter"], verbose=False, stat_groups=["computational"],
               **meta["kwargs"])
    assert [bool(x) for x in h.stats["flags"]["fixed_phase"]] == [bool(x) for x in gold["fixed_history"]]
    # dense pixel-wise WGS is chaotic: the yardstick for the end state is how far the REFERENCE's own fp32
    # and fp64 runs of this fixture pair have drifted apart after the same 8 bodies (holo_*_f64.npz), not a constant;
    # per-step tests are the gate
    tol_p = tol_a = 2e-5
    srt = 2e-3
    if "WGS" in meta["method"]:
        _, gold64 = load_golden(name.replace("_f32", "_f64"))
        tol_p = 2 * phase_rel_l2(gold["final_phase"], gold64["final_phase"])
        tol_a = 2 * rel_l2(gold["final_ampff"], gold64["final_ampff"])
        srt = 2 * max(float(np.max(np.abs(gold[f"stats_computational_{n}"] / gold64[f"stats_computational_{n}"] - 1)))
                      for n in ("efficiency", "uniformity", "pkpk_err", "std_err"))
        report(f"trajectory {name}: 2 x the reference's fp32-fp64 drift", phase=tol_p, amp_ff=tol_a, stats_rtol=srt)
    ep, ea = phase_rel_l2(h.phase, gold["final_phase"]), rel_l2(h.amp_ff, gold["final_ampff"])
    report(f"trajectory {name}", phase=ep, amp_ff=ea)
    assert ep < tol_p and ea < tol_a, (ep, tol_p, ea, tol_a)
    for n in ("efficiency", "uniformity", "pkpk_err", "std_err"):
        np.testing.assert_allclose(h.stats["stats"]["computational"][n], gold[f"stats_computational_{n}"],
                                   rtol=srt, atol=1e-6)
    # fused mode must walk the same flag history
    h2 = Hologram(**hologram_inputs(meta))
    h2.optimize(meta["method"], maxiter=meta["maxiter"], verbose=False, **meta["kwargs"])
    assert [bool(x) for x in h2.stats["flags"]["fixed_phase"]] == [bool(x) for x in gold["fixed_history"]]
    assert phase_rel_l2(h2.phase, gold["final_phase"]) < tol_p


@pytest.mark.parametrize("mode", ["fused", "stepwise", "callback"])
@pytest.mark.parametrize("name", golden_names("mraf_"))
def test_mraf_single_steps(name, mode):
    """MRAF (NaN noise region, zero region, mraf_factor, zero_factor): the fused kernels (zero_factor forces the
    general operators) and the general path (callback) against the recorded steps."""
    meta, gold = load_golden(name)
    cb = (lambda hh: False) if mode != "fused" else None
    prep = force_stepwise if mode == "stepwise" else (lambda hh: hh)       # "callback": the device-resident loop under a callback
    h = prep(Hologram(**hologram_inputs(meta)))
    h.optimize(meta["method"], maxiter=1, verbose=False, callback=cb, **meta["kwargs"])
    assert phase_rel_l2(h.phase, gold["phase_1"]) < 5e-6
    # teacher-forced 2 -> 3 is not recorded (no phase_3); check 1 -> 2 from the recorded state
    h = prep(Hologram(**hologram_inputs(meta)))
    h.phase = gold["phase_1"].copy()
    h.iter = 1
    h.stats["flags"]["fixed_phase"] = [False]
    h.stats["method"] = [meta["method"]]
    if "zero_factor" not in meta["kwargs"]:       # zero_weights state is not part of the snapshot
        h.optimize(meta["method"], maxiter=1, verbose=False, callback=cb, **meta["kwargs"])
        ep = phase_rel_l2(h.phase, gold["phase_2"])
        ew = rel_l2(h.weights, gold["weights_2"]) if "weights_2" in gold else 0.0
        report(f"mraf step 1->2 {name} {mode}", phase=ep, weights=ew)
        if meta["method"] == "GS":
            assert ep < 5e-6 and ew < 3e-6
        else:       # pixel-wise WGS inside the signal region divides by speckle amplitudes
            assert ep < 2e-3 and ew < 1e-3


@pytest.mark.parametrize("name", golden_names("spot_"))
def test_spot_hologram_matches_reference(name):
    """SpotHologram (256^2 pad of 72x120, 8x8 spots): all feedback modes, trajectory + statistics."""
    meta, gold = load_golden(name)
    shape, slm = tuple(meta["shape"]), tuple(meta["slm_shape"])
    h = SpotHologram.make_rectangular_array(shape, tuple(meta["array_shape"]), tuple(meta["array_pitch"]),
                                            basis="knm", slm_shape=slm, phase=synth.seed_phase(meta["seed"], slm))
    assert h.spot_integration_width_knm == meta["width"]
    np.testing.assert_array_equal(h.spot_knm_rounded, gold["spot_knm_rounded"])
    if meta["feedback"] == "external_spot":
        h.external_spot_amp = spot_external_amp(meta, h.spot_amp)
    snaps = {}

    def cb(hh):
        k = hh.iter
        if f"phase_{k}" in gold:
            ky, kx = hh.spot_knm_rounded[1], hh.spot_knm_rounded[0]
            snaps[k] = (hh.phase.copy(), hh.weights[ky, kx].copy(), hh.amp_ff[ky, kx].copy())
        return False

    h.optimize(meta["method"], maxiter=meta["maxiter"], verbose=False, callback=cb, feedback=meta["feedback"],
               stat_groups=meta["stat_groups"], **meta["kwargs"])
    ky, kx = h.spot_knm_rounded[1], h.spot_knm_rounded[0]
    for k, (ph, w, a) in snaps.items():
        assert phase_rel_l2(ph, gold[f"phase_{k}"]) < 2e-5, (name, k)
        assert rel_l2(w, gold[f"weights_{k}_spots"]) < 1e-5, (name, k)
        assert rel_l2(a, gold[f"ampff_{k}_spots"]) < 1e-5, (name, k)
    assert phase_rel_l2(h.phase, gold["final_phase"]) < 2e-5
    assert rel_l2(h.amp_ff[ky, kx], gold["final_ampff_spots"]) < 1e-5      # north_star: 1e-5 on amplitude
    assert rel_l2(h.amp_ff[::4, ::4], gold["final_ampff_sub"]) < 5e-5
    assert rel_l2(h.weights[ky, kx], gold["final_weights_spots"]) < 1e-5
    assert [bool(x) for x in h.stats["flags"]["fixed_phase"]] == [bool(x) for x in gold["fixed_history"]]
    for grp in ("computational", "computational_spot"):
        for n in ("efficiency", "uniformity", "pkpk_err", "std_err"):
            np.testing.assert_allclose(h.stats["stats"][grp][n], gold[f"stats_{grp}_{n}"], rtol=2e-3, atol=2e-6)
    # the fused path (no callback / stats) must land on the same end state
    h2 = SpotHologram.make_rectangular_array(shape, tuple(meta["array_shape"]), tuple(meta["array_pitch"]),
                                             basis="knm", slm_shape=slm, phase=synth.seed_phase(meta["seed"], slm))
    if meta["feedback"] == "external_spot":
        h2.external_spot_amp = spot_external_amp(meta, h2.spot_amp)
    h2.optimize(meta["method"], maxiter=meta["maxiter"], verbose=False, feedback=meta["feedback"], **meta["kwargs"])
    assert phase_rel_l2(h2.phase, gold["final_phase"]) < 2e-5
    assert rel_l2(h2.amp_ff[ky, kx], gold["final_ampff_spots"]) < 1e-5


def test_cfg1_gs_512_matches_reference():
    """BASELINE config 1: Hologram 512^2 random amplitude, GS x20; north_star tolerance 1e-5 on amp_ff."""
    meta, gold = load_golden("cfg1_summary")
    shape = tuple(meta["shape"])
    h = Hologram(synth.random_target(1, shape), phase=synth.seed_phase(1, shape), slm_shape=shape)
    h.optimize("GS", maxiter=20, verbose=False)
    report("cfg1 GS 20 it vs reference", amp_ff=rel_l2(h.amp_ff[::4, ::4], gold["ampff_sub"]),
           phase=phase_rel_l2(h.phase[::4, ::4], gold["phase_sub"]))
    assert rel_l2(h.amp_ff[::4, ::4], gold["ampff_sub"]) < 1e-5
    assert phase_rel_l2(h.phase[::4, ::4], gold["phase_sub"]) < 2e-5
    assert abs(float(np.sqrt(np.sum(h.amp_ff.astype(float) ** 2))) - float(gold["ampff_norm"])) < 1e-5


# (BASELINE configs 2, 3 and 4 at their configured sizes: tests/test_full_configs.py)


# ---- size-independent properties at full size -------------------------------------------------------------
def test_parseval_and_linearity_full_size():
    """||farfield|| = ||nearfield|| = 1 (ortho transforms, unit-norm amp) at 4096^2 and 8192^2 pads."""
    for shape in ((4096, 4096), (8192, 8192)):
        slm = (1152, 1920)
        e = Engine(shape, slm)
        e.set(L.PHASE, synth.seed_phase(5, slm))
        e.nearfield2farfield()
        a = e.get(L.AMP_FF)[0]
        assert abs(float(np.sqrt(np.sum(a.astype(np.float64) ** 2))) - 1.0) < 2e-6
        e.farfield2nearfield()
        assert phase_rel_l2(e.get(L.PHASE)[0], synth.seed_phase(5, slm)) < 5e-6
        e.close()


def test_wgs_improves_uniformity_full_size():
    """WGS on the headline geometry: efficiency stays high and spot uniformity improves (test_gs_convergence)."""
    shape, slm = (4096, 4096), (1152, 1920)
    h = SpotHologram.make_rectangular_array(shape, (32, 32), (64, 64), basis="knm", slm_shape=slm,
                                            phase=synth.seed_phase(9, slm))
    h.optimize("WGS-Kim", maxiter=3, verbose=False, stat_groups=["computational_spot"])
    u0 = h.stats["stats"]["computational_spot"]["uniformity"][1]
    h.optimize("WGS-Kim", maxiter=25, verbose=False)
    h.optimize("WGS-Kim", maxiter=1, verbose=False, stat_groups=["computational_spot"])
    u1 = h.stats["stats"]["computational_spot"]["uniformity"][-1]
    report("cfg2-like WGS-Kim 29 it uniformity", u0=u0, u1=u1)
    assert u1 > u0 and u1 > 0.85
    assert any(h.stats["flags"]["fixed_phase"])


def test_batch_matches_single():
    """A batch engine advances independent holograms exactly like separate engines."""
    shape, slm = (256, 256), (72, 120)
    vec = orc.rectangular_array(shape, (8, 8), (16, 16))
    hs = [SpotHologram(shape, vec, basis="knm", slm_shape=slm, phase=synth.seed_phase(60 + i, slm)) for i in range(3)]
    for h in hs:
        h.optimize("WGS-Leonardo", maxiter=6, verbose=False)
    from slmsuite_amd.batch import optimize_batch
    phases = optimize_batch(shape, slm, hs[0].target, [synth.seed_phase(60 + i, slm) for i in range(3)],
                            method="WGS-Leonardo", maxiter=6)
    for i, h in enumerate(hs):
        assert phase_rel_l2(phases[i], h.phase) < 1e-6


def test_errors_are_loud():
    with pytest.raises(NotImplementedError):
        Engine((9000, 100), (50, 50))             # not a power of two and beyond the Bluestein range (8192)
    with pytest.raises(NotImplementedError):
        Engine((32768, 256), (50, 50))            # longer than the longest line one workgroup transforms
    with pytest.raises(NotImplementedError):
        Engine((16384, 256), (50, 50), dtype=np.float64)    # float64: 8192
    with pytest.raises(ValueError):
        Engine((64, 64), (128, 128))
    h = Hologram(synth.random_target(1, (64, 64)), phase=synth.seed_phase(1, (64, 64)))
    with pytest.raises(ValueError):
        h.optimize("not-a-method", maxiter=1, verbose=False)
    with pytest.raises(NotImplementedError):
        h.optimize("WGS-Leonardo", maxiter=2, verbose=False, feedback="experimental")
    # C-ABI level: bad arguments come back as status codes with a message, never as a crash
    e = Engine((64, 64), (32, 32))
    try:
        with pytest.raises(ValueError):
            e.set(L.PHASE, np.zeros((3, 3), dtype=np.float32))               # wrong size
        with pytest.raises(ValueError):
            e.set_option(99, 1)                                              # unknown option
        with pytest.raises(L.HgsError):
            e.get(L.FARFIELD)                                                # nothing materialised yet
        from slmsuite_amd.engine import make_step
        st = make_step({"method": "WGS-Leonardo", "feedback": "computational", "feedback_exponent": 0.8}, 0)
        with pytest.raises(L.HgsError):
            e.iterate(st, 1)                                                 # no target yet
        e.set(L.TARGET, synth.random_target(1, (64, 64)))
        e.reset_weights()
        e.set(L.PHASE, synth.seed_phase(1, (32, 32)))
        with pytest.raises(ValueError):
            e.iterate(st, -1)
        with pytest.raises(L.HgsError):
            e.iterate_stats(st, 2, ["computational_spot"])                   # spot statistics without spots
        hist, stats = e.iterate_stats(st, 2, ["computational"])
        assert len(hist) == 2 and 0 < stats[1]["computational"][0]["efficiency"] <= 1
        other = Engine((128, 128), (48, 48))
        try:
            with pytest.raises(ValueError):
                Engine.multiplane_farfield2nearfield([e, other], [1.0, 1.0])   # different SLM shapes
        finally:
            other.close()
    finally:
        e.close()


# ---- BASELINE config 5: mixed-region-amplitude-freedom at 8192^2, fp32 vs fp64 --------------------------
def _cfg5_target(n=8192, dtype=np.float32):
    """zeros; centred 3072^2 box = NaN (noise region); centred 2048^2 = uniform(0.2, 1) image."""
    t = np.zeros((n, n), dtype=dtype)
    a, b = n // 2 - 1536, n // 2 + 1536
    t[a:b, a:b] = np.nan
    a, b = n // 2 - 1024, n // 2 + 1024
    t[a:b, a:b] = synth.random_target(5, (2048, 2048), 0.2, 1.0, dtype=dtype)
    return t


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_cfg5_mraf_8192_steps(dtype):
    """
    Config 5 geometry (8192^2 pad of 1152x1920, MRAF mraf_factor=0.5): two loop bodies of GS and one
    weight update of WGS-Leonardo against the CPU oracle on the same inputs (per-step parity; the
    trajectory is chaotic for pixel-wise WGS on dense MRAF targets).
    """
    shape, slm = (8192, 8192), (1152, 1920)
    target = _cfg5_target(dtype=dtype)
    phase0 = synth.seed_phase(5, slm, dtype=dtype)
    tol = 5e-6 if dtype is np.float32 else 1e-11
    for method, n in (("GS", 2), ("WGS-Leonardo", 2)):
        h = Hologram(target, phase=phase0.copy(), slm_shape=slm, dtype=dtype)
        h.optimize(method, maxiter=n, verbose=False, mraf_factor=0.5)
        o = orc.OracleHologram(target, phase=phase0.copy(), slm_shape=slm, dtype=dtype)
        o.optimize(method, maxiter=n, mraf_factor=0.5, populate=False)
        ep = phase_rel_l2(h.phase, o.phase)
        ew = rel_l2(h.weights, o.weights)
        report(f"cfg5 {method} {np.dtype(dtype).name} {n} bodies vs oracle", phase=ep, weights=ew)
        if method == "GS" or dtype is np.float64:
            assert ep < (tol if method == "GS" else 1e-9) and ew < (tol if method == "GS" else 1e-9), (method, ep, ew)
        else:
            # WGS body 2 divides by speckle amplitudes inside the signal region, so two fp32 runs part ways at once.  The
            # yardstick is the algorithm's own sensitivity at this point, not a constant: the reference arithmetic's fp32
            # run against its fp64 run from the same seed (7.8e-5 here; the engine is 9.4e-5 from the fp32 oracle)
            o64 = orc.OracleHologram(target.astype(np.float64), phase=phase0.astype(np.float64), slm_shape=slm, dtype=np.float64)
            o64.optimize(method, maxiter=n, mraf_factor=0.5, populate=False)
            yp, yw = phase_rel_l2(o.phase, o64.phase), rel_l2(o.weights, o64.weights)
            report(f"cfg5 {method} float32 {n} bodies: oracle fp32 vs oracle fp64", phase=yp, weights=yw)
            # (weights: 3 x -- the fused rule's x^p runs on the 1-ulp hardware log2 / exp2)
            assert ep < 2 * yp and ew < 3 * yw, (method, ep, yp, ew, yw)


# ---- persistent-state semantics ----------------------------------------------
def test_state_persists_across_optimize_calls():
    """iter / weights / flags / stats persist; GS after WGS-Kim keeps the frozen phase (A4); reset() clears."""
    meta, gold = load_golden("holo_WGSKim_A_f32")
    kw = hologram_inputs(meta)
    h = Hologram(**kw)
    o = orc.OracleHologram(**kw)
    for method, n, extra in (("WGS-Kim", 6, dict(fix_phase_iteration=4)), ("GS", 2, {}), ("WGS-Leonardo", 1, {})):
        h.optimize(method, maxiter=n, verbose=False, **extra)
        o.optimize(method, maxiter=n, **extra)
        assert h.iter == o.iter
        assert bool(h.flags["fixed_phase"]) == bool(o.flags["fixed_phase"]), method
        assert [bool(x) for x in h.stats["flags"]["fixed_phase"]] == [bool(x) for x in o.stats["flags"]["fixed_phase"]]
        assert h.stats["method"] == o.stats["method"]
    assert h.iter == 9
    np.testing.assert_allclose(h.get_phase(), h.phase + np.pi)
    h.reset()
    assert h.iter == 0 and h.amp_ff is None and h.phase_ff is None and h.stats["method"] == []
    np.testing.assert_array_equal(h.weights, np.nan_to_num(h.target, nan=0))
    h.optimize("GS", maxiter=1, verbose=False)
    assert h.iter == 1 and h.amp_ff is not None


def test_callback_and_set_weights():
    """callback sees the mid-loop farfield and can stop the loop; user edits of weights are honoured."""
    shape = (64, 64)
    h = Hologram(synth.random_pixels_target(3, shape, 20), phase=synth.seed_phase(3, shape))
    seen = []

    def cb(hh):
        seen.append((hh.iter, float(np.sum(hh.amp_ff.astype(float) ** 2))))
        return hh.iter >= 2

    h.optimize("WGS-Leonardo", maxiter=10, verbose=False, callback=cb)
    assert [s[0] for s in seen] == [0, 1, 2] and h.iter == 2
    assert all(abs(s[1] - 1.0) < 1e-5 for s in seen)           # Parseval on the mid-loop farfield
    w = h.weights.copy()
    w[w > 0] = 1.0 / np.sqrt(np.count_nonzero(w))
    h.set_weights(w)
    h.optimize("GS", maxiter=1, verbose=False)
    np.testing.assert_allclose(h.weights, w, rtol=1e-6)
    with pytest.raises(ValueError):
        h.set_weights(np.zeros((3, 3)))


# ---- statistics computed inside the fused pass (hgs_iterate_stats) ---------------------------
STAT_NAMES = ("efficiency", "uniformity", "pkpk_err", "std_err")


def _stats_table(h, group):
    return np.array([h.stats["stats"][group][n] for n in STAT_NAMES], dtype=float)


@pytest.mark.parametrize("name", ["spot_WGSLeonardo_computational", "spot_WGSLeonardo_computational_spot",
                                  "spot_WGSKim_computational_spot"])
def test_in_pass_statistics_match_general_path_and_reference(name):
    """
    optimize(stat_groups=[...]) without a callback keeps the loop on the device: the fused column
    kernel accumulates the "computational" reductions and the spot windows read the amp_ff it
    stores.  Must agree with (i) the general path (HGS_OPT_FORCE_STEPWISE + a callback: the host-driven loop) and (ii) the statistics the
    reference recorded, and walk to the same end state.
    """
    meta, gold = load_golden(name)
    shape, slm = tuple(meta["shape"]), tuple(meta["slm_shape"])

    def make():
        return SpotHologram.make_rectangular_array(shape, tuple(meta["array_shape"]), tuple(meta["array_pitch"]),
                                                   basis="knm", slm_shape=slm,
                                                   phase=synth.seed_phase(meta["seed"], slm))
    groups = ["computational", "computational_spot"]
    h_dev, h_gen = make(), force_stepwise(make())
    h_dev.optimize(meta["method"], maxiter=meta["maxiter"], verbose=False, feedback=meta["feedback"],
                   stat_groups=groups, **meta["kwargs"])
    h_gen.optimize(meta["method"], maxiter=meta["maxiter"], verbose=False, feedback=meta["feedback"],
                   stat_groups=groups, callback=lambda hh: False, **meta["kwargs"])
    worst = 0.0
    for grp in groups:
        a, b = _stats_table(h_dev, grp), _stats_table(h_gen, grp)
        assert a.shape == b.shape == (4, meta["maxiter"])
        np.testing.assert_allclose(a, b, rtol=2e-4, atol=1e-7, err_msg=grp)
        worst = max(worst, float(np.max(np.abs(a - b) / (np.abs(b) + 1e-7))))
        for i, n in enumerate(STAT_NAMES):
            np.testing.assert_allclose(a[i], gold[f"stats_{grp}_{n}"], rtol=2e-3, atol=2e-6, err_msg=f"{grp}.{n}")
    report(f"in-pass statistics {name}", device_vs_general=worst,
           phase=phase_rel_l2(h_dev.phase, h_gen.phase))
    assert phase_rel_l2(h_dev.phase, gold["final_phase"]) < 2e-5
    assert h_dev.stats["flags"]["fixed_phase"] == h_gen.stats["flags"]["fixed_phase"]
    assert h_dev.stats["method"] == h_gen.stats["method"]


@pytest.mark.parametrize("name", ["holo_GS_A_f32", "holo_WGSKim_A_f32", "holo_GS_A_f64", "mraf_WGSLeonardo_mf0.5_zfNone",
                                  "mraf_GS_mf0.5_zfNone"])
def test_in_pass_statistics_dense_target(name):
    """Dense image targets (every pixel in the mask), fp32 and fp64, small transforms (col_fused_kernel);
    MRAF targets (NaN noise region excluded from the mask, two-pass weight update)."""
    if name not in golden_names("holo_") + golden_names("mraf_"):
        pytest.skip("fixture not recorded")
    meta, gold = load_golden(name)
    h_dev, h_gen = Hologram(**hologram_inputs(meta)), force_stepwise(Hologram(**hologram_inputs(meta)))
    n_it = 3     # dense WGS trajectories are chaotic: compare while they are still together
    h_dev.optimize(meta["method"], maxiter=n_it, verbose=False, stat_groups=["computational"], **meta["kwargs"])
    h_gen.optimize(meta["method"], maxiter=n_it, verbose=False, stat_groups=["computational"],
                   callback=lambda hh: False, **meta["kwargs"])
    a, b = _stats_table(h_dev, "computational"), _stats_table(h_gen, "computational")
    tol = 1e-9 if name.endswith("f64") else 5e-4
    np.testing.assert_allclose(a, b, rtol=tol, atol=1e-9 if name.endswith("f64") else 1e-7)
    for i, n in enumerate(STAT_NAMES):
        np.testing.assert_allclose(a[i], np.asarray(gold[f"stats_computational_{n}"])[:n_it],
                                   rtol=0.3 if "WGS" in meta["method"] else 2e-3, atol=1e-6)


def test_in_pass_statistics_full_size_and_batch():
    """cfg 2 geometry (tile-resident kernel): in-pass statistics vs hgs_stats on the materialised farfield."""
    shape, slm = (4096, 4096), (1152, 1920)
    h = SpotHologram.make_rectangular_array(shape, (32, 32), (64, 64), basis="knm", slm_shape=slm,
                                            phase=synth.seed_phase(2, slm))
    h.optimize("WGS-Leonardo", maxiter=4, verbose=False, stat_groups=["computational", "computational_spot"])
    g = force_stepwise(SpotHologram.make_rectangular_array(shape, (32, 32), (64, 64), basis="knm", slm_shape=slm,
                                                           phase=synth.seed_phase(2, slm)))
    g.optimize("WGS-Leonardo", maxiter=4, verbose=False, stat_groups=["computational", "computational_spot"],
               callback=lambda hh: False)
    for grp in ("computational", "computational_spot"):
        a, b = _stats_table(h, grp), _stats_table(g, grp)
        report(f"in-pass statistics cfg2 {grp}", max_rel=float(np.max(np.abs(a - b) / (np.abs(b) + 1e-7))))
        np.testing.assert_allclose(a, b, rtol=5e-4, atol=1e-7, err_msg=grp)
    assert phase_rel_l2(h.phase, g.phase) < 1e-4
    # uniformity must improve and stay in [0, 1]
    u = h.stats["stats"]["computational_spot"]["uniformity"]
    assert 0 < u[0] < u[-1] <= 1


def test_get_farfield_matches_reference_fixture():
    """get_farfield at other shapes / another depth / through the affine resample, recorded from the reference
    (tests/golden/get_farfield.npz, _hologram.py:853-931)."""
    meta, gold = load_golden("get_farfield")
    slm = tuple(meta["slm_shape"])
    h = Hologram(tuple(meta["shape"]), amp=gold["amp"].copy(), phase=synth.seed_phase(meta["seed"], slm), slm_shape=slm)
    aff = dict(M=gold["M"], b=gold["b"])
    kern = gold["kern"]
    cases = dict(ff_default=h.get_farfield(), ff_64x256_kern=h.get_farfield((64, 256), propagation_kernel=kern),
                 ff_256_affine=h.get_farfield((256, 256), propagation_kernel=0, affine=aff),
                 ff_128_kern_affine=h.get_farfield((128, 128), propagation_kernel=kern, affine=aff))
    errs = {k: rel_l2(v, gold[k]) for k, v in cases.items()}
    report("get_farfield vs reference fixture", **errs)
    for k, v in cases.items():
        assert v.shape == gold[k].shape and np.iscomplexobj(v)
        assert errs[k] < (1e-5 if "affine" in k else 2e-6), k


def test_get_farfield_shape_kernel_affine():
    """Hologram.get_farfield (_hologram.py:853-931): other DFT shapes, a depth kernel, the affine resample."""
    from scipy.ndimage import affine_transform
    from slmsuite_amd.holography import toolbox
    slm, shape = (72, 120), (256, 256)
    h = Hologram(synth.random_target(5, shape), phase=synth.seed_phase(5, slm), slm_shape=slm,
                 amp=synth.gaussian_amp(slm))
    h.optimize("GS", maxiter=3, verbose=False)
    kern = (0.3 * synth.seed_phase(6, slm)).astype(np.float32)

    def expected(shp, k):
        nf = toolbox.pad(h.amp * np.exp(1j * (h.phase.astype(np.float64) + k)), shp)
        return np.fft.fftshift(np.fft.fft2(np.fft.fftshift(nf), norm="ortho"))

    for shp, k in (((256, 256), 0), ((512, 1024), kern), ((128, 128), kern), ((256, 256), kern)):
        ff = h.get_farfield(shp, propagation_kernel=k)
        assert ff.shape == shp and np.iscomplexobj(ff)
        err = rel_l2(ff, expected(shp, k))
        report(f"get_farfield {shp} kernel={'array' if not np.isscalar(k) else k}", rel_l2=err)
        assert err < 2e-6
    # the default shape refreshes amp_ff / phase_ff like the reference does
    ff = h.get_farfield()
    assert rel_l2(h.amp_ff, np.abs(ff)) < 1e-6
    aff = dict(M=np.array([[1.02, 0.01], [-0.02, 0.97]]), b=np.array([3.5, -2.25]))
    out = h.get_farfield((256, 256), propagation_kernel=0, affine=aff)
    ref = affine_transform(input=expected((256, 256), 0).astype(np.complex64), matrix=aff["M"], offset=aff["b"],
                           output_shape=(256, 256), order=3, mode="constant", cval=0)
    assert rel_l2(out, ref) < 1e-5
    with pytest.raises(ValueError):
        h.get_farfield((256, 256), propagation_kernel=np.zeros((3, 3)))


# ---- sparse targets: only the columns that hold a weight / target are transformed -------------------------
@pytest.mark.parametrize("method,kw", [("WGS-Leonardo", {}), ("WGS-Kim", {"fix_phase_iteration": 3}), ("GS", {})])
def test_sparse_column_path_matches_dense_path(method, kw):
    """
    4096^2 pad, 300 spots at scattered positions (not a grid): the active-column path (default) against
    the same engine with HGS_OPT_SPARSE_COLUMNS = 0.  Kim walks dense (phase_ff stored) -> sparse (fixed phase);
    statistics ride along in both.
    """
    shape, slm = (4096, 4096), (1152, 1920)
    n = 300
    xy = np.vstack((1024 + np.floor(2048 * synth.uniform01(77, (n,), 0)), 1024 + np.floor(2048 * synth.uniform01(77, (n,), 1))))
    xy = np.unique(xy.astype(int), axis=1).astype(float)
    amp = 0.5 + synth.uniform01(78, (xy.shape[1],), 0)

    def run(sparse):
        h = SpotHologram(shape, xy, basis="knm", spot_amp=amp, slm_shape=slm, phase=synth.seed_phase(79, slm),
                         engine_options={L.OPT_SPARSE_COLUMNS: int(sparse)})
        h.optimize(method, maxiter=7, verbose=False, stat_groups=["computational"], **kw)
        d = dispatch_of(h)
        if sparse:        # per-column kernel over the list, statistics accumulated in the pass
            assert d.count("col_fused_kernel", flags=["list", "stats"], STATS=True) == 7 and d.count("col_tile_kernel") == 0, d
        else:             # dense tile-resident kernel, statistics unit
            assert d.count("col_tile_kernel", flags=["stats"], without=["list"], STATS=True) == 7 and d.count("col_fused_kernel") == 0, d
        h.optimize(method, maxiter=2, verbose=False, **kw)          # state persists; plain fused call
        d = dispatch_of(h)
        rule = 2 if method == "GS" else 1
        if sparse:
            assert d.count("col_fused_kernel", flags=["list"], STATS=False, RULE=rule) == 2, d
        else:
            assert d.count("col_tile_kernel", without=["list"], STATS=False, RULE=rule, LISTED=0) + d.count("col_tile2_kernel", without=["list"], RULE=rule) == 2, d
        return h

    a, b = run(True), run(False)
    ky, kx = a.spot_knm_rounded[1], a.spot_knm_rounded[0]
    errs = dict(phase=phase_rel_l2(a.phase, b.phase), spot_amp=rel_l2(a.amp_ff[ky, kx], b.amp_ff[ky, kx]),
                weights=rel_l2(a.weights, b.weights), amp_ff=rel_l2(a.amp_ff[::8, ::8], b.amp_ff[::8, ::8]))
    report(f"sparse vs dense column path {method}", **errs)
    assert errs["phase"] < 3e-5 and errs["spot_amp"] < 1e-5 and errs["weights"] < 2e-5 and errs["amp_ff"] < 3e-5
    assert a.stats["flags"]["fixed_phase"] == b.stats["flags"]["fixed_phase"]
    for nme in STAT_NAMES:
        np.testing.assert_allclose(a.stats["stats"]["computational"][nme][:7], b.stats["stats"]["computational"][nme][:7],
                                   rtol=2e-4, atol=1e-7)
    assert np.count_nonzero(a.weights) == xy.shape[1]


@pytest.mark.parametrize("method,feedback,kw", [
    ("WGS-Leonardo", "computational_spot", {}),
    ("WGS-Kim", "computational_spot", {"fix_phase_iteration": 3}),
    ("WGS-Nogrette", "computational_spot", {}),
    ("WGS-Leonardo", "external_spot", {}),
    ("WGS-Kim", "computational", {"fix_phase_iteration": 3}),
])
def test_sparse_spot_feedback_matches_general_path(method, feedback, kw):
    """
    Spot feedback on a 4096^2 pad (scattered spots): the sparse path -- forward transform of the spot
    columns dilated by the integration window, N-vector rule, fused constraint + inverse over the spot
    columns -- against the general (materialising) path of the same engine (HGS_OPT_SPARSE_COLUMNS = 0), with both
    statistics groups recorded in the loop and a plain call afterwards.
    """
    shape, slm = (4096, 4096), (1152, 1920)
    n = 200
    xy = np.vstack((1024 + 8 * np.floor(256 * synth.uniform01(87, (n,), 0)), 1024 + 8 * np.floor(256 * synth.uniform01(87, (n,), 1))))
    xy = np.unique(xy.astype(int), axis=1).astype(float)
    amp = 0.5 + synth.uniform01(88, (xy.shape[1],), 0)

    def run(sparse):
        h = SpotHologram(shape, xy, basis="knm", spot_amp=amp, slm_shape=slm, phase=synth.seed_phase(89, slm),
                         engine_options={L.OPT_SPARSE_COLUMNS: int(sparse)})
        if feedback == "external_spot":
            h.external_spot_amp = h.spot_amp * (1 + 0.2 * (synth.uniform01(90, (len(h),), 5) - 0.5))
        h.optimize(method, maxiter=6, verbose=False, feedback=feedback, stat_groups=["computational", "computational_spot"], **kw)
        h.optimize(method, maxiter=2, verbose=False, feedback=feedback, **kw)
        return h

    a, b = run(True), run(False)
    ky, kx = a.spot_knm_rounded[1], a.spot_knm_rounded[0]
    errs = dict(phase=phase_rel_l2(a.phase, b.phase), spot_amp=rel_l2(a.amp_ff[ky, kx], b.amp_ff[ky, kx]),
                weights=rel_l2(a.weights[ky, kx], b.weights[ky, kx]))
    report(f"sparse spot feedback {method} {feedback}", **errs)
    assert errs["phase"] < 3e-5 and errs["spot_amp"] < 1e-5 and errs["weights"] < 2e-5
    assert a.stats["flags"]["fixed_phase"] == b.stats["flags"]["fixed_phase"]
    assert np.count_nonzero(a.weights) == np.count_nonzero(b.weights) == xy.shape[1]
    for grp in ("computational", "computational_spot"):
        for nme in STAT_NAMES:
            # (pkpk_err = N (max - min) of power errors of ~1e-4 each: a difference of nearly equal numbers, the two paths'
            #  2e-7 rounding differences show up as 2e-3 of it once the run has converged)
            np.testing.assert_allclose(a.stats["stats"][grp][nme][:6], b.stats["stats"][grp][nme][:6],
                                       rtol=5e-3 if nme == "pkpk_err" else 3e-4, atol=1e-7, err_msg=f"{grp}.{nme}")


def test_batch_with_different_sparse_targets_full_size():
    """
    cfg 3 slice with per-hologram targets: three spot patterns with different active columns share one
    4096^2 engine (grid.y = hologram; per-hologram column lists and lane masks) and must match three
    single engines; WGS-Kim crosses the phase-fixing iteration.
    """
    from slmsuite_amd.batch import HologramBatch
    shape, slm = (4096, 4096), (1152, 1920)
    singles, targets, phases = [], [], []
    for i, (grid, pitch) in enumerate((((8, 8), (64, 64)), ((5, 12), (96, 40)), ((16, 3), (24, 200)))):
        h = SpotHologram.make_rectangular_array(shape, grid, pitch, basis="knm", slm_shape=slm,
                                                phase=synth.seed_phase(120 + i, slm))
        targets.append(h.target.copy())
        phases.append(synth.seed_phase(120 + i, slm))
        h.optimize("WGS-Kim", maxiter=6, verbose=False, fix_phase_iteration=3)
        singles.append(h)
    hb = HologramBatch(shape, slm, np.stack(targets), np.stack(phases))
    try:
        hb.optimize("WGS-Kim", maxiter=6, fix_phase_iteration=3)
        got = hb.phases()
        w = hb.engine.get(L.WEIGHTS)
    finally:
        hb.close()
    for i, h in enumerate(singles):
        e = phase_rel_l2(got[i], h.phase)
        report(f"batch of different sparse targets, hologram {i}", phase=e, weights=rel_l2(w[i], h.weights))
        assert e < 1e-6 and rel_l2(w[i], h.weights) < 1e-6


@pytest.mark.parametrize("shape,slm,n", [((256, 256), (72, 120), 7), ((512, 512), (100, 200), 13), ((1024, 1024), (288, 480), 30),
                                         ((2048, 2048), (1080, 1920), 45), ((1024, 2048), (500, 1000), 9)])
@pytest.mark.parametrize("method,feedback", [("WGS-Kim", "computational"), ("WGS-Leonardo", "computational_spot"),
                                             ("WGS-Nogrette", "computational")])
def test_sparse_paths_small_grids(shape, slm, n, method, feedback):
    """
    Column lists on grids where a workgroup pass handles 2 or 4 columns side by side (Ph < 4096): odd
    numbers of active columns leave lane groups past the end of the list, which must run on zeros
    and store nothing.  Sparse path vs the same engine with HGS_OPT_SPARSE_COLUMNS = 0.
    """
    xs = 8 + 4 * np.floor((shape[1] / 4 - 4) * synth.uniform01(97, (n,), 0))
    ys = 8 + 4 * np.floor((shape[0] / 4 - 4) * synth.uniform01(97, (n,), 1))
    xy = np.unique(np.vstack((xs, ys)).astype(int), axis=1).astype(float)
    kw = {"fix_phase_iteration": 2} if method == "WGS-Kim" else {}

    def run(sparse):
        h = SpotHologram(shape, xy, basis="knm", slm_shape=slm, phase=synth.seed_phase(98, slm),
                         engine_options={L.OPT_SPARSE_COLUMNS: int(sparse)})
        h.optimize(method, maxiter=5, verbose=False, feedback=feedback, stat_groups=["computational", "computational_spot"], **kw)
        h.optimize(method, maxiter=2, verbose=False, feedback=feedback, **kw)
        return h

    a, b = run(True), run(False)
    ky, kx = a.spot_knm_rounded[1], a.spot_knm_rounded[0]
    errs = dict(phase=phase_rel_l2(a.phase, b.phase), spot_amp=rel_l2(a.amp_ff[ky, kx], b.amp_ff[ky, kx]),
                weights=rel_l2(a.weights, b.weights))
    report(f"sparse small grid {shape} {method} {feedback}", **errs)
    assert errs["phase"] < 3e-5 and errs["spot_amp"] < 1e-5 and errs["weights"] < 2e-5
    assert np.count_nonzero(a.weights) == xy.shape[1]
    for grp in ("computational", "computational_spot"):
        for nme in STAT_NAMES:
            # pkpk_err / std_err of a converged array are differences of nearly equal fp32 powers
            np.testing.assert_allclose(a.stats["stats"][grp][nme][:5], b.stats["stats"][grp][nme][:5], rtol=2e-3, atol=2e-6,
                                       err_msg=f"{grp}.{nme}")



@pytest.mark.parametrize("shape,slm", [((2048, 2048), (1080, 1920)), ((1024, 2048), (288, 480)), ((2048, 1024), (500, 300))])
@pytest.mark.parametrize("method,kw", [("WGS-Leonardo", {}), ("WGS-Kim", {"fix_phase_iteration": 2})])
def test_sparse_column_path_two_and_four_lines_per_workgroup(shape, slm, method, kw):
    """
    1024- and 2048-point lines put four / two of them into one workgroup (row kernel: raw buffer accesses with the
    column mask folded into the offset; column kernel: a lane group past the end of the list): 37 scattered spots,
    active-column path vs dense kernels vs the CPU oracle over five loop bodies.
    """
    n = 37
    lo = (shape[1] // 4, shape[0] // 4)
    xy = np.vstack((lo[0] + np.floor(shape[1] // 2 * synth.uniform01(81, (n,), 0)), lo[1] + np.floor(shape[0] // 2 * synth.uniform01(81, (n,), 1))))
    xy = np.unique(xy.astype(int), axis=1).astype(float)
    amp = 0.5 + synth.uniform01(82, (xy.shape[1],), 0)
    hs = []
    for sparse in (1, 0):
        h = SpotHologram(shape, xy, basis="knm", spot_amp=amp, slm_shape=slm, phase=synth.seed_phase(83, slm),
                         engine_options={L.OPT_SPARSE_COLUMNS: sparse})
        h.optimize(method, maxiter=5, verbose=False, **kw)
        hs.append(h)
    o = orc.OracleSpotHologram(shape, xy, spot_amp=amp, slm_shape=slm, phase=synth.seed_phase(83, slm))
    o.optimize(method, maxiter=5, **kw)
    a, b = hs
    ky, kx = a.spot_knm_rounded[1], a.spot_knm_rounded[0]
    errs = dict(sparse_vs_dense=rel_l2(a.amp_ff[ky, kx], b.amp_ff[ky, kx]), phase_sd=phase_rel_l2(a.phase, b.phase),
                spot_amp=rel_l2(a.amp_ff[ky, kx], o.amp_ff[ky, kx]), weights=rel_l2(a.weights[ky, kx], o.weights[ky, kx]),
                phase=phase_rel_l2(a.phase, o.phase))
    report(f"sparse path {shape} {method}", **errs)
    assert errs["sparse_vs_dense"] < 3e-6 and errs["phase_sd"] < 1e-5
    assert errs["spot_amp"] < 1e-5 and errs["weights"] < 1e-5 and errs["phase"] < 1e-4
    assert np.count_nonzero(a.weights) == xy.shape[1]

# ---- padded shapes that are not powers of two (the reference only warns, _hologram.py:378-384) --------------
GENERAL_SHAPES = [((100, 150), (48, 80), True), ((96, 120), (96, 120), False), ((101, 75), (33, 51), True),
                  ((7, 300), (5, 121), False), ((640, 1000), (300, 500), False), ((1152, 1920), (1152, 1920), False),
                  # beyond 4096: a 16384-point line is one 1024-lane workgroup; Bluestein up to 8192 runs on it
                  ((16384, 256), (1152, 200), False), ((300, 16384), (120, 160), True), ((6000, 320), (1152, 300), False),
                  ((512, 8191), (200, 1920), False), ((16384, 4096), (1152, 1920), False)]


@pytest.mark.parametrize("shape,slm,fancy", GENERAL_SHAPES)
def test_general_shape_transforms_match_numpy_fft(shape, slm, fancy):
    """Even, odd and SLM-sized pads through the Bluestein path: forward vs NumPy, then the inverse restores the phase."""
    phase = synth.seed_phase(31, slm)
    amp = synth.gaussian_amp(slm) if fancy else None
    kern = (0.3 * synth.seed_phase(32, slm)).astype(np.float32) if fancy else None
    ref = oracle_forward(shape, slm, phase, amp, kern)
    e = Engine(shape, slm)
    if amp is not None:
        e.set(L.AMP, ref.amp)
        e.set(L.PROP_KERNEL, kern)
    e.set(L.PHASE, phase)
    e.nearfield2farfield(store_phase_ff=True)
    ff = e.get(L.FARFIELD)[0]
    err = rel_l2(ff, ref.farfield)
    assert rel_l2(e.get(L.AMP_FF)[0], ref.amp_ff) < 3e-6
    e.farfield2nearfield()
    back = phase_rel_l2(e.get(L.PHASE)[0], phase)
    report(f"general shape {shape} {slm}", farfield=err, round_trip_phase=back)
    assert err < 3e-6 and back < 2e-5
    e.close()


@pytest.mark.parametrize("shape,slm", [((8192, 300), (1152, 200)), ((4000, 260), (1000, 250))])
def test_general_shape_long_lines_float64(shape, slm):
    """float64 reaches 8192-point lines (direct) and 4096 (Bluestein): forward and round trip at 1e-12."""
    phase = synth.seed_phase(33, slm, dtype=np.float64)
    ref = oracle_forward(shape, slm, phase, dtype=np.float64)
    e = Engine(shape, slm, dtype=np.float64)
    e.set(L.PHASE, phase)
    e.nearfield2farfield()
    err = rel_l2(e.get(L.FARFIELD)[0], ref.farfield)
    e.farfield2nearfield()
    back = phase_rel_l2(e.get(L.PHASE)[0], phase)
    report(f"general shape fp64 {shape}", farfield=err, round_trip_phase=back)
    assert err < 1e-12 and back < 1e-11
    e.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("method,kw", [("GS", {}), ("WGS-Leonardo", {}), ("WGS-Kim", {"fix_phase_iteration": 3}),
                                       ("WGS-Nogrette", {})])
def test_general_shape_optimize_matches_oracle(method, kw, dtype):
    """optimize() on a 100 x 150 pad of a 48 x 80 SLM (array amp, depth kernel, statistics): engine vs oracle."""
    shape, slm = (100, 150), (48, 80)
    target = synth.random_target(41, shape, dtype=dtype)
    amp = synth.gaussian_amp(slm, dtype=dtype)
    kern = (0.3 * synth.seed_phase(42, slm)).astype(dtype)
    h = Hologram(target.copy(), amp=amp.copy(), phase=synth.seed_phase(43, slm, dtype=dtype), slm_shape=slm, dtype=dtype,
                 propagation_kernel=kern.copy())
    o = orc.OracleHologram(target.copy(), amp=amp.copy(), phase=synth.seed_phase(43, slm, dtype=dtype), slm_shape=slm,
                           dtype=dtype, propagation_kernel=kern.copy())
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        h.optimize(method, maxiter=5, verbose=False, stat_groups=["computational"], **kw)
    o.optimize(method, maxiter=5, stat_groups=["computational"], **kw)
    errs = dict(phase=phase_rel_l2(h.phase, o.phase), amp_ff=rel_l2(h.amp_ff, o.amp_ff), weights=rel_l2(h.weights, o.weights))
    report(f"general shape optimize {method} {np.dtype(dtype).name}", **errs)
    # fp64 pins the logic; in fp32 a dense pixel-wise weight update amplifies rounding within a few bodies (the
    # reference's own fp32 and fp64 runs part at the same rate), GS does not
    tol = 1e-9 if dtype == np.float64 else (2e-5 if method == "GS" else 5e-3)
    assert errs["phase"] < tol and errs["amp_ff"] < tol and errs["weights"] < tol
    assert h.stats["flags"]["fixed_phase"] == o.stats["flags"]["fixed_phase"]
    for n in STAT_NAMES:
        np.testing.assert_allclose(h.stats["stats"]["computational"][n], o.stats["stats"]["computational"][n],
                                   rtol=2e-3 if dtype == np.float32 else 1e-8, atol=1e-6)


def test_general_shape_spot_hologram_at_slm_size():
    """SpotHologram on the bare SLM grid (no padding, odd width): the three feedback modes and both statistics groups."""
    shape = slm = (90, 125)
    vec = orc.rectangular_array(shape, (5, 6), (12, 14))
    for fb in ("computational", "computational_spot", "external_spot"):
        h = SpotHologram(shape, vec, basis="knm", slm_shape=slm, phase=synth.seed_phase(51, slm))
        o = orc.OracleSpotHologram(shape, vec, slm_shape=slm, phase=synth.seed_phase(51, slm))
        h.external_spot_amp = o.external_spot_amp = h.spot_amp * (1 + 0.1 * np.cos(np.arange(len(h))))
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            h.optimize("WGS-Leonardo", maxiter=6, verbose=False, feedback=fb, stat_groups=["computational", "computational_spot"])
        o.optimize("WGS-Leonardo", maxiter=6, feedback=fb, stat_groups=["computational", "computational_spot"])
        ky, kx = h.spot_knm_rounded[1], h.spot_knm_rounded[0]
        errs = dict(phase=phase_rel_l2(h.phase, o.phase), spot_amp=rel_l2(h.amp_ff[ky, kx], o.amp_ff[ky, kx]),
                    weights=rel_l2(h.weights[ky, kx], o.weights[ky, kx]))
        report(f"general shape SpotHologram {fb}", **errs)
        assert errs["phase"] < 5e-5 and errs["spot_amp"] < 1e-5 and errs["weights"] < 2e-5
        for grp in ("computational", "computational_spot"):
            np.testing.assert_allclose(h.stats["stats"][grp]["uniformity"], o.stats["stats"][grp]["uniformity"], rtol=2e-3, atol=1e-6)


def test_roctx_ranges_can_be_switched_on():
    """HGS_OPT_ROCTX: the roctx library is resolved at run time; the operators run unchanged with ranges around them."""
    e = Engine((64, 64), (32, 32))
    try:
        e.set_option(L.OPT_ROCTX, 1)
    except NotImplementedError:
        e.close()
        pytest.skip("no roctx library on this box")
    e.set(L.PHASE, synth.seed_phase(3, (32, 32)))
    e.nearfield2farfield()
    a = e.get(L.AMP_FF)[0]
    e.set_option(L.OPT_ROCTX, 0)
    e.nearfield2farfield()
    np.testing.assert_array_equal(a, e.get(L.AMP_FF)[0])
    e.close()


def test_raw_stats_arrays():
    """raw_stats=True: per-group raw_pwr / raw_pwr_ratio (_stats.py:104-114) and raw_farfield (:193-208) per iteration."""
    shape, slm = (128, 128), (48, 80)
    vec = orc.rectangular_array(shape, (4, 4), (16, 16))
    h = SpotHologram(shape, vec, basis="knm", slm_shape=slm, phase=synth.seed_phase(61, slm))
    o = orc.OracleSpotHologram(shape, vec, slm_shape=slm, phase=synth.seed_phase(61, slm))
    amps = []
    o.optimize("WGS-Leonardo", maxiter=3, callback=lambda oo: amps.append(oo.amp_ff.astype(float).copy()) and False)
    h.optimize("WGS-Leonardo", maxiter=3, verbose=False, raw_stats=True, stat_groups=["computational", "computational_spot"])
    assert len(h.stats["raw_farfield"]) == 3
    tp = np.square(o.target.astype(float))
    tp /= np.nansum(tp)
    for k in range(3):
        fp = np.square(amps[k])
        fp /= fp.sum()
        got = h.stats["stats"]["computational"]["raw_pwr"][k]
        assert got.shape == shape and rel_l2(got, fp) < 2e-5
        ratio = h.stats["stats"]["computational"]["raw_pwr_ratio"][k]
        mask = tp != 0
        assert np.all(np.isnan(ratio[~mask])) and rel_l2(ratio[mask], fp[mask] / tp[mask]) < 2e-5
        assert rel_l2(np.abs(h.stats["raw_farfield"][k]), amps[k]) < 2e-5
        spot = h.stats["stats"]["computational_spot"]["raw_pwr"][k]
        assert spot.shape == (16,) and abs(spot.sum() - 1) < 1e-12
