"""
BASELINE.json configurations at their configured sizes (MI355X): cfg 2 through BOTH column paths (the
engine default over the active-column list AND the dense kernels the bench headline times), its error
distribution over eight seeds next to the reference's own sensitivity to a one-ulp change of the input,
cfg 3 (eight holograms per engine at 4096^2), cfg 4 (CompressedSpotHologram, 1e4 spots at 1152 x 1920,
checked against float64 direct summation on a sample of spots / pixels).
"""
import numpy as np
import pytest

from conftest import dispatch_of, load_golden, rel_l2, phase_rel_l2, report
from oracle import hgs_oracle as orc
from slmsuite_amd import _lib as L
from slmsuite_amd import synth
from slmsuite_amd.batch import HologramBatch
from slmsuite_amd.hardware import SimpleFourierSLM, SimpleSLM
from slmsuite_amd.holography.algorithms import CompressedSpotHologram, SpotHologram

pytestmark = pytest.mark.gpu

SHAPE, SLM = (4096, 4096), (1152, 1920)
PATHS = {"default": {}, "dense": {L.OPT_SPARSE_COLUMNS: 0}, "dense-per-column": {L.OPT_SPARSE_COLUMNS: 0, L.OPT_TILE_KERNEL: 0}}


def assert_cfg2_column_path(h, path, update=True):
    """
    The three policies of PATHS must reach three different kernels -- they agree to the last bit on the cfg 2 grid, so the
    numbers alone would not notice a dispatcher that routed one policy to another's kernel.
      default           col_fused_kernel over the active-column list (rule compiled in), masked row kernel
      dense             col_tile2_kernel (plain passes without a stored farfield phase: half-width tile, three workgroups per CU)
                        or col_tile_kernel with the dense schedule compiled in (LISTED = 0; WGS-Kim), prefetching shifted row kernel
      dense-per-column  col_fused_kernel, every column, no list
    """
    d = dispatch_of(h)
    rule = 1 if update else 2
    if path == "default":
        assert d.count("col_fused_kernel", flags=["list"], N=4096, RULE=rule) > 0, d
        assert d.count("col_tile_kernel") == 0 and d.count("col_fused_kernel", without=["list"]) == 0, d
        assert d.count("row_kernel", flags=["load_mask", "store_mask"], MODE=2) > 0 or d.count("row_kernel", MODE=2) == 0, d
        assert d.count("row_kernel", PREF=True) == 0, d
    elif path == "dense":
        assert d.count("col_tile_kernel", without=["list"], R="float", N=4096, NR=5, RULE=rule, LISTED=0, STATS=False, EXTRAS=False) + \
            d.count("col_tile2_kernel", flags=["xmap"], without=["list", "batch"], R="float", N=4096, NR=5, RULE=rule) > 0, d
        assert d.count("col_fused_kernel") == 0 and d.count("col_tile_kernel", flags=["list"]) == 0, d
        n2 = d.count("row_kernel", MODE=2)
        assert n2 == 0 or d.count("row_kernel", MODE=2, NS=8, PREF=True, SPLIT=False, without=["load_mask", "store_mask"]) == n2, d
    else:
        assert d.count("col_fused_kernel", without=["list"], N=4096, RULE=rule) > 0, d
        assert d.count("col_tile_kernel") == 0 and d.count("col_fused_kernel", flags=["list"]) == 0, d
    return d


def cfg2_hologram(seed, path, phase=None):
    return SpotHologram.make_rectangular_array(SHAPE, (32, 32), (64, 64), basis="knm", slm_shape=SLM,
                                               phase=synth.seed_phase(seed, SLM) if phase is None else phase,
                                               engine_options=PATHS[path])


# ---- cfg 2: every column path against the recorded reference run -------------------------------------------
@pytest.mark.parametrize("path", list(PATHS))
def test_cfg2_leonardo_every_column_path_matches_reference(path):
    """
    The headline (WGS-Leonardo x 50, seed 2) through the engine default (col_fused_kernel over the active
    columns), the dense tile-resident kernel bench.py times (col_tile_kernel) and the dense per-column kernel:
    each directly against tests/golden/cfg2_summary.npz (recorded from the reference).  north_star tolerance
    1e-5 relative L2 on the farfield amplitude at the spots.
    """
    meta, gold = load_golden("cfg2_summary")
    h = cfg2_hologram(2, path)
    h.optimize("WGS-Leonardo", maxiter=50, verbose=False)
    d = assert_cfg2_column_path(h, path)
    assert d.count("col_fused_kernel" if path != "dense" else "col_tile2_kernel", RULE=1) == 49, d     # body 0 has no update
    ky, kx = h.spot_knm_rounded[1], h.spot_knm_rounded[0]
    amp_ff = h.amp_ff
    errs = dict(spot_amp=rel_l2(amp_ff[ky, kx], gold["spot_ampff"]),
                spot_weights=rel_l2(h.weights[ky, kx], gold["spot_weights"]),
                amp_sub=rel_l2(amp_ff[::16, ::16], gold["ampff_sub"]),
                phase_sub=phase_rel_l2(h.phase[::6, ::6], gold["phase_sub"]))
    report(f"cfg2 WGS-Leonardo 50 it vs reference [{path} column path]", **errs)
    # 50 free-phase bodies amplify rounding: an IDEAL fp32 implementation (float64 arithmetic rounded to float32,
    # tests/golden/cfg2_ideal_fp32.json) ends 1.28e-5 from the reference on this seed.  The bound is tied to that
    # yardstick, not to the engine (test_cfg2_error_growth_over_seeds holds the whole curve over eight seeds; the plain
    # north-star 1e-5 is asserted at <= 10 bodies there, on the phase-fixing variant below, and -- at 2e-6 -- on the
    # teacher-forced single bodies of this very run).
    import json
    import os
    from conftest import GOLDEN
    yard = json.load(open(os.path.join(GOLDEN, "cfg2_ideal_fp32.json")))["per_seed_curve"]["2"]["all64"]["50"]
    assert errs["spot_amp"] < max(1e-5, 3 * yard)          # 1.28e-5 for the ideal fp32 implementation; engine 1.6e-5
    assert errs["spot_weights"] < 3e-4 and errs["amp_sub"] < 3e-4 and errs["phase_sub"] < 6e-4


@pytest.mark.parametrize("path", list(PATHS))
def test_cfg2_single_bodies_at_full_size_match_reference(path):
    """
    Teacher-forced at the headline's own size: tests/golden/cfg2_steps.npz holds the reference's complete
    state before bodies 10, 30 and 49 of the seed-2 run and what ONE reference body makes of it.  One body is determined
    by its input, so -- unlike the 50-body end state -- this is held to a flat tolerance on every column path:
    spot amplitudes of the forward transform 2e-6, new phase 2e-6 (unit phasors, every third pixel per axis), new
    weights at the spots 3e-6.  A 2x loss of accuracy anywhere in the long run turns this red.
    """
    meta, gold = load_golden("cfg2_steps")
    ky, kx = gold["spot_knm_rounded"][1], gold["spot_knm_rounded"][0]
    h = cfg2_hologram(2, path, phase=gold[f"phase_{meta['iters'][0]}"].copy())
    assert np.array_equal(h.spot_knm_rounded, gold["spot_knm_rounded"])
    for k in meta["iters"]:
        w = np.zeros(SHAPE, np.float32)
        w[ky, kx] = gold[f"weights_{k}_spots"]          # the weights are zero off the spots: this is the whole array
        h.phase = gold[f"phase_{k}"].copy()
        h.weights = w
        h.iter = k
        seen = {}

        def cb(hh):
            seen["amp"] = hh.amp_ff[ky, kx].copy()
            seen["sub"] = hh.amp_ff[::16, ::16].copy()
            return False

        # (a) the forward transform of the recorded phase, through the stepwise operators
        h.optimize("WGS-Leonardo", maxiter=1, verbose=False, callback=cb)
        ea = rel_l2(seen["amp"], gold[f"ampff_{k}_spots"])
        es = rel_l2(seen["sub"], gold[f"ampff_{k}_sub"])
        # (b) the whole body through this path's fused kernels, from the same state
        h.phase = gold[f"phase_{k}"].copy()
        h.weights = w
        h.iter = k
        dispatch_of(h)                                      # (forget the stepwise launches of (a))
        h.optimize("WGS-Leonardo", maxiter=1, verbose=False)
        assert_cfg2_column_path(h, path)
        ep = phase_rel_l2(h.phase[::3, ::3], gold[f"next_phase_{k}_sub"])
        ew = rel_l2(h.weights[ky, kx], gold[f"next_weights_{k}_spots"])
        report(f"cfg2 teacher-forced body {k} at 4096^2 [{path}]", spot_amp=ea, amp_sub=es, phase=ep, weights=ew)
        assert ea < 2e-6 and es < 2e-6, (path, k, ea, es)
        assert ep < 2e-6 and ew < 3e-6, (path, k, ep, ew)


@pytest.mark.parametrize("path", list(PATHS))
def test_cfg2_kim_every_column_path_matches_reference(path):
    meta, gold = load_golden("cfg2kim_summary")
    h = cfg2_hologram(9, path)
    h.optimize("WGS-Kim", maxiter=30, verbose=False)
    d = assert_cfg2_column_path(h, path)
    fam = "col_tile_kernel" if path == "dense" else "col_fused_kernel"
    # PHASE: 1 = the phase is stored (every free body of WGS-Kim with an update), 2 = the stored phase is used (fixed) -- on
    # the dense path the half-width tile kernel since round 5 (three workgroups per CU; its phase-STORING form does not fit)
    assert d.count(fam, PHASE=1) > 0 and d.count("col_tile2_kernel" if path == "dense" else fam, PHASE=2) > 0, d
    assert [bool(x) for x in h.stats["flags"]["fixed_phase"]] == [bool(x) for x in gold["fixed_history"]]
    ky, kx = h.spot_knm_rounded[1], h.spot_knm_rounded[0]
    errs = dict(spot_amp=rel_l2(h.amp_ff[ky, kx], gold["spot_ampff"]),
                spot_weights=rel_l2(h.weights[ky, kx], gold["spot_weights"]),
                amp_sub=rel_l2(h.amp_ff[::16, ::16], gold["ampff_sub"]),
                phase_sub=phase_rel_l2(h.phase[::6, ::6], gold["phase_sub"]))
    report(f"cfg2 WGS-Kim 30 it vs reference [{path} column path]", **errs)
    assert errs["spot_amp"] < 1e-5 and errs["amp_sub"] < 1e-4 and errs["spot_weights"] < 1e-4
    assert errs["phase_sub"] < 3e-4


# Bounds of the cfg 2 seed sweep, in units of the ideal-fp32 implementation's distance from the reference.  Measured in
# round 5 over 16 seeds x 6 bodies: geometric mean 1.15, log-scatter 0.60 (a factor 1.8 either way), per-seed geometric means
# 0.37 .. 2.26, single points 0.21 .. 3.42 -- at that scatter 3 % of the points of a perfectly ideal implementation lie beyond
# 3 x, so the single-point bound is a backstop and the means are the test.
CFG2_YARD_FACTOR = 3.0           # headline seed, end state (test_cfg2_leonardo_every_column_path_matches_reference); per-seed geometric mean
CFG2_POINT_FACTOR = 5.0          # any single point of the sweep, against the largest distance the ideal run has shown by then
CFG2_MEAN_FACTOR = 1.3           # geometric mean over the sweep


def test_cfg2_error_growth_over_seeds():
    """
    Sixteen seed phases (tests/golden/cfg2_seeds.npz, round 2; cfg2_seeds_b.npz, round 5): the spot amplitudes after 5, 10,
    20, 30, 40 and 50 WGS-Leonardo bodies against the reference's.  Free-phase WGS amplifies rounding differences
    exponentially, so beyond ~10 bodies no fp32 implementation that is not bit-identical to NumPy lands within a flat 1e-5.
    The yardstick is NOT the engine's own behaviour: tests/golden/cfg2_ideal_fp32{,_b}.json (tools/conditioning_cfg2.py
    --variants all64 --curve, CPU only) hold, per seed and iteration, how far an IDEAL fp32 implementation -- the reference's
    op sequence with every FFT / arctan2 / exp evaluated in float64 and rounded once to float32 -- ends up from the reference.
    Asserted:
      * 5 and 10 bodies: the north-star 1e-5 on every seed whose ideal-fp32 run is within 3e-6 there;
      * every recorded iteration: error <= max(1e-5, CFG2_POINT_FACTOR x the largest distance the ideal implementation has
        shown up to that body) -- a run whose distance jumps (a spot crossing a zero of the field) jumps at another body in
        another implementation -- and the geometric mean over a seed's six points <= CFG2_YARD_FACTOR;
      * over all seeds and iterations the GEOMETRIC MEAN of engine error / ideal distance is <= CFG2_MEAN_FACTOR (round 4
        measured a worst ratio of 2.33 over eight seeds and could not say whether that was scatter or a worse operator: over
        the 96 points of sixteen seeds the geometric mean is 1.15 with a log-scatter of 0.60, per-seed means 0.37 .. 2.26 --
        the engine is as far from the reference as an fp32 implementation whose every operator is rounded ONCE, give or take
        the 0.14 standard error of that mean; a 2x loss of accuracy anywhere doubles it);
      * the same for the mean over seeds of the per-seed geometric means (the six points of one run are correlated).
    The distance the reference itself moves when its seed changes by one fp32 ulp (also in the fixtures) is reported.
    """
    import json
    import os
    from conftest import GOLDEN
    worst_ratio = {"default": 0.0, "dense": 0.0}
    logs = {"default": {}, "dense": {}}
    failures = []
    for fixture, yardfile in (("cfg2_seeds", "cfg2_ideal_fp32.json"), ("cfg2_seeds_b", "cfg2_ideal_fp32_b.json")):
        meta, gold = load_golden(fixture)
        ideal = json.load(open(os.path.join(GOLDEN, yardfile)))["per_seed_curve"]
        its = list(meta["curve_iters"]) + [meta["maxiter"]]
        for i, seed in enumerate(meta["seeds"]):
            ref = np.concatenate((gold["curve_ampff"][i], gold["spot_ampff"][i][None]))
            refp = np.concatenate((gold["curve_ampff_perturbed"][i], gold["spot_ampff_perturbed"][i][None]))
            drift = np.array([rel_l2(refp[k], ref[k]) for k in range(len(its))])
            yard = np.array([ideal[str(seed)]["all64"][str(k)] for k in its])
            # what the ideal implementation has reached by then: a run whose distance jumps (one spot crossing a zero of
            # the field) does so at another body in another implementation
            envelope = np.maximum.accumulate(yard)
            for path in ("default", "dense"):
                h = cfg2_hologram(seed, path)
                ky, kx = h.spot_knm_rounded[1], h.spot_knm_rounded[0]
                err, done = [], 0
                for k in its:
                    h.optimize("WGS-Leonardo", maxiter=k - done, verbose=False)
                    done = k
                    err.append(rel_l2(h.amp_ff[ky, kx], ref[len(err)]))
                h._release_engine()
                err = np.array(err)
                report(f"cfg2 seed {seed} [{path}] engine error at bodies {its}", **{f"it{k}": e for k, e in zip(its, err)})
                report(f"cfg2 seed {seed} ideal fp32 implementation at bodies {its}", **{f"it{k}": e for k, e in zip(its, yard)})
                report(f"cfg2 seed {seed} reference 1-ulp drift at bodies {its}", **{f"it{k}": e for k, e in zip(its, drift)})
                worst_ratio[path] = max(worst_ratio[path], float((err / yard).max()))
                logs[path][seed] = np.log(err / yard)
                if np.all(yard[:2] < 3e-6) and not np.all(err[:2] < 1e-5):
                    failures.append(("north-star at 5 / 10 bodies", seed, path, err[:2].tolist()))
                if not np.all(err < np.maximum(1e-5, CFG2_POINT_FACTOR * envelope)):
                    failures.append(("pointwise bound", seed, path, err.tolist(), yard.tolist()))
                if np.exp(np.log(err / yard).mean()) > CFG2_YARD_FACTOR:
                    failures.append(("per-seed geometric mean", seed, path, float(np.exp(np.log(err / yard).mean()))))
    report("cfg2 seed sweep: worst engine error / ideal-fp32 distance", **worst_ratio)
    for path, per_seed in logs.items():
        allp = np.concatenate(list(per_seed.values()))
        gm_all = float(np.exp(allp.mean()))
        gm_seeds = float(np.exp(np.mean([v.mean() for v in per_seed.values()])))
        report(f"cfg2 seed sweep [{path}]: engine error / ideal-fp32 distance over {len(per_seed)} seeds x 6 bodies",
               geometric_mean=gm_all, mean_of_per_seed_geometric_means=gm_seeds, log_scatter=float(allp.std()),
               per_seed_min=float(np.exp(min(v.mean() for v in per_seed.values()))),
               per_seed_max=float(np.exp(max(v.mean() for v in per_seed.values()))))
        if not (gm_all <= CFG2_MEAN_FACTOR and gm_seeds <= CFG2_MEAN_FACTOR):
            failures.append(("geometric mean over the sweep", path, gm_all, gm_seeds))
    assert not failures, failures


# ---- cfg 3: eight holograms per engine at 4096^2 ------------------------------------------------------------
@pytest.mark.parametrize("path", ["default", "dense"])
def test_cfg3_batch_of_eight_matches_single_engines(path):
    """BASELINE config 3's per-GPU shard: batch = 8 at the cfg 2 geometry against eight single engines."""
    n, iters = 8, 12
    host = cfg2_hologram(100, "default")
    phases = np.stack([synth.seed_phase(100 + i, SLM) for i in range(n)])
    hb = HologramBatch(SHAPE, SLM, host.target, phases, spot_index=host.spot_knm_rounded, spot_amp=host.spot_amp)
    for opt, val in PATHS[path].items():
        hb.engine.set_option(opt, val)
    try:
        hb.optimize("WGS-Leonardo", maxiter=iters)
        d = dispatch_of(hb)
        if path == "dense":      # the batch-shaped column kernel: half-width tiles, three workgroups per CU (96 per hologram)
            assert d.count("col_tile2_kernel", flags=["batch", "xmap"], N=4096, NR=5) == iters and d.count("col_tile_kernel") == 0, d
        else:
            assert d.count("col_fused_kernel", flags=["batch", "list"]) == iters and d.count("col_tile2_kernel") == 0, d
        got = hb.phases()
        w = hb.engine.get(L.WEIGHTS)
    finally:
        hb.close()
    ky, kx = host.spot_knm_rounded[1], host.spot_knm_rounded[0]
    for i in range(n):
        h = cfg2_hologram(100 + i, path)
        h.optimize("WGS-Leonardo", maxiter=iters, verbose=False)
        e_ph, e_w = phase_rel_l2(got[i], h.phase), rel_l2(w[i][ky, kx], h.weights[ky, kx])
        report(f"cfg3 batch of 8 [{path}] hologram {i}", phase=e_ph, weights=e_w)
        # (the weight-norm partial sums are folded over a different number of workgroups in the batch engine)
        assert e_ph < 5e-6 and e_w < 5e-6
        h._release_engine()
    # the holograms are independent: different seeds must give different masks
    assert phase_rel_l2(got[0], got[1]) > 0.5


# ---- cfg 4: CompressedSpotHologram at its configured size ---------------------------------------------------
def _cfg4_spots(D, N):
    v = synth.uniform01(4, (D, N), 9) * 2 - 1
    v[:2] *= 0.02
    if D == 3:
        v[2] *= 1e-6
    return v


def _kernel_phase(h, spots, pix):
    """phi_n(p) in float64 for the given spot / flat pixel indices, from the oracle's monomial tables."""
    terms, wts = orc.monomial_weights(h.zernike_basis, h.spot_zernike)
    x = h._xg.ravel()[pix].astype(np.float64)
    y = h._yg.ravel()[pix].astype(np.float64)
    phi = np.zeros((len(spots), len(pix)))
    for m, (px, py) in enumerate(terms):
        phi += wts[m, spots][:, None] * (x ** int(px) * y ** int(py))[None, :]
    return phi


@pytest.mark.parametrize("D", [2, 3])
@pytest.mark.parametrize("sep", [1, 0])
def test_cfg4_full_size_against_direct_summation(D, sep):
    """
    N = 1e4 spots, S = 1152 x 1920 (BASELINE config 4), WGS-Kim across the phase-fixing iteration.  One more loop
    body is then taken operator by operator and checked against float64 DIRECT (non-separable) summation on
    the host: the farfield of 64 random spots over all 2.2 M pixels, and the new phase at 1,000 random pixels
    over all 1e4 spots -- for the matrix-core (separable) form and for the direct kernels.
    """
    N = 10000
    slm = SimpleSLM(SLM, pitch_um=(8, 8), wav_um=0.78)
    h = CompressedSpotHologram(_cfg4_spots(D, N), basis="kxy", cameraslm=SimpleFourierSLM(slm),
                               engine_options={L.OPT_SEPARABLE: sep})
    h.reset_phase(synth.seed_phase(4, SLM))
    h.optimize("WGS-Kim", maxiter=12 if sep else 11, verbose=False)       # phase fixes at iteration 10
    d = dispatch_of(h)
    if sep:        # both transforms as complex GEMMs on the matrix cores (EPI 1: n2f with the y contraction in the epilogue)
        assert d.count("cgemm_streamk", EPI=1) == 12 and d.count("cgemm_streamk", EPI=0) == 12, d
        assert d.families() == {"cgemm_streamk"}, d
    else:          # direct kernels: runs of 16 pixels by recurrence (degree 1: tilts; 2: with focus)
        assert d.count("c_n2f_run", DEG=D - 1) == 11 and d.count("c_f2n_run", DEG=D - 1) == 11, d
        assert d.families() == {"c_n2f_run", "c_f2n_run"}, d
    assert h.flags["fixed_phase"] and h.stats["flags"]["fixed_phase"][:10] == [False] * 10
    e = h._get_engine()
    S = SLM[0] * SLM[1]
    rng = np.random.default_rng(44)
    spots = np.sort(rng.choice(N, 64, replace=False))
    pix = np.sort(rng.choice(S, 1000, replace=False))

    # forward: ff_n = sum_p amp e^{i phase_p} e^{-i phi_n(p)} / sqrt(S), then ff /= ||ff||  (_spots.py:767-824)
    phase = h.phase.astype(np.float64).ravel()
    e.nearfield2farfield()
    ff = e.get(L.FARFIELD)[0].astype(np.complex128)
    assert abs(np.sqrt(np.sum(np.abs(ff) ** 2)) - 1) < 1e-5
    ref = np.zeros(len(spots), dtype=np.complex128)
    allp = np.arange(S)
    amp = np.full(S, float(h.amp)) if np.isscalar(h.amp) else np.asarray(h.amp, dtype=np.float64).ravel()
    for c0 in range(0, S, 1 << 18):                # chunks of pixels: 64 x 262144 float64 at a time
        pp = allp[c0:c0 + (1 << 18)]
        ref += np.sum(amp[pp][None, :] * np.exp(1j * (phase[pp][None, :] - _kernel_phase(h, spots, pp))), axis=1)
    ref *= 1 / np.sqrt(S)
    got = ff[spots]
    scale = np.real(np.vdot(ref, got)) / np.real(np.vdot(ref, ref))         # the positive factor 1 / ||ff||
    err_ff = rel_l2(got, scale * ref)

    # constraint: farfield = weights * exp(i phase_ff) with the FIXED phase (_hologram.py:1601-1605)
    st = h._make_step()
    pff_before = e.get(L.PHASE_FF)[0].copy()
    e.farfield_constraint(st)
    assert st.fixed_phase == 1
    ffc = e.get(L.FARFIELD)[0].astype(np.complex128)
    wts = e.get(L.WEIGHTS)[0].astype(np.float64)
    np.testing.assert_array_equal(e.get(L.PHASE_FF)[0], pff_before)
    err_cons = rel_l2(ffc, wts * np.exp(1j * pff_before.astype(np.float64)))

    # inverse: nf_p = sum_n ff_n e^{+i phi_n(p)} / sqrt(S), phase = atan2(nf)  (_spots.py:887-914)
    e.farfield2nearfield()
    ph_new = e.get(L.PHASE)[0].ravel()[pix]
    nf = np.zeros(len(pix), dtype=np.complex128)
    alln = np.arange(N)
    for c0 in range(0, N, 2000):
        nn = alln[c0:c0 + 2000]
        nf += np.sum(ffc[nn][:, None] * np.exp(1j * _kernel_phase(h, nn, pix)), axis=0)
    err_ph = phase_rel_l2(ph_new, np.angle(nf))
    report(f"cfg4 full size D={D} {'matrix-core' if sep else 'direct'} path vs float64 direct summation",
           farfield_64_spots=err_ff, constraint=err_cons, phase_1000_pixels=err_ph)
    assert err_ff < 2e-5 and err_cons < 1e-6 and err_ph < 1e-4
    h._release_engine()


def test_cfg4_configured_run_reaches_its_end_state():
    """
    BASELINE config 4 as configured: N = 1e4 spots, S = 1152 x 1920, WGS-Kim (phase fixed at 10), **200 iterations**.
    No CPU restatement can follow that (2.2e10 kernel evaluations per transform).  The end state is pinned by what is
    determinate: the flag history; the farfield the run ends on is the float64 direct sum over all 2.2 M pixels of its own
    final phase (64 random spots); and the weighted loop did its job -- the spot powers are more uniform after 200 iterations
    than after 12 (0.78 -> 0.83 for these 1e4 densely packed spots).  On the way the two arithmetics of this class -- the
    matrix-core path and the run kernels, each within 2e-6 .. 4e-6 of float64 direct summation per operator
    (test_cfg4_full_size_against_direct_summation) -- are compared after 14 iterations from the same start, next to how far
    ONE of them moves when its start phase changes by one fp32 ulp.
    """
    N, D = 10000, 2
    slm = SimpleSLM(SLM, pitch_um=(8, 8), wav_um=0.78)

    def make(sep):
        hh = CompressedSpotHologram(_cfg4_spots(D, N), basis="kxy", cameraslm=SimpleFourierSLM(slm), engine_options={L.OPT_SEPARABLE: sep})
        hh.reset_phase(synth.seed_phase(4, SLM))
        return hh

    def uniformity(hh):
        p = np.abs(hh.farfield.astype(np.complex128)) ** 2 / np.asarray(hh.target, dtype=np.float64) ** 2
        return 1 - (p.max() - p.min()) / (p.max() + p.min())

    h, hd = make(1), make(0)
    h.optimize("WGS-Kim", maxiter=12, verbose=False)
    u12 = uniformity(h)
    h.optimize("WGS-Kim", maxiter=2, verbose=False)
    hd.optimize("WGS-Kim", maxiter=14, verbose=False)
    # the two arithmetics from the same start, 14 iterations in ONE call each.  (Not against `h`: its 12 + 2 is two calls,
    # and the trailing transform of a call refreshes the frozen phase_ff of WGS-Kim -- reference quirk, _hologram.py:949 --
    # which alone moves the spot amplitudes by 1.4e-3; round 3 reported that as the distance between the arithmetics.)
    ha = make(1)
    ha.optimize("WGS-Kim", maxiter=14, verbose=False)
    e_paths = dict(spot_amp=rel_l2(np.abs(ha.farfield), np.abs(hd.farfield)), weights=rel_l2(ha.weights, hd.weights),
                   phase=phase_rel_l2(ha.phase, hd.phase))
    e_calls = dict(spot_amp=rel_l2(np.abs(h.farfield), np.abs(ha.farfield)))
    hd._release_engine()
    # ... and how far ONE arithmetic moves when its start phase changes by one fp32 ulp: the yardstick for the distance
    # between the two arithmetics (each within 2e-6 .. 4e-6 of float64 direct summation per operator, see the test above)
    hp = make(1)
    hp.reset_phase(np.nextafter(synth.seed_phase(4, SLM), np.float32(4.0)))
    hp.optimize("WGS-Kim", maxiter=14, verbose=False)
    e_ulp = dict(spot_amp=rel_l2(np.abs(hp.farfield), np.abs(ha.farfield)), weights=rel_l2(hp.weights, ha.weights),
                 phase=phase_rel_l2(hp.phase, ha.phase))
    hp._release_engine()
    ha._release_engine()
    h.optimize("WGS-Kim", maxiter=186, verbose=False)
    assert h.iter == 200 and h.flags["fixed_phase"] and sum(bool(x) for x in h.stats["flags"]["fixed_phase"]) == 190
    u200 = uniformity(h)
    # the farfield of the final phase by float64 direct summation, 64 spots
    S = SLM[0] * SLM[1]
    spots = np.sort(np.random.default_rng(45).choice(N, 64, replace=False))
    phase = h.phase.astype(np.float64).ravel()
    amp = np.full(S, float(h.amp)) if np.isscalar(h.amp) else np.asarray(h.amp, dtype=np.float64).ravel()
    ref = np.zeros(len(spots), dtype=np.complex128)
    allp = np.arange(S)
    for c0 in range(0, S, 1 << 18):
        pp = allp[c0:c0 + (1 << 18)]
        ref += np.sum(amp[pp][None, :] * np.exp(1j * (phase[pp][None, :] - _kernel_phase(h, spots, pp))), axis=1)
    got = h.farfield.astype(np.complex128)[spots]
    scale = np.real(np.vdot(ref, got)) / np.real(np.vdot(ref, ref))
    err_end = rel_l2(got, scale * ref)
    report("cfg4 configured run (200 it): paths at 14 it, end state vs float64 direct sum, uniformity", end_farfield=err_end,
           uniformity_12=u12, uniformity_200=u200, **{f"paths14_{k}": v for k, v in e_paths.items()},
           **{f"one_ulp14_{k}": v for k, v in e_ulp.items()}, split_12_plus_2_vs_14_spot_amp=e_calls["spot_amp"])
    assert e_paths["spot_amp"] < 1e-3, e_paths
    # the two arithmetics part no faster than one of them parts from itself under a one-ulp change of the input (x 20:
    # the perturbation is one rounding, the paths differ by a few per operator)
    # (measured: 1.7e-5 between the arithmetics, 6.7e-7 for the one-ulp change, 1.4e-3 for the 12 + 2 call split)
    assert e_paths["spot_amp"] < 35 * max(e_ulp["spot_amp"], 1e-6), (e_paths, e_ulp)      # (26 x in rounds 4 and 5; the factor was 50 until round 5)
    assert err_end < 2e-5
    assert u200 > u12, (u12, u200)
    h._release_engine()


# ---- cfg 4's DFT-grid companion: 1e4 spots at distinct pixels of an 8192^2 pad, WGS-Kim ---------
def _grid_spots(shape, box, n):
    lin = np.unique((synth.uniform01(4, (4 * n,), stream=7) * box * box).astype(np.int64))
    lin = np.sort(lin[np.argsort(synth.uniform01(5, (lin.size,), stream=8), kind="stable")[:n]])
    return np.stack([lin % box + (shape[1] - box) // 2, lin // box + (shape[0] - box) // 2]).astype(np.float64)


_GRID_ORACLE = {}


@pytest.mark.parametrize("path", ["default", "dense"])
def test_cfg4_grid_companion_follows_oracle(path):
    """
    SpotHologram with 1e4 spots at distinct pixels of the centred 3360^2 box of an 8192^2 pad (what |k| <= 0.02 rad
    spans there), WGS-Kim with the phase fixed from iteration 2: four loop bodies on the engine (active-column list
    and dense kernels) and on the CPU oracle from the same seed.
    """
    shape, n = (8192, 8192), 10000
    vec = _grid_spots(shape, 3360, n)
    kw = dict(fix_phase_iteration=2)
    h = SpotHologram(shape, vec, basis="knm", slm_shape=SLM, phase=synth.seed_phase(4, SLM), engine_options=PATHS[path])
    h.optimize("WGS-Kim", maxiter=4, verbose=False, **kw)
    ky, kx = h.spot_knm_rounded[1], h.spot_knm_rounded[0]
    if "o" not in _GRID_ORACLE:          # (the oracle's four bodies at 8192^2 are 18 s of CPU: once for both column paths)
        o = orc.OracleSpotHologram(shape, vec, slm_shape=SLM, phase=synth.seed_phase(4, SLM))
        o.optimize("WGS-Kim", maxiter=4, **kw)
        _GRID_ORACLE["o"] = (o.amp_ff[ky, kx].copy(), o.weights[ky, kx].copy(), o.phase.copy(), list(o.stats["flags"]["fixed_phase"]))
    o_amp, o_w, o_phase, o_fixed = _GRID_ORACLE["o"]
    errs = dict(spot_amp=rel_l2(h.amp_ff[ky, kx], o_amp), spot_weights=rel_l2(h.weights[ky, kx], o_w),
                phase=phase_rel_l2(h.phase, o_phase))
    report(f"cfg4 grid companion WGS-Kim 4 it [{path}]", **errs)
    assert h.stats["flags"]["fixed_phase"] == o_fixed
    assert errs["spot_amp"] < 1e-5 and errs["spot_weights"] < 1e-5 and errs["phase"] < 1e-4


# ---- cfg 5: fp32 vs fp64 tolerance sweep (reduced; the full curve is tools/cfg5_sweep.py -> profiles/r03) ------------
def test_cfg5_precision_sweep_per_step():
    """
    BASELINE config 5 is a *sweep*: at 8192^2 (MRAF, mraf_factor 0.5), from the engine's own fp32 state before body k,
    ONE body computed by the oracle in float64 (the truth), the oracle in float32, the engine in float32 and in float64.
    Asserted per step:
      * engine fp64 vs oracle fp64: 1e-11 (GS) / 1e-9 (WGS: the weight rule's pow), relative L2, phase and weights;
      * GS, engine fp32: 5e-6 against the fp32 oracle and an error against the truth within 1.5 x NumPy's own
        (median and 99th percentile of the per-pixel phase error; measured 1.00 .. 1.12 x);
      * WGS-Leonardo, engine fp32 against the truth: per-pixel phase and weight errors with median <= 1e-5 (the
        north-star tolerance) and 99th percentile <= 1e-4.  No tighter: pixel-wise WGS on a dense MRAF image divides by
        speckle amplitudes, NumPy's own fp32 body is 0.9 .. 2.5e-5 (L2) from the truth here, and L2 figures of two fp32
        implementations scatter by 4 x from state to state (a handful of pixels near a zero of the field carry them),
        so medians / percentiles are compared.  The engine's weight rule runs x^p on the hardware log2 / exp2 (1 ulp each,
        on the ratio (|F| c / T)^2): its per-pixel error is 0.15 .. 10 x that of NumPy's correctly rounded powf
        (median 1.9e-7 vs 1.8e-8 at worst, both far under the tolerance) -- reported, see DESIGN.md.
    tools/cfg5_sweep.py runs k = 1..20 plus the free-running divergence curves; profiles/r03/cfg5_sweep.json keeps them.
    """
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import cfg5_sweep

    res = cfg5_sweep.sweep(steps=(1, 10), free_run=False, log=lambda *_: None)      # (round 6: 2 and 20 dropped -- 19 s of CPU oracle each; tools/cfg5_sweep.py runs 1 .. 20)
    for method, entry in res["methods"].items():
        for row in entry["teacher_forced"]:
            k = row["k"]
            report(f"cfg5 sweep {method} k={k}", **{f"{t}_{q}_{st}": row[t][q][st] for t in row if t != "k"
                                                    for q in ("phase", "weights") for st in ("l2", "median", "p99")})
            e64, e32 = row["engine64_vs_oracle64"], row["engine32_vs_oracle32"]
            eng, ref = row["engine32_vs_truth"], row["oracle32_vs_truth"]
            tol64 = 1e-11 if method == "GS" else 1e-9
            assert e64["phase"]["l2"] < tol64 and e64["weights"]["l2"] < tol64, (method, k, e64)
            if method == "GS":
                assert e32["phase"]["l2"] < 5e-6 and e32["weights"]["l2"] < 1e-6, (method, k, e32)
                for st in ("median", "p99"):
                    assert eng["phase"][st] <= 1.5 * ref["phase"][st], (method, k, st, eng, ref)
            else:
                for q in ("phase", "weights"):
                    assert eng[q]["median"] <= 1e-5 and eng[q]["p99"] <= 1e-4, (method, k, q, eng, ref)
