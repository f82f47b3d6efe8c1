"""
Round 6 (``-m gpu``): what the round-5 review found around the device-resident callback loop and the half-width tile kernel's
dispatch; the single-inverse MRAF pass (pre-summed ||w'||) against the two-inverse form it replaces.
"""
import numpy as np
import pytest

from conftest import dispatch_of, force_stepwise, phase_rel_l2, rel_l2, report
from slmsuite_amd import _lib as L
from slmsuite_amd import synth
from slmsuite_amd.holography.algorithms import Hologram, SpotHologram

pytestmark = pytest.mark.gpu


def test_callback_reads_phase_ff_of_a_body_the_general_operators_ran():
    """
    A callback against the device-resident loop whose bodies do NOT run the fused kernels (WGS-Wu: the general operators
    inside the engine call; a shape that is no power of two: Bluestein lines): the engine keeps no previous phase then --
    HGS_PHASE_FF itself is what the body stored, and that is what ``hologram.phase_ff`` has to show at every invocation, as
    the host-driven loop shows it (before round 6 the callback got the host copy of an earlier read, or None).
    """
    def views_of(h, method):
        seen = []

        def cb(hh):
            pf = hh.phase_ff
            seen.append(None if pf is None else pf.copy())
            return False

        h.optimize(method, maxiter=4, verbose=False, callback=cb)
        return seen

    cases = {
        "WGS-Wu 128^2": (lambda: Hologram(synth.random_target(3, (128, 128), 0.2, 1.0), phase=synth.seed_phase(3, (48, 80)), slm_shape=(48, 80)), "WGS-Wu"),
        "GS 100x150 (Bluestein)": (lambda: Hologram(synth.random_target(4, (100, 150), 0.2, 1.0), phase=synth.seed_phase(4, (40, 60)), slm_shape=(40, 60)), "GS"),
    }
    for name, (make, method) in cases.items():
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            fast, slow = make(), force_stepwise(make())
        vf, vs = views_of(fast, method), views_of(slow, method)
        d = dispatch_of(fast)
        assert d.count("col_fused_kernel") + d.count("col_tile_kernel") + d.count("col_tile2_kernel") == 0, d   # general operators
        assert [v is None for v in vf] == [v is None for v in vs], name
        worst = 0.0
        for k, (a, b) in enumerate(zip(vf, vs)):
            if a is None:
                continue
            assert k == 0 or not np.array_equal(a, vf[k - 1]), (name, k)          # a fresh array per body, not a stale copy
            worst = max(worst, phase_rel_l2(a, b))
        report(f"callback phase_ff on the general operators: {name}", phase_ff=worst)
        assert worst < 5e-4, (name, worst)


def test_callback_phase_assignment_does_not_depend_on_what_it_reads():
    """
    A phase assigned inside a callback is overwritten by the body that follows (_hologram.py:1483-1487) -- whether or not
    the callback looked at ``farfield`` / ``amp_ff`` after assigning it, and what it sees there is the field of the phase the
    body starts from, formed before the callback (:1465-1477).  (Until round 6 a read after the assignment uploaded the
    assigned phase and the body iterated from it.)
    """
    shape, slm = (256, 256), (72, 120)
    other = synth.seed_phase(11, slm)
    seen = {}

    def run(look):
        h = SpotHologram.make_rectangular_array(shape, (6, 6), (24, 24), basis="knm", slm_shape=slm, phase=synth.seed_phase(5, slm))

        def cb(hh):
            if hh.iter == 2:
                before = hh.amp_ff.copy() if look == "before" else None
                hh.phase = other.copy()
                if look == "after":
                    seen["after"] = hh.amp_ff.copy()
                if before is not None:
                    seen["before"] = before
            return False

        h.optimize("WGS-Leonardo", maxiter=5, verbose=False, callback=cb)
        return h.phase.copy(), np.array(h.weights, copy=True)

    p_none, w_none = run(None)
    p_before, w_before = run("before")
    p_after, w_after = run("after")
    np.testing.assert_array_equal(p_none, p_before)
    np.testing.assert_array_equal(p_none, p_after)
    np.testing.assert_array_equal(w_none, w_after)
    np.testing.assert_array_equal(w_none, w_before)
    np.testing.assert_array_equal(seen["before"], seen["after"])


def test_tile_option_off_reaches_the_per_column_kernel_at_2048_rows():
    """HGS_OPT_TILE_KERNEL = 0 is the tests' A/B reference: it has to leave col_tile2_kernel at 2048 rows too (round 5: only
    at 4096), and the two kernels agree to rounding."""
    shape, slm = (2048, 2048), (1080, 1920)
    out = {}
    for tile in (1, 0):
        h = SpotHologram.make_rectangular_array(shape, (8, 8), (64, 64), basis="knm", slm_shape=slm, phase=synth.seed_phase(2, slm),
                                                engine_options={L.OPT_SPARSE_COLUMNS: 0, L.OPT_TILE_KERNEL: tile})
        h.optimize("WGS-Leonardo", maxiter=4, verbose=False)
        d = dispatch_of(h)
        if tile:
            assert d.count("col_tile2_kernel", N=2048) == 4 and d.count("col_fused_kernel") == 0, d
        else:
            assert d.count("col_tile2_kernel") == 0 and d.count("col_fused_kernel", N=2048) == 4, d
        out[tile] = (h.phase.copy(), np.array(h.weights, copy=True))
    assert phase_rel_l2(out[1][0], out[0][0]) < 5e-6 and rel_l2(out[1][1], out[0][1]) < 5e-6


def test_trailing_transform_from_the_kept_g_against_the_stored_phase():
    """
    After a fused float32 loop the last row launch leaves G of the next body behind (row_kernel MODE 3) and the trailing
    transform of optimize() starts from it: amp * v / |v| un-rounded, where a fresh engine (get_farfield's side engine, a new
    hologram) starts from exp(i * stored phase).  The two describe the same field to float32 rounding -- consistent with the
    loop rather than bit-identical with the stored phase (include/hgs.h, HGS_KEEP_G).
    """
    shape, slm = (1024, 1024), (288, 480)
    for sparse in (0, 1):
        h = SpotHologram.make_rectangular_array(shape, (8, 8), (64, 64), basis="knm", slm_shape=slm, phase=synth.seed_phase(9, slm),
                                                engine_options={L.OPT_SPARSE_COLUMNS: sparse})
        h.optimize("WGS-Leonardo", maxiter=6, verbose=False)
        kept = h.farfield.copy()                               # _populate_results, from the G left behind
        d = dispatch_of(h)
        assert d.count("row_kernel", MODE=0) == 1, d           # only the first body built G from a phase
        g = Hologram(np.array(h.target, copy=True), phase=h.phase.copy(), slm_shape=slm)
        fresh = g.get_farfield()                               # exp(i * the stored phase) on an engine of its own
        err = rel_l2(kept, fresh)
        report(f"trailing transform, kept G vs stored phase (sparse={sparse})", farfield=err)
        assert err < 2e-6, err
