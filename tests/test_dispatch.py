"""
The dispatcher's table, observed (``-m gpu``): every policy the engine documents -- shape, precision, slot count, list
density, options -- must reach the kernel instance it names.  ``hgs_dispatch_read`` reports the template arguments of
every launch (include/hgs.h); results alone cannot tell neighbouring variants apart (several agree to the last bit), so
a dispatcher condition that silently changes turns THESE tests red while the parity tests stay green.

Column-kernel modes: col_kernel MODE 3 = forward + store the farfield (C_FWD | C_STORE), 24 = load + inverse
(C_LOAD | C_INV).  Phase modes of the fused kernels: 0 = phase taken from the field, 1 = ... and stored, 2 = stored phase used.
"""
import numpy as np
import pytest

from conftest import dispatch_of
from slmsuite_amd import _lib as L
from slmsuite_amd import synth
from slmsuite_amd.batch import HologramBatch
from slmsuite_amd.hardware import SimpleFourierSLM, SimpleSLM
from slmsuite_amd.holography.algorithms import CompressedSpotHologram, Hologram, SpotHologram

pytestmark = pytest.mark.gpu


def _image(n, seed=3):
    return synth.random_target(seed, (n, n), 0.2, 1.0)


def test_record_names_template_arguments_and_clears():
    h = Hologram(_image(256), phase=synth.seed_phase(1, (256, 256)))
    h.optimize("GS", maxiter=3, verbose=False)
    d = dispatch_of(h)
    recs = {r["name"]: r for r in d.records}
    assert any(n.startswith("col_fused_kernel<R=float,N=256,PHASE=0,STATS=false,RULE=2,NRS=16>") for n in recs), d
    # (the last row launch of a float32 call is MODE 3: it writes the phase and leaves G of the next body behind)
    assert d.count("row_kernel", R="float", N=256, MODE=0) == 1 and d.count("row_kernel", MODE=2) == 2 and d.count("row_kernel", MODE=3) == 1, d
    assert dispatch_of(h).records == []                         # reading clears
    _ = h.amp_ff                                                # the trailing transform (_populate_results): stepwise operators
    d = dispatch_of(h)
    # ... which is why this transform starts with its column pass (G of every column is already there)
    assert d.count("col_kernel", N=256, MODE=3) == 1 and d.count("row_kernel") == 0 and len(d.records) == 1, d


@pytest.mark.parametrize("n, dtype, family, extra", [
    (512, np.float32, "col_fused_kernel", dict(R="float", N=512)),
    (2048, np.float32, "col_tile2_kernel", dict(R="float", N=2048, NR=8)),     # half-width tile-resident kernel (520 SLM rows: within eight slots of 128)
    (4096, np.float32, "col_tile2_kernel", dict(R="float", N=4096, NR=5)),     # 1032 SLM rows from row 1532: five slots of 256; plain passes without a stored
                                                                                 # farfield phase run the half-width tile kernel (three workgroups per CU) since round 5
    (1024, np.float64, "col_fused_kernel", dict(R="double", N=1024, RULE=0)),
    (4096, np.float64, "col_fused_kernel", dict(R="double", N=4096, RULE=0)),      # the tile-resident kernel is fp32 only
])
def test_dense_image_reaches_the_kernel_of_its_size_and_precision(n, dtype, family, extra):
    slm = (n // 4 + 8, n // 2 - 64)
    h = Hologram(_image(n).astype(dtype), phase=synth.seed_phase(2, slm, dtype=dtype), slm_shape=slm, dtype=dtype)
    h.optimize("WGS-Leonardo", maxiter=3, verbose=False)
    d = dispatch_of(h)
    assert d.count(family, without=["list"], **extra) == 3, d
    assert len(d.families() - {family, "row_kernel"}) == 0, d
    if dtype == np.float32:       # rule compiled in: body 0 without an update (2), then Leonardo (1)
        assert d.count(family, RULE=2) == 1 and d.count(family, RULE=1) == 2, d
    assert d.count("row_kernel", R=extra["R"], N=n) == 4, d


def test_tall_slm_leaves_the_tile_kernel():
    """More than six occupied register slots (SLM rows beyond 6/16 of the pad): the per-column kernel takes the pass."""
    n, slm = 4096, (1800, 1920)
    h = Hologram(_image(n), phase=synth.seed_phase(2, slm), slm_shape=slm)
    h.optimize("WGS-Leonardo", maxiter=2, verbose=False)
    d = dispatch_of(h)
    assert d.count("col_fused_kernel", N=4096, without=["list"]) == 2 and d.count("col_tile_kernel") == 0, d
    h2 = Hologram(_image(n), phase=synth.seed_phase(2, (1500, 1920)), slm_shape=(1500, 1920))       # 6 slots: still tile-resident
    h2.optimize("WGS-Leonardo", maxiter=2, verbose=False)
    d = dispatch_of(h2)
    assert d.count("col_tile2_kernel", N=4096, NR=6) == 2 and d.count("col_fused_kernel") + d.count("col_tile_kernel") == 0, d


@pytest.mark.parametrize("method", ["WGS-Nogrette", "WGS-Wu", "WGS-tanh", "WGS-Kim", "GS"])
def test_rule_specialisation_follows_the_method(method):
    """RULE 1 is compiled for the Leonardo / Kim power rule only, RULE 2 for passes without an update; the other updates run
    the generic kernel (RULE 0), WGS-Nogrette with one more forward-only pass (EXTRAS unit) that sums feedback / target.
    (Plain passes that neither store nor read the farfield phase take the half-width tile kernel: all of GS, body 0 of the
    Wu / tanh / Nogrette / Kim runs; WGS-Kim's later bodies store the phase, so they stay on col_tile_kernel.)"""
    n, slm = 4096, (1152, 1920)
    h = Hologram(_image(n), phase=synth.seed_phase(5, slm), slm_shape=slm)
    h.optimize(method, maxiter=3, verbose=False)
    d = dispatch_of(h)
    want = {"GS": {(2, False): 3}, "WGS-Kim": {(2, False): 1, (1, False): 2},
            "WGS-Wu": {(2, False): 1, (0, False): 2}, "WGS-tanh": {(2, False): 1, (0, False): 2},
            "WGS-Nogrette": {(2, False): 1, (0, False): 2, (0, True): 2}}[method]
    got = {}
    for r in d.records:
        if r["kernel"] in ("col_tile_kernel", "col_tile2_kernel"):
            key = (int(r["args"]["RULE"]), r["args"].get("EXTRAS") == "true")
            got[key] = got.get(key, 0) + r["count"]
    assert got == want, d
    assert d.count("col_tile2_kernel") > 0 and (d.count("col_tile_kernel") > 0) == (method != "GS"), d      # (body 0 of every run is a plain pass)


def test_stepwise_option_and_callbacks_run_the_three_operators():
    n = 512
    h = Hologram(_image(n), phase=synth.seed_phase(1, (n, n)), engine_options={L.OPT_FORCE_STEPWISE: 1})
    h.optimize("WGS-Leonardo", maxiter=3, verbose=False)
    d = dispatch_of(h)
    assert d.count("col_kernel", N=n, MODE=3) == 3 and d.count("col_kernel", N=n, MODE=24) == 3, d
    assert d.count("col_fused_kernel") + d.count("col_tile_kernel") == 0, d
    # the same option keeps a callback on the host-driven loop (three engine calls per iteration) ...
    h2 = Hologram(_image(n), phase=synth.seed_phase(1, (n, n)), engine_options={L.OPT_FORCE_STEPWISE: 1})
    h2.optimize("WGS-Leonardo", maxiter=2, verbose=False, callback=lambda hh: False)
    d = dispatch_of(h2)
    assert d.count("col_kernel", MODE=3) == 2 and d.count("col_kernel", MODE=24) == 2 and d.count("col_fused_kernel") == 0, d
    # ... without it a callback runs against the device-resident loop: one fused call per iteration, whose last row launch
    # (MODE 3) leaves G behind, so that only the first call rebuilds it from the phase; nothing is materialised for a
    # callback that reads nothing
    h3 = Hologram(_image(n), phase=synth.seed_phase(1, (n, n)))
    h3.optimize("WGS-Leonardo", maxiter=3, verbose=False, callback=lambda hh: False)
    d = dispatch_of(h3)
    assert d.count("col_fused_kernel", N=n) == 3 and d.count("col_kernel") == 0, d
    assert d.count("row_kernel", MODE=0) == 1 and d.count("row_kernel", MODE=3) == 3 and d.count("row_kernel", MODE=2) == 0, d
    # a callback that looks at the farfield gets one forward transform per look, from the G the loop left behind
    seen = []
    h3.optimize("WGS-Leonardo", maxiter=2, verbose=False, callback=lambda hh: seen.append(float(hh.amp_ff[3, 5])) and False)
    d = dispatch_of(h3)
    assert len(seen) == 2 and d.count("col_kernel", MODE=3) == 2 and d.count("col_fused_kernel", N=n) == 2, d
    assert d.count("row_kernel", MODE=0) == 0, d              # dense path: G of every column is always there


@pytest.mark.parametrize("shape, lines", [((100, 150), {256, 512}), ((300, 16384), {1024, 16384}), ((101, 75), {256})])
def test_other_padded_shapes_run_bluestein_lines(shape, lines):
    slm = (shape[0] // 2, shape[1] // 2)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        h = Hologram(synth.random_target(4, shape, 0.2, 1.0), phase=synth.seed_phase(3, slm), slm_shape=slm)
    h.optimize("GS", maxiter=2, verbose=False)
    d = dispatch_of(h)
    assert d.families() == {"bluestein_lines"}, d
    assert {int(r["args"]["M"]) for r in d.records} == lines, d
    assert d.count("bluestein_lines") == 2 * 4, d          # x and y, forward and inverse, per body


def test_spot_feedback_on_a_sparse_target():
    """computational_spot: forward transform of the window-dilated spot columns (col_kernel over a list), then the fused
    kernel over the spot columns with the update compiled out; the general path with sparse columns off."""
    shape, slm = (1024, 1024), (288, 480)
    for sparse in (1, 0):
        h = SpotHologram.make_rectangular_array(shape, (8, 8), (64, 64), basis="knm", slm_shape=slm,
                                                phase=synth.seed_phase(9, slm), engine_options={L.OPT_SPARSE_COLUMNS: sparse})
        h.optimize("WGS-Leonardo", maxiter=3, verbose=False, feedback="computational_spot")
        d = dispatch_of(h)
        if sparse:
            assert d.count("col_kernel", flags=["list"], N=1024, MODE=3) == 2, d            # bodies 1, 2 update
            assert d.count("col_fused_kernel", flags=["list"], N=1024, RULE=2) == 3, d
            assert d.count("col_kernel", MODE=24) == 0, d
        else:
            assert d.count("col_kernel", without=["list"], MODE=3) == 3 and d.count("col_kernel", without=["list"], MODE=24) == 3, d
            assert d.count("col_fused_kernel") == 0, d


def test_batches_carry_the_batch_flag_and_keep_one_row_workgroups():
    shape, slm = (4096, 4096), (1152, 1920)
    host = SpotHologram.make_rectangular_array(shape, (8, 8), (64, 64), basis="knm", slm_shape=slm, phase=synth.seed_phase(2, slm))
    phases = np.stack([synth.seed_phase(40 + i, slm) for i in range(3)])
    hb = HologramBatch(shape, slm, host.target, phases)
    hb.engine.set_option(L.OPT_SPARSE_COLUMNS, 0)
    hb.optimize("WGS-Leonardo", 3)
    d = dispatch_of(hb)
    # a batch runs the half-width tile kernel (three workgroups per CU, the halves of a tile on one XCD) on its plain passes
    assert d.count("col_tile2_kernel", flags=["batch", "xmap"], N=4096, NR=5, RULE=1) == 2 and d.count("col_tile2_kernel", RULE=2) == 1, d
    assert d.count("col_tile_kernel") == 0, d
    assert d.count("row_kernel", flags=["batch"], MODE=2, NS=8, PREF=False) == 2 and d.count("row_kernel", PREF=True) == 0, d
    hb.close()


def test_compressed_forms():
    """≥ 96 spots on a separable basis: matrix cores; fewer: the run kernels; fp64 or a vortex term: per-pixel kernels."""
    slm_shape = (96, 128)
    fs = SimpleFourierSLM(SimpleSLM(slm_shape, pitch_um=(8, 8), wav_um=0.78))

    def spots(n, d=2):
        v = np.vstack([0.03 * (synth.uniform01(51, (n,), k) - 0.5) for k in range(2)])
        return v if d == 2 else np.vstack((v, 4e-6 * (synth.uniform01(51, (n,), 2) - 0.5)))

    def run(n, d=2, dtype=np.float32, opts=None):
        h = CompressedSpotHologram(spots(n, d), basis="kxy", cameraslm=fs, dtype=dtype, engine_options=opts)
        h.optimize("WGS-Leonardo", maxiter=2, verbose=False)
        return dispatch_of(h)

    d = run(120)
    assert d.families() == {"cgemm_streamk"} and d.count("cgemm_streamk") == 4, d
    d = run(40)                                               # below HGS_OPT_SEPARABLE_MIN_SPOTS
    assert d.families() == {"c_n2f_run", "c_f2n_run"} and d.count("c_n2f_run", DEG=1) == 2, d
    d = run(40, 3)
    assert d.count("c_n2f_run", DEG=2) == 2 and d.count("c_f2n_run", DEG=2) == 2, d
    d = run(120, opts={L.OPT_SEPARABLE_MIN_SPOTS: 200})
    assert d.families() == {"c_n2f_run", "c_f2n_run"}, d
    d = run(120, dtype=np.float64)                            # fp64: per-pixel kernels
    assert d.families() == {"c_n2f_partial", "c_f2n"} and d.count("c_n2f_partial", R="double", DEG=1) == 2, d
    d = run(40, opts={L.OPT_RUN_KERNELS: 0})
    assert d.families() == {"c_n2f_partial", "c_f2n"} and d.count("c_f2n", R="float", DEG=1) == 2, d
