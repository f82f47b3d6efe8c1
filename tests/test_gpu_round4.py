"""
Round 4 (``-m gpu``): arrays that stay on the device at the class surface (hgs_set_array_device, hgs_copy_phase,
get_farfield(get=False), torch CUDA tensors as phase), a hand-assigned raster on a SpotHologram, and the progress-bar
path of optimize().
"""
import numpy as np
import pytest
import torch          # before the first engine: torch's HIP runtime opens the GPU first (slmsuite_amd._lib)

from conftest import Dispatch, dispatch_of, rel_l2, phase_rel_l2, report
from slmsuite_amd import _lib as L
from slmsuite_amd import synth
from slmsuite_amd.engine import Engine
from slmsuite_amd.holography.algorithms import Hologram, SpotHologram

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_device_uploads_equal_host_uploads(dtype):
    """hgs_set_array_device: every array the engine takes from device memory lands exactly as from host memory."""
    shape, slm = (256, 512), (96, 160)
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    a, b = Engine(shape, slm, dtype, batch=2), Engine(shape, slm, dtype, batch=2)
    rng = np.random.default_rng(3)
    arrays = {L.PHASE: rng.uniform(-3, 3, (2,) + slm), L.TARGET: rng.uniform(0, 1, (2,) + shape), L.WEIGHTS: rng.uniform(0, 1, (2,) + shape),
              L.PHASE_FF: rng.uniform(-3, 3, (2,) + shape), L.PROP_KERNEL: rng.uniform(-1, 1, slm)}
    amp = rng.uniform(0.5, 1, slm)
    amp /= np.sqrt(np.sum(amp ** 2))
    a.set(L.AMP, amp.astype(dtype))
    b.set_tensor(L.AMP, torch.from_numpy(amp.astype(dtype)).cuda())
    for which, arr in arrays.items():
        arr = arr.astype(dtype)
        a.set(which, arr)
        b.set_tensor(which, torch.from_numpy(arr).cuda())
        if which != L.PROP_KERNEL:
            np.testing.assert_array_equal(a.get(which), b.get(which))
    # a wrong dtype is converted on the device; one hologram's bytes broadcast over the batch, as with hgs_set_array
    b.set_tensor(L.PHASE, torch.from_numpy(arrays[L.PHASE][:1].astype(np.float64 if dtype == np.float32 else np.float32)).cuda())
    a.set(L.PHASE, arrays[L.PHASE][:1].astype(np.float64 if dtype == np.float32 else np.float32).astype(dtype))
    np.testing.assert_array_equal(a.get(L.PHASE), b.get(L.PHASE))
    a.nearfield2farfield(True)
    b.nearfield2farfield(True)
    # (||amp||^2 is folded on the device for a device upload: same farfield to rounding)
    assert rel_l2(a.get(L.FARFIELD), b.get(L.FARFIELD)) < (1e-6 if dtype == np.float32 else 1e-14)
    t = b.get_tensor(L.FARFIELD)
    assert t.is_cuda and t.dtype == (torch.complex64 if dtype == np.float32 else torch.complex128) and tuple(t.shape) == (2,) + shape
    np.testing.assert_array_equal(t.cpu().numpy(), b.get(L.FARFIELD))
    with pytest.raises(NotImplementedError):
        b.set_from_device(L.AMP_SCALAR, t.data_ptr(), dtype().itemsize)
    a.close()
    b.close()


def test_copy_phase_between_engines_and_kernel_clear():
    shape, slm = (256, 256), (100, 120)
    a, b = Engine(shape, slm, np.float32), Engine((512, 512), slm, np.float32, batch=3)
    ph = synth.seed_phase(5, slm)
    a.set(L.PHASE, ph)
    b.copy_phase_from(a)                                       # one source hologram broadcasts over the batch
    np.testing.assert_array_equal(b.get(L.PHASE), np.stack([ph] * 3))
    c = Engine(shape, (64, 64), np.float32)
    with pytest.raises(ValueError):
        c.copy_phase_from(a)
    kern = (0.2 * synth.seed_phase(6, slm)).astype(np.float32)
    a.set(L.PROP_KERNEL, kern)
    a.nearfield2farfield()
    with_kernel = a.get(L.FARFIELD)
    a.clear_propagation_kernel()
    a.nearfield2farfield()
    plain = a.get(L.FARFIELD)
    ref = Engine(shape, slm, np.float32)
    ref.set(L.PHASE, ph)
    ref.nearfield2farfield()
    np.testing.assert_array_equal(plain, ref.get(L.FARFIELD))
    assert rel_l2(with_kernel, plain) > 1e-2
    for e in (a, b, c, ref):
        e.close()


def test_get_farfield_stays_on_the_device():
    """get=False: a torch tensor on the GPU with the values of get=True; after an optimize() the phase goes engine -> engine
    (no host copy of it exists then); amplitude and kernel are only re-sent when they are other objects."""
    shape, slm = (512, 512), (144, 240)
    amp = np.ones(slm, np.float32)
    h = Hologram(synth.random_target(3, shape), amp=amp, phase=synth.seed_phase(2, slm), slm_shape=slm)
    host = h.get_farfield((1024, 1024))
    dev = h.get_farfield((1024, 1024), get=False)
    assert isinstance(dev, torch.Tensor) and dev.is_cuda and dev.dtype == torch.complex64 and tuple(dev.shape) == (1024, 1024)
    np.testing.assert_array_equal(dev.cpu().numpy(), host)
    e_ff = h._ff_engines[(1024, 1024)]["engine"]
    sets = []
    orig = Engine.set
    Engine.set = lambda self, which, arr: (sets.append(which), orig(self, which, arr))[1]
    try:
        h.optimize("WGS-Leonardo", maxiter=3, verbose=False)
        assert h._host.get("phase") is None or "phase" in h._stale          # only the engine holds the new phase
        sets.clear()
        after = h.get_farfield((1024, 1024), get=False)
        assert L.PHASE not in sets and L.AMP not in sets and L.PROP_KERNEL not in sets, sets       # nothing went host -> device
    finally:
        Engine.set = orig
    want = Engine((1024, 1024), slm, np.float32)
    want.set(L.AMP, h.amp)
    want.set(L.PHASE, h.phase)
    want.nearfield2farfield()
    np.testing.assert_array_equal(after.cpu().numpy(), want.get(L.FARFIELD)[0])
    assert dispatch_of(e_ff).count("row_kernel", N=1024, MODE=0) >= 1
    # another depth, then none again: the kernel is sent, then cleared
    k = (0.1 * synth.seed_phase(4, slm)).astype(np.float32)
    deep = h.get_farfield((1024, 1024), propagation_kernel=k)
    flat = h.get_farfield((1024, 1024))
    np.testing.assert_array_equal(flat, after.cpu().numpy())
    assert rel_l2(deep, flat) > 1e-3
    want.close()


def test_phase_given_as_a_gpu_tensor():
    """``phase=`` / ``reset_phase`` accept a torch CUDA tensor (the reference: a CuPy array kept without a copy); it goes to
    the engine device to device and the run equals the one from the same values given as NumPy."""
    shape, slm = (512, 512), (144, 240)
    p0 = synth.seed_phase(8, slm)
    t = synth.random_target(6, shape)
    a = Hologram(t, phase=p0.copy(), slm_shape=slm)
    b = Hologram(t, phase=np.zeros(slm, np.float32), slm_shape=slm)
    b.reset_phase(torch.from_numpy(p0).cuda())
    np.testing.assert_array_equal(b.phase, p0)                 # readable before any engine exists
    a.optimize("WGS-Leonardo", maxiter=4, verbose=False)
    b.optimize("WGS-Leonardo", maxiter=4, verbose=False)
    np.testing.assert_array_equal(a.phase, b.phase)
    b.phase = torch.from_numpy(p0).cuda()                      # with an engine alive
    a.phase = p0.copy()
    a.optimize("GS", maxiter=2, verbose=False)
    b.optimize("GS", maxiter=2, verbose=False)
    np.testing.assert_array_equal(a.phase, b.phase)
    with pytest.raises(ValueError):
        b.reset_phase(torch.zeros((3, 3), device="cuda"))


def test_spot_hologram_with_a_hand_assigned_raster():
    """A SpotHologram's target normally goes up as its spot list; a raster assigned by hand (``_set_target(image)``, or
    ``h.target = ...``) must reach the engine whole -- and a NaN anywhere in it must switch MRAF on."""
    shape, slm = (256, 256), (64, 96)
    h = SpotHologram.make_rectangular_array(shape, (4, 4), (32, 32), basis="knm", slm_shape=slm, phase=synth.seed_phase(3, slm))
    h.optimize("WGS-Leonardo", maxiter=2, verbose=False)            # engine alive, spot-list upload
    image = synth.random_target(12, shape, 0.2, 1.0)
    image[:40] = np.nan
    h._set_target(image, reset_weights=True)
    assert not h._sparse_target() and h._mraf_enabled()
    h.reset_phase(synth.seed_phase(3, slm))
    h.optimize("WGS-Leonardo", maxiter=3, verbose=False, mraf_factor=0.5)
    ref = Hologram(image, phase=synth.seed_phase(3, slm), slm_shape=slm)
    ref.optimize("WGS-Leonardo", maxiter=3, verbose=False, mraf_factor=0.5)
    # iteration counters differ (h had taken two bodies before): compare from equal footing
    h2 = SpotHologram.make_rectangular_array(shape, (4, 4), (32, 32), basis="knm", slm_shape=slm, phase=synth.seed_phase(3, slm))
    h2._set_target(image, reset_weights=True)
    h2.optimize("WGS-Leonardo", maxiter=3, verbose=False, mraf_factor=0.5)
    np.testing.assert_array_equal(h2.phase, ref.phase)
    np.testing.assert_array_equal(np.nan_to_num(h2.weights), np.nan_to_num(ref.weights))
    h2.set_target(reset_weights=True)                               # back to the spot list
    assert h2._sparse_target() and not h2._mraf_enabled()


def test_progress_bar_does_not_chop_the_loop():
    """verbose=True (the default): the device loop is cut by time, not into maxiter // 20 pieces -- 20 iterations are one or
    two engine calls, 400 a handful -- with the history of a verbose=False run.  (A cut is not free of rounding: the phase
    passes through atan2 / sincos at a call boundary instead of staying a unit phasor, so the runs agree closely, not bit
    for bit; a spot array keeps that difference small, a dense pixel-wise image would amplify it 50 - 500 x per body.)"""
    shape, slm = (512, 512), (144, 240)

    def make():
        return SpotHologram.make_rectangular_array(shape, (8, 8), (32, 32), basis="knm", slm_shape=slm, phase=synth.seed_phase(8, slm))

    calls = []
    orig = Engine.iterate
    Engine.iterate = lambda self, st, n: (calls.append(n), orig(self, st, n))[1]
    try:
        a = make()
        a.optimize("WGS-Kim", maxiter=20, verbose=True)
        assert sum(calls) == 20 and len(calls) <= 2, calls
        first = a.phase.copy()
        calls.clear()
        a.optimize("WGS-Kim", maxiter=400, verbose=True)
        assert sum(calls) == 400 and len(calls) <= 12, calls
        b = make()
        b.optimize("WGS-Kim", maxiter=20, verbose=False)
        err = phase_rel_l2(first, b.phase)
        b.optimize("WGS-Kim", maxiter=400, verbose=False)
    finally:
        Engine.iterate = orig
    report("optimize(verbose=True) vs verbose=False, 20 bodies of WGS-Kim on a spot array", phase=err)
    assert err < 1e-4, err
    assert a.stats["flags"]["fixed_phase"] == b.stats["flags"]["fixed_phase"] and len(a.stats["method"]) == 420


def test_verbose_two_prints_the_method_flags(capsys):
    h = Hologram(synth.random_target(6, (128, 128)), phase=synth.seed_phase(8, (128, 128)))
    h.optimize("WGS-Kim", maxiter=1, verbose=2, fix_phase_iteration=7)
    out = capsys.readouterr().out
    assert "Optimizing with 'WGS-Kim' using the following method-specific flags:" in out
    assert "'fix_phase_iteration': 7" in out and "'feedback_exponent': 0.8" in out and "'method'" not in out


def test_stream_groups_leave_every_hologram_unchanged():
    """HologramBatch(streams=G): the batch is split over G engines whose launches overlap on the device; every hologram's
    masks are those of the single-engine batch, whatever the split (also an uneven one), and the device-side gather of the
    masks puts them in batch order."""
    from slmsuite_amd.batch import HologramBatch
    shape, slm = (1024, 1024), (288, 480)
    host = SpotHologram.make_rectangular_array(shape, (8, 8), (64, 64), basis="knm", slm_shape=slm, phase=synth.seed_phase(2, slm))
    phases = np.stack([synth.seed_phase(300 + i, slm) for i in range(5)])
    want = None
    for groups in (1, 2, 3):
        hb = HologramBatch(shape, slm, host.target, phases, spot_index=host.spot_knm_rounded, spot_amp=host.spot_amp, streams=groups)
        assert len(hb.engines) == groups and [hi - lo for lo, hi in hb.bounds] == {1: [5], 2: [3, 2], 3: [2, 2, 1]}[groups]
        hb.optimize("WGS-Kim", 12, fix_phase_iteration=5)
        hb.optimize("WGS-Kim", 3)
        got = hb.phases()
        t = torch.empty((5,) + slm, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        hb.phases_into_device(t.data_ptr(), t.numel() * 4)
        np.testing.assert_array_equal(t.cpu().numpy(), got)
        assert hb.iter == 15 and hb.flags["fixed_phase"]
        ms = hb.time_iterations("WGS-Kim", 4)
        assert ms > 0 and hb.iter == 19
        hb.close()
        if want is None:
            want = got
        else:
            np.testing.assert_array_equal(got, want)


def test_dispatch_read_buffer_protocol_and_device_upload_errors():
    """The C ABI's edge behaviour behind the Python wrappers: size query keeps the record, a short buffer fails and keeps it,
    success clears it; device uploads check sizes like host uploads."""
    import ctypes as C
    e = Engine((256, 256), (64, 96), np.float32)
    e.set(L.PHASE, synth.seed_phase(1, (64, 96)))
    e.nearfield2farfield()
    lib = e.lib
    need = C.c_size_t(0)
    assert lib.hgs_dispatch_read(e._h, None, 0, C.byref(need)) == 0 and need.value > 10
    small = C.create_string_buffer(4)
    assert lib.hgs_dispatch_read(e._h, small, 4, None) == L.HGS_ERR_ARG and small.value == b""
    again = C.c_size_t(0)
    assert lib.hgs_dispatch_read(e._h, None, 0, C.byref(again)) == 0 and again.value == need.value       # still there
    buf = C.create_string_buffer(need.value)
    assert lib.hgs_dispatch_read(e._h, buf, len(buf), None) == 0
    text = buf.value.decode()
    assert "row_kernel<R=float,N=256,MODE=0" in text and "col_kernel<R=float,N=256,MODE=3>" in text and text.endswith("\n")
    assert lib.hgs_dispatch_read(e._h, None, 0, C.byref(again)) == 0 and again.value == 1                 # cleared: just the terminator
    assert lib.hgs_dispatch_read(None, buf, len(buf), None) == L.HGS_ERR_ARG
    t = torch.zeros((64, 96), dtype=torch.float32, device="cuda")
    with pytest.raises(ValueError):
        e.set_from_device(L.PHASE, t.data_ptr(), 17)
    with pytest.raises(ValueError):
        e.set_from_device(L.PHASE, 0, t.numel() * 4)
    with pytest.raises(ValueError):
        e.set_from_device(99, t.data_ptr(), t.numel() * 4)
    e.close()


# ---- float64 MRAF with a weight update in one column pass ---------------------------------------------------------------
def _mraf_frame(shape, dtype, box=True):
    """Image in the middle with a NaN (noise) patch above part of it, zeros outside -- or (box False) NaN rows across the whole width."""
    H, W = shape
    t = np.zeros(shape, dtype)
    if box:      # (noise in a quarter of the image's columns: a column list takes the single pass where at most half of it holds noise)
        t[H // 3:2 * H // 3, W // 2 - W // 8:W // 2 + W // 8] = synth.random_target(5, (2 * H // 3 - H // 3, W // 4), 0.2, 1.0, dtype=dtype)
        t[H // 8:H // 3, W // 2 - W // 32:W // 2 + W // 32] = np.nan
    else:
        t[:] = synth.random_target(5, shape, 0.2, 1.0, dtype=dtype)
        t[: H // 5, :] = np.nan
        t[:, : W // 6] = 0
    return t


@pytest.mark.parametrize("shape, slm", [((64, 4096), (40, 1500)), ((256, 8192), (100, 3000)), ((4096, 4096), (600, 900))])
@pytest.mark.parametrize("method, extra", [("WGS-Leonardo", {}), ("WGS-Kim", dict(fix_phase_iteration=1)), ("WGS-Nogrette", {})])
@pytest.mark.parametrize("sparse", [0, 1])
def test_float64_single_pass_mraf(shape, slm, method, extra, sparse, monkeypatch):
    """
    float64 has no tile-resident kernel; its per-column kernel takes an MRAF weight update in one pass as well
    (CParams::split): signal part transformed back in place, noise part mraf_factor * F out as farfield values, one
    col_kernel<LOAD | INV> launch over the columns that hold a NaN target, row_kernel<double, ..., SPLIT> joins the two once
    ||w'|| is known.  Against the two-pass form (HGS_MRAF_SPLIT64 = 0) and the oracle, dense launches and column lists, a
    noise box (few noise columns) and noise rows across the whole width; dispatch asserted.
    """
    from oracle import hgs_oracle as orc
    if shape == (4096, 4096) and (method != "WGS-Leonardo"):
        pytest.skip("one method at the large size")
    dt = np.float64
    target = _mraf_frame(shape, dt, box=bool(sparse))
    phase0 = synth.seed_phase(17, slm, dtype=dt)
    out = {}
    monkeypatch.setenv("HGS_MRAF_PRESUM", "0")        # (round 6: the second update would take the single-inverse form -- tests/test_gpu_round6.py)
    for split in ("1", "0"):
        monkeypatch.setenv("HGS_MRAF_SPLIT64", split)
        h = Hologram(target.copy(), phase=phase0.copy(), slm_shape=slm, dtype=dt, engine_options={L.OPT_SPARSE_COLUMNS: sparse})
        h.optimize(method, maxiter=3, verbose=False, mraf_factor=0.5, **extra)
        d = dispatch_of(h)
        nog = 2 if method == "WGS-Nogrette" else 0           # its forward-only pass, in each of the two updating bodies
        lst = ["list"] if sparse else []
        if split == "1":     # body 0: one plain pass; bodies 1, 2: one pass + the inverse of the noise part + a SPLIT row launch
            assert d.count("col_fused_kernel", N=shape[0], flags=lst) == 3 + nog, d
            assert d.count("col_kernel", N=shape[0], MODE=24, flags=["list"]) == 2, d
            assert d.count("row_kernel", N=shape[1], SPLIT=True) == 2, d
        else:
            assert d.count("col_fused_kernel", N=shape[0], flags=lst) == 5 + nog, d
            assert d.count("col_kernel", MODE=24) == 0 and d.count("row_kernel", SPLIT=True) == 0, d
        out[split] = (h.phase.copy(), np.nan_to_num(np.array(h.weights, copy=True)), h.amp_ff.copy())
        h._release_engine()
    o = orc.OracleHologram(target.copy(), phase=phase0.copy(), slm_shape=slm, dtype=dt)
    o.optimize(method, maxiter=3, mraf_factor=0.5, **extra)
    errs = dict(phase_vs_two_pass=phase_rel_l2(out["1"][0], out["0"][0]), weights_vs_two_pass=rel_l2(out["1"][1], out["0"][1]),
                phase=phase_rel_l2(out["1"][0], o.phase), weights=rel_l2(out["1"][1], np.nan_to_num(o.weights)),
                amp_ff=rel_l2(out["1"][2], o.amp_ff))
    report(f"float64 single-pass MRAF {shape} {slm} {method} sparse={sparse}", **errs)
    assert max(errs.values()) < 1e-9, errs
    assert errs["phase_vs_two_pass"] > 0          # (two different sequences of launches)


@pytest.mark.parametrize("shape, slm", [((256, 4096), (100, 1500)), ((4096, 4096), (2048, 1920))])
@pytest.mark.parametrize("method, extra", [("WGS-Leonardo", {}), ("WGS-Kim", dict(fix_phase_iteration=1))])
def test_float32_single_pass_mraf_on_the_per_column_kernel(shape, slm, method, extra, monkeypatch):
    """The same route in float32 where the tile-resident kernel does not run (short columns; an SLM whose rows spread over
    eight register slots): one pass of col_fused_kernel + the inverse of the noise part + a SPLIT row launch, against the
    two-pass form (the body that splits differs by the rounding of the join) and the oracle's float32 <-> float64 distance."""
    from oracle import hgs_oracle as orc
    target = _mraf_frame(shape, np.float32, box=False)
    phase0 = synth.seed_phase(19, slm)
    out = {}
    for split in ("1", "0"):
        monkeypatch.setenv("HGS_MRAF_SPLIT64", split)
        h = Hologram(target.copy(), phase=phase0.copy(), slm_shape=slm, dtype=np.float32, engine_options={L.OPT_SPARSE_COLUMNS: 0})
        h.optimize(method, maxiter=2, verbose=False, mraf_factor=0.5, **extra)
        d = dispatch_of(h)
        assert d.count("col_tile_kernel") == 0, d
        if split == "1":
            assert d.count("col_fused_kernel", N=shape[0]) == 2 and d.count("col_kernel", N=shape[0], MODE=24, flags=["list"]) == 1, d
            assert d.count("row_kernel", N=shape[1], SPLIT=True) == 1, d
        else:
            assert d.count("col_fused_kernel", N=shape[0]) == 3 and d.count("col_kernel", MODE=24) == 0 and d.count("row_kernel", SPLIT=True) == 0, d
        out[split] = (h.phase.copy(), np.nan_to_num(np.array(h.weights, copy=True)))
        h._release_engine()
    ep, ew = phase_rel_l2(out["1"][0], out["0"][0]), rel_l2(out["1"][1], out["0"][1])
    runs = {}
    for dt in (np.float32, np.float64):
        o = orc.OracleHologram(target.astype(dt), phase=phase0.astype(dt), slm_shape=slm, dtype=dt)
        o.optimize(method, maxiter=2, mraf_factor=0.5, populate=False, **extra)
        runs[dt] = (o.phase, np.nan_to_num(o.weights))
    yp, yw = phase_rel_l2(runs[np.float32][0], runs[np.float64][0]), rel_l2(runs[np.float32][1], runs[np.float64][1])
    gp, gw = phase_rel_l2(out["1"][0], runs[np.float64][0]), rel_l2(out["1"][1], runs[np.float64][1])
    report(f"float32 single-pass MRAF on the per-column kernel {shape} {slm} {method}", phase_vs_two_pass=ep, weights_vs_two_pass=ew,
           phase_vs_f64=gp, weights_vs_f64=gw, oracle32_phase_vs_f64=yp, oracle32_weights_vs_f64=yw)
    assert ew < 1e-6 and ep < 3e-6, (ep, ew)
    assert gp < max(3e-5, 3 * yp) and gw < max(1e-5, 3 * yw), (gp, yp, gw, yw)


def test_float64_single_pass_mraf_in_a_batch(monkeypatch):
    """Three holograms in one engine, per-hologram noise columns (one of them has no NaN at all): the noise list, the
    farfield buffer of the noise part and the SPLIT row launch are per hologram -- against the same holograms one at a time."""
    from slmsuite_amd.batch import HologramBatch
    monkeypatch.setenv("HGS_MRAF_PRESUM", "0")        # (the split form on both updates; round 6's form: tests/test_gpu_round6.py)
    shape, slm, dt = (64, 4096), (40, 1500), np.float64
    targets = np.stack([_mraf_frame(shape, dt, box=False), _mraf_frame(shape, dt, box=True),
                        synth.random_target(9, shape, 0.2, 1.0, dtype=dt)])
    phases = np.stack([synth.seed_phase(70 + i, slm, dtype=dt) for i in range(3)])
    hb = HologramBatch(shape, slm, targets, phases, dtype=dt)
    try:
        hb.set_option(L.OPT_SPARSE_COLUMNS, 0)
        hb.optimize("WGS-Leonardo", maxiter=3, mraf_factor=0.5)
        d = Dispatch(hb.engine.dispatch_read())
        got = hb.phases()
    finally:
        hb.close()
    assert d.count("col_kernel", N=64, MODE=24, flags=["list"]) == 2 and d.count("row_kernel", N=4096, SPLIT=True) == 2, d
    for i in range(3):
        h = Hologram(targets[i].copy(), phase=phases[i].copy(), slm_shape=slm, dtype=dt, engine_options={L.OPT_SPARSE_COLUMNS: 0})
        h.optimize("WGS-Leonardo", maxiter=3, verbose=False, mraf_factor=0.5)
        assert phase_rel_l2(got[i], h.phase) < 1e-10, i
        h._release_engine()


@pytest.mark.parametrize("sparse", [0, 1])
def test_float64_single_pass_mraf_with_statistics(sparse):
    """... with stat_groups = ["computational"] (the statistics unit of the per-column kernel takes the same single pass) and
    WGS-Kim fixing by efficiency off: history of efficiency / uniformity and end state against the oracle."""
    from oracle import hgs_oracle as orc
    shape, slm, dt = (128, 4096), (70, 1800), np.float64
    target = _mraf_frame(shape, dt, box=bool(sparse))
    phase0 = synth.seed_phase(23, slm, dtype=dt)
    h = Hologram(target.copy(), phase=phase0.copy(), slm_shape=slm, dtype=dt, engine_options={L.OPT_SPARSE_COLUMNS: sparse})
    h.optimize("WGS-Leonardo", maxiter=4, verbose=False, mraf_factor=0.5, stat_groups=["computational"])
    d = dispatch_of(h)
    assert d.count("col_fused_kernel", N=128, STATS=True) == 4 and d.count("col_kernel", MODE=24) == 3 and d.count("row_kernel", SPLIT=True) == 3, d
    o = orc.OracleHologram(target.copy(), phase=phase0.copy(), slm_shape=slm, dtype=dt)
    o.optimize("WGS-Leonardo", maxiter=4, mraf_factor=0.5, stat_groups=["computational"])
    assert phase_rel_l2(h.phase, o.phase) < 1e-9 and rel_l2(np.nan_to_num(h.weights), np.nan_to_num(o.weights)) < 1e-9
    for key in ("efficiency", "uniformity", "pkpk_err", "std_err"):
        np.testing.assert_allclose(h.stats["stats"]["computational"][key], o.stats["stats"]["computational"][key], rtol=1e-8, atol=1e-12, err_msg=key)
