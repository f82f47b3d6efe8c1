"""
Wall time of optimize() as a user sees it (cfg 2 geometry, engine default path AND dense kernels): construction, the
cold first call (engine creation + uploads + 50 bodies + the trailing transform), warm calls, the resident rate
(hgs_iterate_timed), reset() + re-optimise, and the wavefront-calibration re-optimisation pattern
(cameraslms.py:1840-1930: a CompressedSpotHologram re-optimised with "GS" x 3 while spot_zernike changes).

    python tools/e2e_timing.py [out.json]

One JSON object; `*_ms` are host wall-clock milliseconds, `*_its` iterations per second.
"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402,F401  (first: torch's HIP runtime must open the GPU before libhgs.so does, see _lib.load)
from slmsuite_amd import _lib as L  # noqa: E402
from slmsuite_amd import synth  # noqa: E402
from slmsuite_amd.hardware import SimpleFourierSLM, SimpleSLM  # noqa: E402
from slmsuite_amd.holography.algorithms import CompressedSpotHologram, SpotHologram  # noqa: E402

SH, SLM = (4096, 4096), (1152, 1920)
K = 50


def ms(t0):
    return 1e3 * (time.perf_counter() - t0)


def run(dense):
    opts = {L.OPT_SPARSE_COLUMNS: 0} if dense else {}
    out = {}
    p0 = synth.seed_phase(2, SLM)
    t = time.perf_counter()
    h = SpotHologram.make_rectangular_array(SH, (32, 32), (64, 64), basis="knm", slm_shape=SLM, phase=p0, engine_options=opts)
    out["construct_ms"] = ms(t)
    t = time.perf_counter()
    h.optimize("WGS-Leonardo", maxiter=K, verbose=False)
    h._engine.sync()
    out["cold_optimize50_ms"] = ms(t)
    t = time.perf_counter()
    _ = h.phase
    out["read_phase_ms"] = ms(t)
    warm = []
    for _i in range(5):
        t = time.perf_counter()
        h.optimize("WGS-Leonardo", maxiter=K, verbose=False)
        h._engine.sync()
        warm.append(ms(t))
    out["warm_optimize50_ms"] = float(np.median(warm))
    # resident rate: the loop alone, HIP events on the engine stream
    st = h._make_step()
    e = h._get_engine()
    e.iterate_timed(st, 20)
    res = [e.iterate_timed(st, K) for _i in range(10)]
    out["resident_50_ms"] = float(np.median(res))
    out["resident_its"] = K / (1e-3 * out["resident_50_ms"])
    out["cold_its"] = K / (1e-3 * out["cold_optimize50_ms"])
    out["warm_its"] = K / (1e-3 * out["warm_optimize50_ms"])
    out["cold_over_resident"] = out["cold_optimize50_ms"] / out["resident_50_ms"]
    # reset() keeps the engine: weights from the target on the device, nothing re-uploaded but a new phase
    t = time.perf_counter()
    h.reset(reset_phase=False)
    out["reset_keep_phase_ms"] = ms(t)
    t = time.perf_counter()
    h.optimize("WGS-Leonardo", maxiter=K, verbose=False)
    h._engine.sync()
    out["optimize50_after_reset_ms"] = ms(t)
    t = time.perf_counter()
    h.reset_phase(p0)
    h.reset(reset_phase=False)
    h.optimize("WGS-Leonardo", maxiter=K, verbose=False)
    h._engine.sync()
    out["reset_new_phase_optimize50_ms"] = ms(t)
    t = time.perf_counter()
    h.optimize("WGS-Leonardo", maxiter=K, verbose=False, stat_groups=["computational_spot"])
    out["optimize50_spot_stats_ms"] = ms(t)
    t = time.perf_counter()
    _ = h.weights
    out["read_weights_ms"] = ms(t)
    # a second hologram of the same geometry in the same process (allocator warm)
    t = time.perf_counter()
    h2 = SpotHologram.make_rectangular_array(SH, (32, 32), (64, 64), basis="knm", slm_shape=SLM, phase=synth.seed_phase(3, SLM),
                                             engine_options=opts)
    h2.optimize("WGS-Leonardo", maxiter=K, verbose=False)
    h2._engine.sync()
    out["second_hologram_construct_plus_optimize50_ms"] = ms(t)
    return out


def breakdown():
    """Where a cold optimize(50) goes (engine default path): every Engine call of the class, synchronised and timed."""
    from slmsuite_amd.engine import Engine
    acc = {}

    def wrap(name):
        orig = getattr(Engine, name)

        def f(self, *a, **k):
            t = time.perf_counter()
            r = orig(self, *a, **k)
            if name != "__init__" and getattr(self, "_h", None) is not None and self._h.value:
                self.sync()
            acc[name] = acc.get(name, 0.0) + ms(t)
            return r
        setattr(Engine, name, f)
        return orig

    names = ["__init__", "set", "set_sparse", "reset_weights", "reset", "iterate", "nearfield2farfield", "get", "set_option"]
    saved = {n: wrap(n) for n in names}
    try:
        t = time.perf_counter()
        h = SpotHologram.make_rectangular_array(SH, (32, 32), (64, 64), basis="knm", slm_shape=SLM, phase=synth.seed_phase(5, SLM))
        construct = ms(t)
        t = time.perf_counter()
        h.optimize("WGS-Leonardo", maxiter=K, verbose=False)
        total = ms(t)
        t = time.perf_counter()
        _ = h.phase
        rd = ms(t)
    finally:
        for n, o in saved.items():
            setattr(Engine, n, o)
    out = {"construct_ms": construct, "optimize50_total_ms": total, "read_phase_ms": rd}
    out.update({f"engine.{k}_ms": v for k, v in acc.items()})
    out["host_side_ms"] = total - sum(v for k, v in acc.items() if k != "get")
    return out


def wavefront_pattern():
    """wavefront_calibrate_zernike's loop body: GS x 3 per measurement while the coefficients of the spots change."""
    slm_shape = (1152, 1920)
    fs = SimpleFourierSLM(SimpleSLM(slm_shape, pitch_um=(8, 8), wav_um=0.78))
    basis = np.array([2, 1, 4, 3, 5, 7, 8, 6, 9, 12])
    N = 16
    z = np.zeros((len(basis), N))
    z[:2] = 600 * (synth.uniform01(41, (2, N), 0) - 0.5)
    z[2:] = 1.0 * (synth.uniform01(42, (len(basis) - 2, N), 0) - 0.5)
    t = time.perf_counter()
    h = CompressedSpotHologram(z.copy(), basis=basis, cameraslm=fs)
    h.reset_phase(synth.seed_phase(40, slm_shape))
    out = {"construct_ms": ms(t)}
    t = time.perf_counter()
    h.optimize("GS", maxiter=3, verbose=False)
    h._engine.sync()
    out["first_gs3_ms"] = ms(t)
    rounds = []
    for rnd in range(12):
        z[2 + rnd % 8, :] += 0.05
        h.spot_zernike = z.copy()
        t = time.perf_counter()
        h.optimize("GS", maxiter=3, verbose=False)
        _ = h.get_phase()
        rounds.append(ms(t))
    out["reoptimise_gs3_plus_get_phase_ms"] = float(np.median(rounds))
    out["n_spots"], out["n_terms"], out["slm_shape"] = N, len(basis), list(slm_shape)
    return out


def callback_pattern():
    """
    optimize(callback=...) at cfg 2 (_hologram.py:1465-1490: the callback fires every iteration).  Since round 5 it runs
    against the device-resident loop -- one fused engine call per iteration, what the callback reads materialised on demand;
    the host-driven loop of the general operators (rounds 1 - 4, now behind HGS_OPT_FORCE_STEPWISE) beside it.
    """
    out = {}
    for name, opts in (("engine_default", {}), ("dense_kernels", {L.OPT_SPARSE_COLUMNS: 0})):
        h = SpotHologram.make_rectangular_array(SH, (32, 32), (64, 64), basis="knm", slm_shape=SLM, phase=synth.seed_phase(2, SLM),
                                                engine_options=opts)
        h.optimize("WGS-Leonardo", maxiter=K, verbose=False)
        h._engine.sync()

        def timed(**kw):
            ts = []
            for _i in range(5):
                t = time.perf_counter()
                h.optimize("WGS-Leonardo", maxiter=K, verbose=False, **kw)
                h._engine.sync()
                ts.append(ms(t))
            return float(np.median(ts))
        r = {"no_callback_optimize50_ms": timed()}
        r["callback_reads_nothing_ms"] = timed(callback=lambda hh: False)
        r["callback_reads_iter_and_stats_ms"] = timed(callback=lambda hh: hh.iter < 0 or len(hh.stats["method"]) < 0)
        ky, kx = h.spot_knm_rounded[1], h.spot_knm_rounded[0]
        r["callback_reads_amp_ff_every_10th_ms"] = timed(callback=lambda hh: hh.iter % 10 == 0 and float(hh.amp_ff[ky[0], kx[0]]) < 0)
        r["callback_over_no_callback"] = r["callback_reads_nothing_ms"] / r["no_callback_optimize50_ms"]
        g = SpotHologram.make_rectangular_array(SH, (32, 32), (64, 64), basis="knm", slm_shape=SLM, phase=synth.seed_phase(2, SLM),
                                                engine_options={**opts, L.OPT_FORCE_STEPWISE: 1})
        g.optimize("WGS-Leonardo", maxiter=K, verbose=False, callback=lambda hh: False)
        t = time.perf_counter()
        g.optimize("WGS-Leonardo", maxiter=K, verbose=False, callback=lambda hh: False)
        g._engine.sync()
        r["host_driven_loop_callback_ms"] = ms(t)
        out[name] = r
    return out


def wavefront_breakdown():
    """Where one round of the wavefront-calibration pattern goes: every Engine call of the round, synchronised and timed."""
    from slmsuite_amd.engine import Engine
    slm_shape = (1152, 1920)
    fs = SimpleFourierSLM(SimpleSLM(slm_shape, pitch_um=(8, 8), wav_um=0.78))
    basis = np.array([2, 1, 4, 3, 5, 7, 8, 6, 9, 12])
    N = 16
    z = np.zeros((len(basis), N))
    z[:2] = 600 * (synth.uniform01(41, (2, N), 0) - 0.5)
    z[2:] = 1.0 * (synth.uniform01(42, (len(basis) - 2, N), 0) - 0.5)
    h = CompressedSpotHologram(z.copy(), basis=basis, cameraslm=fs)
    h.reset_phase(synth.seed_phase(40, slm_shape))
    h.optimize("GS", maxiter=3, verbose=False)
    _ = h.get_phase()
    acc, cnt = {}, {}

    def wrap(name):
        orig = getattr(Engine, name)

        def f(self, *a, **k):
            t = time.perf_counter()
            r = orig(self, *a, **k)
            self.sync()
            acc[name] = acc.get(name, 0.0) + ms(t)
            cnt[name] = cnt.get(name, 0) + 1
            return r
        setattr(Engine, name, f)
        return orig

    names = ["set", "set_sparse", "reset_weights", "reset", "iterate", "nearfield2farfield", "get", "set_option"]
    saved = {n: wrap(n) for n in names}
    rounds = 8
    try:
        t0 = time.perf_counter()
        for rnd in range(rounds):
            z[2 + rnd % 8, :] += 0.05
            h.spot_zernike = z.copy()
            h.optimize("GS", maxiter=3, verbose=False)
            _ = h.get_phase()
        total = ms(t0) / rounds
    finally:
        for n, o in saved.items():
            setattr(Engine, n, o)
    out = {"round_ms_with_a_sync_after_every_engine_call": total}
    out.update({f"engine.{k}_ms_per_round": v / rounds for k, v in acc.items()})
    out.update({f"engine.{k}_calls_per_round": cnt[k] / rounds for k in cnt})
    out["host_side_ms_per_round"] = total - sum(acc.values()) / rounds
    return out


def camera_frame_pattern():
    """
    SimulatedCamera's per-frame work (hardware/cameras/simulated.py:344-376): a fresh amplitude array and phase are handed
    to a Hologram, ``get_farfield(get=False)`` and |ff|^2 follow.  4096^2 grid, S = 1152 x 1920.  Four ways:
      host_out            get=True: the 134 MB field comes to the host (what every call did before round 4)
      device_out          get=False: the field stays on the GPU as a torch tensor, |ff|^2 formed there
      device_out_same_amp ... and the amplitude object is the one of the previous frame (not re-sent)
      device_phase        ... and the phase is handed over as a torch CUDA tensor (no host -> device copy either)
    plus the frame after an optimize() (phase taken from the hologram's own engine on the device).
    """
    import torch
    from slmsuite_amd.holography.algorithms import Hologram
    amp0 = np.ones(SLM, dtype=np.float32)
    h = Hologram(SH, amp=amp0, phase=synth.seed_phase(9, SLM), slm_shape=SLM)
    out = {}

    def frames(n, fresh_amp, get, device_phase=False):
        ts = []
        for i in range(n):
            ph = synth.seed_phase(100 + i, SLM)
            ph_dev = torch.from_numpy(ph).cuda() if device_phase else None
            torch.cuda.synchronize()
            t = time.perf_counter()
            if fresh_amp:
                h.amp = np.array(amp0, copy=True) * (1 / np.sqrt(amp0.size))
            h.reset_phase(ph_dev if device_phase else ph)
            ff = h.get_farfield(get=get)
            img = (ff.abs() ** 2) if not get else np.abs(ff) ** 2
            if not get:
                torch.cuda.synchronize()
            ts.append(ms(t))
            del img
        return float(np.median(ts[1:]))

    out["host_out_ms"] = frames(6, True, True)
    out["device_out_ms"] = frames(8, True, False)
    out["device_out_same_amp_ms"] = frames(8, False, False)
    out["device_phase_ms"] = frames(8, False, False, device_phase=True)
    target = np.zeros(SH, dtype=np.float32)
    target[1800:2300, 1800:2300] = 1
    h2 = Hologram(target, phase=synth.seed_phase(9, SLM), slm_shape=SLM)
    ts = []
    for _i in range(6):
        h2.optimize("GS", maxiter=2, verbose=False)
        torch.cuda.synchronize()
        t = time.perf_counter()
        ff = h2.get_farfield(get=False)
        torch.cuda.synchronize()
        ts.append(ms(t))
    out["after_optimize_device_out_ms"] = float(np.median(ts[1:]))
    return out


if __name__ == "__main__":
    res = {"workload": "cfg2: SpotHologram 32x32 on 4096^2, S = 1152x1920, WGS-Leonardo x 50",
           "engine_default": run(False), "dense_kernels": run(True), "cold_call_breakdown": breakdown(),
           "wavefront_calibration_pattern": wavefront_pattern(), "wavefront_calibration_breakdown": wavefront_breakdown(),
           "callback_pattern": callback_pattern(), "camera_frame_pattern": camera_frame_pattern()}
    txt = json.dumps(res, indent=1)
    print(txt)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(txt + "\n")
