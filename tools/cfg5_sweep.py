"""
BASELINE config 5: "Mixed-region-amplitude-freedom Hologram, 2048^2 image target, 8192^2 pad, fp64 vs fp32 tolerance
sweep" (SURVEY 8d row 5).  Runs on the GPU box (engine + the CPU oracle as the checker); writes one JSON.

    python tools/cfg5_sweep.py profiles/r03/cfg5_sweep.json          # iterations 1..20, GS and WGS-Leonardo

For every iteration k the state S_k (phase, weights) of the engine's own fp32 run is the teacher.  From S_k, ONE loop
body is computed four ways -- oracle float64 (taken as the truth), oracle float32 (what the reference's arithmetic
does), engine float32, engine float64 -- and compared on the new phase (distance of unit phasors) and the new weights, each as
relative L2 norm, median and 99th percentile of the per-pixel error (see `spread`):

    engine32_vs_oracle32   the per-step parity number (north-star tolerance 1e-5 on amplitudes; SURVEY 7-5: 2e-6)
    engine64_vs_oracle64   the same in double precision
    oracle32_vs_truth      the reference's own fp32 rounding error on this step   } the yardstick: an fp32 engine is
    engine32_vs_truth      the engine's fp32 rounding error on the same step       } done when these two are alike

and, free-running from the common seed (no teacher), the divergence curves fp32 <-> fp64 of the engine and of the
oracle: how fast ANY fp32 implementation of this loop leaves its fp64 twin (why end states are not compared).
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import hgs_oracle as orc          # noqa: E402  (the checker)
from slmsuite_amd import synth                 # noqa: E402
from slmsuite_amd.holography.algorithms import Hologram   # noqa: E402

SHAPE, SLM = (8192, 8192), (1152, 1920)
KW = dict(mraf_factor=0.5)


def cfg5_target(n=8192, dtype=np.float32):
    """zeros; centred 3072^2 box = NaN (noise region); centred 2048^2 = uniform(0.2, 1) image."""
    t = np.zeros((n, n), dtype=dtype)
    a, b = n // 2 - 1536, n // 2 + 1536
    t[a:b, a:b] = np.nan
    a, b = n // 2 - 1024, n // 2 + 1024
    t[a:b, a:b] = synth.random_target(5, (2048, 2048), 0.2, 1.0, dtype=dtype)
    return t


def phasor_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.sqrt(np.mean(np.abs(np.exp(1j * a) - np.exp(1j * b)) ** 2)))


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.sqrt(np.nansum((a - b) ** 2)) / np.sqrt(np.nansum(b ** 2)))


def spread(a, b, kind):
    """
    Distance of a from b three ways: the relative L2 norm, and -- pixel-wise WGS on a dense MRAF image divides by speckle
    amplitudes, so a handful of pixels near a zero of the field carry most of any L2 difference, and WHICH pixels they
    are changes with every rounding -- two robust figures: the median and the 99th percentile of the per-pixel error
    (phase: |e^{ia} - e^{ib}|; weights: |a - b| / rms(b) over the pixels where b is not zero).
    """
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if kind == "phase":
        d = np.abs(np.exp(1j * a) - np.exp(1j * b)).ravel()
        l2 = float(np.sqrt(np.mean(d ** 2)))
    else:
        nz = b != 0
        l2 = rel_l2(a, b)
        if not np.any(nz):
            return {"l2": l2, "median": 0.0, "p99": 0.0}
        d = np.abs(a[nz] - b[nz]) / np.sqrt(np.mean(b[nz] ** 2))
    sub = d[:: max(1, d.size // 2_000_000)]          # percentiles on a 2 M-pixel subsample
    return {"l2": l2, "median": float(np.median(sub)), "p99": float(np.percentile(sub, 99))}


def _force(h, phase, weights, k, dtype):
    """Put a hologram (product or oracle: same attribute names) into the state before loop body k."""
    h.phase = np.array(phase, dtype=dtype)
    h.weights = np.array(weights, dtype=dtype)
    h.iter = k


def sweep(steps=tuple(range(1, 21)), methods=("GS", "WGS-Leonardo"), free_run=True, log=print):
    res = {"workload": "cfg5: Hologram MRAF (mraf_factor 0.5), 8192^2 pad of 1152x1920, NaN box 3072^2, image 2048^2",
           "steps": list(steps), "methods": {}}
    t32, t64 = cfg5_target(dtype=np.float32), cfg5_target(dtype=np.float64)
    p0 = synth.seed_phase(5, SLM, dtype=np.float32)
    for method in methods:
        t0 = time.time()
        run32 = Hologram(t32, phase=p0.copy(), slm_shape=SLM, dtype=np.float32)        # teacher + free run
        step32 = Hologram(t32, phase=p0.copy(), slm_shape=SLM, dtype=np.float32)
        step64 = Hologram(t64, phase=p0.astype(np.float64), slm_shape=SLM, dtype=np.float64)
        o32 = orc.OracleHologram(t32, phase=p0.copy(), slm_shape=SLM, dtype=np.float32)
        o64 = orc.OracleHologram(t64, phase=p0.astype(np.float64), slm_shape=SLM, dtype=np.float64)
        rows = []
        done = 0
        for k in steps:
            # state before body k of the engine's fp32 run (iteration numbers as the reference counts them: body 0 is
            # the first; "iteration k" of the sweep is the body that starts from S_k, k = 1..20)
            run32.optimize(method, maxiter=k - done, verbose=False, **KW)
            done = k
            ph, w = run32.phase, run32.weights
            outs = {}
            for name, h, dt in (("e32", step32, np.float32), ("e64", step64, np.float64)):
                _force(h, ph, w, k, dt)
                h.optimize(method, maxiter=1, verbose=False, **KW)
                outs[name] = (h.phase.astype(np.float64), h.weights.astype(np.float64))
            for name, o, dt in (("o32", o32, np.float32), ("o64", o64, np.float64)):
                _force(o, ph, w, k, dt)
                o.optimize(method, maxiter=1, populate=False, **KW)
                outs[name] = (o.phase.astype(np.float64), o.weights.astype(np.float64))
            row = {"k": k}
            for a, b, tag in (("e32", "o32", "engine32_vs_oracle32"), ("e64", "o64", "engine64_vs_oracle64"),
                              ("o32", "o64", "oracle32_vs_truth"), ("e32", "o64", "engine32_vs_truth")):
                row[tag] = {"phase": spread(outs[a][0], outs[b][0], "phase"), "weights": spread(outs[a][1], outs[b][1], "weights")}
            rows.append(row)
            log(f"{method} k={k}: " + "  ".join(f"{t}={row[t]['phase']['l2']:.2e}/{row[t]['weights']['l2']:.2e}" for t in row if t != "k"))
        entry = {"teacher_forced": rows}
        del step32, step64
        if free_run:
            n = max(steps)
            f32 = Hologram(t32, phase=p0.copy(), slm_shape=SLM, dtype=np.float32)
            f64 = Hologram(t64, phase=p0.astype(np.float64), slm_shape=SLM, dtype=np.float64)
            q32 = orc.OracleHologram(t32, phase=p0.copy(), slm_shape=SLM, dtype=np.float32)
            q64 = orc.OracleHologram(t64, phase=p0.astype(np.float64), slm_shape=SLM, dtype=np.float64)
            curve = []
            for k in range(1, n + 1):
                for h in (f32, f64):
                    h.optimize(method, maxiter=1, verbose=False, **KW)
                for o in (q32, q64):
                    o.optimize(method, maxiter=1, populate=False, **KW)
                curve.append({"bodies": k,
                              "engine32_vs_engine64": phasor_l2(f32.phase, f64.phase),
                              "oracle32_vs_oracle64": phasor_l2(q32.phase, q64.phase),
                              "engine32_vs_oracle32": phasor_l2(f32.phase, q32.phase),
                              "engine64_vs_oracle64": phasor_l2(f64.phase, q64.phase)})
                log(f"{method} free run {k}: " + "  ".join(f"{t}={v:.2e}" for t, v in curve[-1].items() if t != "bodies"))
            entry["free_running_phase_divergence"] = curve
            del f32, f64
        entry["wall_s"] = time.time() - t0
        res["methods"][method] = entry
        del run32
    return res


if __name__ == "__main__":
    out = sweep()
    txt = json.dumps(out, indent=1)
    if len(sys.argv) > 1:
        os.makedirs(os.path.dirname(os.path.abspath(sys.argv[1])), exist_ok=True)
        open(sys.argv[1], "w").write(txt + "\n")
    else:
        print(txt)
