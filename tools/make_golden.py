#!/usr/bin/env python
"""
Generate the golden fixtures under tests/golden/ by running the REAL reference
(/root/reference, slmsuite 0.4.1, NumPy backend) on seeded inputs.

Runs only in the build container (the reference never travels to the GPU box; the
fixtures do).  Nothing from the reference is copied: fixtures hold inputs and outputs only.

    python tools/make_golden.py            # small fixtures (seconds)
    python tools/make_golden.py --cfg2     # additionally the 4096^2 config-2 summaries: cfg2 (~3 min), cfg2kim
                                           # (~2 min), cfg2seeds, cfg2seedsb (16 runs each, ~12 min on 4 cores), cfg2steps (~3 min, 27 MB); --only NAME picks one
"""
import argparse
import json
import os
import sys
import types
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slmsuite_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def import_reference():
    """SURVEY Appendix B recipe: empty cv2/h5py stubs, Agg backend, reference on sys.path."""
    warnings.simplefilter("ignore")
    import matplotlib
    matplotlib.use("Agg")
    for name in ("cv2", "h5py"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.path.insert(0, "/root/reference")
    from slmsuite.holography import algorithms, toolbox, analysis
    return algorithms, toolbox, analysis


def record_run(holo, method, maxiter, save_iters, **kw):
    """
    One optimize(maxiter) call with a snapshot callback.  The callback fires after the forward
    transform of loop body k (_hologram.py:1473), i.e. it sees the state *before* body k:
    phase_k, weights_k, phase_ff as left by body k-1, plus amp_ff = |FFT(phase_k)|.
    (Calling optimize(maxiter=1) repeatedly would NOT be equivalent for WGS-Kim: the trailing
    _populate_results overwrites the frozen phase_ff, _hologram.py:949.)
    """
    out = {}

    def snap(h):
        k = h.iter - it0
        if k in save_iters:
            out[f"phase_{k}"] = np.array(h.phase, copy=True)
            out[f"weights_{k}"] = np.array(h.weights, copy=True)
            if h.phase_ff is not None:
                out[f"phaseff_{k}"] = np.array(h.phase_ff, copy=True)
            out[f"ampff_{k}"] = np.array(h.amp_ff, copy=True)
            out[f"fixed_{k}"] = np.array(bool(h.flags.get("fixed_phase", False)))
        return False

    it0 = holo.iter
    holo.optimize(method, maxiter=maxiter, verbose=False, callback=snap, **kw)
    out["final_phase"] = np.array(holo.phase, copy=True)
    out["final_weights"] = np.array(holo.weights, copy=True)
    out["final_ampff"] = np.array(holo.amp_ff, copy=True)
    out["final_phaseff"] = np.array(holo.phase_ff, copy=True)
    out["fixed_history"] = np.array([bool(x) for x in holo.stats["flags"]["fixed_phase"]])
    for g, d in holo.stats["stats"].items():
        for n, lst in d.items():
            out[f"stats_{g}_{n}"] = np.array(lst, dtype=float)
    return out


def slim(meta, arrays):
    """Keep the fixtures small: P-sized per-iteration snapshots only where a test needs them."""
    kind, variant = meta["kind"], meta.get("variant")
    keep = {}
    for k, v in arrays.items():
        head, _, tail = k.rpartition("_")
        if head in ("phase", "weights", "phaseff", "ampff", "fixed") and tail.isdigit():
            it = int(tail)
            if head == "fixed":
                keep[k] = v
            elif kind == "hologram" and variant == "A":
                if (head in ("phase", "weights") and it in (1, 2, 5, 6)) or (head == "phaseff" and it == 5):
                    keep[k] = v
            elif kind in ("hologram", "mraf"):
                if (head == "phase" and it in (1, 2, 5)) or (head == "weights" and it == 2):
                    keep[k] = v
            else:
                keep[k] = v
        elif k == "final_phaseff" and not (kind == "hologram" and variant == "A"):
            continue
        else:
            keep[k] = v
    return keep


def save(name, meta, arrays):
    arrays = slim(meta, dict(arrays))
    arrays["meta"] = np.array(json.dumps(meta))
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  {name}.npz  {os.path.getsize(path) / 1024:.0f} KiB")


METHOD_KW = {
    "GS": {},
    "WGS-Leonardo": {},
    "WGS-Kim": {"fix_phase_iteration": 4},
    "WGS-Nogrette": {},
    "WGS-Wu": {},
    "WGS-tanh": {},
}


def gen_hologram_cases(alg):
    """F1/F2: every method x {A: 64^2 S=P scalar amp, B: 128^2 pad of 48x80, array amp + kernel}."""
    for mi, (method, kw) in enumerate(METHOD_KW.items()):
        for variant in ("A", "B"):
            for dt in (np.float32, np.float64):
                if dt is np.float64 and not (variant == "A" and method in ("GS", "WGS-Kim")):
                    continue
                seed = 100 + mi
                if variant == "A":
                    shape, slm = (64, 64), (64, 64)
                    amp, kernel = None, None
                else:
                    shape, slm = (128, 128), (48, 80)
                    amp = synth.gaussian_amp(slm, dtype=dt)
                    kernel = (0.3 * synth.seed_phase(seed + 50, slm)).astype(dt)
                target = synth.random_target(seed, shape, dtype=dt)
                phase0 = synth.seed_phase(seed, slm, dtype=dt)
                h = alg.Hologram(target.copy(), amp=None if amp is None else amp.copy(),
                                 phase=phase0.copy(), slm_shape=slm, dtype=dt,
                                 propagation_kernel=None if kernel is None else kernel.copy())
                maxiter = 8
                rec = record_run(h, method, maxiter, save_iters=(0, 1, 2, 3, 5, 6, 7),
                                 stat_groups=["computational"], **kw)
                meta = dict(kind="hologram", method=method, variant=variant, seed=seed,
                            shape=shape, slm_shape=slm, dtype=np.dtype(dt).name, maxiter=maxiter,
                            kwargs=kw, amp="gaussian" if amp is not None else None,
                            kernel_seed=None if kernel is None else seed + 50)
                tag = f"holo_{method.replace('-', '')}_{variant}_{'f32' if dt is np.float32 else 'f64'}"
                save(tag, meta, rec)


def mraf_target(seed, n=128, dtype=np.float32):
    """zeros; centred (3n/4)^2 box = NaN (noise); centred (n/2)^2 = uniform(0.2, 1) image."""
    t = np.zeros((n, n), dtype=dtype)
    a, b = n // 8, n - n // 8
    t[a:b, a:b] = np.nan
    a, b = n // 4, n - n // 4
    t[a:b, a:b] = synth.random_target(seed, (b - a, b - a), 0.2, 1.0, dtype=dtype)
    return t


def gen_mraf_cases(alg):
    """F3: MRAF with mraf_factor in {None, 0.5}, zero_factor in {absent, 1}."""
    for method in ("GS", "WGS-Leonardo"):
        for mf in (None, 0.5):
            for zf in (None, 1.0):
                if zf is not None and mf is None:
                    continue
                if method == "GS" and zf is not None:
                    continue
                if method == "WGS-Leonardo" and mf is None:
                    continue
                seed = 300
                shape, slm = (128, 128), (64, 96)
                target = mraf_target(seed)
                phase0 = synth.seed_phase(seed, slm)
                kw = {}
                if mf is not None:
                    kw["mraf_factor"] = mf
                if zf is not None:
                    kw["zero_factor"] = zf
                h = alg.Hologram(target.copy(), phase=phase0.copy(), slm_shape=slm)
                rec = record_run(h, method, 5, save_iters=(0, 1, 2, 3, 5),
                                 stat_groups=["computational"], **kw)
                if hasattr(h, "zero_weights"):
                    rec["final_zero_weights"] = np.array(h.zero_weights)
                meta = dict(kind="mraf", method=method, seed=seed, shape=shape, slm_shape=slm,
                            dtype="float32", maxiter=5, kwargs=kw)
                tag = f"mraf_{method.replace('-', '')}_mf{mf}_zf{zf}"
                save(tag, meta, rec)


def gen_spot_cases(alg):
    """F4: SpotHologram 256^2 pad of 72x120, 8x8 grid pitch 16; three feedback modes."""
    shape, slm = (256, 256), (72, 120)
    for fb in ("computational", "computational_spot", "external_spot"):
        for method in ("WGS-Leonardo", "WGS-Kim"):
            if method == "WGS-Kim" and fb != "computational_spot":
                continue
            seed = 400
            phase0 = synth.seed_phase(seed, slm)
            h = alg.SpotHologram.make_rectangular_array(
                shape, array_shape=(8, 8), array_pitch=(16, 16), basis="knm",
                slm_shape=slm, phase=phase0.copy())
            kw = dict(METHOD_KW[method])
            if fb == "external_spot":
                h.external_spot_amp = h.spot_amp * (1 + 0.2 * (synth.uniform01(seed, (64,), 5) - 0.5))
            sg = ["computational", "computational_spot"]
            rec = record_run(h, method, 8, save_iters=(0, 1, 2, 3, 5, 6, 7), feedback=fb,
                             stat_groups=sg, **kw)
            rec["spot_knm"] = np.array(h.spot_knm)
            rec["spot_knm_rounded"] = np.array(h.spot_knm_rounded)
            rec["spot_amp"] = np.array(h.spot_amp)
            rec["external_spot_amp"] = np.array(h.external_spot_amp)
            rec["target_spots"] = np.array(h.target[h.spot_knm_rounded[1], h.spot_knm_rounded[0]])
            rec["final_ampff_sub"] = rec["final_ampff"][::4, ::4].copy()
            rec["final_ampff_spots"] = rec["final_ampff"][h.spot_knm_rounded[1], h.spot_knm_rounded[0]]
            rec["final_weights_spots"] = rec["final_weights"][h.spot_knm_rounded[1], h.spot_knm_rounded[0]]
            rec["final_weights_sum"] = np.array(float(np.sum(rec["final_weights"].astype(float))))
            del rec["final_ampff"], rec["final_weights"]
            # drop the big P-sized per-iteration arrays except weights at spots
            ky, kx = h.spot_knm_rounded[1], h.spot_knm_rounded[0]
            for k in list(rec.keys()):
                if k.startswith(("weights_", "ampff_", "phaseff_")):
                    if k.startswith("phaseff_"):
                        del rec[k]
                        continue
                    if k.startswith(("weights_", "ampff_")):
                        rec[k + "_spots"] = rec[k][ky, kx]
                        del rec[k]
            meta = dict(kind="spot", method=method, feedback=fb, seed=seed, shape=shape, slm_shape=slm,
                        dtype="float32", maxiter=8, kwargs=kw, array_shape=(8, 8), array_pitch=(16, 16),
                        width=int(h.spot_integration_width_knm), stat_groups=sg)
            save(f"spot_{method.replace('-', '')}_{fb}", meta, rec)


def _spot_record_slim(h, rec):
    """Spot fixtures keep P-sized per-iteration arrays only at the spots (+ a strided amp_ff sample of the end state)."""
    ky, kx = h.spot_knm_rounded[1], h.spot_knm_rounded[0]
    rec["spot_knm"] = np.array(h.spot_knm)
    rec["spot_knm_rounded"] = np.array(h.spot_knm_rounded)
    rec["spot_amp"] = np.array(h.spot_amp)
    rec["final_ampff_sub"] = rec["final_ampff"][::4, ::4].copy()
    rec["final_ampff_spots"] = rec["final_ampff"][ky, kx]
    rec["final_weights_spots"] = rec["final_weights"][ky, kx]
    rec["final_weights_sum"] = np.array(float(np.sum(rec["final_weights"].astype(float))))
    del rec["final_ampff"], rec["final_weights"]
    for k in list(rec.keys()):
        if k.startswith("phaseff_"):
            del rec[k]
        elif k.startswith(("weights_", "ampff_")):
            rec[k + "_spots"] = rec[k][ky, kx]
            del rec[k]
    return rec


def gen_kimeff_cases(alg):
    """
    WGS-Kim fixed by efficiency (_hologram.py:1560-1569): the phase is frozen at the first iteration whose recorded
    efficiency -- of the LAST statistics group -- exceeds fix_phase_efficiency.  Two cases: a dense 64^2 Hologram with
    the iteration rule out of reach (fix_phase_iteration = 100, threshold crossed at iteration 4), and the 256^2
    SpotHologram with the spot group (threshold 0.575 between its iterations 3 and 4; the pixel group never exceeds
    0.13) and the iteration rule left at its default.  Only ONE group is requested: with two, "last" is the iteration
    order of a Python set of strings (_stats.py:162-171), which changes with PYTHONHASHSEED from process to process.
    """
    seed = 150
    shape = slm = (64, 64)
    h = alg.Hologram(synth.random_target(seed, shape), phase=synth.seed_phase(seed, slm), slm_shape=slm)
    kw = dict(fix_phase_efficiency=0.95, fix_phase_iteration=100)
    rec = record_run(h, "WGS-Kim", 9, save_iters=(0, 1, 2, 3, 4, 5, 6, 8), stat_groups=["computational"], **kw)
    meta = dict(kind="kimeff_hologram", method="WGS-Kim", seed=seed, shape=shape, slm_shape=slm, dtype="float32",
                maxiter=9, kwargs=kw, stat_groups=["computational"])
    save("kimeff_hologram", meta, rec)

    seed = 401
    shape, slm = (256, 256), (72, 120)
    h = alg.SpotHologram.make_rectangular_array(shape, array_shape=(8, 8), array_pitch=(16, 16), basis="knm",
                                                slm_shape=slm, phase=synth.seed_phase(seed, slm))
    kw = dict(fix_phase_efficiency=0.575)
    sg = ["computational_spot"]
    rec = record_run(h, "WGS-Kim", 9, save_iters=(0, 1, 2, 3, 4, 5, 6, 8), feedback="computational_spot",
                     stat_groups=sg, **kw)
    rec = _spot_record_slim(h, rec)
    meta = dict(kind="kimeff_spot", method="WGS-Kim", feedback="computational_spot", seed=seed, shape=shape,
                slm_shape=slm, dtype="float32", maxiter=9, kwargs=kw, array_shape=(8, 8), array_pitch=(16, 16),
                width=int(h.spot_integration_width_knm), stat_groups=sg)
    save("kimeff_spot", meta, rec)


SPOT_NULL_VECTORS = ((40.0, 100.4, 200.0, 128.0), (60.0, 180.6, 90.0, 128.0))
# two more null points whose disks cross the array edge: window_slice clips the bounding box first and centres the
# disk on the CLIPPED box (toolbox/__init__.py:503-528), so these pin that behaviour
SPOT_NULL_VECTORS_EDGE = ((40.0, 100.4, 200.0, 128.0, 1.0, 254.0), (60.0, 180.6, 90.0, 128.0, 128.0, 254.0))


def spot_null_region(shape):
    """The blanket null region of the spot_null fixtures: a 24-pixel frame plus one off-centre block."""
    m = np.zeros(shape, dtype=bool)
    m[:24, :] = m[-24:, :] = m[:, :24] = m[:, -24:] = True
    m[150:170, 40:90] = True
    return m


def gen_spot_null_cases(alg):
    """
    SpotHologram with null points / a null region (_spots.py:1300-1373, 1514-1538): the background of the target is
    NaN (amplitude freedom), the null region and a disk around every null point AND every spot are zero.  Spot
    feedback then meets the MRAF branch of the farfield routines.  Variants: an explicit null_radius with a region;
    the automatic radius (smallest distance / 4) with null_region_radius_frac and a noise attenuation.
    """
    shape, slm = (256, 256), (72, 120)
    seed = 410
    for tag, ctor, fb, method, kw in (
        ("radius", dict(null_radius=3, null_region=spot_null_region(shape)), "computational_spot", "WGS-Leonardo", {}),
        ("auto", dict(null_region_radius_frac=0.8), "computational", "WGS-Leonardo", {"mraf_factor": 0.5}),
        ("kim", dict(null_radius=2.5), "computational_spot", "WGS-Kim", {"fix_phase_iteration": 3}),
    ):
        nulls = np.array(SPOT_NULL_VECTORS_EDGE if tag == "kim" else SPOT_NULL_VECTORS)
        h = alg.SpotHologram.make_rectangular_array(
            shape, array_shape=(6, 6), array_pitch=(20, 20), basis="knm", slm_shape=slm,
            phase=synth.seed_phase(seed, slm), null_vectors=nulls, **ctor)
        sg = ["computational", "computational_spot"]
        target0 = np.array(h.target)
        rec = record_run(h, method, 6, save_iters=(0, 1, 2, 3, 4, 5), feedback=fb, stat_groups=sg, **kw)
        rec = _spot_record_slim(h, rec)
        rec["target"] = target0
        rec["null_radius_knm"] = np.array(h.null_radius_knm)
        rec["null_vectors"] = nulls
        meta = dict(kind="spot_null", variant=tag, method=method, feedback=fb, seed=seed, shape=shape, slm_shape=slm,
                    dtype="float32", maxiter=6, kwargs=kw, array_shape=(6, 6), array_pitch=(20, 20),
                    width=int(h.spot_integration_width_knm), stat_groups=sg,
                    ctor={k: (v if not isinstance(v, np.ndarray) else "spot_null_region") for k, v in ctor.items()})
        save(f"spotnull_{tag}", meta, rec)


def gen_multiplane_cases(alg):
    """MultiplaneHologram (_multiplane.py): three children of different pad shapes on one 48x80 SLM."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_cases as gc
    for method, maxiter, kw in (("GS", 4, {}), ("WGS-Leonardo", 5, {}), ("WGS-Kim", 6, {"fix_phase_iteration": 3})):
        children = gc.multiplane_children(alg.Hologram, alg.SpotHologram)
        mp = alg.MultiplaneHologram(children, weights=list(gc.MULTIPLANE_WEIGHTS))
        rec = {}

        def snap(h):
            rec[f"phase_{h.iter}"] = np.array(h.phase, copy=True)
            return False

        mp.optimize(method, maxiter=maxiter, verbose=False, callback=snap, stat_groups=["computational"], **kw)
        rec["final_phase"] = np.array(mp.phase, copy=True)
        rec["weights"] = np.array(mp.weights, copy=True)
        for i, h in enumerate(children):
            rec[f"child{i}_final_ampff"] = np.array(h.amp_ff, copy=True)
            rec[f"child{i}_final_weights"] = np.array(h.weights, copy=True)
            rec[f"child{i}_fixed_history"] = np.array([bool(x) for x in h.stats["flags"]["fixed_phase"]])
            rec[f"child{i}_iter"] = np.array(h.iter)
            for n, lst in h.stats["stats"]["computational"].items():
                rec[f"child{i}_stats_{n}"] = np.array(lst, dtype=float)
        rec["spot_knm_rounded"] = np.array(children[1].spot_knm_rounded)
        meta = dict(kind="multiplane", method=method, maxiter=maxiter, kwargs=kw, dtype="float32",
                    slm_shape=gc.MULTIPLANE_SLM, weights=list(gc.MULTIPLANE_WEIGHTS))
        save(f"multiplane_{method.replace('-', '')}", meta, rec)


def gen_fourier_cases(alg):
    """Callers in FourierSLM: ij-basis SpotHologram and the set-up of fourier_grid_project."""
    fs = make_fourier_slm()
    ij = np.array([[100., 150, 128, 171.5], [90, 128, 170, 66.25]])
    phase0 = synth.seed_phase(800, (48, 64))
    h = alg.SpotHologram((128, 128), ij, basis="ij", cameraslm=fs, phase=phase0.copy())
    rec = dict(spot_ij=ij, spot_knm=np.array(h.spot_knm), spot_kxy=np.array(h.spot_kxy),
               spot_knm_rounded=np.array(h.spot_knm_rounded), width=np.array(h.spot_integration_width_knm),
               amp=np.array(h.amp))
    h.optimize("WGS-Kim", maxiter=6, verbose=False, feedback="computational_spot", fix_phase_iteration=3,
               stat_groups=["computational_spot"])
    rec["final_phase"] = np.array(h.phase)
    ky, kx = h.spot_knm_rounded[1], h.spot_knm_rounded[0]
    rec["final_ampff_spots"] = np.array(h.amp_ff[ky, kx])
    rec["final_weights_spots"] = np.array(h.weights[ky, kx])
    for n, lst in h.stats["stats"]["computational_spot"].items():
        rec[f"stats_{n}"] = np.array(lst, dtype=float)
    g = fs.fourier_grid_project(array_shape=(4, 3), array_pitch=(3, 4), array_center=(2, -1), maxiter=2, verbose=False)
    rec["grid_width"] = np.array([g.spot_integration_width_knm, g.spot_integration_width_ij])
    rec["width_ij"] = np.array(h.spot_integration_width_ij)
    rec["psf_kxy"] = np.array(fs.slm.get_spot_radius_kxy())
    rec["grid_shape"] = np.array(g.shape)
    rec["grid_spot_knm"] = np.array(g.spot_knm)
    rec["grid_spot_ij"] = np.array(g.spot_ij)
    rec["grid_target_nonzero"] = np.array(np.nonzero(g.target))
    rec["grid_target_values"] = np.array(g.target[np.nonzero(g.target)])
    save("fourier_callers", dict(kind="fourier", slm_shape=(48, 64), seed=800, M=[[6000., 0], [0, 6000.]],
                                 b=[128., 128.]), rec)


def gen_helper_cases(alg, toolbox, analysis):
    """F7: unpad/pad index tuples, get_padded_shape table, take windows, integration width."""
    out = {}
    pads = [((64, 64), (64, 64)), ((128, 128), (48, 80)), ((4096, 4096), (1152, 1920)),
            ((65, 64), (30, 31)), ((9, 12), (4, 7)), ((256, 256), (72, 120)), ((8192, 8192), (1152, 1920))]
    out["unpad_in"] = np.array([[a[0], a[1], b[0], b[1]] for a, b in pads])
    out["unpad_out"] = np.array([toolbox.unpad(a, b) for a, b in pads])
    ps = [((1152, 1920), 1, True), ((1152, 1920), 2, True), ((1152, 1920), 3, True),
          ((720, 1280), 1, False), ((720, 1280), 2, False), ((512, 512), 1, True), ((600, 800), 0, False)]
    out["padshape_in"] = np.array([[s[0], s[1], o, int(q)] for s, o, q in ps])
    out["padshape_out"] = np.array([alg.Hologram.get_padded_shape(s, o, q) for s, o, q in ps])
    img = synth.random_target(7, (40, 50))
    vec = np.array([[5.2, 20.7, 44.0, 10.5], [6.9, 30.1, 3.0, 35.5]])
    out["take_img"] = img
    out["take_vec"] = vec
    for w in (1, 3, 5):
        out[f"take_w{w}"] = analysis.take(img, vec, w, centered=True, integrate=True)
    save("helpers", dict(kind="helpers"), out)


def gen_cfg1(alg):
    """F6/cfg1: Hologram 512^2 random amplitude, GS, 20 iterations, f32."""
    shape = (512, 512)
    target = synth.random_target(1, shape)
    phase0 = synth.seed_phase(1, shape)
    h = alg.Hologram(target.copy(), phase=phase0.copy(), slm_shape=shape)
    h.optimize("GS", maxiter=20, verbose=False, stat_groups=["computational"])
    out = dict(phase_sub=np.array(h.phase[::4, ::4]), ampff_sub=np.array(h.amp_ff[::4, ::4]),
               ampff_norm=np.array(float(np.sqrt(np.sum(np.square(h.amp_ff.astype(float)))))),
               efficiency=np.array(h.stats["stats"]["computational"]["efficiency"]),
               std_err=np.array(h.stats["stats"]["computational"]["std_err"]))
    save("cfg1_summary", dict(kind="cfg1", seed=1, shape=shape, maxiter=20, method="GS", sub=4), out)


def gen_cfg2(alg):
    """F6/cfg2: SpotHologram 32x32 pitch 64 on 4096^2, S=1152x1920, WGS-Leonardo 50 it."""
    import time
    shape, slm = (4096, 4096), (1152, 1920)
    phase0 = synth.seed_phase(2, slm)
    h = alg.SpotHologram.make_rectangular_array(
        shape, array_shape=(32, 32), array_pitch=(64, 64), basis="knm", slm_shape=slm, phase=phase0.copy())
    t0 = time.time()
    h.optimize("WGS-Leonardo", maxiter=50, verbose=False, stat_groups=[])
    dt = time.time() - t0
    ky, kx = h.spot_knm_rounded[1], h.spot_knm_rounded[0]
    out = dict(spot_ampff=np.array(h.amp_ff[ky, kx]), spot_weights=np.array(h.weights[ky, kx]),
               ampff_sub=np.array(h.amp_ff[::16, ::16]), phase_sub=np.array(h.phase[::6, ::6]),
               ampff_norm=np.array(float(np.sqrt(np.sum(np.square(h.amp_ff.astype(float)))))),
               spot_knm_rounded=np.array(h.spot_knm_rounded), wall_s=np.array(dt))
    save("cfg2_summary", dict(kind="cfg2", seed=2, shape=shape, slm_shape=slm, maxiter=50,
                              method="WGS-Leonardo", sub_ampff=16, sub_phase=6, wall_s=dt), out)


def gen_misc_cases(alg):
    """
    Off-loop rows of SURVEY 8(a)-18 recorded from the reference: the quadratic initial phase (_hologram.py:480-527,
    581-601) and get_farfield at other shapes, another depth and through an affine resample (:853-931).
    """
    slm = (48, 80)
    yy, xx = np.mgrid[0:128, 0:128]
    target = np.exp(-(((xx - 75.5) / 14.0) ** 2 + ((yy - 52.0) / 9.0) ** 2)).astype(np.float32)
    target[100:110, 20:40] = np.nan                       # a noise region: the moments use nansum
    amp = synth.gaussian_amp(slm, frac=0.4)
    h = alg.Hologram(target.copy(), amp=amp.copy(), phase=synth.seed_phase(31, slm), slm_shape=slm)
    out = dict(target=target, amp=amp, q1=np.array(h._get_quadratic_initial_phase(1)),
               q17=np.array(h._get_quadratic_initial_phase(1.7)))
    c, sd = h._get_target_moments_knm_norm()
    out.update(center_knm_norm=np.array(c), std_knm_norm=np.array(sd))
    h.reset_phase(quadratic_phase=True, random_phase=0)
    out["phase_quadratic"] = np.array(h.phase)
    save("quadratic_phase", dict(kind="quadratic_phase", slm_shape=slm, shape=(128, 128)), out)

    # get_farfield: shape variants, a depth kernel, an affine resample (scipy order-3 spline, in place on the farfield)
    h = alg.Hologram((128, 128), amp=amp.copy(), phase=synth.seed_phase(32, slm), slm_shape=slm)
    kern = (0.4 * synth.seed_phase(33, slm)).astype(np.float32)
    aff = dict(M=np.array([[1.02, 0.01], [-0.02, 0.97]]), b=np.array([3.5, -2.25]))
    out = dict(amp=amp, kern=kern, M=aff["M"], b=aff["b"],
               ff_default=np.array(h.get_farfield()),
               ff_64x256_kern=np.array(h.get_farfield((64, 256), propagation_kernel=kern)),
               ff_256_affine=np.array(h.get_farfield((256, 256), propagation_kernel=0, affine=aff)),
               ff_128_kern_affine=np.array(h.get_farfield((128, 128), propagation_kernel=kern, affine=aff)))
    save("get_farfield", dict(kind="get_farfield", slm_shape=slm, shape=(128, 128), seed=32), out)


def gen_cfg2kim(alg):
    """cfg 2 geometry with WGS-Kim (phase fixed at iteration 10), 30 it, seed 9 -> cfg2kim_summary.npz."""
    from slmsuite.holography import analysis
    shape, slm = (4096, 4096), (1152, 1920)
    h = alg.SpotHologram.make_rectangular_array(shape, array_shape=(32, 32), array_pitch=(64, 64), basis="knm",
                                                slm_shape=slm, phase=synth.seed_phase(9, slm).copy())
    h.optimize("WGS-Kim", maxiter=30, verbose=False, stat_groups=[])
    ky, kx = h.spot_knm_rounded[1], h.spot_knm_rounded[0]
    fb = np.sqrt(analysis.take(np.square(h.amp_ff), h.spot_knm, h.spot_integration_width_knm, centered=True, integrate=True))
    st = alg.Hologram._calculate_stats(fb, h.spot_amp, xp=np, efficiency_compensation=False,
                                       total=np.sum(np.square(h.amp_ff)))
    out = dict(spot_ampff=np.array(h.amp_ff[ky, kx]), spot_weights=np.array(h.weights[ky, kx]),
               ampff_sub=np.array(h.amp_ff[::16, ::16]), phase_sub=np.array(h.phase[::6, ::6]),
               fixed_history=np.array([bool(x) for x in h.stats["flags"]["fixed_phase"]]),
               uniformity=np.array(st["uniformity"]), efficiency=np.array(st["efficiency"]))
    save("cfg2kim_summary", dict(kind="cfg2kim", seed=9, shape=shape, slm_shape=slm, maxiter=30, method="WGS-Kim",
                                 sub_ampff=16, sub_phase=6), out)


CFG2_STEP_ITERS = (10, 30, 49)


def gen_cfg2_steps(alg):
    """
    Teacher-forcing material for the headline run at FULL size (cfg 2, seed 2, WGS-Leonardo x 50): the complete state
    before bodies 10, 30 and 49 -- the phase (8.8 MB each; the weights are zero off the 1,024 spots, so the spot values
    are the whole array; WGS-Leonardo keeps no phase_ff) -- and what ONE reference body makes of it: the spot
    amplitudes of the forward transform, the updated weights at the spots, the next phase (every third pixel per axis).
    A single body is determined by its input to a few 1e-7, unlike the 50-body trajectory -> cfg2_steps.npz.
    """
    shape, slm = (4096, 4096), (1152, 1920)
    h = alg.SpotHologram.make_rectangular_array(shape, array_shape=(32, 32), array_pitch=(64, 64), basis="knm",
                                                slm_shape=slm, phase=synth.seed_phase(2, slm).copy())
    ky, kx = h.spot_knm_rounded[1], h.spot_knm_rounded[0]
    out = {}

    def snap(hh):       # fires after the forward transform of body k: phase_k, weights_k, amp_ff = |FFT(phase_k)|
        k = hh.iter
        if k in CFG2_STEP_ITERS:
            out[f"phase_{k}"] = np.array(hh.phase, copy=True)
            out[f"weights_{k}_spots"] = np.array(hh.weights[ky, kx])
            out[f"ampff_{k}_spots"] = np.array(hh.amp_ff[ky, kx])
            out[f"ampff_{k}_sub"] = np.array(hh.amp_ff[::16, ::16])
            assert np.count_nonzero(hh.weights) == 1024
        if k - 1 in CFG2_STEP_ITERS:
            out[f"next_phase_{k - 1}_sub"] = np.array(hh.phase[::3, ::3])
            out[f"next_weights_{k - 1}_spots"] = np.array(hh.weights[ky, kx])
        return False

    h.optimize("WGS-Leonardo", maxiter=50, verbose=False, stat_groups=[], callback=snap)
    out["next_phase_49_sub"] = np.array(h.phase[::3, ::3])          # body 49 is the last: its output is the end state
    out["next_weights_49_spots"] = np.array(h.weights[ky, kx])
    out["spot_knm_rounded"] = np.array(h.spot_knm_rounded)
    save("cfg2_steps", dict(kind="cfg2_steps", seed=2, shape=shape, slm_shape=slm, maxiter=50, method="WGS-Leonardo",
                            iters=list(CFG2_STEP_ITERS), sub_phase=3, sub_ampff=16), out)


CFG2_SEEDS = (2, 10, 11, 12, 13, 14, 15, 16)
CFG2_CURVE_ITERS = (5, 10, 20, 30, 40)


def _cfg2_seed_run(job):
    """One reference run of cfg 2 (WGS-Leonardo x 50) from seed phase `seed`, optionally perturbed by ~1 ulp."""
    seed, perturbed = job
    alg, _, _ = import_reference()
    shape, slm = (4096, 4096), (1152, 1920)
    p0 = synth.seed_phase(seed, slm)
    if perturbed:
        # the same perturbation tests/test_conditioning.py applies: relative 1e-7 * N(0,1), i.e. <= ~2 ulp
        rng = np.random.default_rng(1000 + seed)
        p0 = (p0.astype(np.float64) * (1 + 1e-7 * rng.standard_normal(p0.shape))).astype(np.float32)
    h = alg.SpotHologram.make_rectangular_array(shape, array_shape=(32, 32), array_pitch=(64, 64), basis="knm",
                                                slm_shape=slm, phase=p0.copy())
    ky, kx = h.spot_knm_rounded[1], h.spot_knm_rounded[0]
    curve = {}

    def snap(hh):       # fires after the forward transform of body k: amp_ff = |FFT(phase_k)|
        if hh.iter in CFG2_CURVE_ITERS:
            curve[hh.iter] = np.array(hh.amp_ff[ky, kx])
        return False

    h.optimize("WGS-Leonardo", maxiter=50, verbose=False, stat_groups=[], callback=snap)
    return seed, perturbed, np.array(h.amp_ff[ky, kx]), np.array(h.weights[ky, kx]), \
        np.stack([curve[k] for k in CFG2_CURVE_ITERS])


CFG2_SEEDS_B = (17, 18, 19, 20, 21, 22, 23, 24)     # round 5: eight more, for the statistics of engine / ideal-fp32


def gen_cfg2_seeds(alg, seeds=None, name="cfg2_seeds"):
    """
    cfg 2 from eight seed phases, each also from the seed perturbed by about one fp32 ulp: the spot amplitudes the
    reference ends on, and how far its OWN result moves under a 1-ulp change of the input (the floor under which no
    fp32 implementation that is not bit-identical to NumPy can be expected to land).  ~2.5 min per run, 4 at a time.
    """
    import multiprocessing as mp
    CFG2_SEEDS = tuple(seeds) if seeds is not None else globals()["CFG2_SEEDS"]
    jobs = [(s, p) for s in CFG2_SEEDS for p in (False, True)]
    with mp.get_context("spawn").Pool(4) as pool:
        res = pool.map(_cfg2_seed_run, jobs)
    n = len(CFG2_SEEDS)
    amp = np.zeros((n, 1024), np.float32); amp_p = np.zeros_like(amp)
    w = np.zeros_like(amp); w_p = np.zeros_like(amp)
    cur = np.zeros((n, len(CFG2_CURVE_ITERS), 1024), np.float32); cur_p = np.zeros_like(cur)
    for seed, pert, a, ww, cv in res:
        i = CFG2_SEEDS.index(seed)
        (amp_p if pert else amp)[i] = a
        (w_p if pert else w)[i] = ww
        (cur_p if pert else cur)[i] = cv
    save(name, dict(kind="cfg2_seeds", seeds=list(CFG2_SEEDS), shape=(4096, 4096), slm_shape=(1152, 1920),
                            maxiter=50, method="WGS-Leonardo", curve_iters=list(CFG2_CURVE_ITERS),
                            perturbation="phase * (1 + 1e-7 * N(0,1)), default_rng(1000 + seed), rounded to fp32"),
         dict(spot_ampff=amp, spot_ampff_perturbed=amp_p, spot_weights=w, spot_weights_perturbed=w_p,
              curve_ampff=cur, curve_ampff_perturbed=cur_p))


def camera_image(seed, shape=(256, 256)):
    """A camera-basis target: three Gaussian blobs and a bar, zero elsewhere (float32, deterministic)."""
    yy, xx = np.mgrid[:shape[0], :shape[1]].astype(np.float64)
    u = synth.uniform01(seed, (3, 3), 1)
    img = np.zeros(shape)
    for k in range(3):
        cx, cy, s = 60 + 130 * u[k, 0], 60 + 130 * u[k, 1], 6 + 6 * u[k, 2]
        img += (0.5 + k / 3) * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
    img[120:136, 40:200] += 0.8
    img[img < 1e-3] = 0
    return img.astype(np.float32)


def gen_feedback_ij_cases(alg):
    """Camera-basis targets (FeedbackHologram target_ij / update_target / ijcam_to_knmslm, _feedback.py:75-330), the ij
    null region of a SpotHologram (_spots.py:1352-1357) and the depth row of the Fourier calibration (cameraslms.py:1221-1354)."""
    fs = make_fourier_slm()
    shape = (128, 128)
    img = camera_image(811)
    phase0 = synth.seed_phase(810, (48, 64))
    h = alg.FeedbackHologram(shape, target_ij=img.copy(), cameraslm=fs, phase=phase0.copy())
    rec = dict(img=img, target_ctor=np.array(h.target), cam_points=np.array(h._cam_points),
               knm_cubic=np.array(h.ijcam_to_knmslm(img.copy())),
               knm_blur=np.array(h.ijcam_to_knmslm(img.copy(), blur_ij=2)),
               knm_nearest=np.array(h.ijcam_to_knmslm(img.copy(), order=0)))
    h.optimize("WGS-Leonardo", maxiter=4, verbose=False)
    rec["phase_plain"] = np.array(h.phase)
    # a null region: NaN (noise region) stays where the camera cannot see AND the radius mask is off
    h2 = alg.FeedbackHologram(shape, target_ij=img.copy(), cameraslm=fs, phase=phase0.copy(), null_region_radius_frac=0.6)
    rec["target_frac"] = np.array(h2.target)
    region = np.zeros(shape, dtype=bool)
    region[:, :20] = True
    h2.update_target(img[::-1].copy(), null_region=region, null_region_radius_frac=0.8, reset_weights=True)
    rec["target_update"] = np.array(h2.target)
    rec["weights_update"] = np.array(h2.weights)
    h2.optimize("WGS-Leonardo", maxiter=4, verbose=False, mraf_factor=0.5)
    rec["phase_mraf"] = np.array(h2.phase)
    # SpotHologram in the ij basis with a camera-shaped null region
    ij = np.array([[100., 150, 128, 171.5], [90, 128, 170, 66.25]])
    null_ij = np.array([[110., 140.], [120., 100.]])
    cam_region = np.zeros((256, 256), dtype=np.float32)
    cam_region[40:220, 30:230] = 1
    # (the reference transforms the region in place into an array of ITS shape, so this path needs shape == cam.shape)
    s = alg.SpotHologram((256, 256), ij, basis="ij", cameraslm=fs, phase=phase0.copy(), null_vectors=null_ij, null_radius=9.0,
                         null_region=cam_region.copy())
    rec.update(spot_ij=ij, null_ij=null_ij, cam_region=cam_region, spot_target=np.array(s.target),
               spot_null_knm=np.array(s.null_knm), spot_null_radius=np.array(s.null_radius_knm),
               spot_null_region=np.array(s.null_region_knm))
    # depth rows
    k3 = np.array([[0.001, -0.004, 0.0], [0.002, 0.003, -0.001], [1e-6, -2e-6, 5e-7]])
    i3 = np.array([[100., 150, 128], [90, 128, 170], [3.0, -8.0, 0.5]])
    rec.update(kxy3=k3, ij_of_kxy3=np.array(fs.kxyslm_to_ijcam(k3)), ij3=i3, kxy_of_ij3=np.array(fs.ijcam_to_kxyslm(i3)),
               f_eff=np.array([fs.get_effective_focal_length("ij"), np.mean(fs.get_effective_focal_length("norm"))]))
    # CompressedSpotHologram specified on the camera, with depth (3-vectors through the calibration)
    c = alg.CompressedSpotHologram(i3, basis="ij", cameraslm=fs)
    rec.update(comp_zernike=np.array(c.spot_zernike), comp_kxy=np.array(c.spot_kxy), comp_ij=np.array(c.spot_ij),
               comp_width_ij=np.array(c.spot_integration_width_ij), comp_basis=np.array(c.zernike_basis))
    save("feedback_ij", dict(kind="feedback_ij", shape=list(shape), slm_shape=(48, 64), seed=810, M=[[6000., 0], [0, 6000.]],
                             b=[128., 128.]), rec)


def make_fourier_slm(res_wh=(64, 48)):
    """SimulatedSLM + SimulatedCamera + analytic Fourier calibration (SURVEY 8c recipe)."""
    from slmsuite.hardware.slms.simulated import SimulatedSLM
    from slmsuite.hardware.cameras.simulated import SimulatedCamera
    from slmsuite.hardware.cameraslms import FourierSLM
    slm = SimulatedSLM(res_wh, pitch_um=(8, 8), wav_um=0.78)
    cam = SimulatedCamera(slm, resolution=(256, 256), pitch_um=(4, 4))
    fs = FourierSLM(cam, slm)
    fs.fourier_calibrate_analytic(np.array([[6000., 0], [0, 6000.]]), np.array([128., 128.]))
    return fs


def gen_compressed_cases(alg):
    """F5: CompressedSpotHologram on a 48x64 SLM: N=50 (2-D), N=50 (3-D), N=300 (> N_BATCH_MAX), custom basis."""
    from slmsuite.holography.toolbox import phase as tphase
    fs = make_fourier_slm()
    slm = fs.slm
    scale = slm.get_source_zernike_scaling()
    xg, yg = slm.grid[0] * scale, slm.grid[1] * scale
    out = {"zernike_coeff_json": np.array(json.dumps(
        {str(j): {f"{k[0]},{k[1]}": int(v) for k, v in tphase._zernike_coefficients(j).items()} for j in range(45)}))}
    save("compressed_helpers", dict(kind="compressed_helpers", scale=float(scale), pitch=[float(p) for p in slm.pitch],
                                    shape=[int(x) for x in slm.shape]), out)
    cases = [("2d50", 2, 50, "kxy", None, "WGS-Kim", dict(fix_phase_iteration=4)),
             ("3d50", 3, 50, "kxy", None, "WGS-Leonardo", {}),
             ("2d300", 2, 300, "kxy", None, "WGS-Kim", dict(fix_phase_iteration=4)),
             ("zern5", 5, 24, [2, 1, 4, 3, 5], None, "WGS-Nogrette", {}),
             ("mraf2d", 2, 40, "kxy", "mraf", "WGS-Leonardo", dict(mraf_factor=0.5)),
             # ANSI tilt x, tilt y and the vortex pseudo-index -1 (phase.py:1783-1790): fractional charges, the
             # negative ones of which the NumPy path ignores
             ("vortex", 3, 30, [2, 1, -1], None, "WGS-Leonardo", {})]
    for tag, D, N, basis, special, method, kw in cases:
        seed = 500 + N + D
        v = (synth.uniform01(seed, (D, N), 7) * 2 - 1)
        if basis == "kxy":
            v[:2] *= 0.012
            if D == 3:
                v[2] *= 2e-6
        else:
            v *= np.array([12, 12, 2, 1.5, 1.5])[:D, None]
        spot_amp = None
        if special == "mraf":
            spot_amp = np.full(N, 1.0)
            spot_amp[::5] = np.nan
            spot_amp[3::7] = 0
        phase0 = synth.seed_phase(seed, tuple(int(x) for x in slm.shape))
        h = alg.CompressedSpotHologram(v.copy(), basis=basis, spot_amp=None if spot_amp is None else spot_amp.copy(),
                                       cameraslm=fs, phase=phase0.copy())
        # CompressedSpotHologram.__init__ ends with self.reset() (_spots.py:495), which re-randomises
        # the phase it was given; put the seed phase back so the run is reproducible.
        h.reset_phase(phase0.copy())
        rec = {}

        def snap(hh):
            k = hh.iter
            rec[f"ff_{k}"] = np.array(hh.farfield, copy=True)
            rec[f"weights_{k}"] = np.array(hh.weights, copy=True)
            if k in (1, 2, 5, 6):
                rec[f"phase_{k}"] = np.array(hh.phase, copy=True)
            rec[f"fixed_{k}"] = np.array(bool(hh.flags.get("fixed_phase", False)))
            return False

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            h.optimize(method, maxiter=8, verbose=False, callback=snap, **kw)
        rec.update(final_phase=np.array(h.phase), final_weights=np.array(h.weights), final_ff=np.array(h.farfield),
                   final_ampff=np.array(h.amp_ff), spot_zernike=np.array(h.spot_zernike),
                   zernike_basis=np.array(h.zernike_basis), spot_vectors=v, target=np.array(h.target),
                   xg=np.array(xg), yg=np.array(yg), spot_kxy=np.array(h.spot_kxy),
                   fixed_history=np.array([bool(x) for x in h.stats["flags"]["fixed_phase"]]))
        if spot_amp is not None:
            rec["spot_amp_in"] = spot_amp
        meta = dict(kind="compressed", tag=tag, D=D, N=N, basis=basis if isinstance(basis, str) else list(basis),
                    method=method, kwargs=kw, seed=seed, maxiter=8, slm_shape=[int(x) for x in slm.shape],
                    scale=float(scale), feedback=h.flags["feedback"])
        save(f"compressed_{tag}", meta, rec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg2", action="store_true")
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    alg, toolbox, analysis = import_reference()
    steps = {
        "hologram": lambda: gen_hologram_cases(alg),
        "mraf": lambda: gen_mraf_cases(alg),
        "spot": lambda: gen_spot_cases(alg),
        "helpers": lambda: gen_helper_cases(alg, toolbox, analysis),
        "cfg1": lambda: gen_cfg1(alg),
        "compressed": lambda: gen_compressed_cases(alg),
        "multiplane": lambda: gen_multiplane_cases(alg),
        "fourier": lambda: gen_fourier_cases(alg),
        "misc": lambda: gen_misc_cases(alg),
        "kimeff": lambda: gen_kimeff_cases(alg),
        "spotnull": lambda: gen_spot_null_cases(alg),
        "feedback_ij": lambda: gen_feedback_ij_cases(alg),
    }
    if args.cfg2:
        steps["cfg2"] = lambda: gen_cfg2(alg)
        steps["cfg2kim"] = lambda: gen_cfg2kim(alg)
        steps["cfg2seeds"] = lambda: gen_cfg2_seeds(alg)
        steps["cfg2seedsb"] = lambda: gen_cfg2_seeds(alg, CFG2_SEEDS_B, "cfg2_seeds_b")
        steps["cfg2steps"] = lambda: gen_cfg2_steps(alg)
    for name, fn in steps.items():
        if args.only and name != args.only:
            continue
        print(name)
        fn()


if __name__ == "__main__":
    main()
