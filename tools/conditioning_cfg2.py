#!/usr/bin/env python
"""
How close can ANY fp32 implementation come to the reference's cfg-2 end state?

The CPU oracle (bit-faithful to the reference on cfg 2) is re-run with single operators replaced by a MORE
accurate version of themselves -- computed in float64 and rounded once to float32 -- everything else, including
every storage type, unchanged:

    fft64    forward and inverse 2-D FFT in complex128, result rounded to complex64
             (a correctly rounded fp32-output FFT; pocketfft's own fp32 rounding differs from it by ~1e-7 rel.)
    atan64   arctan2 of nearfield / farfield in float64, rounded
    exp64    exp(i phase) of nearfield / farfield in complex128, rounded
    all64    the three together ("fp32 storage, exact arithmetic")

The distance of each variant's spot amplitudes from the unmodified oracle after 50 WGS-Leonardo bodies is what a
perfect implementation of that operator would score against the reference: the floor for implementations that
are not bit-identical to NumPy's kernels.  Runs on the CPU only (build container), ~2-3 min per run.

    python tools/conditioning_cfg2.py [--seeds 2 10 11 12] [--iters 50] [--procs 4] [--out profiles/r02/conditioning_cfg2.json]
    python tools/conditioning_cfg2.py --seeds 2 10 11 12 13 14 15 16 --variants all64 --curve \
        --out tests/golden/cfg2_ideal_fp32.json     # the yardstick of tests/test_full_configs.py (16 runs, ~15 min)

--curve also records the distance after 5, 10, 20, 30 and 40 bodies (the iterations cfg2_seeds.npz holds).
"""
import argparse
import json
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = ("plain", "fft64", "atan64", "exp64", "all64")
CURVE_ITERS = (5, 10, 20, 30, 40)


def run(job):
    seed, variant, iters, small, curve = job
    from oracle import hgs_oracle as orc
    from slmsuite_amd import synth

    class V(orc.OracleSpotHologram):
        def nearfield2farfield(self):
            if variant in ("fft64", "all64"):
                nf = self.build_nearfield()
                ff = np.fft.fftshift(np.fft.fft2(np.fft.fftshift(nf.astype(np.complex128)), norm="ortho"))
                self.farfield = ff.astype(self.ctype)
                self.amp_ff = np.abs(self.farfield, out=self.amp_ff)
            else:
                super().nearfield2farfield()

        def build_nearfield(self):
            if variant in ("exp64", "all64"):
                r0, r1, c0, c1 = orc.unpad_slices(self.shape, self.slm_shape)
                self.nearfield.fill(0)
                self.nearfield[r0:r1, c0:c1] = (self.amp * np.exp(1j * self.phase.astype(np.float64))).astype(self.ctype)
                return self.nearfield
            return super().build_nearfield()

        def farfield2nearfield(self):
            ff = self.farfield.astype(np.complex128) if variant in ("fft64", "all64") else self.farfield
            nf = np.fft.ifftshift(np.fft.ifft2(np.fft.ifftshift(ff), norm="ortho")).astype(self.ctype)
            self.nearfield = nf
            r0, r1, c0, c1 = orc.unpad_slices(self.shape, self.slm_shape)
            if variant in ("atan64", "all64"):
                self.phase = np.arctan2(nf.imag[r0:r1, c0:c1].astype(np.float64),
                                        nf.real[r0:r1, c0:c1].astype(np.float64)).astype(self.dtype)
            else:
                self.phase = np.arctan2(nf.imag[r0:r1, c0:c1], nf.real[r0:r1, c0:c1], out=self.phase)

        def gs_farfield_routines(self, masks):
            if variant in ("plain", "fft64"):
                return super().gs_farfield_routines(masks)
            fl = self.flags
            if "WGS" in fl["method"] and self.iter > 0:
                self.update_weights()
                fl["fixed_phase"] = False
            if variant in ("atan64", "all64"):
                self.phase_ff = np.arctan2(self.farfield.imag.astype(np.float64),
                                           self.farfield.real.astype(np.float64)).astype(self.dtype)
            else:
                self.phase_ff = np.arctan2(self.farfield.imag, self.farfield.real, out=self.phase_ff)
            if variant in ("exp64", "all64"):
                self.farfield = np.exp(1j * self.phase_ff.astype(np.float64)).astype(self.ctype)
            else:
                np.exp(1j * self.phase_ff, out=self.farfield)
            np.multiply(self.farfield, self.weights, out=self.farfield)

    if small:
        shape, slm, grid, pitch = (512, 512), (144, 240), (16, 16), (16, 16)
    else:
        shape, slm, grid, pitch = (4096, 4096), (1152, 1920), (32, 32), (64, 64)
    h = V(shape, orc.rectangular_array(shape, grid, pitch), slm_shape=slm, phase=synth.seed_phase(seed, slm))
    ky, kx = h.spot_knm_rounded[1], h.spot_knm_rounded[0]
    snaps = {}

    def cb(hh):         # after the forward transform of body k: amp_ff = |FFT(phase_k)|, as cfg2_seeds.npz records it
        if curve and hh.iter in CURVE_ITERS:
            snaps[hh.iter] = hh.amp_ff[ky, kx].astype(np.float64)
        return False

    h.optimize("WGS-Leonardo", maxiter=iters, callback=cb)
    snaps[iters] = h.amp_ff[ky, kx].astype(np.float64)
    return seed, variant, snaps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs="+", default=[2, 10, 11, 12])
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--procs", type=int, default=4)
    ap.add_argument("--small", action="store_true", help="512^2 pad (quick look)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--variants", nargs="+", default=[v for v in VARIANTS if v != "plain"])
    ap.add_argument("--curve", action="store_true", help="also after 5, 10, 20, 30, 40 bodies")
    a = ap.parse_args()
    variants = ["plain"] + [v for v in a.variants if v != "plain"]
    jobs = [(s, v, a.iters, a.small, a.curve) for s in a.seeds for v in variants]
    with mp.get_context("spawn").Pool(a.procs) as pool:
        res = pool.map(run, jobs)
    amps = {(s, v): x for s, v, x in res}
    dist = lambda x, base: float(np.linalg.norm(x - base) / np.linalg.norm(base))   # noqa: E731
    table, curves = {}, {}
    for s in a.seeds:
        base = amps[(s, "plain")]
        table[s] = {v: dist(amps[(s, v)][a.iters], base[a.iters]) for v in variants if v != "plain"}
        if a.curve:
            curves[s] = {v: {str(k): dist(amps[(s, v)][k], base[k]) for k in sorted(base)} for v in variants if v != "plain"}
    out = {"what": "rel. L2 distance of the spot amplitudes from the unmodified oracle (= the reference) after "
                   f"{a.iters} WGS-Leonardo bodies, cfg 2 geometry{' (512^2 stand-in)' if a.small else ''}",
           "variants": {"fft64": "FFT/IFFT in complex128, rounded to complex64", "atan64": "arctan2 in float64, rounded",
                        "exp64": "exp(i phase) in complex128, rounded", "all64": "all three"},
           "numpy": np.__version__, "per_seed": table}
    if a.curve:
        out["per_seed_curve"] = curves
    print(json.dumps(out, indent=1))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
