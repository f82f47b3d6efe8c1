#!/usr/bin/env python
"""
Benchmark of the hologram optimize() hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload cfg2|...] [--batch B]

``--gpus N`` with N > 1 needs N ranks, one per GPU.  Under a launcher (the driver's ``python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N``) the ranks come from RANK / LOCAL_RANK / WORLD_SIZE; called plainly,
``python bench.py --gpus N`` launches exactly that command itself and returns its exit code.  It never prints a line for
fewer GPUs than asked for: fewer visible devices, a WORLD_SIZE that differs from N or a process group of another size end
the run non-zero.  The default workload is then ``cfg3`` (BASELINE configs[2]: cfg 2 with 8 holograms per GPU, weak
scaling); ``process_group`` / ``rccl_ranks`` are read back from the group, ``gathered`` says what the final all-gather of
the phase masks moved and that every rank checked it, and ``one_hologram_per_gpu`` is the cfg 2 rate on every rank (the
``--gpus 1`` default workload) so that an efficiency on equal per-GPU work can be formed from the N = 1 line.

A *step* is one loop body of optimize_gs (nearfield -> farfield -> constraint / weight update ->
nearfield) of the named workload, state resident in HBM when the timed region starts.  The default
workload is BASELINE.json config 2 (the configuration the metric is quoted on): SpotHologram, 32 x 32
spots (pitch 64 px) on a 4096 x 4096 padded grid, SLM 1152 x 1920, WGS-Leonardo, fp32.  Each rank owns
``--batch`` independent holograms (weak scaling; ``--workload cfg3`` = cfg 2 with 8 per
GPU); the only collective is the final all-gather of the phase masks (reported as gather_ms, outside
the timed region, fed straight from device memory).  It runs after the line has been built from the timed
regions and under a watchdog (``--gather-timeout``): a communicator that cannot be created, or a collective
that never returns, costs the line its gather fields (``gathered.error`` says why), not the measurement.

Workloads (``--workload``):
  cfg2       headline (above)                 cfg3      the same with --batch 8
  cfg1       Hologram 512 x 512 on a 512 x 512 SLM, GS (BASELINE configs[0], the reference's CPU-runnable case)
  cfg2dense  cfg 2 geometry, dense random image target (nothing to skip)
  cfg4       CompressedSpotHologram, 1e4 spots, SLM 1152 x 1920, D = 2, WGS-Kim (cfg4d3: D = 3) -- MFMA bound
  cfg4zern   CompressedSpotHologram on a basis with a cross term (ANSI 2, 1, 4, 3, 5), 1e3 spots: not separable, the
             direct kernels (VALU / transcendental bound; roofline against the 157.3 TFLOP/s fp32 vector peak)
  cfg4grid   the DFT-grid companion of cfg 4: SpotHologram with 1e4 spots at distinct pixels of an 8192^2 pad, WGS-Kim
  cfg5mraf   Hologram with MRAF (NaN noise box 3072^2, image 2048^2) on an 8192^2 pad; --dtype f32|f64,
             --method GS|WGS-Leonardo
  cfg5pad / hd / small   spot arrays on 8192^2 / 2048^2 (1080 x 1920 SLM) / 1024^2 pads
  refbench   the reference's OWN speed benchmark (tests/holography/test_algorithms.py:121-145): Hologram on 1024 x 1024
             (S = P), 20 random unit pixels, whole ``optimize(method, maxiter=20, stat_groups=[])`` calls as a user makes
             them (host bookkeeping and the trailing transform included); --steps = maxiter; the four methods of the
             reference's parametrisation are all timed (``methods``), ``value`` is --method (default WGS-Leonardo)

Timing protocol: W warm-up steps, then ``--reps`` (default 40) repetitions of the timed region -- EXACTLY K
steps between barrier + synchronize on both sides, the slowest rank counting -- and ``value`` is K / the MEDIAN
repetition; the spread is reported (``ms_per_step_min`` / ``_max``; the first regions after a short warm-up run on clocks
that are still settling, which is why there are forty of them), with more than one rank also the rate of every
rank (``per_rank_its``) so that a straggler shows.
The region holds the K steps and nothing else (``hgs_iterate``); ``event_ms_per_step`` -- HIP events around the same K steps,
``hgs_iterate_timed`` -- comes from a few repetitions of its own behind the timed ones (an event pair with its wait and
read-out is 26 - 32 us per call: 2 % of a 20-step region, tools/region_probe.py).

For spot workloads (and the MRAF target, whose frame outside the noise box is empty) the headline is
timed with the dense kernels forced (every farfield column transformed); the engine's default for
such targets -- transform only the columns that hold a non-zero target, identical results -- is timed in an extra pass and reported as ``engine_default_path``.

Rank 0 prints one JSON line: metric / value (whole-job iterations/s) plus
  roofline      the dominant kernel against its bound.  HBM-bound kernels: ``achieved`` = the bytes this
                kernel moves across the fabric per launch (its own loads + stores, stated in
                ``bytes_model``) / its mean launch duration, measured here with HIP
                events on the engine stream over a second pass of the K steps; ``frac`` = achieved / 8 TB/s.
                ``canonical_equivalent`` restates the same duration against SURVEY 8(d)'s un-pruned byte
                count (a throughput figure: it exceeds the peak when pruning skips bytes).  ``traffic`` is
                measured in this run: bench.py re-runs a short pass of itself under
                ``rocprofv3 --pmc FETCH_SIZE`` and ``--pmc WRITE_SIZE`` (separate passes; FETCH_SIZE x 2 on
                gfx950 as MI355X_MICROARCH.md prescribes) and reports the per-launch mean of the same kernel;
                null (with ``traffic_note``) when rocprofv3 is unavailable or --pmc 0.
                cfg4: bound "mfma", flop = 6 N S ISSUED per GEMM launch (three real matrix products per complex one)
                against the 157.3 TFLOP/s fp32 matrix peak; the algorithmic 8 N S over the same time is reported as
                ``algorithmic_equivalent``.  cfg4zern: bound "valu" (direct kernels), SURVEY's 20 flop per evaluation.
  cpu_baseline  the CPU oracle (NumPy restatement of the reference path, kind "port") timed on this box's
                host cores on a bounded sample of the same workload (rank 0, N = 1).
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

sys.path.insert(0, os.path.join(ROOT, "tools"))
from benchlib.byte_models import grid_bytes_models  # noqa: E402
from benchlib.launcher import _gather_ints, apply_opts, self_launch as _self_launch  # noqa: E402
from benchlib.pmc import pmc_traffic as _pmc_traffic  # noqa: E402
from benchlib.workloads import (ALL_WORKLOADS, COMPRESSED_WORKLOADS, HBM_PEAK, IMAGE_WORKLOADS, MALL_BYTES, MFMA_F32_PEAK,  # noqa: E402
                                REFBENCH_METHODS, SPOT_WORKLOADS, VALU_F32_PEAK, VECTOR_WORKLOADS, ZERN_BASIS, grid_spots)


def pmc_traffic(args, kernel_substrings):
    return _pmc_traffic(args, kernel_substrings, __file__)


def self_launch(args):
    return _self_launch(args, __file__)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="K steps per timed repetition (default 200; refbench: 20)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--reps", type=int, default=40,
                    help="repetitions of the K-step timed region; the median is reported (min / max beside it).  40 because the "
                         "GPU's clocks take about 15 ms of work to settle after an idle spell: with ten 20-step regions (16 ms) the "
                         "median still sat on the ramp (12.6 k it/s against 13.1 k after 200 warm-up steps, same regions)")
    ap.add_argument("--batch", type=int, default=None, help="independent holograms per GPU (cfg3: 8)")
    ap.add_argument("--streams", type=int, default=None,
                    help="stream groups a rank's holograms are split over (one engine and HIP stream each; launches of "
                         "different groups overlap on the device).  Default: 2 from four holograms per GPU, else 1")
    ap.add_argument("--workload", default=None, choices=ALL_WORKLOADS,
                    help="default: cfg2 on one GPU, cfg3 (cfg 2 with 8 holograms per GPU, BASELINE configs[2]) on several")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for --gpus N > 1 (nccl = RCCL; gloo: self-test of the launcher on one device)")
    ap.add_argument("--gather-timeout", type=float, default=180.0,
                    help="seconds the final all-gather of the phase masks may take before the line is printed without it")
    ap.add_argument("--share-devices", action="store_true",
                    help="let ranks share GPUs (device = LOCAL_RANK %% device_count); needs --backend gloo -- RCCL wants one "
                         "device per rank.  Launcher self-test only: the line is marked, its value is not a scaling figure")
    ap.add_argument("--method", default=None)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--spots", type=int, default=None, help="cfg4 / cfg4grid: number of spots (default 10000; cfg4zern: 1000)")
    ap.add_argument("--cpu-iters", type=int, default=None, help="iterations of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-roofline-pass", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL even for one rank (self-test)")
    ap.add_argument("--sparse-columns", type=int, default=0,
                    help="1: time the engine's default (sparse-target aware) path as the headline; 0 (default): "
                         "force the dense kernels (every farfield column transformed) for the headline and "
                         "report the default path separately")
    ap.add_argument("--no-extra-pass", action="store_true")
    ap.add_argument("--pmc", type=int, default=None, help="1: measure roofline.traffic with rocprofv3 PMC child passes "
                                                          "(default: on for one rank, off otherwise)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    # test infrastructure (tests/test_host_logic.py): the rank protocol -- self-launch, barriers, the solo pass of rank 0, the
    # gather and the fields derived from them -- on a box without a GPU, against a stand-in that sleeps instead of
    # launching kernels.  The line says so in `metric` and `stub_engine`; nothing it reports is a measurement.
    ap.add_argument("--stub-engine", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--opt", action="append", default=[], help="engine option NAME=VALUE (hgs_set_option), e.g. "
                                                               "TILE_KERNEL=0")
    a = ap.parse_args()
    if a.gpus < 1:
        ap.error("--gpus must be >= 1")
    if a.share_devices and a.backend != "gloo":
        ap.error("--share-devices needs --backend gloo (RCCL refuses two ranks on one device)")
    if a.stub_engine and not (a.backend == "gloo" and a.share_devices and a.no_roofline_pass):
        ap.error("--stub-engine is the CPU self-test of the rank protocol: --backend gloo --share-devices --no-roofline-pass")
    if a.workload is None:
        a.workload = "cfg2" if a.gpus == 1 else "cfg3"
    if a.steps is None:
        a.steps = 20 if a.workload == "refbench" else 200
    a.reps = max(1, a.reps)
    if a.spots is None:
        a.spots = 1000 if a.workload == "cfg4zern" else 10000
    if a.batch is None:
        a.batch = 8 if a.workload == "cfg3" else 1
    if a.streams is None:
        a.streams = 2 if a.batch >= 4 else 1
    a.streams = max(1, min(a.streams, a.batch))
    if a.method is None:
        a.method = ("WGS-Kim" if (a.workload in COMPRESSED_WORKLOADS or a.workload in VECTOR_WORKLOADS) else
                    "GS" if a.workload == "cfg1" else "WGS-Leonardo")
    return a


# ---------------------------------------------------------------------------------------------------
# problems: each exposes warm(n), run(n) -> milliseconds (HIP events on the engine stream), engine
# ---------------------------------------------------------------------------------------------------
class StubProblem:
    """--stub-engine: the interface of the problems below with nothing behind it (a step is a sleep, the phase masks are a
    host tensor that names the rank).  For the CPU test of the rank protocol only."""
    STEP_S = 2e-4

    class _Engine:
        def sync(self): pass
        def version(self): return "stub (no device, no kernels)"
        def set_option(self, *a): pass

    def __init__(self, args, rank, local_rank):
        self.args, self.rank = args, rank
        self.engine = self._Engine()
        self.shape, self.slm = (64, 64), (16, 24)
        self.desc = "STUB ENGINE"
        self.mraf = False

    def warm(self, n):
        pass

    def run(self, n):
        time.sleep(self.STEP_S * n * self.args.batch)
        return self.STEP_S * n * self.args.batch * 1e3

    def phases_device(self, torch, device):
        return torch.full((self.args.batch,) + self.slm, float(self.rank), dtype=torch.float32)

    def close(self):
        pass


class GridProblem:
    """DFT-grid workloads through HologramBatch (one engine, --batch holograms in grid.y)."""

    def __init__(self, args, rank, local_rank):
        from slmsuite_amd import synth
        from slmsuite_amd.batch import HologramBatch
        from slmsuite_amd.holography.algorithms import SpotHologram
        self.args = args
        self.np_dtype = np.float32 if args.dtype == "f32" else np.float64
        w = args.workload
        self.flags = {}
        if w in SPOT_WORKLOADS:
            self.shape, self.slm, grid, pitch = SPOT_WORKLOADS[w]
            host = SpotHologram.make_rectangular_array(self.shape, grid, pitch, basis="knm", slm_shape=self.slm,
                                                       phase=synth.seed_phase(2, self.slm), dtype=self.np_dtype)
            target, kw = host.target, dict(spot_index=host.spot_knm_rounded, spot_amp=host.spot_amp)
            self.n_targets = grid[0] * grid[1]
            self.sparse_target = True
            self.desc = f"SpotHologram {grid} spots, pitch {pitch}"
        elif w in VECTOR_WORKLOADS:
            self.shape, self.slm, box = VECTOR_WORKLOADS[w]
            host = SpotHologram(self.shape, grid_spots(self.shape, box, args.spots), basis="knm", slm_shape=self.slm,
                                phase=synth.seed_phase(2, self.slm), dtype=self.np_dtype)
            target, kw = host.target, dict(spot_index=host.spot_knm_rounded, spot_amp=host.spot_amp)
            self.n_targets = args.spots
            self.sparse_target = True
            self.desc = f"SpotHologram {args.spots} spots at distinct pixels of the centred {box}^2 box"
        else:
            self.shape, self.slm = IMAGE_WORKLOADS[w]
            n = self.shape[0]
            if w == "cfg5mraf":
                target = np.zeros(self.shape, dtype=self.np_dtype)
                a, b = (n - 3072) // 2, (n + 3072) // 2
                target[a:b, a:b] = np.nan
                a, b = (n - 2048) // 2, (n + 2048) // 2
                target[a:b, a:b] = synth.random_target(5, (b - a, b - a), 0.2, 1.0, dtype=self.np_dtype)
                self.flags = {"mraf_factor": 0.5}
                self.signal_cols, self.noise_cols = 2048, 3072      # columns with a finite non-zero / a NaN target
                self.noise_pixels = 3072 * 3072 - 2048 * 2048
                self.n_targets = 2048 * 2048
                self.desc = "Hologram MRAF (mraf_factor 0.5; NaN noise box 3072^2, image 2048^2)"
            elif w == "cfg1":
                target = synth.random_target(1, self.shape, dtype=self.np_dtype)
                self.n_targets = n * n
                self.desc = "Hologram, random amplitude image (the reference's CPU-runnable case; launch / latency bound here)"
            else:
                target = synth.random_target(11, self.shape, 0.2, 1.0, dtype=self.np_dtype)
                self.n_targets = n * n
                self.desc = "Hologram, dense random image target"
            kw = {}
            self.sparse_target = False
        phases = np.stack([synth.seed_phase(1000 * rank + 2 + i, self.slm, dtype=self.np_dtype) for i in range(args.batch)])
        self.hb = HologramBatch(self.shape, self.slm, target, phases, dtype=self.np_dtype, device=local_rank,
                                streams=args.streams, **kw)
        self.engine = self.hb.engine
        self.mraf = self.hb.mraf

    def warm(self, n):
        self.hb.time_iterations(self.args.method, max(1, n), **self.flags)

    def run(self, n):
        return self.hb.time_iterations(self.args.method, n, **self.flags)

    def run_plain(self, n):
        """The same K steps without the HIP-event pair around them (hgs_iterate instead of hgs_iterate_timed): what the
        wall-clock regions run -- two event records, the wait on the second and the read-out cost a 20-step region 2 %
        (tools/region_probe.py: 68.6 -> 67.2 us per step), and they are instrumentation, not steps."""
        self.hb.optimize(self.args.method, maxiter=n, **self.flags)

    def run_profiled(self, n):
        """The K steps of the roofline pass: stream groups one after the other, so that a launch's HIP-event interval holds
        that launch only."""
        if len(self.hb.engines) > 1:
            self.hb.run_groups_one_by_one(self.args.method, n, **self.flags)
        else:
            self.run(n)

    def phases_device(self, torch, device):
        t = torch.empty((self.args.batch,) + tuple(self.slm), dtype=torch.float32 if self.args.dtype == "f32" else torch.float64,
                        device=torch.device("cuda", device))
        self.hb.phases_into_device(t.data_ptr(), t.numel() * t.element_size())
        return t

    def close(self):
        self.hb.close()

    # ---- byte models: tools/benchlib/byte_models.py, unit-tested on CPU ----
    def bytes_models(self):
        return grid_bytes_models(self.shape, self.slm, self.args.dtype, self.args.batch, self.args.streams, self.args.method,
                                 sparse_target=self.sparse_target, n_targets=self.n_targets, mraf=self.mraf,
                                 signal_cols=getattr(self, "signal_cols", None), noise_cols=getattr(self, "noise_cols", None),
                                 noise_pixels=getattr(self, "noise_pixels", None))


class CompressedProblem:
    """cfg 4: CompressedSpotHologram at SLM size, separable bases on the matrix cores."""

    def __init__(self, args, rank, local_rank):
        from slmsuite_amd import synth
        from slmsuite_amd.hardware import SimpleFourierSLM, SimpleSLM
        from slmsuite_amd.holography.algorithms import CompressedSpotHologram
        if args.batch != 1:
            raise SystemExit("cfg4 runs one hologram per GPU")
        if local_rank != 0:
            raise SystemExit("cfg4 is a single-GPU workload")
        self.args = args
        self.D = COMPRESSED_WORKLOADS[args.workload]
        self.slm = (1152, 1920)
        self.N = args.spots
        slm = SimpleSLM(self.slm, pitch_um=(8, 8), wav_um=0.78)
        self.v, basis = compressed_vectors(args.workload, self.N, SimpleFourierSLM(slm))
        self.h = CompressedSpotHologram(self.v, basis=basis, cameraslm=SimpleFourierSLM(slm),
                                        dtype=np.float32 if args.dtype == "f32" else np.float64)
        self.h.reset_phase(synth.seed_phase(4, self.slm))
        self.engine = self.h._get_engine()
        self.desc = f"CompressedSpotHologram {self.N} spots, D = {self.D}"
        self.warmed = False

    def warm(self, n):
        self.h.optimize(self.args.method, maxiter=max(1, n), verbose=False)
        self.engine = self.h._get_engine()

    def run(self, n, events=True):
        e = self.h._get_engine()
        st = self.h._make_step()
        ms = e.iterate_timed(st, n) if events else e.iterate(st, n)
        self.h.iter = st.iter
        self.h.flags["fixed_phase"] = bool(st.fixed_phase)
        self.h._mark_device_fresh(["phase", "weights"])
        return ms

    def run_plain(self, n):
        self.run(n, events=False)

    def close(self):
        self.h._release_engine()


class RefBenchProblem:
    """The reference's own benchmark (test_algorithms.py:121-145) through the product's class surface: whole optimize() calls."""

    def __init__(self, args, rank, local_rank):
        from slmsuite_amd import synth
        from slmsuite_amd.holography.algorithms import Hologram
        if args.batch != 1:
            raise SystemExit("refbench runs one hologram per GPU")
        self.args = args
        self.shape = self.slm = (1024, 1024)
        dt = np.float32 if args.dtype == "f32" else np.float64
        self.target = synth.random_pixels_target(1000 * rank + 7, self.shape, 20, dtype=dt)
        self.h = Hologram(target=self.target, phase=synth.seed_phase(1000 * rank + 7, self.slm, dtype=dt), dtype=dt)
        self.h._get_engine()
        self.engine = self.h._engine
        self.desc = "Hologram, 20 random unit pixels, S = P (test_algorithms.py:121-145), whole optimize() calls"
        self.method = args.method

    # optimize() leaves its trailing transform (_populate_results, _hologram.py:934-949) to whoever reads the results;
    # the reference runs it inside the call and so does the CPU baseline, so the timed region flushes it: every timed
    # call is maxiter loop bodies + one forward transform, on both sides
    def warm(self, n):
        self.h.optimize(self.method, maxiter=max(1, n), verbose=False, stat_groups=[])
        self.h._flush_populate()
        self.engine.sync()

    def run(self, n):
        t0 = time.perf_counter()
        self.h.optimize(self.method, maxiter=n, verbose=False, stat_groups=[])
        self.h._flush_populate()
        self.engine.sync()
        return (time.perf_counter() - t0) * 1e3

    def time_method(self, method, n, reps):
        """Median wall time of optimize(method, maxiter=n) on a fresh state (reset(): the engine survives)."""
        self.h.reset(reset_phase=False, reset_flags=True)
        keep, self.method = self.method, method
        self.warm(n)
        ts = sorted(self.run(n) for _ in range(reps))
        self.method = keep
        return ts[len(ts) // 2]

    def close(self):
        self.h._release_engine()


def compressed_spots(D, N):
    from slmsuite_amd import synth
    v = synth.uniform01(4, (D, N), 9) * 2 - 1
    v[:2] *= 0.02
    if D == 3:
        v[2] *= 1e-6
    return v


def compressed_vectors(workload, N, fourier_slm):
    """(spot_vectors, basis) of a compressed workload: kxy vectors, or Zernike coefficients on ZERN_BASIS (cfg4zern:
    the cfg 4 positions with depth, plus up to half a wave of either astigmatism per spot)."""
    D = COMPRESSED_WORKLOADS[workload]
    if workload != "cfg4zern":
        return compressed_spots(D, N), "kxy"
    from slmsuite_amd import synth
    from slmsuite_amd.holography import toolbox
    zern, _ = toolbox.convert_vector_zernike(compressed_spots(3, N), "kxy", fourier_slm)      # rows: ANSI 2, 1, 4
    astig = (synth.uniform01(4, (2, N), 10) * 2 - 1) * np.pi
    return np.vstack([zern, astig]), np.array(ZERN_BASIS)


# ---------------------------------------------------------------------------------------------------
# CPU baseline: the oracle on a bounded sample of the same workload
# ---------------------------------------------------------------------------------------------------
def cpu_baseline(args):
    from oracle import hgs_oracle as orc          # checker / baseline only; never the product path
    from slmsuite_amd import synth
    w = args.workload
    dt = np.float32 if args.dtype == "f32" else np.float64
    cores_note = f"NumPy {np.__version__}, {os.cpu_count()} host cores visible, 1 used"
    if w == "refbench":
        calls = args.cpu_iters if args.cpu_iters is not None else 6
        if calls <= 0:
            return None
        shape = (1024, 1024)
        o = orc.OracleHologram(synth.random_pixels_target(7, shape, 20, dtype=dt), phase=synth.seed_phase(7, shape, dtype=dt), dtype=dt)
        o.optimize(args.method, maxiter=args.steps, stat_groups=[])
        t0 = time.perf_counter()
        for _ in range(calls):
            o.optimize(args.method, maxiter=args.steps, stat_groups=[])
        dtm = time.perf_counter() - t0
        return {"value": calls * args.steps / dtm, "unit": "iterations/s", "cores": 1, "kind": "port",
                "sample": f"{calls} optimize({args.method}, maxiter={args.steps}, stat_groups=[]) calls incl. the trailing transform "
                          f"({cores_note}), {dtm:.1f} s"}
    if w in SPOT_WORKLOADS:
        iters = args.cpu_iters if args.cpu_iters is not None else (16 if SPOT_WORKLOADS[w][0][0] <= 4096 else 4)
        if iters <= 0:
            return None
        shape, slm, grid, pitch = SPOT_WORKLOADS[w]
        o = orc.OracleSpotHologram(shape, orc.rectangular_array(shape, grid, pitch), slm_shape=slm,
                                   phase=synth.seed_phase(2, slm), dtype=dt)
        flags = {}
        sample = f"{iters} {args.method} loop bodies of {w} (one hologram)"
    elif w in VECTOR_WORKLOADS:
        iters = args.cpu_iters if args.cpu_iters is not None else 3
        if iters <= 0:
            return None
        shape, slm, box = VECTOR_WORKLOADS[w]
        o = orc.OracleSpotHologram(shape, grid_spots(shape, box, args.spots), slm_shape=slm, phase=synth.seed_phase(2, slm), dtype=dt)
        flags = {}
        sample = f"{iters} {args.method} loop bodies of {w} (one hologram)"
    elif w in IMAGE_WORKLOADS:
        iters = args.cpu_iters if args.cpu_iters is not None else (200 if w == "cfg1" else 12 if w == "cfg2dense" else 3)
        if iters <= 0:
            return None
        shape, slm = IMAGE_WORKLOADS[w]
        n = shape[0]
        if w == "cfg5mraf":
            t = np.zeros(shape, dtype=dt)
            a, b = (n - 3072) // 2, (n + 3072) // 2
            t[a:b, a:b] = np.nan
            a, b = (n - 2048) // 2, (n + 2048) // 2
            t[a:b, a:b] = synth.random_target(5, (b - a, b - a), 0.2, 1.0, dtype=dt)
            flags = {"mraf_factor": 0.5}
        elif w == "cfg1":
            t = synth.random_target(1, shape, dtype=dt)
            flags = {}
        else:
            t = synth.random_target(11, shape, 0.2, 1.0, dtype=dt)
            flags = {}
        o = orc.OracleHologram(t, phase=synth.seed_phase(2, slm, dtype=dt), slm_shape=slm, dtype=dt)
        sample = f"{iters} {args.method} loop bodies of {w} (one hologram)"
    else:
        # cfg 4: the oracle holds the dense N x S kernel matrix -- time a subset of the spots and scale
        # linearly in N (both directions are N x S contractions)
        iters = args.cpu_iters if args.cpu_iters is not None else 5
        if iters <= 0:
            return None
        from slmsuite_amd.hardware import SimpleFourierSLM, SimpleSLM
        from slmsuite_amd.holography import toolbox
        D = COMPRESSED_WORKLOADS[w]
        n_s = 96
        slm_shape = (1152, 1920)
        slm = SimpleSLM(slm_shape, pitch_um=(8, 8), wav_um=0.78)
        v, basis = compressed_vectors(w, args.spots, SimpleFourierSLM(slm))
        v = v[:, :n_s]
        zern = v if w == "cfg4zern" else toolbox.convert_vector_zernike(v, "kxy", SimpleFourierSLM(slm))[0]
        xg, yg = toolbox.process_grid(slm)
        sc = slm.get_source_zernike_scaling()
        sx, sy = (sc, sc) if np.isscalar(sc) else (sc[0], sc[1])
        kw_basis = {"zernike_basis": basis} if w == "cfg4zern" else {}
        o = orc.OracleCompressedSpotHologram(zern, np.asarray(xg) * sx, np.asarray(yg) * sy,
                                             phase=synth.seed_phase(4, slm_shape), dtype=dt, **kw_basis)
        o.kernel()                                  # the reference caches K too (_spots.py:595-636)
        o.optimize(args.method, maxiter=1, populate=False)
        t0 = time.perf_counter()
        o.optimize(args.method, maxiter=iters, populate=False)
        dtm = time.perf_counter() - t0
        per = dtm / iters * (args.spots / n_s)
        return {"value": 1.0 / per, "unit": "iterations/s", "cores": 1, "kind": "port",
                "sample": f"{iters} {args.method} loop bodies on {n_s} of the {args.spots} spots (cached kernel matrix), "
                          f"scaled linearly to {args.spots} spots ({cores_note}), {dtm:.1f} s"}
    o.optimize(args.method, maxiter=1, populate=False, **flags)      # warm the caches / first-touch pages
    t0 = time.perf_counter()
    o.optimize(args.method, maxiter=iters, populate=False, **flags)
    dtm = time.perf_counter() - t0
    return {"value": iters / dtm, "unit": "iterations/s", "cores": 1, "kind": "port",
            "sample": f"{sample} ({cores_note}), {dtm:.1f} s"}


# ---------------------------------------------------------------------------------------------------
# PMC traffic, measured in this run: child passes of this script under rocprofv3
# ---------------------------------------------------------------------------------------------------
def main():
    args = parse()
    launched = "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched and not args.pmc_child:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not args.pmc_child:
        raise SystemExit(f"bench.py --gpus {args.gpus} was launched with WORLD_SIZE={world}: the line would report the wrong "
                         f"number of GPUs (launch {args.gpus} ranks, or let bench.py launch them itself)")
    import torch
    stub = args.stub_engine
    if not stub and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    n_dev = 1 if stub else torch.cuda.device_count()
    cuda_sync = (lambda: None) if stub else torch.cuda.synchronize
    if args.share_devices:
        local_rank %= n_dev
    elif local_rank >= n_dev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} has no device ({n_dev} visible); --gpus {args.gpus} needs one GPU per rank")
    dist = None
    ctl = None                 # control group (None = the default group, gloo)
    data_group = None          # the group of the final gather: RCCL (created after the timed regions) or the gloo default
    on_device = True           # the phase masks of the final gather: device memory over RCCL, host memory over gloo
    if (world > 1 or args.force_dist) and not args.pmc_child:
        import torch.distributed as dist
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533")):
            os.environ.setdefault(k, v)         # --force-dist outside a launcher: a one-rank group
        if not stub:
            torch.cuda.set_device(local_rank)
        # The process group the timed regions see is gloo: the barriers of the protocol are host rendezvous (every rank has
        # synchronised its engine streams before it arrives) and the scalars of the max-over-ranks are host numbers.  RCCL
        # carries what north_star gives it -- the gather of the phase masks -- and its communicator is created only then:
        # measured on the one-GPU box, an initialised RCCL communicator costs the two stream groups of a rank their overlap
        # (cfg 3: 15.4 k -> 14.2 k it/s, the one-group rate) and an RCCL barrier 2.2 ms per region, neither of which the
        # one-GPU line of the same bench pays.
        dist.init_process_group("gloo")
        on_device = args.backend == "nccl"
        if dist.get_world_size() != args.gpus and not args.force_dist:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus {args.gpus} asked for")
    coll_dev = torch.device("cpu")

    from slmsuite_amd import _lib as L

    compressed = args.workload in COMPRESSED_WORKLOADS
    refbench = args.workload == "refbench"
    prob = (StubProblem if stub else CompressedProblem if compressed else RefBenchProblem if refbench else GridProblem)(args, rank, local_rank)
    apply_opts(prob.engine, args.opt)
    # targets with empty farfield columns (spot arrays; the zero frame outside an MRAF noise box): the engine would
    # skip those columns, the byte model of the roofline counts all of them - time the dense kernels, report the
    # default separately
    spot = args.workload in SPOT_WORKLOADS or args.workload in VECTOR_WORKLOADS or bool(getattr(prob, "mraf", False))
    if refbench:            # the class surface as a user drives it: engine defaults
        args.sparse_columns = 1

    if spot:
        prob.engine.set_option(L.OPT_SPARSE_COLUMNS, args.sparse_columns)
    # warmup (also takes the hologram past iteration 0 so every timed step updates weights)
    prob.warm(args.warmup)
    if args.pmc_child:                       # profiled child of pmc_traffic(): just launch the kernels
        prob.run(args.steps)
        prob.engine.sync()
        prob.close()
        return
    # timed region, --reps times: EXACTLY K steps between barrier + synchronize, the slowest rank counts
    # (alone: this rank by itself -- no barrier, nobody else's time -- while the others wait, see single_rank_same_job)
    def timed_region(pb, alone=False):
        def sync_all():
            pb.engine.sync()
            cuda_sync()
            if dist is not None and not alone:
                dist.barrier(group=ctl)
        ws, evs, rws = [], [], []
        # the wall-clock repetitions run the K steps and nothing else; the HIP-event figure of the same K steps (the roofline's
        # time base) comes from a few repetitions of its own behind them
        plain = getattr(pb, "run_plain", None)
        for _ in range(args.reps):
            sync_all()
            t0 = time.perf_counter()
            ms_ev = pb.run(args.steps) if plain is None else plain(args.steps)
            sync_all()
            mine = time.perf_counter() - t0
            if dist is not None and not alone:        # the slowest rank counts
                tmax = torch.tensor([mine], dtype=torch.float64, device=coll_dev)
                every = [torch.zeros_like(tmax) for _ in range(world)]
                dist.all_gather(every, tmax, group=ctl)
                rws.append([float(x.item()) for x in every])
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=ctl)
                mine = float(tmax.item())
            ws.append(mine)
            evs.append(ms_ev)
        if plain is not None:
            ev = []
            for _ in range(max(3, min(10, args.reps // 4))):
                sync_all()
                ev.append(pb.run(args.steps))
            sync_all()
            evs = [sorted(ev)[len(ev) // 2]] * len(ws)
        mid_ = sorted(range(args.reps), key=lambda i: ws[i])[len(ws) // 2]
        return ws, evs, rws, mid_

    # Several ranks: first rank 0 ALONE on its own shard -- the very job it runs in the all-rank regions, the other ranks idle
    # at a barrier -- so that this one line carries its own scaling figure: `scaling_efficiency` = value / (N x that rate).
    # (The N = 1 line of the same bench is another workload by default -- cfg 2, one hologram -- and another process.)
    solo = None
    if world > 1 and dist is not None and not args.no_extra_pass:
        if rank == 0:
            ws_, _, _, m_ = timed_region(prob, alone=True)
            solo = {"value": args.batch * args.steps / ws_[m_], "unit": "iterations/s", "ms_per_step": ws_[m_] * 1e3 / args.steps,
                    "what": "rank 0 alone on its shard of the job (same holograms, steps and repetitions; the other ranks wait at a "
                            "barrier), timed before the all-rank regions"}
        dist.barrier(group=ctl)

    walls, events, rank_walls, mid = timed_region(prob)
    wall, ms_events = walls[mid], events[mid]

    # several ranks on the default workload (cfg 3: eight holograms per GPU): the one-hologram-per-GPU rate as well, i.e.
    # the N = 1 default workload (cfg 2) on every rank, so that a scaling figure on EQUAL per-GPU work can be formed
    one_per_gpu = None
    if world > 1 and args.workload == "cfg3" and args.batch != 1 and not args.no_extra_pass and not stub:
        a1 = argparse.Namespace(**vars(args))
        a1.workload, a1.batch, a1.streams = "cfg2", 1, 1
        p1 = GridProblem(a1, rank, local_rank)
        apply_opts(p1.engine, args.opt)
        p1.engine.set_option(L.OPT_SPARSE_COLUMNS, args.sparse_columns)
        p1.warm(args.warmup)
        w1, _, _, m1 = timed_region(p1)
        p1.close()
        one_per_gpu = {"workload": "cfg2 on every rank (one hologram per GPU: the --gpus 1 default)", "value": world * args.steps / w1[m1],
                       "unit": "iterations/s", "ms_per_step": w1[m1] * 1e3 / args.steps}

    # roofline pass: same K steps again with per-launch HIP events on the engine stream
    prof, ran = None, None
    if not args.no_roofline_pass:
        prob.engine.profile_enable(True)
        if hasattr(prob.engine, "dispatch_read"):
            prob.engine.dispatch_read()              # (clear: the record of the roofline pass alone)
        getattr(prob, "run_profiled", prob.run)(args.steps)
        prof = prob.engine.profile_read()
        prob.engine.profile_enable(False)
        # which template instances that pass launched -- the engine's own record (hgs_dispatch_read), not a guess
        if hasattr(prob.engine, "dispatch_read"):
            ran = sorted(prob.engine.dispatch_read(), key=lambda r: -r["count"])

    # the engine's default for spot targets: only the columns that hold a spot are transformed
    sparse_ms, sprof = None, None
    if spot and not args.no_extra_pass and not args.sparse_columns:
        prob.engine.set_option(L.OPT_SPARSE_COLUMNS, 1)
        prob.warm(args.warmup)
        sparse_ms = prob.run(args.steps)
        if not args.no_roofline_pass:
            prob.engine.profile_enable(True)
            getattr(prob, "run_profiled", prob.run)(args.steps)
            sprof = prob.engine.profile_read()
            prob.engine.profile_enable(False)
        prob.engine.set_option(L.OPT_SPARSE_COLUMNS, 0)

    gather_ms, gathered, group_info = None, None, None       # filled in by the final gather, after the line is built

    ref_methods = None
    if refbench and not args.no_extra_pass:
        ref_methods = {m: args.steps / (prob.time_method(m, args.steps, max(3, args.reps)) * 1e-3) for m in REFBENCH_METHODS}

    if rank == 0:
        iters_total = world * args.batch * args.steps
        value = iters_total / wall
        want_pmc = args.pmc if args.pmc is not None else (1 if (world == 1 and dist is None) else 0)
        roof = None
        bm = {}
        if compressed and prof is not None and args.workload == "cfg4zern":
            # direct kernels: one value of exp(+-i phi_n(p)) and one complex MAC per (spot, pixel) and direction; counted as
            # SURVEY 8(d) counts the reference's evaluation: (2 D + 8 + 2) flop each (phase polynomial, sin, cos, complex MAC)
            S = prob.slm[0] * prob.slm[1]
            flop_launch = (2.0 * prob.D + 10.0) * prob.N * S
            n_l = prof["col_fwd"]["launches"] + prof["col_inv"]["launches"]
            dur = (prof["col_fwd"]["ms"] + prof["col_inv"]["ms"]) * 1e-3 / max(1, n_l)
            ach = flop_launch / dur
            roof = {"bound": "valu", "kernel": "c_n2f_run / c_f2n_run direct kernels (exp(i phi) of every pixel and spot advanced along "
                                               "16-pixel runs by recurrence, start values in double, nothing tabulated) + their "
                                               "reductions, one timed unit per transform direction",
                    "achieved": ach / 1e12, "peak": VALU_F32_PEAK / 1e12, "unit": "TFLOP/s", "frac": ach / VALU_F32_PEAK,
                    "traffic": None, "traffic_note": "VALU / transcendental bound: 2 N S kernel evaluations per iteration, "
                                                     "compulsory traffic ~ 35 MB",
                    "flop_per_launch": flop_launch, "launch_us": dur * 1e6, "launches": n_l,
                    "evaluations_per_s": prob.N * S / dur,
                    "timing": "HIP events per transform on the engine stream, second pass of K steps"}
        elif compressed and prof is not None:
            # dominant kernel: the two complex GEMMs (cgemm_streamk), timed under col_fwd / col_inv together with their small
            # helper kernels.  Algorithmic work of a GEMM launch = 8 N S real flop (a complex multiply-add = 4 real ones);
            # the kernel forms each complex product from THREE real matrix products (P1 = Ar Br, P2 = Ai Bi,
            # P3 = (Ar + Ai)(Br + Bi)), i.e. it issues 6 N S flop to the matrix pipe.  `achieved` / `frac` count what is
            # issued (what the pipe does); the algorithmic count over the same time is `algorithmic_equivalent` -- a
            # throughput figure that may exceed the peak, like `canonical_equivalent` of the HBM-bound kernels.
            S = prob.slm[0] * prob.slm[1]
            flop_alg = 8.0 * prob.N * S
            flop_launch = 6.0 * prob.N * S
            n_l = prof["col_fwd"]["launches"] + prof["col_inv"]["launches"]
            dur = (prof["col_fwd"]["ms"] + prof["col_inv"]["ms"]) * 1e-3 / max(1, n_l)
            ach = flop_launch / dur
            roof = {"bound": "mfma", "kernel": "cgemm_streamk (complex fp32 GEMM, v_mfma_f32_32x32x2_f32, three real products per "
                                               "complex one, stream-K over 2 x #CU workgroups) + its table/contraction helpers, one "
                                               "timed unit per transform direction",
                    "achieved": ach / 1e12, "peak": MFMA_F32_PEAK / 1e12, "unit": "TFLOP/s", "frac": ach / MFMA_F32_PEAK,
                    "traffic": None, "traffic_note": "matrix-core bound; HBM traffic not the limiter (tables 153 + 153 + 92 MB)",
                    "flop_per_launch": flop_launch, "launch_us": dur * 1e6, "launches": n_l,
                    "flop_note": "issued to the matrix pipe: 6 N S per transform (three real products per complex one)",
                    "algorithmic_equivalent": {"flop_per_launch": flop_alg, "achieved": flop_alg / dur / 1e12,
                                               "frac": flop_alg / dur / MFMA_F32_PEAK,
                                               "note": "8 N S real flop per transform (four real products per complex one), "
                                                       "same duration"},
                    "iteration": {"flop": 2 * flop_launch, "achieved": 2 * flop_launch * args.steps / (ms_events * 1e-3) / 1e12,
                                  "frac": 2 * flop_launch * args.steps / (ms_events * 1e-3) / MFMA_F32_PEAK},
                    "timing": "HIP events per transform on the engine stream, second pass of K steps"}
        elif refbench and prof is not None:
            P = prob.shape[0] * prob.shape[1]
            r = 4 if args.dtype == "f32" else 8
            canon = ((13 if args.method == "GS" else 15) * P + 2 * P) * r
            its = args.steps / (wall)
            roof = {"bound": "hbm", "kernel": "col_fused_kernel over the active-column list + row_kernel (masked), launch / latency bound",
                    "achieved": canon * its / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": canon * its / HBM_PEAK,
                    "traffic": None,
                    "traffic_note": "latency-bound workload (the whole state is 21 MB): SURVEY 8(d) asks for absolute it/s here; "
                                    "`achieved` = SURVEY's canonical bytes per iteration x iterations/s of whole optimize() calls",
                    "bytes_per_iteration_canonical": canon,
                    "kernels_us": {k: (v["ms"] * 1e3 / v["launches"] if v["launches"] else None) for k, v in prof.items()},
                    "launches_per_call": {k: v["launches"] for k, v in prof.items()}}
        elif prof is not None and prof["col_fused"]["launches"] > 0:
            bm = prob.bytes_models()
            col, rowk = prof["col_fused"], prof["row"]
            # MRAF + weight update launches two column passes per iteration: time and bytes are per iteration there
            per = bm["col_passes"]
            dur = col["ms"] * 1e-3 / col["launches"] * per
            achieved = bm["col"] / dur
            row_dur = rowk["ms"] * 1e-3 / max(1, rowk["launches"])
            traffic, tnote, tr_row, kernel_ran = None, "--pmc 0", None, None
            if want_pmc:
                # which fused column kernel ran is the engine's choice (tile-resident, per column, sparse list):
                # count both and keep the one that was launched
                subs = {"col_tile": "col_tile_kernel", "col_tile2": "col_tile2_kernel", "col_fused": "col_fused_kernel",
                        "row": "row_kernel<" + ("float" if args.dtype == "f32" else "double") + f", {prob.shape[1]}, 2"}   # (<R, N, MODE 2[, NS]>)
                res, tnote = pmc_traffic(args, subs)
                if res is not None:
                    res["col"] = max((res["col_tile"], res["col_tile2"], res["col_fused"]), key=lambda r: r["launches"])
                if res is not None and res["col"]["fetch"] is not None and res["col"]["write"] is not None:
                    traffic = (res["col"]["fetch"] + res["col"]["write"]) * per
                    tr_row = None if res["row"]["fetch"] is None else res["row"]["fetch"] + res["row"]["write"]
                    tnote += f"; kernel = {res['col']['kernel_name']}"
                    if args.dtype != "f32":
                        tnote += ("; the x 2 FETCH_SIZE correction is calibrated for 16-B-per-lane fp32 streams, fp64 access "
                                  "widths are uncalibrated (guide) - treat this figure as indicative")
                    kernel_ran = res["col"]["kernel_name"]
                elif res is not None:
                    tnote = "kernel not found in the PMC pass: " + json.dumps(res)
            iter_s = args.steps / (ms_events * 1e-3)
            launched = [r for r in (ran or []) if r["kernel"] in ("col_tile_kernel", "col_tile2_kernel", "col_fused_kernel")]
            roof = {"bound": "hbm", "kernel": kernel_ran or (launched[0]["name"] if launched else "(no dispatch record)"),
                    "kernels_launched": [f"{r['name']} x{r['count']}" for r in (ran or [])],
                    "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK,
                    "traffic": traffic, "traffic_note": tnote,
                    # scalars of the nested objects below (a parser that keeps only flat fields still carries them):
                    # the whole iteration on moved bytes -- the figure north_star's 50 % is about --, the row launch, and
                    # SURVEY's canonical (un-pruned) count over the same time (a throughput equivalent, may exceed 1)
                    "frac_iteration": (bm["col"] + bm["row"] + bm.get("other", 0)) * args.streams * iter_s / HBM_PEAK,
                    "row_frac": bm["row"] / row_dur / HBM_PEAK,
                    "row_launch_us": row_dur * 1e6,
                    "canonical_equivalent_frac": bm["canon_iter"] * args.streams * iter_s / HBM_PEAK,
                    "traffic_over_model": None if traffic is None else traffic / bm["col"],
                    "frac_on_traffic": None if traffic is None else traffic / dur / HBM_PEAK,
                    "bytes_per_launch": bm["col"], "bytes_model": bm["col_model"],
                    "launch_us": dur * 1e6, "launches": col["launches"] // per,
                    "canonical_equivalent": {"bytes_per_launch": bm["canon_col"], "achieved": bm["canon_col"] / dur / 1e9,
                                             "frac_of_peak": bm["canon_col"] / dur / HBM_PEAK,
                                             "note": "SURVEY 8(d) un-pruned count 4*P*c + 3*P*r per column launch; a throughput "
                                                     "equivalent, NOT bytes moved (rows outside the SLM are never stored)"},
                    "valu_view": valu_view(args, prob, dur, row_dur),
                    "working_set_bytes": bm["working_set"],
                    "infinity_cache_resident": bool(bm["working_set"] <= MALL_BYTES),
                    "infinity_cache_note": "FETCH_SIZE/WRITE_SIZE count fabric requests including Infinity-Cache (256 MiB) hits; "
                                           "when the working set fits, true HBM-pin traffic is lower than `traffic`",
                    "row_kernel": {"launch_us": row_dur * 1e6, "bytes_per_launch": bm["row"], "bytes_model": bm["row_model"],
                                   "achieved": bm["row"] / row_dur / 1e9, "frac": bm["row"] / row_dur / HBM_PEAK,
                                   "traffic": tr_row, "traffic_over_model": None if tr_row is None else tr_row / bm["row"]},
                    "iteration": {"moved_bytes": (bm["col"] + bm["row"] + bm.get("other", 0)) * args.streams,
                                  "achieved": (bm["col"] + bm["row"] + bm.get("other", 0)) * args.streams * iter_s / 1e9,
                                  "frac": (bm["col"] + bm["row"] + bm.get("other", 0)) * args.streams * iter_s / HBM_PEAK,
                                  "canonical_bytes": bm["canon_iter"] * args.streams,
                                  "canonical_equivalent_frac": bm["canon_iter"] * args.streams * iter_s / HBM_PEAK,
                                  "note": "all stream groups of this rank together" if args.streams > 1 else None},
                    "timing": "HIP events per launch on the engine stream, second pass of K steps; an event pair adds about "
                              "1 - 2 us to a launch (the rocprofv3 --kernel-trace average of the same kernel, profiles/, is "
                              "the sharper figure)" + ("; this pass runs the stream groups one after the other, so that a launch's "
                              "event interval holds that launch only" if args.streams > 1 else "")}
        ok_ = bm.get("other_kind", "col_inv")
        if roof is not None and prof is not None and bm.get("other") and prof.get(ok_, {}).get("launches", 0) > 0:
            inv_dur = prof[ok_]["ms"] * 1e-3 / prof[ok_]["launches"]
            roof["noise_inverse_launch" if ok_ == "col_inv" else "presum_launch"] = {
                "kernel": bm.get("other_name"), "launch_us": inv_dur * 1e6, "bytes_per_launch": bm["other"],
                "frac": bm["other"] / inv_dur / HBM_PEAK}
        if (roof is not None and roof.get("presum_launch") and roof.get("traffic") is not None
                and str(bm.get("other_name", "")).startswith("col_fused_kernel")):
            # the pre-pass IS the column kernel (same instance, forward only): the PMC pass knows kernels by name, so its mean per
            # dispatch runs over both launches of an iteration -- held against the mean of the two models
            mean_model = (bm["col"] + bm["other"]) / 2
            roof["traffic_over_model"] = roof["traffic"] / mean_model
            roof["frac_on_traffic"] = None
            roof["traffic_note"] += ("; the forward-only pre-pass is a launch of the same kernel: `traffic` is the mean over both "
                                     f"launches of an iteration, `traffic_over_model` holds it against (col + pre-pass) / 2 = {int(mean_model)} B")
        if roof is not None and roof.get("traffic_over_model") is not None and abs(roof["traffic_over_model"] - 1) > 0.05:
            roof["traffic_explanation"] = traffic_explanation(args, prob, roof)
        cpu = cpu_baseline(args) if world == 1 else None
        shape_txt = "" if compressed else f" padded to {prob.shape[0]}x{prob.shape[1]}"
        line = {
            "metric": "WGS iterations/sec (4096^2 padded field)" if args.workload in ("cfg2", "cfg3") else
                      f"{args.method} iterations/sec ({args.workload}" + (": whole optimize() calls)" if refbench else ")"),
            "value": value, "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.workload}: {prob.desc}, SLM {prob.slm[0]}x{prob.slm[1]}{shape_txt}, "
                                   f"{args.method}, {'fp32' if args.dtype == 'f32' else 'fp64'}",
                       "holograms_per_gpu": args.batch, "stream_groups_per_gpu": args.streams,
                       "parallelism": f"independent holograms x{world}"},
            "repetitions": args.reps, "ms_per_step_min": min(walls) * 1e3 / args.steps,
            "ms_per_step_max": max(walls) * 1e3 / args.steps,
            "timing": f"median of {args.reps} repetitions of the {args.steps}-step region (barrier + synchronize on both sides, "
                      "max over ranks); state resident in HBM; the region holds the K steps only -- `event_ms_per_step` (HIP events "
                      "around the same K steps) is taken in repetitions of its own, since round 6",
            "event_ms_per_step": ms_events / args.steps, "gather_ms": gather_ms,
            "engine": prob.engine.version(),
            "roofline": roof, "cpu_baseline": cpu,
        }
        if rank_walls:
            med = [sorted(r[i] for r in rank_walls)[len(rank_walls) // 2] for i in range(world)]
            line["per_rank_its"] = [args.batch * args.steps / t for t in med]
        if one_per_gpu is not None:
            line["one_hologram_per_gpu"] = one_per_gpu
        if solo is not None:
            line["single_rank_same_job"] = solo
            line["scaling_efficiency"] = value / (world * solo["value"])
        if stub:
            line["metric"] = "STUB ENGINE -- rank-protocol self-test, not a measurement"
            line["stub_engine"] = True
        if args.share_devices:
            line["launcher_self_test"] = (f"{world} ranks share {n_dev} device(s) over {args.backend}: exercises the launcher and the "
                                          "collectives, NOT a scaling measurement")
        if ref_methods is not None:
            line["methods"] = ref_methods
            line["methods_note"] = ("iterations/s of whole optimize(method, maxiter=K, stat_groups=[]) calls from a reset state, median; "
                                    "the reference's parametrisation (tests/holography/test_algorithms.py:121)")
        if spot:
            line["column_mode"] = ("sparse-aware (engine default)" if args.sparse_columns else
                                   "dense kernels forced (HGS_OPT_SPARSE_COLUMNS=0): every farfield column transformed")
        if sparse_ms is not None:
            line["engine_default_path"] = {
                "what": "same workload with the engine default HGS_OPT_SPARSE_COLUMNS=1: only the farfield columns "
                        "holding a non-zero weight/target are transformed and moved (identical results)",
                "value": world * args.batch * args.steps / (sparse_ms * 1e-3), "unit": "iterations/s (rank-0 HIP events)",
                "ms_per_step": sparse_ms / args.steps,
                "col_kernel_us": None if sprof is None else sprof["col_fused"]["ms"] * 1e3 / max(1, sprof["col_fused"]["launches"]),
                "row_kernel_us": None if sprof is None else sprof["row"]["ms"] * 1e3 / max(1, sprof["row"]["launches"]),
            }
    else:
        line = None

    # Final gather of the phase masks over RCCL, device memory -> RCCL, timed separately.  It runs AFTER the
    # line has been built from the timed regions and under a watchdog: a communicator that cannot be created or a collective
    # that never returns costs the line its gather fields, not the measurement.
    if dist is not None:
        import threading
        finished = threading.Event()

        def give_up():
            if finished.is_set():
                return
            if rank == 0:
                line["gathered"] = {"error": f"the all-gather of the phase masks did not return within {args.gather_timeout:g} s; "
                                             "the timed regions do not depend on it"}
                line["rccl_ranks"] = 0
                print(json.dumps(line), flush=True)
            os._exit(0)

        watchdog = threading.Timer(args.gather_timeout, give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            if not compressed and not refbench:
                if on_device:
                    try:
                        data_group = dist.new_group(backend="nccl", device_id=torch.device("cuda", local_rank))
                    except TypeError:          # (older torch: no device_id argument)
                        data_group = dist.new_group(backend="nccl")
                cuda_sync()
                dist.barrier(group=ctl)
                t1 = time.perf_counter()
                ph = prob.phases_device(torch, local_rank)
                if not on_device:
                    ph = ph.cpu()
                out = [torch.empty_like(ph) for _ in range(world)]
                dist.all_gather(out, ph, group=data_group)
                cuda_sync()
                gather_ms = (time.perf_counter() - t1) * 1e3
                # every rank must now hold every rank's masks: finite, and its own shard back unchanged
                ok = all(bool(torch.isfinite(o).all().item()) for o in out) and bool(torch.equal(out[rank], ph))
                flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=coll_dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=ctl)
                gathered = {"masks": world * args.batch, "bytes": world * ph.numel() * ph.element_size(),
                            "verified_on_every_rank": bool(flag.item() == 1.0)}
            group_info = {"backend": dist.get_backend(data_group), "ranks": dist.get_world_size(data_group),
                          "devices": sorted(set(_gather_ints(dist, torch, coll_dev, local_rank, world, ctl))),
                          "barriers": "gloo (host rendezvous after hgs_sync; the RCCL communicator of the gather is created after "
                                      "the timed regions)" if on_device else "gloo"}
        except Exception as exc:           # noqa: BLE001 -- whatever the communicator raises: reported, not fatal
            gathered = {"error": f"{type(exc).__name__}: {exc}"[:400]}
        finished.set()
        watchdog.cancel()
        if rank == 0:
            line["gather_ms"] = gather_ms
            if group_info is not None:
                line["process_group"] = group_info      # read back from the group: backend ("nccl" = RCCL), ranks, device per rank
                line["rccl_ranks"] = group_info["ranks"] if group_info["backend"] == "nccl" else 0
            line["gathered"] = gathered
            if gather_ms is not None:
                # "cfg3 reports aggregate it/s = sum over GPUs of (holograms x iterations) / wall-time, including the
                # final RCCL gather".  `value` is the rate of the K-step regions; here the gather is charged once to the region
                # (K steps) and once to the job as BASELINE configures it (50 iterations per hologram).
                its = world * args.batch
                line["value_including_gather"] = {
                    "value": its * args.steps / (wall + gather_ms * 1e-3), "unit": "iterations/s",
                    "what": f"{args.steps} steps per hologram + one all-gather of the phase masks",
                    "job_of_50_iterations": its * 50 / (50 * wall / args.steps + gather_ms * 1e-3),
                    "gather_ms": gather_ms}
    if rank == 0:
        print(json.dumps(line), flush=True)
    prob.close()
    if dist is not None:
        dist.destroy_process_group()


def valu_view(args, prob, col_dur, row_dur):
    """
    The same two launches against the VECTOR pipe (they do not wait on HBM: DESIGN.md section 6).  Nominal transform work
    5 N log2 N real operations per length-N complex transform (the textbook count; zero rows are pruned, so fewer are
    issued): a column launch is 2 Pw transforms of length Ph per hologram, a row launch 2 Sh of length Pw.  Peaks: 157.3
    TFLOP/s is packed fp32 FMA (4 flop per lane and issue); a butterfly network is mostly additions (2 flop per lane and
    issue), for which the pipe's ceiling is 78.6 TFLOP/s; fp64: 78.6 / 39.3.
    """
    import math
    Ph, Pw = prob.shape
    Sh = prob.slm[0]
    B = -(-args.batch // args.streams)
    fma_peak = 157.3e12 if args.dtype == "f32" else 78.6e12
    col_flop = 2 * Pw * 5.0 * Ph * math.log2(Ph) * B * prob.bytes_models()["col_passes"]
    row_flop = 2 * Sh * 5.0 * Pw * math.log2(Pw) * B
    out = {}
    for name, flop, dur in (("column_launch", col_flop, col_dur), ("row_launch", row_flop, row_dur)):
        out[name] = {"nominal_fft_flop": flop, "achieved_tflops": flop / dur / 1e12, "frac_of_fma_peak": flop / dur / fma_peak,
                     "frac_of_add_peak": flop / dur / (fma_peak / 2)}
    out["note"] = ("nominal 5 N log2 N per transform; two-pass launches counted with both forward transforms; the constraint "
                   "arithmetic (up to ~45 instructions per evaluated pixel) is not included")
    return out


def traffic_explanation(args, prob, roof):
    """Why the fabric counters and the byte model of the column launch differ by more than 5 % (known cases)."""
    ratio = roof["traffic_over_model"]
    if args.dtype != "f32":
        return ("fp64: the x 2 FETCH_SIZE correction of gfx950 is calibrated for 16-byte-per-lane fp32 streams; with 8-byte "
                "elements the counter under-reports (guide: uncalibrated), the model is the better figure")
    if ratio < 1 and roof.get("infinity_cache_resident"):
        return "working set inside the Infinity Cache: part of the weight / target re-reads never reach the fabric counters"
    if ratio > 1:
        return ("partial-line accesses: 32-byte tile rows of GH and 64-byte lane groups of the weights are fetched as whole "
                "128-byte lines where the neighbouring pieces are not in the same L2 at the same time")
    return "unexplained: treat `traffic` as the moved bytes and `frac_on_traffic` as the fraction"


if __name__ == "__main__":
    main()
