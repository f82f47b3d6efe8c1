#!/usr/bin/env python
"""
Benchmark of the hologram optimize() hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A *step* is one WGS iteration (nearfield -> farfield -> constraint/weight update -> nearfield) of
BASELINE.json config 2: SpotHologram, 32x32 spots (pitch 64 px) on a 4096 x 4096 padded grid,
SLM 1152 x 1920, WGS-Leonardo, fp32, synthetic seed phase, state resident in HBM.  Each rank owns
``--batch`` independent holograms (weak scaling); the only collective is the final
all-gather of the phase masks (reported as gather_ms, outside the timed region).

The headline is timed with the dense kernels forced (every farfield column transformed, as the
canonical byte count assumes); the engine's default for spot targets -- transform only the columns
that hold a spot, identical results -- is timed in an extra pass and reported as
``engine_default_path``.

Rank 0 prints one JSON line: metric/value (whole-job iterations/s), plus
  roofline      the dominant kernel (fused column kernel) against the 8 TB/s HBM peak:
                ALGORITHMIC bytes per launch (44 * P * r * batch) / its mean duration,
                measured here with HIP events on the engine stream in a second pass of K steps;
  cpu_baseline  the CPU oracle (NumPy restatement of the reference path, kind "port") timed on
                this box's host cores on a bounded sample of the same workload (rank 0, N = 1).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # bytes/s, MI355X_MICROARCH.md

WORKLOADS = {
    # name: (padded shape, slm shape, spot grid, pitch)
    "cfg2": ((4096, 4096), (1152, 1920), (32, 32), (64, 64)),
    "small": ((1024, 1024), (288, 480), (16, 16), (32, 32)),
    "hd": ((2048, 2048), (1080, 1920), (16, 16), (64, 64)),        # a 1920x1080 SLM at padding_order = 1
    "cfg5pad": ((8192, 8192), (1152, 1920), (32, 32), (128, 128)),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1, help="independent holograms per GPU")
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--method", default="WGS-Leonardo")
    ap.add_argument("--cpu-iters", type=int, default=16, help="iterations of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-roofline-pass", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL even for one rank (self-test)")
    ap.add_argument("--sparse-columns", type=int, default=0,
                    help="1: time the engine's default (sparse-target aware) path as the headline; 0 (default): "
                         "force the dense kernels (every farfield column transformed) for the headline and "
                         "report the default path separately")
    ap.add_argument("--no-extra-pass", action="store_true")
    return ap.parse_args()


def build_problem(workload, rank, batch):
    from slmsuite_amd import synth
    from slmsuite_amd.holography.algorithms import SpotHologram
    shape, slm, grid, pitch = WORKLOADS[workload]
    host = SpotHologram.make_rectangular_array(shape, grid, pitch, basis="knm", slm_shape=slm,
                                               phase=synth.seed_phase(2, slm))
    phases = np.stack([synth.seed_phase(1000 * rank + 2 + i, slm) for i in range(batch)])
    return shape, slm, host, phases


def cpu_baseline(workload, iters):
    """The CPU oracle on a bounded sample: `iters` loop bodies of the same workload, one core."""
    from oracle import hgs_oracle as orc          # checker / baseline only; never the product path
    from slmsuite_amd import synth
    shape, slm, grid, pitch = WORKLOADS[workload]
    o = orc.OracleSpotHologram(shape, orc.rectangular_array(shape, grid, pitch), slm_shape=slm,
                               phase=synth.seed_phase(2, slm))
    o.optimize("WGS-Leonardo", maxiter=1, populate=False)      # warm the caches / first-touch pages
    t0 = time.perf_counter()
    o.optimize("WGS-Leonardo", maxiter=iters, populate=False)
    dt = time.perf_counter() - t0
    return {"value": iters / dt, "unit": "iterations/s", "cores": 1, "kind": "port",
            "sample": f"{iters} WGS-Leonardo loop bodies of {workload} (NumPy {np.__version__}, "
                      f"{os.cpu_count()} host cores visible, 1 used), {dt:.1f} s"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")

    from slmsuite_amd import _lib as L
    from slmsuite_amd.batch import HologramBatch

    shape, slm, host, phases = build_problem(args.workload, rank, args.batch)
    hb = HologramBatch(shape, slm, host.target, phases, device=local_rank,
                       spot_index=host.spot_knm_rounded, spot_amp=host.spot_amp)

    def barrier():
        hb.engine.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    hb.engine.set_option(L.OPT_SPARSE_COLUMNS, args.sparse_columns)
    # warmup (also takes the hologram past iteration 0 so every timed step updates weights)
    hb.time_iterations(args.method, max(1, args.warmup))
    barrier()
    t0 = time.perf_counter()
    ms_events = hb.time_iterations(args.method, args.steps)
    barrier()
    wall = time.perf_counter() - t0
    tmax = torch.tensor([wall], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    wall = float(tmax.item())

    # roofline pass: same K steps again with per-launch HIP events on the engine stream
    prof = None
    if not args.no_roofline_pass:
        hb.engine.profile_enable(True)
        hb.time_iterations(args.method, args.steps)
        prof = hb.engine.profile_read()
        hb.engine.profile_enable(False)

    # the engine's default for this workload: only the columns that hold a spot are transformed
    sparse_ms = None
    if not args.no_extra_pass and not args.sparse_columns:
        hb.engine.set_option(L.OPT_SPARSE_COLUMNS, 1)
        hb.time_iterations(args.method, max(1, args.warmup))
        sparse_ms = hb.time_iterations(args.method, args.steps)
        sprof = None
        if not args.no_roofline_pass:
            hb.engine.profile_enable(True)
            hb.time_iterations(args.method, args.steps)
            sprof = hb.engine.profile_read()
            hb.engine.profile_enable(False)
        hb.engine.set_option(L.OPT_SPARSE_COLUMNS, 0)

    # final gather of the phase masks over RCCL, timed separately
    gather_ms = None
    if dist is not None:
        ph = torch.from_numpy(hb.phases()).cuda()
        out = [torch.empty_like(ph) for _ in range(world)]
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        dist.all_gather(out, ph)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - t1) * 1e3

    if rank == 0:
        P = shape[0] * shape[1]
        S = slm[0] * slm[1]
        r = 4
        iters_total = world * args.batch * args.steps
        value = iters_total / wall
        bytes_iter = (15 * P + 2 * S) * r            # canonical B_WGS
        roof = None
        if prof is not None and prof["col_fused"]["launches"] > 0:
            col = prof["col_fused"]
            dur = col["ms"] * 1e-3 / col["launches"]
            alg = 44 * P * args.batch                  # 4*P*c + 3*P*r bytes per launch
            achieved = alg / dur
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
            if os.path.exists(pmc) and args.workload == "cfg2" and args.batch == 1 and args.method == "WGS-Leonardo":
                try:
                    traffic = json.load(open(pmc)).get("col_fused_bytes_per_launch")
                except Exception:
                    traffic = None
            rowk = prof["row"]
            roof = {"bound": "hbm", "kernel": "fused column kernel (col_tile_kernel<float,N,PHASE,6> / col_fused_kernel)",
                    "achieved": achieved / 1e9,
                    "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": traffic,
                    "launch_us": dur * 1e6, "launches": col["launches"],
                    "algorithmic_bytes_per_launch": alg,
                    "row_kernel_us": rowk["ms"] * 1e3 / max(1, rowk["launches"]),
                    "iteration": {"algorithmic_bytes": bytes_iter * args.batch,
                                  "achieved": bytes_iter * args.batch * (args.steps / (ms_events * 1e-3)) / 1e9,
                                  "frac": bytes_iter * args.batch * (args.steps / (ms_events * 1e-3)) / HBM_PEAK,
                                  # what the pruned/fused design actually moves per iteration:
                                  # GH read+write by both kernels, weights r/w, target r, phase w
                                  "designed_bytes": (4 * 8 * slm[0] * shape[1] + 3 * 4 * P + 4 * S) * args.batch},
                    "timing": "HIP events per launch on the engine stream, second pass of K steps"}
        cpu = None
        if world == 1 and args.cpu_iters > 0:
            cpu = cpu_baseline(args.workload, args.cpu_iters)
        line = {
            "metric": "WGS iterations/sec (4096^2 padded field)", "value": value, "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: SpotHologram {WORKLOADS[args.workload][2]} spots, "
                                   f"SLM {slm[0]}x{slm[1]} padded to {shape[0]}x{shape[1]}, {args.method}, fp32",
                       "holograms_per_gpu": args.batch, "parallelism": f"independent holograms x{world}"},
            "event_ms_per_step": ms_events / args.steps, "gather_ms": gather_ms,
            "engine": hb.engine.version(),
            "roofline": roof, "cpu_baseline": cpu,
            "column_mode": "sparse-aware (engine default)" if args.sparse_columns else
                           "dense kernels forced (HGS_OPT_SPARSE_COLUMNS=0): all 4096 columns transformed",
        }
        if sparse_ms is not None:
            line["engine_default_path"] = {
                "what": "same workload with the engine default HGS_OPT_SPARSE_COLUMNS=1: only the farfield columns "
                        "holding a non-zero weight/target are transformed and moved (identical results)",
                "value": world * args.batch * args.steps / (sparse_ms * 1e-3), "unit": "iterations/s (rank-0 HIP events)",
                "ms_per_step": sparse_ms / args.steps,
                "col_kernel_us": None if sprof is None else sprof["col_fused"]["ms"] * 1e3 / max(1, sprof["col_fused"]["launches"]),
                "row_kernel_us": None if sprof is None else sprof["row"]["ms"] * 1e3 / max(1, sprof["row"]["launches"]),
            }
        print(json.dumps(line))
    hb.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
