"""
Two processes driving the HIP engine at once (``-m gpu``): two gloo ranks shard a batch of holograms, each optimises its
shard on the GPU through its own engine (the box has one GPU, so both ranks use device 0; RCCL itself needs one device
per rank and is not what this test is about) and the phase masks are all-gathered.  Every rank must end up with the
masks a single process computes for the whole batch (independent holograms, no data-path collective).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu

SHAPE, SLM, N_HOLO, ITERS = (512, 512), (288, 480), 5, 6


def _inputs():
    from slmsuite_amd import synth
    target = synth.random_pixels_target(3, SHAPE, 24)
    phases = np.stack([synth.seed_phase(170 + i, SLM) for i in range(N_HOLO)])
    return target, phases


def _run(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from slmsuite_amd.batch import optimize_batch_distributed, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    target, phases = _inputs()
    res = optimize_batch_distributed(SHAPE, SLM, target, phases, "WGS-Leonardo", ITERS, device=0)
    q.put((rank, res, shard_range(N_HOLO, rank, world)))
    dist.destroy_process_group()


def test_two_processes_drive_the_engine_and_gather():
    from slmsuite_amd.batch import optimize_batch
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_run, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    target, phases = _inputs()
    want = optimize_batch(SHAPE, SLM, target, phases, "WGS-Leonardo", ITERS, device=0)
    assert sorted(g[2] for g in got) == [(0, 3), (3, 5)]
    for rank, res, _ in got:
        assert res.shape == (N_HOLO,) + SLM
        # a shard is a smaller batch of the same kernels: identical arithmetic per hologram
        np.testing.assert_array_equal(res, want)


def _bench(argv, timeout=900):
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True,
                       timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, [json.loads(l) for l in lines]


def test_bench_launches_its_own_ranks():
    """
    ``python bench.py --gpus 2`` with no launcher around it (the form the driver uses for N = 1) must come back with a
    TWO-rank line.  The box has one GPU, so the two ranks share it over gloo (RCCL wants a device per rank); everything
    else -- the self-launch under torch.distributed.run, rank binding, barriers, max-over-ranks timing, the gather of
    the phase masks and its check on every rank -- is the path an 8-GPU node takes.
    """
    r, lines = _bench(["--gpus", "2", "--backend", "gloo", "--share-devices", "--steps", "4", "--warmup", "2", "--reps", "2",
                       "--no-roofline-pass"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout
    d = lines[0]
    assert d["n_gpus"] == 2 and d["process_group"]["ranks"] == 2 and d["process_group"]["backend"] == "gloo"
    assert d["rccl_ranks"] == 0                                  # honest: this was not RCCL
    assert d["config"]["holograms_per_gpu"] == 8 and "cfg3" in d["config"]["workload"]
    assert len(d["per_rank_its"]) == 2 and all(v > 0 for v in d["per_rank_its"])
    assert d["gathered"]["masks"] == 16 and d["gathered"]["verified_on_every_rank"] is True
    assert d["gather_ms"] > 0 and d["value"] > 0
    assert d["one_hologram_per_gpu"]["value"] > 0
    assert "launcher_self_test" in d
    # the line carries its own scaling figure (rank 0 alone on its shard, timed first) and the aggregate with the gather
    solo = d["single_rank_same_job"]
    assert solo["value"] > 0 and abs(d["scaling_efficiency"] - d["value"] / (2 * solo["value"])) < 1e-12
    # (two ranks on ONE device: about half each -- 0.45 .. 0.6 in most runs, but the regions are four steps long and rank 0's
    #  solo region is the first work the device sees, so only the order of magnitude is asserted: 0.82 was seen once)
    assert 0.1 < d["scaling_efficiency"] < 1.5, d["scaling_efficiency"]
    g = d["value_including_gather"]
    assert 0 < g["value"] < g["job_of_50_iterations"] < d["value"] and g["gather_ms"] == d["gather_ms"]


def test_bench_one_rank_rccl_group():
    """The RCCL branch itself on the one device there is: a one-rank nccl group, gather included."""
    r, lines = _bench(["--gpus", "1", "--force-dist", "--workload", "small", "--batch", "3", "--steps", "4", "--warmup", "2",
                       "--reps", "2", "--no-roofline-pass", "--cpu-iters", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = lines[0]
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["process_group"]["backend"] == "nccl"
    assert d["gathered"]["masks"] == 3 and d["gathered"]["verified_on_every_rank"] is True


def test_bench_refuses_more_gpus_than_the_box_has():
    import torch
    n = torch.cuda.device_count()
    r, lines = _bench(["--gpus", str(n + 1), "--steps", "2", "--warmup", "1"])
    assert r.returncode != 0 and lines == []
    assert f"only {n} GPU" in r.stderr


def test_bench_line_survives_a_gather_that_does_not_return():
    """The final all-gather runs after the line is built and under a watchdog: with no time at all for it (the two ranks
    are still creating the group when the timer fires) the line comes out anyway -- value from the timed regions, the gather
    fields replaced by the reason -- and the launch ends with exit code 0."""
    r, lines = _bench(["--gpus", "2", "--backend", "gloo", "--share-devices", "--steps", "4", "--warmup", "2", "--reps", "2",
                       "--no-roofline-pass", "--no-extra-pass", "--gather-timeout", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout
    d = lines[0]
    assert d["n_gpus"] == 2 and d["value"] > 0 and len(d["per_rank_its"]) == 2
    assert "did not return" in d["gathered"]["error"] and d["rccl_ranks"] == 0 and d["gather_ms"] is None
