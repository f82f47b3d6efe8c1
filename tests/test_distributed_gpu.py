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
