"""
The N > 1 path on CPU: two gloo ranks shard a batch of holograms, each runs its shard, and the
final phase masks are all-gathered.  The per-rank worker is injected (the CPU oracle stands in for
the HIP engine here: this test is about the sharding and the collective).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT


def _oracle_worker(shape, slm_shape, target, local_phases, method, maxiter, dtype=np.float32, **kw):
    from oracle import hgs_oracle as orc
    out = []
    for ph in local_phases:
        h = orc.OracleHologram(target, phase=ph, slm_shape=slm_shape, dtype=dtype)
        h.optimize(method, maxiter=maxiter, populate=False)
        out.append(h.phase.copy())
    return np.stack(out) if out else np.zeros((0,) + tuple(slm_shape), dtype=dtype)


def _run(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from slmsuite_amd import synth
    from slmsuite_amd.batch import optimize_batch_distributed, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shape, slm = (64, 64), (32, 48)
    target = synth.random_pixels_target(3, shape, 12)
    phases = np.stack([synth.seed_phase(70 + i, slm) for i in range(n)])
    res = optimize_batch_distributed(shape, slm, target, phases, "WGS-Leonardo", 4, compute=_oracle_worker)
    q.put((rank, res, shard_range(n, rank, world)))
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [4, 5])
def test_two_rank_shard_and_gather(n):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_run, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from slmsuite_amd import synth
    shape, slm = (64, 64), (32, 48)
    target = synth.random_pixels_target(3, shape, 12)
    phases = np.stack([synth.seed_phase(70 + i, slm) for i in range(n)])
    want = _oracle_worker(shape, slm, target, phases, "WGS-Leonardo", 4)
    ranges = sorted(g[2] for g in got)
    assert ranges[0][0] == 0 and ranges[-1][1] == n and ranges[0][1] == ranges[1][0]
    for rank, res, _ in got:
        assert res.shape == (n, 32, 48)
        np.testing.assert_array_equal(res, want)      # every rank holds every mask


def test_shard_range_partitions():
    from slmsuite_amd.batch import shard_range
    for n in (1, 7, 8, 64):
        for w in (1, 2, 4, 8):
            pieces = [shard_range(n, r, w) for r in range(w)]
            assert pieces[0][0] == 0 and pieces[-1][1] == n
            assert all(pieces[i][1] == pieces[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in pieces) - min(h - l for l, h in pieces) <= 1
