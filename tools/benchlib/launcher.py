"""Rank plumbing of bench.py --gpus N: the self-launch under torch.distributed.run and small collectives."""
import os
import subprocess
import sys


def _gather_ints(dist, torch, dev, value, world, group=None):
    t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    return [int(o.item()) for o in out]


def apply_opts(engine, opts):
    from slmsuite_amd import _lib as L
    for o in opts:
        name, val = o.split("=")
        engine.set_option(getattr(L, "OPT_" + name.upper()), int(val))


def self_launch(args, bench_path):
    """
    ``python bench.py --gpus N`` (N > 1) outside a launcher: re-run this very command line under
    ``python -m torch.distributed.run`` with one rank per GPU (the form the driver uses itself) and hand its exit code
    back.  Fails loudly -- non-zero, nothing printed on stdout -- when the box has fewer than N devices.
    """
    import socket
    import torch
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if getattr(args, "stub_engine", False):       # the CPU self-test of the rank protocol: no device behind the ranks
        n_dev = max(n_dev, 1)
    if n_dev < 1:
        print(f"bench.py --gpus {args.gpus}: no GPU visible (the engine has no CPU fallback)", file=sys.stderr)
        return 2
    if n_dev < args.gpus and not args.share_devices:
        print(f"bench.py --gpus {args.gpus}: only {n_dev} GPU(s) visible; refusing to report a {args.gpus}-GPU figure "
              f"from fewer devices", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(bench_path)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode
