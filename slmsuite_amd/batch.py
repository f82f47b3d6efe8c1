"""
Batches of independent holograms: one engine advances ``batch`` holograms of the same geometry
together, and a node shards a batch over its GPUs, one process per GPU.

The holograms never exchange data while iterating; the only collective is the final all-gather of
the phase masks over RCCL/xGMI (``torch.distributed`` backend "nccl" on ROCm), which moves
``n * S * r`` bytes once (70.8 MB for 8 x 1152x1920 fp32 masks) and is irrelevant to throughput.
PyTorch is used for device memory and the process group only.
"""
import numpy as np

from slmsuite_amd import _lib as L
from slmsuite_amd.engine import Engine, make_step
from slmsuite_amd.holography.algorithms import ALGORITHM_DEFAULTS


def batch_flags(method, **flags):
    """Flag dictionary as Hologram._update_flags builds it (_hologram.py:1370-1393)."""
    if method not in ALGORITHM_DEFAULTS or method == "CG":
        raise ValueError(f"Unrecognized method '{method}'")
    fl = dict(ALGORITHM_DEFAULTS[method])
    fl.update(flags)
    fl["method"] = method
    fl.setdefault("fixed_phase", False)
    return fl


def normalize_targets(target, dtype=np.float32):
    """
    Hologram._set_target (_hologram.py:741-760): target = |target| / sqrt(nansum(target^2)), per hologram.
    The additive rules (WGS-Wu, WGS-tanh) depend on the target scale, so an un-normalised image must not reach
    the engine.  Targets that already have unit norm (e.g. ``Hologram.target``) pass through bit-identically.
    """
    t = np.abs(np.asarray(target, dtype=dtype))
    flat = t.reshape((-1,) + t.shape[-2:]) if t.ndim >= 2 else t.reshape(1, -1)
    out = flat.copy()
    for i in range(flat.shape[0]):
        nrm = float(np.sqrt(np.nansum(np.square(flat[i].astype(np.float64)))))
        if nrm > 0 and abs(nrm - 1.0) > 1e-6:
            out[i] = flat[i] * np.dtype(dtype).type(1.0 / nrm)
    return out.reshape(t.shape).astype(dtype, copy=False)


class EngineGroup:
    """The engines of a HologramBatch addressed as one (options, synchronisation, per-kernel event timing summed)."""

    def __init__(self, engines):
        self.engines = list(engines)

    def set_option(self, option, value):
        for e in self.engines:
            e.set_option(option, value)

    def sync(self):
        for e in self.engines:
            e.sync()

    def profile_enable(self, on=True):
        for e in self.engines:
            e.profile_enable(on)

    def profile_read(self):
        total = None
        for e in self.engines:
            one = e.profile_read()
            if total is None:
                total = one
            else:
                for k, v in one.items():
                    total[k]["ms"] += v["ms"]
                    total[k]["launches"] += v["launches"]
        return total

    def dispatch_read(self):
        return [r for e in self.engines for r in e.dispatch_read()]

    def version(self):
        return self.engines[0].version()

    # what callers of a single-engine batch use on ``HologramBatch.engine``: fanned out over the groups in hologram order
    def get(self, which):
        """Array ``which`` of every hologram, [n, ...] (the groups' batches concatenated; shared arrays from the first)."""
        if which in (L.AMP, L.AMP_SCALAR, L.PROP_KERNEL, L.SPOT_INDEX, L.SPOT_AMP, L.EXTERNAL_AMP):
            return self.engines[0].get(which)
        return np.concatenate([e.get(which) for e in self.engines], axis=0)

    def nearfield2farfield(self, store_phase_ff=False):
        for e in self.engines:
            e.nearfield2farfield(store_phase_ff)

    def farfield2nearfield(self):
        for e in self.engines:
            e.farfield2nearfield()

    def iterate(self, step, n_iter):
        """``n_iter`` bodies on every group from copies of ``step``; the flag history (the same for all) of the first, and
        ``step`` left as the first group's call leaves it."""
        import copy
        hist = None
        for k, e in enumerate(self.engines):
            st = step if k == 0 else copy.copy(step)
            h = e.iterate(st, n_iter)
            hist = h if hist is None else hist
        return hist

    def __getattr__(self, name):
        raise AttributeError(f"EngineGroup has no '{name}': a HologramBatch with several stream groups holds one engine per "
                             "group (HologramBatch.engines); address them one by one or use the batch's own methods")


class HologramBatch:
    """
    ``batch`` holograms sharing geometry / amp (and optionally target) on one GPU.

    ``streams`` > 1 splits them over that many engines (contiguous groups, one HIP stream each): launches of different
    groups overlap on the device -- the VALU-bound column launch of one group runs under the latency-bound row launch of
    another (cfg 3, eight holograms at 4096^2: +5 % with two groups, +6.5 % with four; ``tools/two_stream_probe.py``).
    Every hologram's result is that of a single-engine batch (the kernels never mix holograms).
    """

    def __init__(self, shape, slm_shape, target, phases, dtype=np.float32, amp=None,
                 propagation_kernel=None, spot_index=None, spot_amp=None, device=0, streams=1):
        phases = np.asarray(phases, dtype=dtype)
        self.n = phases.shape[0]
        self.slm_shape = tuple(int(v) for v in slm_shape)
        groups = max(1, min(int(streams), self.n))
        self.bounds = [shard_range(self.n, k, groups) for k in range(groups)]
        tg = np.asarray(target)
        self.engines = []
        for lo, hi in self.bounds:
            e = Engine(shape, slm_shape, dtype, batch=hi - lo,
                       n_spots=0 if spot_index is None else np.shape(spot_index)[1], device=device)
            self.engines.append(e)
            if amp is None:
                e.set(L.AMP_SCALAR, np.array([1 / np.sqrt(np.prod(slm_shape))], dtype=dtype))
            else:
                a = np.array(amp, dtype=dtype)
                e.set(L.AMP, a * (1 / np.sqrt(np.nansum(np.square(a)))))
            if propagation_kernel is not None:
                e.set(L.PROP_KERNEL, propagation_kernel)
            # one target broadcasts to the whole batch; per-hologram targets go to their group
            e.set(L.TARGET, normalize_targets(tg[lo:hi] if tg.ndim == 3 else tg, dtype))
            e.reset_weights()
            e.set(L.PHASE, phases[lo:hi])
            if spot_index is not None:
                e.set(L.SPOT_INDEX, spot_index)
                e.set(L.SPOT_AMP, spot_amp)
                e.set(L.EXTERNAL_AMP, spot_amp)
        self.engine = self.engines[0] if groups == 1 else EngineGroup(self.engines)
        self.iter = 0
        self.mraf = bool(np.isnan(np.sum(target)))
        self.false_run = 0
        self.flags = None

    def set_option(self, option, value):
        for e in self.engines:
            e.set_option(option, value)

    def sync(self):
        for e in self.engines:
            e.sync()

    def _steps(self, spot_window):
        return [make_step(self.flags, self.iter, false_run=self.false_run, mraf_enabled=self.mraf, spot_window=spot_window)
                for _ in self.engines]

    def _advance(self, steps):
        st = steps[0]                              # (every group walks the same flag history)
        self.iter, self.false_run = st.iter, st.false_run
        self.flags["fixed_phase"] = bool(st.fixed_phase)

    def optimize(self, method="WGS-Leonardo", maxiter=50, spot_window=3, **flags):
        if self.flags is None:
            self.flags = batch_flags(method, **flags)
        else:
            kept = {k: v for k, v in self.flags.items() if k != "method"}      # flags persist between calls, like Hologram.flags
            self.flags = batch_flags(method, **{**kept, **flags})
        steps = self._steps(spot_window)
        for e, st in zip(self.engines, steps):     # hgs_iterate only enqueues: the groups' launches interleave on the device
            e.iterate(st, maxiter)
        self._advance(steps)
        return self

    def time_iterations(self, method, n_iter, spot_window=3, **flags):
        """Milliseconds for n_iter loop bodies of every hologram: HIP events on the engine stream with one
        group; with several, host clock from the first enqueue to the last stream's completion."""
        if self.flags is None:
            self.flags = batch_flags(method, **flags)
        else:
            kept = {k: v for k, v in self.flags.items() if k != "method"}      # (as optimize(): another method on the next call)
            self.flags = batch_flags(method, **{**kept, **flags})
        steps = self._steps(spot_window)
        if len(self.engines) == 1:
            ms = self.engines[0].iterate_timed(steps[0], n_iter)
        else:
            import time
            self.sync()
            t0 = time.perf_counter()
            for e, st in zip(self.engines, steps):
                e.iterate(st, n_iter)
            self.sync()
            ms = (time.perf_counter() - t0) * 1e3
        self._advance(steps)
        return ms

    def run_groups_one_by_one(self, method, n_iter, spot_window=3, **flags):
        """``n_iter`` loop bodies of every hologram with the stream groups run to completion one after the other: per-kernel
        event timing (``engine.profile_enable``) then sees every launch alone, not stretched by a neighbour's."""
        if self.flags is None:
            self.flags = batch_flags(method, **flags)
        steps = self._steps(spot_window)
        for e, st in zip(self.engines, steps):
            e.iterate(st, n_iter)
            e.sync()
        self._advance(steps)

    def phases(self):
        return np.concatenate([e.get(L.PHASE) for e in self.engines], axis=0)

    def phases_into_device(self, dev_ptr, nbytes):
        """Copy the [n, Sh, Sw] phase masks into caller-owned DEVICE memory (e.g. a torch tensor handed to RCCL)."""
        one = int(np.prod(self.slm_shape)) * self.engines[0].dtype.itemsize
        if nbytes != self.n * one:
            raise ValueError(f"phases_into_device: expected {self.n * one} bytes, got {nbytes}")
        for e, (lo, hi) in zip(self.engines, self.bounds):
            e.get_into_device(L.PHASE, dev_ptr + lo * one, (hi - lo) * one)

    def close(self):
        for e in self.engines:
            e.close()


def optimize_batch(shape, slm_shape, target, phases, method="WGS-Leonardo", maxiter=50,
                   dtype=np.float32, device=0, **kw):
    """Single-GPU convenience: returns the [n, Sh, Sw] phase masks after ``maxiter`` iterations."""
    ctor = {k: kw.pop(k) for k in ("amp", "propagation_kernel", "spot_index", "spot_amp") if k in kw}
    hb = HologramBatch(shape, slm_shape, target, phases, dtype=dtype, device=device, **ctor)
    try:
        hb.optimize(method, maxiter, **kw)
        return hb.phases()
    finally:
        hb.close()


def shard_range(n_items, rank, world_size):
    """Contiguous block partition [lo, hi) of n_items holograms for ``rank``."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def optimize_batch_distributed(shape, slm_shape, target, phases, method="WGS-Leonardo", maxiter=50,
                               dtype=np.float32, compute=None, device=None, **kw):
    """
    Shard ``phases`` ([n, Sh, Sw]) over the ranks of the default process group, optimise each
    shard locally, and all-gather the final phase masks so every rank returns the full [n, Sh, Sw].

    ``compute(shape, slm_shape, target, local_phases, method, maxiter, **kw) -> local result`` is the
    per-rank worker; it defaults to the HIP engine on this rank's GPU.  (CPU-only tests inject a
    stand-in so the sharding and the collective are exercised under gloo.)
    """
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()
    phases = np.asarray(phases, dtype=dtype)
    n = phases.shape[0]
    lo, hi = shard_range(n, rank, world)
    use_cuda = dist.get_backend() == "nccl"
    # equal-sized contributions for all_gather: pad to the largest shard
    per = (n + world - 1) // world
    if compute is None and use_cuda:
        # the engine's phase masks go device -> device into the tensor RCCL sends: no host bounce
        if device is None:
            device = torch.cuda.current_device()
        tdt = torch.float32 if np.dtype(dtype) == np.float32 else torch.float64
        t = torch.zeros((per,) + tuple(slm_shape), dtype=tdt, device=torch.device("cuda", device))
        if hi > lo:
            ctor = {k: kw.pop(k) for k in ("amp", "propagation_kernel", "spot_index", "spot_amp") if k in kw}
            tg = np.asarray(target)
            hb = HologramBatch(shape, slm_shape, tg[lo:hi] if tg.ndim == 3 else tg, phases[lo:hi], dtype=dtype,
                               device=device, **ctor)
            try:
                hb.optimize(method, maxiter, **kw)
                torch.cuda.synchronize(device)
                hb.phases_into_device(t.data_ptr(), (hi - lo) * int(np.prod(slm_shape)) * np.dtype(dtype).itemsize)
            finally:
                hb.close()
    else:
        if compute is None:
            if device is None:      # this rank's GPU, as on the RCCL branch (a gloo job may still own one GPU per rank)
                device = torch.cuda.current_device() if torch.cuda.is_available() else 0
            compute = lambda *a, **k: optimize_batch(*a, device=device, **k)   # noqa: E731
        tg = np.asarray(target)
        local_target = tg[lo:hi] if tg.ndim == 3 else target      # per-hologram targets shard with the phases
        local = np.asarray(compute(shape, slm_shape, local_target, phases[lo:hi], method, maxiter, dtype=dtype, **kw),
                           dtype=dtype) if hi > lo else np.zeros((0,) + tuple(slm_shape), dtype=dtype)
        buf = np.zeros((per,) + tuple(slm_shape), dtype=dtype)
        buf[: hi - lo] = local
        t = torch.from_numpy(buf)
        if use_cuda:
            t = t.cuda()
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    res = np.empty((n,) + tuple(slm_shape), dtype=dtype)
    for r in range(world):
        l2, h2 = shard_range(n, r, world)
        res[l2:h2] = out[r][: h2 - l2].cpu().numpy()
    return res
