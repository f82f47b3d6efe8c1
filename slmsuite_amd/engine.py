"""
Thin NumPy-facing wrapper of one engine handle (include/hgs.h).  All compute happens in the HIP
library; this class only marshals buffers and flags.
"""
import ctypes as C

import numpy as np

from . import _lib as L

ALGORITHM_INDEX = {"GS": 0, "WGS-Leonardo": 1, "WGS-Kim": 2, "WGS-Nogrette": 3, "WGS-Wu": 4, "WGS-tanh": 5}
FEEDBACK_INDEX = {"computational": L.FB_PIXEL, "computational_spot": L.FB_SPOT_WINDOW,
                  "external_spot": L.FB_EXTERNAL}


class Engine:
    def __init__(self, shape, slm_shape, dtype=np.float32, batch=1, n_spots=0, device=0, kind=0, n_monomials=0):
        self.lib = L.load()
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise ValueError(f"Data type {dtype} not supported.")
        self.ctype = np.dtype(np.complex64 if self.dtype == np.float32 else np.complex128)
        self.kind = int(kind)
        self.device = int(device)
        self.slm_shape = (int(slm_shape[0]), int(slm_shape[1]))
        self.batch = int(batch)
        self.n_spots = int(n_spots)
        self.n_monomials = int(n_monomials)
        if self.kind == 1:      # CompressedSpotHologram: farfield-sized arrays are N-vectors
            self.shape = (self.n_spots,)
            pad = self.slm_shape
        else:
            self.shape = pad = (int(shape[0]), int(shape[1]))
        cfg = L.hgs_config(device=int(device), pad_h=pad[0], pad_w=pad[1],
                           slm_h=self.slm_shape[0], slm_w=self.slm_shape[1],
                           real_bytes=self.dtype.itemsize, batch=self.batch, n_spots=self.n_spots,
                           kind=self.kind, n_monomials=self.n_monomials)
        self._h = C.c_void_p()
        L.check(self.lib.hgs_create(C.byref(cfg), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.hgs_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- arrays ----------------------------------------------------------------------------
    def _np(self, which, arr):
        if which in (L.SPOT_INDEX, L.MONOMIALS):
            return np.ascontiguousarray(arr, dtype=np.int32)
        if which in (L.SPOT_AMP, L.EXTERNAL_AMP):
            return np.ascontiguousarray(arr, dtype=np.float64)
        if which in (L.FARFIELD, L.ZERO_WEIGHTS):
            return np.ascontiguousarray(arr, dtype=self.ctype)
        return np.ascontiguousarray(arr, dtype=self.dtype)

    def set(self, which, arr):
        a = self._np(which, arr)
        L.check(self.lib.hgs_set_array(self._h, which, a.ctypes.data_as(C.c_void_p), a.nbytes))

    def get(self, which):
        if which == L.PHASE:
            out = np.empty((self.batch,) + self.slm_shape, dtype=self.dtype)
        elif which in (L.FARFIELD, L.ZERO_WEIGHTS):
            out = np.empty((self.batch,) + self.shape, dtype=self.ctype)
        else:
            out = np.empty((self.batch,) + self.shape, dtype=self.dtype)
        L.check(self.lib.hgs_get_array(self._h, which, out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def try_get(self, which):
        """``get`` that returns None where the engine holds no such array yet (HGS_ERR_STATE) instead of raising."""
        out = np.empty((self.batch,) + (self.slm_shape if which == L.PHASE else self.shape),
                       dtype=self.ctype if which in (L.FARFIELD, L.ZERO_WEIGHTS) else self.dtype)
        code = self.lib.hgs_get_array(self._h, which, out.ctypes.data_as(C.c_void_p), out.nbytes)
        if code == L.HGS_ERR_STATE:
            return None
        L.check(code)
        return out

    def get_prev_phase(self):
        """HGS_PHASE_PREV (HGS_OPT_KEEP_PREV_PHASE): the phase the last one-iteration fused call started from, or None when
        the engine holds none (the general operators ran -- HGS_PHASE_FF itself is up to date then -- or nothing ran yet)."""
        out = np.empty((self.batch,) + self.slm_shape, dtype=self.dtype)
        code = self.lib.hgs_get_array(self._h, L.PHASE_PREV, out.ctypes.data_as(C.c_void_p), out.nbytes)
        if code == L.HGS_ERR_STATE:
            return None
        L.check(code)
        return out

    def get_into_device(self, which, dev_ptr, nbytes):
        L.check(self.lib.hgs_get_array_device(self._h, which, C.c_void_p(dev_ptr), nbytes))

    def set_from_device(self, which, dev_ptr, nbytes):
        """hgs_set_array_device: ``nbytes`` at device address ``dev_ptr`` (e.g. ``tensor.data_ptr()``), layouts as ``set``."""
        L.check(self.lib.hgs_set_array_device(self._h, which, C.c_void_p(dev_ptr), nbytes))

    def set_tensor(self, which, tensor):
        """A contiguous torch CUDA tensor of the engine's element type, device to device."""
        want = {4: "float32", 8: "float64"}[self.dtype.itemsize]
        if which in (L.FARFIELD, L.ZERO_WEIGHTS):
            want = {4: "complex64", 8: "complex128"}[self.dtype.itemsize]
        if str(tensor.dtype).split(".")[-1] != want:
            tensor = tensor.to(getattr(__import__("torch"), want))
        tensor = tensor.contiguous()
        # the engine copies on ITS stream: whatever torch still has in flight for this tensor (its producer, the
        # conversions above) must have landed first
        __import__("torch").cuda.current_stream(tensor.device).synchronize()
        self.set_from_device(which, tensor.data_ptr(), tensor.numel() * tensor.element_size())

    def clear_propagation_kernel(self):
        """hgs_set_array(HGS_PROP_KERNEL, nbytes = 0): no kernel."""
        L.check(self.lib.hgs_set_array(self._h, L.PROP_KERNEL, None, 0))

    def copy_phase_from(self, other):
        """hgs_copy_phase: this engine's phase <- ``other``'s, on the device."""
        L.check(self.lib.hgs_copy_phase(self._h, other._h))

    def get_tensor(self, which):
        """Array ``which`` as a torch tensor on the engine's GPU ([batch, ...], natural layout): no host copy."""
        import torch
        real = {4: torch.float32, 8: torch.float64}[self.dtype.itemsize]
        cplx = {4: torch.complex64, 8: torch.complex128}[self.dtype.itemsize]
        shape = (self.batch,) + (self.slm_shape if which == L.PHASE else self.shape)
        t = torch.empty(shape, dtype=cplx if which in (L.FARFIELD, L.ZERO_WEIGHTS) else real, device=torch.device("cuda", self.device))
        torch.cuda.current_stream(t.device).synchronize()     # (the block may be a recycled one torch is still reading)
        self.get_into_device(which, t.data_ptr(), t.numel() * t.element_size())
        return t

    def set_sparse(self, which, xy, values):
        """hgs_set_array_sparse: ``values[k]`` at pixel (kx = xy[0, k], ky = xy[1, k]), zero elsewhere."""
        xy = np.ascontiguousarray(xy, dtype=np.int32)
        v = np.ascontiguousarray(values, dtype=self.dtype)
        if xy.ndim != 2 or xy.shape[0] != 2 or xy.shape[1] != v.size:
            raise ValueError("sparse upload needs xy of shape (2, n) and n values")
        L.check(self.lib.hgs_set_array_sparse(self._h, which, xy.ctypes.data_as(C.POINTER(C.c_int32)),
                                              v.ctypes.data_as(C.c_void_p), int(v.size)))

    def reset_weights(self):
        L.check(self.lib.hgs_reset_weights(self._h))

    def reset(self):
        """hgs_reset: weights from the target, phase_ff / farfield / amp_ff forgotten (Hologram.reset)."""
        L.check(self.lib.hgs_reset(self._h))

    # -- operators ---------------------------------------------------------------------------
    def nearfield2farfield(self, store_phase_ff=False):
        L.check(self.lib.hgs_nearfield2farfield(self._h, int(bool(store_phase_ff))))

    def farfield_constraint(self, step):
        L.check(self.lib.hgs_farfield_constraint(self._h, C.byref(step)))

    def farfield2nearfield(self):
        L.check(self.lib.hgs_farfield2nearfield(self._h))

    def iterate(self, step, n_iter):
        hist = (C.c_uint8 * max(1, n_iter))()
        L.check(self.lib.hgs_iterate(self._h, C.byref(step), int(n_iter), hist))
        return [bool(hist[i]) for i in range(n_iter)]

    STAT_GROUPS = ("computational", "computational_spot")

    def iterate_stats(self, step, n_iter, groups, width=1, spot_xy=None):
        """
        hgs_iterate_stats: n_iter loop bodies plus the statistics recorded in each of them.
        ``groups``: iterable of "computational" / "computational_spot".  Returns
        (fixed_phase history, [per iteration {group: [per hologram stats dict]}]).
        """
        mask = sum(1 << self.STAT_GROUPS.index(g) for g in groups)
        hist = (C.c_uint8 * max(1, n_iter))()
        out = (C.c_double * max(1, n_iter * 2 * self.batch * 4))()
        xy = None
        if spot_xy is not None:
            xy_np = np.ascontiguousarray(spot_xy, dtype=np.float64)
            xy = xy_np.ctypes.data_as(C.POINTER(C.c_double))
        L.check(self.lib.hgs_iterate_stats(self._h, C.byref(step), int(n_iter), hist, int(mask), int(width), xy, out))
        res = np.array(out[:]).reshape(max(1, n_iter), 2, self.batch, 4)
        per_iter = []
        for i in range(n_iter):
            d = {}
            for gi, g in enumerate(self.STAT_GROUPS):
                if mask & (1 << gi):
                    d[g] = [dict(efficiency=float(r[0]), uniformity=float(r[1]), pkpk_err=float(r[2]),
                                 std_err=float(r[3])) for r in res[i, gi]]
            per_iter.append(d)
        return [bool(hist[i]) for i in range(n_iter)], per_iter

    def iterate_timed(self, step, n_iter):
        ms = C.c_double()
        L.check(self.lib.hgs_iterate_timed(self._h, C.byref(step), int(n_iter), C.byref(ms)))
        return ms.value

    def stats(self, group, width=1, spot_xy=None):
        out = (C.c_double * (4 * self.batch))()
        xy = None
        if spot_xy is not None:
            xy_np = np.ascontiguousarray(spot_xy, dtype=np.float64)
            xy = xy_np.ctypes.data_as(C.POINTER(C.c_double))
        L.check(self.lib.hgs_stats(self._h, int(group), int(width), xy, out))
        res = np.array(out[:]).reshape(self.batch, 4)
        return [dict(efficiency=float(r[0]), uniformity=float(r[1]), pkpk_err=float(r[2]), std_err=float(r[3]))
                for r in res]

    @staticmethod
    def multiplane_farfield2nearfield(engines, weights):
        """hgs_multiplane_farfield2nearfield: common phase of the children from their weighted complex nearfields."""
        n = len(engines)
        handles = (C.c_void_p * n)(*[e._h for e in engines])
        w = (C.c_double * n)(*[float(x) for x in weights])
        L.check(engines[0].lib.hgs_multiplane_farfield2nearfield(handles, w, n))

    def set_option(self, option, value):
        L.check(self.lib.hgs_set_option(self._h, int(option), int(value)))

    def sync(self):
        L.check(self.lib.hgs_sync(self._h))

    def profile_enable(self, on=True):
        L.check(self.lib.hgs_profile_enable(self._h, int(bool(on))))

    def profile_read(self):
        out = (C.c_double * (2 * len(L.K_NAMES)))()
        L.check(self.lib.hgs_profile_read(self._h, out))
        return {n: dict(ms=out[2 * i], launches=int(out[2 * i + 1])) for i, n in enumerate(L.K_NAMES)}

    def dispatch_read(self):
        """
        hgs_dispatch_read: the kernel instances launched since the previous call, as a list of
        ``{"kernel": family, "args": {template parameter: value}, "flags": set, "count": launches, "name": text}``.
        """
        need = C.c_size_t(0)
        L.check(self.lib.hgs_dispatch_read(self._h, None, 0, C.byref(need)))
        buf = C.create_string_buffer(max(1, need.value))
        L.check(self.lib.hgs_dispatch_read(self._h, buf, len(buf), None))
        out = []
        for line in buf.value.decode().splitlines():
            name, count = line.rsplit("\t", 1)
            head, _, flags = name.partition("> ")
            family, _, arglist = head.rstrip(">").partition("<")
            args = dict(kv.split("=", 1) for kv in arglist.split(",") if kv)
            out.append({"kernel": family, "args": args, "flags": set(flags.split()), "count": int(count), "name": name})
        return out

    def version(self):
        return self.lib.hgs_version().decode()


def make_step(flags, iteration, false_run=0, mraf_enabled=False, spot_window=3, efficiency_group=None):
    """
    POD image of Hologram.flags for one engine call (see hgs_step in include/hgs.h).  ``efficiency_group``
    (index into Engine.STAT_GROUPS) hands flags["fix_phase_efficiency"] to the engine for a device-resident loop
    with statistics; None leaves that gate with the caller (the stepwise path evaluates it on the host).
    """
    method = flags["method"]
    if method not in ALGORITHM_INDEX:
        raise ValueError(f"Unsupported optimization method '{method}'")
    fb = flags.get("feedback", "computational")
    if fb not in FEEDBACK_INDEX:
        raise NotImplementedError(
            f"Feedback '{fb}' needs camera hardware; the MI355X engine covers the computational and "
            "external_spot feedback modes")
    mf = flags.get("mraf_factor", None)
    zf = flags.get("zero_factor", 0)
    st = L.hgs_step()
    st.method = ALGORITHM_INDEX[method]
    st.feedback = FEEDBACK_INDEX[fb]
    st.iter = int(iteration)
    st.fixed_phase = int(bool(flags.get("fixed_phase", False)))
    st.fix_phase_iteration = int(flags.get("fix_phase_iteration", 10) or 0)
    st.false_run = int(false_run)
    st.mraf_enabled = int(bool(mraf_enabled))
    st.has_mraf_factor = int(mf is not None)
    st.zero_mode = int(bool(mraf_enabled) and ("zero_factor" in flags) and zf != 0)
    st.spot_window = int(spot_window)
    st.feedback_exponent = float(flags.get("feedback_exponent", 1.0) or 0.0)
    st.feedback_factor = float(flags.get("feedback_factor", 1.0) or 0.0)
    st.mraf_factor = float(mf) if mf is not None else float("nan")
    st.zero_factor = float(zf if zf is not None else 0.0)
    st.fix_phase_efficiency = float("nan")
    st.efficiency_group = 0
    st.reserved = 0
    if efficiency_group is not None and flags.get("fix_phase_efficiency", None) is not None and "Kim" in method:
        st.fix_phase_efficiency = float(flags["fix_phase_efficiency"])
        st.efficiency_group = int(efficiency_group)
    return st
