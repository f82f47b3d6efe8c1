"""
CPU oracle for the GS / WGS hologram optimisation hot path.

**This file is TEST INFRASTRUCTURE, not product code.**  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; nothing under ``slmsuite_amd/`` does (the product path fails loudly when
the HIP library is missing, it never falls back to this file).

It is an independent NumPy restatement of the reference algorithm in
``slmsuite/holography/algorithms`` (paths relative to the reference checkout);
every function cites the reference lines it follows.  It deliberately issues
the same *unfused* NumPy operation sequence as the reference (separate ufunc
passes, ``np.fft.fft2`` + ``fftshift``), so that (i) its floating-point results
track the reference to rounding and (ii) its wall-clock is a fair stand-in for
the reference NumPy path when timed as ``cpu_baseline`` (kind = "port").

Pinning: ``tests/test_oracle_golden.py`` checks this oracle against the golden
vectors in ``tests/golden/*.npz``, which were produced by importing the real
reference (``tools/make_golden.py``; the reference cannot travel to the GPU
box, the fixtures do).
"""
import numpy as np

# Method tables -- _header.py:53-81 (order defines ALGORITHM_INDEX).
ALGORITHM_DEFAULTS = {
    "GS": {"feedback": "computational"},
    "WGS-Leonardo": {"feedback": "computational", "feedback_exponent": 0.8},
    "WGS-Kim": {
        "feedback": "computational",
        "fix_phase_efficiency": None,
        "fix_phase_iteration": 10,
        "feedback_exponent": 0.8,
    },
    "WGS-Nogrette": {"feedback": "computational", "feedback_factor": 0.1},
    "WGS-Wu": {"feedback": "computational", "feedback_exponent": 0.5},
    "WGS-tanh": {"feedback": "computational", "feedback_factor": 0.2, "feedback_exponent": 0.5},
}
FEEDBACK_OPTIONS = (
    "computational", "computational_spot", "experimental", "experimental_spot", "external_spot",
)


# --------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------
def unpad_slices(shape, slm_shape):
    """Centred crop indices (r0, r1, c0, c1).  toolbox/__init__.py:1699-1712."""
    dh = (shape[0] - slm_shape[0]) / 2.0
    dw = (shape[1] - slm_shape[1]) / 2.0
    if dh < 0 or dw < 0:
        raise ValueError("slm_shape larger than shape")
    return (
        int(np.floor(dh)), int(shape[0] - np.ceil(dh)),
        int(np.floor(dw)), int(shape[1] - np.ceil(dw)),
    )


def padded_shape(slm_shape, padding_order=1, square_padding=True):
    """Power-of-two padding rule.  _hologram.py:712-725 (precision=inf branch)."""
    if padding_order > 0:
        shp = np.power(2, np.ceil(np.log2(slm_shape)) + padding_order - 1).astype(int)
    else:
        shp = np.asarray(slm_shape)
    shp = tuple(int(s) for s in shp)
    if square_padding:
        shp = (max(shp), max(shp))
    return shp


def l2norm(x):
    """sqrt(nansum(|x|^2)).  _hologram.py:1979-2011."""
    if np.iscomplexobj(x):
        return np.sqrt(np.nansum(np.square(np.abs(x))))
    return np.sqrt(np.nansum(np.square(x)))


def take_sum(image, vectors, width):
    """
    Sum of ``width x width`` windows centred on floor(vectors) (float64 accumulation).
    analysis/__init__.py:61-204 with centered=True, integrate=True, clip=False;
    window offsets are floor(arange(w) - (w-1)/2)  (analysis._coordinates).
    """
    v = np.floor(np.asarray(vectors, dtype=float).reshape(2, -1)).astype(int)
    off = np.floor(np.arange(width) - (width - 1) / 2.0).astype(int)
    ox, oy = np.meshgrid(off, off)
    ix = ox.ravel()[None, :] + v[0][:, None]
    iy = oy.ravel()[None, :] + v[1][:, None]
    return np.sum(image[iy, ix].astype(float), axis=-1)


def calculate_stats(feedback_amp, target_amp, total=None, alias=True):
    """
    efficiency / uniformity / pkpk_err / std_err.  _stats.py:7-116 with
    efficiency_compensation=False (the only value the computational groups use,
    _stats.py:123-128, _spots.py:1634-1679).

    ``alias=True`` reproduces quirk A23: the reference wraps its inputs with ``copy=None``
    (_stats.py:51-52) and then rescales them IN PLACE (:65, :69), so requesting the
    "computational" group nudges ``amp_ff`` and ``target`` by a factor within 1 ulp of 1
    every iteration.  The golden fixtures were recorded with stats on, so the oracle keeps the
    aliasing to stay bit-faithful; ``alias=False`` computes on copies (what the HIP engine does).
    """
    f_amp = np.array(feedback_amp, copy=None if alias else True)
    t_amp = np.array(target_amp, copy=None if alias else True)
    f_pwr = np.square(f_amp)
    t_pwr = np.square(t_amp)
    if total is not None:
        efficiency = np.nansum(f_pwr) / total
    s = np.sum(f_pwr)
    f_pwr *= 1 / s
    f_amp *= 1 / np.sqrt(s)
    ts = np.nansum(t_pwr)
    t_pwr *= 1 / ts
    t_amp *= 1 / np.sqrt(ts)
    if total is None:
        efficiency = np.square(float(np.nansum(np.multiply(t_amp, f_amp))))
    mask = np.logical_and(t_pwr != 0, np.logical_not(np.isnan(t_pwr)))
    fm = f_pwr[mask]
    tm = t_pwr[mask]
    ratio = fm / tm
    err = tm - fm
    rmin, rmax = float(np.amin(ratio)), float(np.amax(ratio))
    return {
        "efficiency": float(efficiency),
        "uniformity": float(1 - (rmax - rmin) / (rmax + rmin)),
        "pkpk_err": float(err.size * float(np.amax(err) - np.amin(err))),
        "std_err": float(err.size * float(np.std(err))),
    }


# --------------------------------------------------------------------------------------
# the weight update (rows 9-10 of SURVEY 8a)
# --------------------------------------------------------------------------------------
def update_weights_generic(weight_amp, feedback_amp, target_amp, method, flags, dtype):
    """
    In-place WGS weight update.  _hologram.py:1822-1879 (nan_checks=True).
    ``method`` is the full name ("WGS-Leonardo", ...).
    """
    m = method.lower()
    if m[:4] != "wgs-":
        raise ValueError("Weighting is only for WGS.")
    m = m[4:]

    fc = np.array(feedback_amp, copy=True, dtype=dtype)
    fc *= 1 / l2norm(fc)
    tgt = np.asarray(target_amp)

    if "wu" in m or "tanh" in m:            # additive rules: target - p*feedback  (:1833-1835)
        fc *= -flags["feedback_exponent"]
        fc += tgt
    else:                                   # multiplicative rules (:1837-1843)
        with np.errstate(divide="ignore", invalid="ignore"):
            np.divide(fc, tgt, out=fc)
        fc[fc == np.inf] = 1
        fc[tgt == 0] = 1
        np.nan_to_num(fc, copy=False, nan=1)

    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        if "leonardo" in m or "kim" in m:   # (:1846-1848)
            np.power(fc, -flags["feedback_exponent"], out=fc)
        elif "nogrette" in m:               # (:1849-1855)
            fc *= -(1 / np.nanmean(fc))
            fc += 1
            fc *= -flags["feedback_factor"]
            fc += 1
            np.reciprocal(fc, out=fc)
        elif "wu" in m:                     # (:1856-1857)  exponent applied a second time
            fc = np.exp(flags["feedback_exponent"] * fc)
        elif "tanh" in m:                   # (:1858-1860)
            fc = flags["feedback_factor"] * np.tanh(flags["feedback_exponent"] * fc)
            fc += 1
        else:
            raise ValueError(f"Method '{method}' not recognized")

        fc[fc == np.inf] = 1
        weight_amp *= fc
    np.nan_to_num(weight_amp, copy=False, nan=0.0001)
    weight_amp *= 1 / l2norm(weight_amp)
    return weight_amp


# --------------------------------------------------------------------------------------
# hologram state + loop
# --------------------------------------------------------------------------------------
class OracleHologram:
    """
    State of one DFT-grid hologram and the GS/WGS loop.

    Constructor conventions follow Hologram.__init__ (_hologram.py:196-439):
    ``target`` is an array of the padded shape or an ``(h, w)`` tuple; ``amp=None`` means the
    *scalar* 1/sqrt(S) (float64, :401-402); an array amp is L2-normalised; ``phase`` must be
    given (the reference default is an unseeded RNG, quirk A14).
    """

    def __init__(self, target, amp=None, phase=None, slm_shape=None, dtype=np.float32,
                 propagation_kernel=None, **flags):
        self.dtype = np.float32 if np.dtype(dtype).itemsize == 4 else np.float64
        self.ctype = np.complex64 if self.dtype is np.float32 else np.complex128
        if isinstance(target, tuple) or (np.ndim(target) == 1 and len(target) == 2):
            self.shape = (int(target[0]), int(target[1]))
            target = None
        else:
            self.shape = tuple(np.shape(target))
        if slm_shape is None:
            if amp is not None:
                slm_shape = np.shape(amp)
            elif phase is not None:
                slm_shape = np.shape(phase)
            else:
                slm_shape = self.shape
        self.slm_shape = (int(slm_shape[0]), int(slm_shape[1]))

        if amp is None:
            self.amp = 1 / np.sqrt(np.prod(self.slm_shape))           # np.float64 scalar
        else:
            self.amp = np.array(amp, dtype=self.dtype)
            self.amp *= 1 / l2norm(self.amp)
        self.propagation_kernel = (
            None if propagation_kernel is None else np.array(propagation_kernel, dtype=self.dtype)
        )
        self.flags = dict(flags)
        self.set_target(target)
        if phase is None:
            raise ValueError("oracle runs need an explicit seed phase")
        self.phase = np.array(phase, dtype=self.dtype)
        self.reset()

    # -- _set_target :741-769, reset_weights :603-614, reset :442-478 ----------------------
    def set_target(self, target, reset_weights=False):
        if target is None:
            self.target = np.zeros(self.shape, dtype=self.dtype)
        else:
            self.target = np.array(target, dtype=self.dtype)
            np.abs(self.target, out=self.target)
            self.target *= 1 / l2norm(self.target)
        if reset_weights:
            self.reset_weights()

    def reset_weights(self):
        self.weights = self.target.copy()
        if hasattr(self, "zero_weights"):
            self.zero_weights *= 0
        np.nan_to_num(self.weights, copy=False, nan=0)

    def reset(self):
        self.reset_weights()
        self.iter = 0
        self.stats = {"method": [], "flags": {}, "stats": {}}
        self.amp_ff = None
        self.phase_ff = None
        self.nearfield = np.zeros(self.shape, dtype=self.ctype)
        self.farfield = np.zeros(self.shape, dtype=self.ctype)

    # -- operators ------------------------------------------------------------------------
    def build_nearfield(self):
        """_hologram.py:1000-1011."""
        r0, r1, c0, c1 = unpad_slices(self.shape, self.slm_shape)
        self.nearfield.fill(0)
        if self.propagation_kernel is None:
            self.nearfield[r0:r1, c0:c1] = self.amp * np.exp(1j * self.phase)
        else:
            self.nearfield[r0:r1, c0:c1] = self.amp * np.exp(1j * (self.phase + self.propagation_kernel))
        return self.nearfield

    def nearfield2farfield(self):
        """_hologram.py:1038-1056 + _midloop_cleaning :951-953."""
        nf = self.build_nearfield()
        self.farfield = np.fft.fftshift(np.fft.fft2(np.fft.fftshift(nf), norm="ortho"))
        self.amp_ff = np.abs(self.farfield, out=self.amp_ff)

    def farfield2nearfield(self):
        """_hologram.py:1058-1073 + _nearfield_extract :1026-1036."""
        self.nearfield = np.fft.ifftshift(np.fft.ifft2(np.fft.ifftshift(self.farfield), norm="ortho"))
        r0, r1, c0, c1 = unpad_slices(self.shape, self.slm_shape)
        self.phase = np.arctan2(
            self.nearfield.imag[r0:r1, c0:c1], self.nearfield.real[r0:r1, c0:c1], out=self.phase
        )
        if self.propagation_kernel is not None:
            self.phase -= self.propagation_kernel

    def inverse_nearfield(self):
        """_farfield2nearfield(extract=False) (:1058-1073): complex nearfield cropped to the SLM."""
        self.nearfield = np.fft.ifftshift(np.fft.ifft2(np.fft.ifftshift(self.farfield), norm="ortho"))
        r0, r1, c0, c1 = unpad_slices(self.shape, self.slm_shape)
        return self.nearfield[r0:r1, c0:c1]

    def populate_results(self):
        """_hologram.py:934-949."""
        self.nearfield2farfield()
        self.phase_ff = np.arctan2(self.farfield.imag, self.farfield.real, out=self.phase_ff)

    def update_weights(self):
        """Hologram._update_weights, _hologram.py:1914-1922."""
        if self.flags["feedback"] == "computational":
            update_weights_generic(self.weights, self.amp_ff, self.target,
                                   self.flags["method"], self.flags, self.dtype)

    # -- stats dictionary bookkeeping (drives WGS-Kim) ---------------------------------------
    def compute_stats(self, stat_groups):
        stats = {}
        if "computational" in stat_groups:          # _stats.py:118-128
            stats["computational"] = calculate_stats(self.amp_ff, self.target)
        return stats

    def update_stats(self, stat_groups):
        """_stats.py:130-190 (raw_stats omitted)."""
        stats = self.compute_stats(stat_groups)
        it = self.iter
        M = len(self.stats["method"])
        if it + 1 > M:
            self.stats["method"].extend([""] * (it + 1 - M))
            M = it + 1
        self.stats["method"][it] = self.flags["method"]
        for flag in set(self.flags) | set(self.stats["flags"]):
            lst = self.stats["flags"].setdefault(flag, [np.nan] * M)
            if it + 1 > len(lst):
                lst.extend([np.nan] * (it + 1 - len(lst)))
            if flag in self.flags:
                lst[it] = self.flags[flag]
        groups = set(stats) | set(self.stats["stats"])
        if groups:
            names = set()
            for g in stats:
                names |= set(stats[g])
            if self.stats["stats"]:
                names |= set(self.stats["stats"][next(iter(self.stats["stats"]))])
            for g in groups:
                d = self.stats["stats"].setdefault(g, {})
                for n in names:
                    lst = d.setdefault(n, [np.nan] * M)
                    if it + 1 > len(lst):
                        lst.extend([np.nan] * (it + 1 - len(lst)))
                    if g in stats and n in stats[g]:
                        lst[it] = stats[g][n]

    # -- the farfield routines (row 8) ---------------------------------------------------------
    def mraf_masks(self):
        """_hologram.py:1495-1548."""
        if not np.isnan(np.sum(self.target)):
            return None
        noise = np.isnan(self.target)
        zero = np.abs(self.target) == 0
        if self.flags.get("zero_factor", 0) != 0:
            Z = int(np.sum(zero))
            if Z > 0 and not hasattr(self, "zero_weights"):
                self.zero_weights = np.zeros((Z,), dtype=self.ctype)
        signal = np.logical_not(np.logical_or(noise, zero))
        return {"noise": noise, "zero": zero, "signal": signal}

    def gs_farfield_routines(self, masks):
        """_hologram.py:1550-1653."""
        fl = self.flags
        if "WGS" in fl["method"] and self.iter > 0:
            self.update_weights()
            if "Kim" in fl["method"]:
                was_not_fixed = not fl["fixed_phase"]
                if fl["fix_phase_efficiency"] is not None:
                    st = self.stats["stats"]
                    if len(st) == 0:
                        raise ValueError("Must track statistics to fix phase based on efficiency!")
                    eff = st[tuple(st.keys())[-1]]["efficiency"][self.iter]
                    if eff > fl["fix_phase_efficiency"]:
                        fl["fixed_phase"] = True
                if was_not_fixed and self.iter >= fl["fix_phase_iteration"] - 1:
                    prev = self.stats["flags"]["fixed_phase"]
                    if all(not prev[-1 - i] for i in range(fl["fix_phase_iteration"])):
                        fl["fixed_phase"] = True
                if (fl["fixed_phase"] and self.phase_ff is None) or was_not_fixed:
                    self.phase_ff = np.arctan2(self.farfield.imag, self.farfield.real, out=self.phase_ff)
            else:
                fl["fixed_phase"] = False

        fixed = bool(fl.get("fixed_phase", False))
        if masks is None:                           # :1590-1605
            if (not fixed) or self.phase_ff is None:
                self.phase_ff = np.arctan2(self.farfield.imag, self.farfield.real, out=self.phase_ff)
            np.exp(1j * self.phase_ff, out=self.farfield)
            np.multiply(self.farfield, self.weights, out=self.farfield)
        else:                                       # MRAF :1606-1653
            zero, noise, signal = masks["zero"], masks["noise"], masks["signal"]
            mraf_factor = fl.get("mraf_factor", None)
            if hasattr(self, "zero_weights"):
                fz = self.farfield[zero]
                self.zero_weights -= fl.get("zero_factor", 1) * np.abs(fz) * fz
                self.farfield[zero] = self.zero_weights
            else:
                self.farfield[zero] = 0
            if not fixed:
                self.phase_ff = np.arctan2(self.farfield.imag, self.farfield.real, out=self.phase_ff)
            np.exp(1j * self.phase_ff, where=signal, out=self.farfield)
            np.multiply(self.farfield, self.weights, where=signal, out=self.farfield)
            if mraf_factor is not None:
                np.multiply(self.farfield, mraf_factor, where=noise, out=self.farfield)

    # -- optimize (rows 2-3) ---------------------------------------------------------------------
    def update_flags(self, method, feedback, stat_groups, **kwargs):
        """_hologram.py:1370-1410."""
        if method not in ALGORITHM_DEFAULTS:
            raise ValueError(f"Unrecognized method '{method}'")
        self.flags["method"] = method
        for k, v in ALGORITHM_DEFAULTS[method].items():
            if k not in self.flags:
                self.flags[k] = v
        if "fixed_phase" not in self.flags:
            self.flags["fixed_phase"] = False
        for k in kwargs:
            self.flags[k] = kwargs[k]
        for g in stat_groups:
            if g not in FEEDBACK_OPTIONS:
                raise ValueError(f"Statistics group '{g}' not recognized")
        self.flags["stat_groups"] = stat_groups
        if feedback is not None:
            if feedback not in FEEDBACK_OPTIONS:
                raise ValueError(f"Feedback '{feedback}' not recognized")
            self.flags["feedback"] = feedback

    def optimize(self, method="GS", maxiter=20, callback=None, feedback=None, stat_groups=(),
                 populate=True, trace=None, **kwargs):
        """
        _hologram.py:1076-1368 + optimize_gs :1427-1493.  ``trace`` (optional callable) is
        invoked as trace(self, stage) after each operator for step-level fixtures.
        """
        kwargs.pop("name", None)
        self.update_flags(method, feedback, list(stat_groups), **kwargs)
        masks = self.mraf_masks()
        for _ in range(maxiter):
            self.nearfield2farfield()
            if trace is not None:
                trace(self, "forward")
            if callback is not None and callback(self):
                break
            self.update_stats(self.flags["stat_groups"])
            self.gs_farfield_routines(masks)
            if trace is not None:
                trace(self, "constraint")
            self.farfield2nearfield()
            if trace is not None:
                trace(self, "inverse")
            self.iter += 1
        if populate:
            self.populate_results()


class OracleMultiplaneHologram:
    """
    Composite of several oracle holograms sharing one phase mask (MultiplaneHologram,
    _multiplane.py:8-289): per iteration every child transforms forward (:245-249), records its
    own statistics (:224-226) and applies its own constraint (:281-283); the children's complex
    nearfields over the SLM, each with its propagation kernel removed, are summed with the
    L2-normalised weights and the common phase is the argument of the sum (:251-276).
    """

    def __init__(self, holograms, weights=None):
        self.holograms = list(holograms)
        h0 = self.holograms[0]
        self.dtype, self.ctype, self.slm_shape = h0.dtype, h0.ctype, h0.slm_shape
        self.amp = h0.amp
        self.phase = h0.phase
        for h in self.holograms:                       # shared data (:73-76)
            h.amp = self.amp
            h.phase = self.phase
        if weights is None:
            weights = np.ones(len(self.holograms), dtype=self.dtype)
        self.weights = np.array(weights, dtype=self.dtype)
        self.weights /= l2norm(self.weights)
        self.nearfield = np.zeros(self.slm_shape, dtype=self.ctype)
        self.flags = {}
        self.iter = 0

    def optimize(self, method="GS", maxiter=20, callback=None, feedback=None, stat_groups=(), **kwargs):
        self.holograms[0].update_flags(method, feedback, list(stat_groups), **kwargs)
        self.flags = dict(self.holograms[0].flags)
        for h in self.holograms:                       # _update_flags :176-182
            h.flags.update(self.flags)
        masks = [h.mraf_masks() for h in self.holograms]
        for _ in range(maxiter):
            for h in self.holograms:
                h.nearfield2farfield()
                h.iter = self.iter
            if callback is not None and callback(self):
                break
            for h in self.holograms:
                h.update_stats(self.flags["stat_groups"])
            for h, m in zip(self.holograms, masks):
                h.gs_farfield_routines(m)
            self.nearfield.fill(0)
            for h, w in zip(self.holograms, self.weights):
                nf = h.inverse_nearfield()
                if h.propagation_kernel is None:
                    self.nearfield += w * nf
                else:
                    self.nearfield += w * nf * np.exp(-1j * h.propagation_kernel)
                h.iter = self.iter
            # one shared array: every child sees the new phase (:1026-1036 on the parent)
            np.arctan2(self.nearfield.imag, self.nearfield.real, out=self.phase)
            self.iter += 1
        for h in self.holograms:                       # _populate_results through the overload
            h.nearfield2farfield()
            h.iter = self.iter


class OracleSpotHologram(OracleHologram):
    """
    DFT-grid spot array with one pixel per spot (basis "knm", no camera).
    SpotHologram.__init__ _spots.py:1090-1373; _set_target_spots :1490-1546;
    _update_weights :1573-1624; _calculate_stats_computational_spot :1626-1679.
    """

    def __init__(self, shape, spot_vectors, spot_amp=None, slm_shape=None, phase=None,
                 amp=None, dtype=np.float32, null_vectors=None, null_radius=None, null_region=None,
                 null_region_radius_frac=None, **flags):
        v = np.asarray(spot_vectors, dtype=float).reshape(2, -1)
        N = v.shape[1]
        self.spot_knm = v
        self.spot_amp = np.full(N, 1.0 / np.sqrt(N)) if spot_amp is None else np.ravel(spot_amp)
        self.external_spot_amp = np.copy(self.spot_amp)
        # null parameters, basis "knm" (:1196-1199)
        self.null_knm = None if null_vectors is None else np.asarray(null_vectors, dtype=float).reshape(2, -1)
        self.null_radius_knm = null_radius if self.null_knm is not None else None
        self.null_region_knm = null_region
        self.spot_integration_width_knm = spot_integration_width(v)
        if np.any(v[0] < 0) or np.any(v[1] < 0) or np.any(v[0] >= shape[1]) or np.any(v[1] >= shape[0]):
            raise ValueError("Spots outside SLM computational space bounds!")
        if self.null_knm is not None:               # :1343-1349
            if self.null_radius_knm is None:
                self.null_radius_knm = smallest_distance_chebyshev(np.hstack((self.null_knm, self.spot_knm))) / 4
            self.null_radius_knm = int(np.ceil(self.null_radius_knm))
        super().__init__((int(shape[0]), int(shape[1])), amp=amp, phase=phase,
                         slm_shape=slm_shape, dtype=dtype, **flags)
        if null_region_radius_frac is not None:     # :1361-1373 (shape[0] points along x: only square grids agree)
            if self.null_region_knm is None:
                self.null_region_knm = np.zeros(self.shape, dtype=bool)
            xg, yg = np.meshgrid(np.linspace(-1, 1, self.shape[0]), np.linspace(-1, 1, self.shape[1]))
            self.null_region_knm[np.square(xg) + np.square(yg) > null_region_radius_frac ** 2] = True
        self.set_target_spots(reset_weights=True)

    def set_target_spots(self, reset_weights=False):
        self.spot_knm_rounded = np.rint(self.spot_knm).astype(int)
        if self.null_knm is None:                   # :1514-1515 (a null region alone changes nothing)
            self.target.fill(0)
        else:   # MRAF background: NaN = free; null region and disks around null points AND spots = 0 (:1516-1538)
            self.target.fill(np.nan)
            if self.null_region_knm is not None:
                self.target[self.null_region_knm] = 0
            w = int(2 * self.null_radius_knm + 1)
            for x, y in np.hstack((self.null_knm, self.spot_knm)).T:
                imprint_zero_disk(self.target, np.rint(x), np.rint(y), w)
        self.target[self.spot_knm_rounded[1], self.spot_knm_rounded[0]] = self.spot_amp
        self.target /= l2norm(self.target)
        if reset_weights:
            self.reset_weights()

    def update_weights(self):
        fb = self.flags["feedback"]
        if fb == "computational":
            update_weights_generic(self.weights, self.amp_ff, self.target,
                                   self.flags["method"], self.flags, self.dtype)
            return
        if fb == "computational_spot":
            amp_fb = np.sqrt(take_sum(np.square(self.amp_ff), self.spot_knm_rounded,
                                      self.spot_integration_width_knm))
        elif fb == "external_spot":
            amp_fb = self.external_spot_amp
        else:
            raise ValueError(f"Feedback '{fb}' not recognized.")
        ky, kx = self.spot_knm_rounded[1], self.spot_knm_rounded[0]
        self.weights[ky, kx] = update_weights_generic(
            self.weights[ky, kx], np.array(amp_fb, dtype=self.dtype), self.spot_amp,
            self.flags["method"], self.flags, self.dtype)

    def compute_stats(self, stat_groups):
        stats = super().compute_stats(stat_groups)
        if "computational_spot" in stat_groups:
            if self.shape == self.slm_shape:
                ky, kx = self.spot_knm_rounded[1], self.spot_knm_rounded[0]
                stats["computational_spot"] = calculate_stats(
                    self.amp_ff[ky, kx], self.spot_amp, total=np.sum(np.square(self.amp_ff)))
            else:
                pwr = np.square(self.amp_ff)
                fb = take_sum(pwr, self.spot_knm, self.spot_integration_width_knm)
                stats["computational_spot"] = calculate_stats(
                    np.sqrt(fb), self.spot_amp, total=np.sum(pwr))
        return stats


def imprint_zero_disk(matrix, x, y, w):
    """
    toolbox.imprint(matrix, (x, w, y, w), 0, centered=True, circular=True) -> window_slice
    (toolbox/__init__.py:499-528): the bounding box is clipped to [0, n - 1] FIRST, the disk is centred on the
    clipped box's corner + (w - 1) // 2, and the clipped upper bound is exclusive.
    """
    xi = int(x - (w - 2) / 2)
    xf = xi + int(w)
    yi = int(y - (w - 2) / 2)
    yf = yi + int(w)
    xi, xf = np.clip([xi, xf], 0, matrix.shape[1] - 1)
    yi, yf = np.clip([yi, yf], 0, matrix.shape[0] - 1)
    xg, yg = np.meshgrid(np.arange(xi, xf), np.arange(yi, yf))
    xc = xi + int((w - 1) / 2)
    yc = yi + int((w - 1) / 2)
    rr = (w ** 2) * np.square(xg.astype(float) - xc) + (w ** 2) * np.square(yg.astype(float) - yc)
    mask = rr <= (w ** 2) * (w ** 2) / 4.0
    matrix[yg[mask], xg[mask]] = 0


def smallest_distance_chebyshev(vectors):
    """Brute-force restatement of toolbox.smallest_distance (toolbox/__init__.py:1127-1250)."""
    v = np.asarray(vectors, dtype=float).reshape(2, -1)
    n = v.shape[1]
    if n < 2:
        return np.inf
    best = np.inf
    for i in range(n - 1):
        d = np.max(np.abs(v[:, i + 1:] - v[:, i:i + 1]), axis=0)
        best = min(best, float(np.min(d)))
    return best


def spot_integration_width(spot_knm, psf_knm=0.0):
    """Default integration width without hardware.  _spots.py:1284-1297 (quirk A19)."""
    min_psf = 3
    dist = np.max([smallest_distance_chebyshev(spot_knm) / 1.5, min_psf])
    w = np.clip(10 * psf_knm, min_psf, dist)
    return int(2 * np.floor(w / 2) + 1)


def rectangular_array(shape, array_shape, array_pitch, array_center=None):
    """Spot grid in "knm".  SpotHologram.make_rectangular_array, _spots.py:1441-1485."""
    if np.isscalar(array_shape):
        array_shape = (int(array_shape), int(array_shape))
    if np.isscalar(array_pitch):
        array_pitch = (array_pitch, array_pitch)
    if array_center is None:
        array_center = (shape[1] / 2.0, shape[0] / 2.0)
    xe = (np.arange(array_shape[0]) - (array_shape[0] - 1) / 2.0) * array_pitch[0] + array_center[0]
    ye = (np.arange(array_shape[1]) - (array_shape[1] - 1) / 2.0) * array_pitch[1] + array_center[1]
    xg, yg = np.meshgrid(xe, ye, sparse=False, indexing="xy")
    return np.vstack((xg.ravel(), yg.ravel()))


# ======================================================================================
# CompressedSpotHologram path: non-uniform DFT with per-spot Zernike kernels
# ======================================================================================
def ansi_to_radial(j):
    """ANSI single index -> (n, l).  phase.zernike_convert_index (phase.py:570-680)."""
    j = int(j)
    n = int(np.ceil((-3 + np.sqrt(9 + 8 * j)) / 2))
    return n, 2 * j - n * (n + 2)


def zernike_cartesian(j):
    """
    Integer coefficients {(px, py): c} of the un-normalised real Zernike polynomial of ANSI index j
    (value +-1 at the pupil edge; Z1 = y, Z2 = x, Z4 = 2x^2 + 2y^2 - 1), as produced by
    phase._zernike_coefficients (phase.py:1357-1442).  Independent derivation:
    R_n^|l|(r) = sum_s (-1)^s (n-s)! / (s! ((n+|l|)/2-s)! ((n-|l|)/2-s)!) r^(n-2s), times
    Re/Im (x+iy)^|l| for l >= 0 / l < 0, with r^2k = (x^2+y^2)^k expanded binomially.
    """
    from math import comb, factorial
    n, l = ansi_to_radial(j)
    al = abs(l)
    out = {}
    for s in range((n - al) // 2 + 1):
        rc = (-1) ** s * factorial(n - s) // (factorial(s) * factorial((n + al) // 2 - s) * factorial((n - al) // 2 - s))
        k = (n - 2 * s - al) // 2                       # r^(n-2s) = r^al * (x^2+y^2)^k
        for t in range(k + 1):                          # (x^2+y^2)^k = sum_t C(k,t) x^(2(k-t)) y^(2t)
            for q in range(al + 1):                     # (x+iy)^al = sum_q C(al,q) x^(al-q) (iy)^q
                if (l >= 0 and q % 2 == 1) or (l < 0 and q % 2 == 0):
                    continue
                sign = (-1) ** (q // 2)                 # i^q: real part for even q, imaginary for odd q
                key = (al - q + 2 * (k - t), q + 2 * t)
                out[key] = out.get(key, 0) + rc * comb(k, t) * comb(al, q) * sign
    return {k: v for k, v in out.items() if v != 0}


def zernike_basis_default(D):
    """phase._zernike_indices_parse(None, D) (phase.py:923-961): [2,1], [2,1,4], [2,1,4,3,5,...]."""
    if D == 2:
        return np.array([2, 1])
    if D == 3:
        return np.array([2, 1, 4])
    if D == 4:
        return np.array([2, 1, 4, 3])
    return np.hstack((np.array([2, 1, 4, 3]), np.arange(5, D + 1)))


def cantor_pairing(px, py):
    return (px + py) * (px + py + 1) // 2 + py


def monomial_weights(basis, spot_zernike):
    """
    (terms [M,2], weights [M,N]): phi_n = sum_m weights[m,n] x^terms[m,0] y^terms[m,1].
    phase._zernike_get_cantor (phase.py:850-920): monomials in ascending Cantor order.
    """
    a = np.asarray(spot_zernike, dtype=float)
    acc = {}
    special_terms, special_weights = [], []
    for d, idx in enumerate(np.ravel(basis)):
        if int(idx) < 0:        # special (non-polynomial) indices ride behind the monomials, phase.py:909-918
            special_terms.append((int(idx), 0))
            special_weights.append(a[d])
            continue
        for key, c in zernike_cartesian(int(idx)).items():
            acc[key] = acc.get(key, 0) + c * a[d]
    keys = sorted(acc, key=lambda k: cantor_pairing(*k))
    terms = list(keys) + special_terms
    rows = [acc[k] for k in keys] + special_weights
    return np.array(terms, dtype=int).reshape(-1, 2), np.array(rows, dtype=float).reshape(len(terms), -1)


def compressed_kernel(xg, yg, terms, weights, ctype):
    """
    K[n, p] = exp(i phi_n(p)) / sqrt(S) in the hologram's complex dtype.
    _build_cupy_kernel_batched (_spots.py:595-636) + phase.polynomial (phase.py:1672-1795):
    the grids are cast to the complex dtype and the monomial sum is accumulated in that precision.
    """
    x = np.asarray(xg).astype(ctype)
    y = np.asarray(yg).astype(ctype)
    N = weights.shape[1]
    out = np.zeros((N,) + x.shape, dtype=ctype)
    w = weights.astype(ctype)
    for m, (px, py) in enumerate(terms):
        if px < 0:              # phase.py:1783-1792: the vortex plate, positive charges only
            if not (px == -1 and py == 0):
                raise ValueError(f"Unrecognized terms {(px, py)} for index {m}.")
            lg = np.arctan2(np.real(y), np.real(x))
            for n in range(N):
                if w[m, n] > 0:
                    out[n] += w[m, n] * lg
            continue
        mono = np.ones_like(x)
        for _ in range(px):
            mono *= x
        for _ in range(py):
            mono *= y
        for n in range(N):
            if w[m, n] != 0:
                out[n] += w[m, n] * mono
    out = out.reshape(N, -1)
    out *= ctype(1j)
    np.exp(out, out=out)
    out /= np.sqrt(out.shape[1])
    return out


class OracleCompressedSpotHologram(OracleHologram):
    """
    N free-floating spots with per-spot Zernike kernels, no padded grid (CompressedSpotHologram,
    _spots.py:178-1018).  Inputs are taken after unit conversion: ``spot_zernike`` [D,N] in Zernike
    radians, the pupil-scaled grids ``xg``, ``yg`` [H,W] (slm.grid * zernike scaling, :614-618).
    """

    def __init__(self, spot_zernike, xg, yg, zernike_basis=None, spot_amp=None, amp=None, phase=None,
                 dtype=np.float32, **flags):
        self.spot_zernike = np.asarray(spot_zernike, dtype=float)
        D, N = self.spot_zernike.shape
        self.zernike_basis = zernike_basis_default(D) if zernike_basis is None else np.ravel(zernike_basis)
        self.xg, self.yg = np.asarray(xg, dtype=float), np.asarray(yg, dtype=float)
        self.spot_amp = np.full(N, 1.0 / np.sqrt(N)) if spot_amp is None else np.array(spot_amp)
        slm_shape = self.xg.shape
        super().__init__(slm_shape, amp=amp, phase=phase, slm_shape=slm_shape, dtype=dtype, **flags)
        self.set_target_spots(self.spot_amp)
        self.reset()
        self.external_spot_amp = np.ones(self.target.shape)
        self._kernel = None

    def set_target_spots(self, new_target):       # set_target :917-947
        self.target = np.array(new_target, dtype=self.dtype)
        np.abs(self.target, out=self.target)
        self.target *= 1 / l2norm(self.target)

    def reset(self):
        super().reset()
        self.nearfield = np.zeros(self.slm_shape, dtype=self.ctype)
        self.farfield = np.zeros(self.target.shape, dtype=self.ctype)

    def kernel(self):
        if self._kernel is None:
            terms, weights = monomial_weights(self.zernike_basis, self.spot_zernike)
            self._kernel = compressed_kernel(self.xg, self.yg, terms, weights, self.ctype)
        return self._kernel

    def nearfield2farfield(self):                 # _nearfield2farfield_cupy :767-824
        nf = self.build_nearfield()
        K = self.kernel()
        ff = np.conj(K @ np.conj(nf).ravel())
        ff *= 1 / l2norm(ff)
        self.farfield = ff.astype(self.ctype)
        self.amp_ff = np.abs(self.farfield, out=self.amp_ff)

    def farfield2nearfield(self):                 # _farfield2nearfield_cupy :887-914
        K = self.kernel()
        self.nearfield = (self.farfield[np.newaxis, :] @ K).reshape(self.slm_shape).astype(self.ctype)
        self.phase = np.arctan2(self.nearfield.imag, self.nearfield.real, out=self.phase)
        if self.propagation_kernel is not None:
            self.phase -= self.propagation_kernel

    def inverse_nearfield(self):                  # _farfield2nearfield_cupy(extract=False) :887-914
        K = self.kernel()
        self.nearfield = (self.farfield[np.newaxis, :] @ K).reshape(self.slm_shape).astype(self.ctype)
        return self.nearfield

    def update_weights(self):                     # _update_weights :950-989
        fb = self.flags["feedback"]
        if fb == "computational":
            fb = self.flags["feedback"] = "computational_spot"
        if fb == "computational_spot":
            amp_fb = self.amp_ff
        elif fb == "external_spot":
            amp_fb = self.external_spot_amp
        else:
            raise ValueError(f"Feedback '{fb}' not recognized.")
        update_weights_generic(self.weights, np.array(amp_fb, dtype=self.dtype), self.target,
                               self.flags["method"], self.flags, self.dtype)

    def compute_stats(self, stat_groups):         # _update_stats :1004-1018 ignores computational_spot
        return {}
