"""
MI355X-native hologram optimisation: the class surface of ``slmsuite.holography.algorithms``
(``Hologram``, ``FeedbackHologram``, ``SpotHologram``) over the HIP engine in ``libhgs.so``.

Same constructors, ``optimize(method=...)`` signature, ``flags`` / ``stats`` dictionaries and
persistent-state semantics as the reference (slmsuite 0.4.1; citations are relative to its
checkout).  Everything per-iteration -- nearfield build, padded 2-D FFT, farfield constraint and
weight update, inverse FFT, phase extraction, statistics reductions -- runs on the GPU; this file only
keeps the host-side bookkeeping (flag parsing, WGS-Kim history, stats lists).  There is no CPU
fallback: without the built library or a gfx950 device ``optimize()`` raises.
"""
import os
import sys
import time
import warnings

import numpy as np

from slmsuite_amd import _lib as L
from slmsuite_amd.engine import Engine, make_step
from slmsuite_amd.holography import toolbox

try:  # progress bars are optional
    from tqdm.auto import tqdm
except Exception:  # pragma: no cover
    tqdm = None

# _header.py:53-81
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
    "WGS-Wu": {"feedback": "computational", "feedback_exponent": .5},
    "WGS-tanh": {"feedback": "computational", "feedback_factor": .2, "feedback_exponent": .5},
    "CG": {"feedback": "computational", "optimizer": "Adam", "optimizer_kwargs": {"lr": .1}, "loss": None},
}
ALGORITHM_INDEX = {key: i for i, key in enumerate(ALGORITHM_DEFAULTS.keys())}
FEEDBACK_OPTIONS = [
    "computational",
    "computational_spot",
    "experimental",
    "experimental_spot",
    "external_spot",
]

_DEVICE_ARRAYS = {"phase": L.PHASE, "weights": L.WEIGHTS, "amp_ff": L.AMP_FF, "phase_ff": L.PHASE_FF,
                  "farfield": L.FARFIELD}


def _norm(matrix):
    """sqrt(nansum(|x|^2)).  Hologram._norm, _hologram.py:1979-2011."""
    if np.iscomplexobj(matrix):
        return np.sqrt(np.nansum(np.square(np.abs(matrix))))
    return np.sqrt(np.nansum(np.square(matrix)))


def _is_device_tensor(value):
    """A torch tensor that lives on a GPU (the engine takes those device to device)."""
    return type(value).__module__.split(".")[0] == "torch" and bool(getattr(value, "is_cuda", False))


def _fingerprint(arr):
    """
    (identity, cheap content sample) of a host array a side engine holds a device copy of: the copy is re-sent when either
    changes.  The sample is every 64th row plus the corner values -- it notices the in-place edits that keep the object
    (``kernel += defocus``, ``amp *= mask``, ``kernel[:] = ...``) without a pass over the whole array per camera frame; an
    edit confined to other rows needs :meth:`Hologram.refresh_farfield_inputs`.
    """
    if arr is None or np.isscalar(arr):
        return (None, arr)
    a = np.asarray(arr)
    sample = a[::64] if a.ndim == 2 else a
    return (id(arr), a.shape, float(np.sum(sample, dtype=np.float64)), float(a.flat[0]), float(a.flat[-1]))


def _history_put(seq, start, values, fill=np.nan):
    """``seq[start:start + len(values)] = values``, growing ``seq`` with ``fill`` as far as needed (at least to ``start``)."""
    missing = start + len(values) - len(seq)
    if missing > 0:
        seq.extend([fill] * missing)
    seq[start:start + len(values)] = values


class Hologram:
    """
    Phase-retrieval hologram on a padded DFT grid (reference: ``Hologram``, _hologram.py:26-2011).

    State arrays (``phase``, ``weights``, ``amp_ff``, ``phase_ff``, ``farfield``) live on the GPU
    once optimisation has started; the attributes of the same name download them on access and
    upload on assignment, so user code written against the reference keeps working.
    """

    # ---- construction (_hologram.py:196-439) -------------------------------------------------------
    def __init__(self, target, amp=None, phase=None, slm_shape=None, dtype=np.float32,
                 propagation_kernel=None, **kwargs):
        # SLM geometry: whatever states it -- an SLM / FourierSLM object (which may also supply the source amplitude),
        # an explicit pair, the amplitude array, the initial phase -- must agree
        device = getattr(slm_shape, "slm", slm_shape)
        if hasattr(device, "_get_source_amplitude") and hasattr(device, "shape"):
            amp = device._get_source_amplitude() if amp is None else amp
            slm_shape = device.shape
        stated = (("amplitude (via `amp` or SLM)", None if amp is None else np.shape(amp)),
                  ("initial phase (`phase`)", None if phase is None else np.shape(phase)),
                  ("SLM (via `target` or `slm_shape`)", slm_shape))
        stated = [(what, tuple(int(v) for v in shp)) for what, shp in stated if shp is not None and len(shp) == 2]
        self.slm_shape = None
        if stated:
            self.slm_shape = tuple(int(v) for v in np.rint(np.mean([shp for _, shp in stated], axis=0)))
            for what, shp in stated:
                if shp != self.slm_shape:
                    raise ValueError(f"The shape of the {what} is not equal to the other provided SLM shapes")

        # computational shape: the target image's, a bare (h, w) pair, or -- target=None -- the SLM's own
        own_shape = target is not None
        if not own_shape:
            if self.slm_shape is None:
                raise ValueError("SLM shape must be provided through cameraslm=")
            self.shape, target = self.slm_shape, []
        elif np.ndim(target) == 1 and len(target) == 2:
            self.shape, target = (int(target[0]), int(target[1])), None
        elif np.ndim(target) == 2:
            self.shape = tuple(int(v) for v in np.shape(target))
        else:
            raise ValueError(f"Unexpected target {target}.")
        if own_shape and any(v & (v - 1) for v in self.shape):
            warnings.warn(
                f"Hologram target shape {self.shape} is not a power of 2; consider using "
                ".get_padded_shape() to pad to powers of 2 and speed up FFT computation.")
        self.slm_shape = self.slm_shape or self.shape

        try:
            self.dtype, self.dtype_complex = {4: (np.float32, np.complex64), 8: (np.float64, np.complex128)}[np.dtype(dtype).itemsize]
        except KeyError:
            raise ValueError(f"Data type {dtype} not supported.") from None

        self._engine = None
        self._host = {}        # name -> host copy
        self._stale = set()    # names whose device copy is newer than the host copy
        self._upload = set()   # names whose host copy must reach the device before the next op
        self._n_spots_engine = 0
        self._weights_reset = False   # weights == target with NaN -> 0 (reset_weights): host copy built on access
        self._populate_pending = False   # the trailing transform of optimize() (_populate_results) not yet run
        # engine policy applied whenever this hologram creates an engine: {L.OPT_*: value} (hgs_set_option)
        self.engine_options = dict(kwargs.pop("engine_options", {}) or {})

        # source amplitude, unit L2 norm: an array, or the scalar of a uniform beam (:401-405)
        if amp is not None:
            self.amp = np.array(amp, dtype=self.dtype)
            self.amp *= 1 / _norm(self.amp)
        else:
            self.amp = 1 / np.sqrt(np.prod(self.slm_shape))

        self.propagation_kernel = None
        if isinstance(propagation_kernel, toolbox.REAL_TYPES):
            raise ValueError("propagation_kernel must be an array of slm_shape (scalars are rejected)")
        if propagation_kernel is not None:
            self.propagation_kernel = np.array(propagation_kernel, dtype=self.dtype)
            if self.propagation_kernel.shape != self.slm_shape:
                raise ValueError("Expected the propagation kernel to be the same shape as the SLM.")

        self.flags = kwargs
        self._host["phase"] = None
        self._set_target(target, reset_weights=False)
        self.reset_phase(phase)
        self.reset(reset_phase=False, reset_flags=False)

    # ---- device-backed attributes -----------------------------------------------------------------
    def _get_dev(self, name):
        if name in ("farfield", "amp_ff", "phase_ff"):
            if self.__dict__.get("_midloop") is not None:
                self._midloop_materialise(name)          # inside a callback of the device-resident loop
            else:
                self._flush_populate()
        if name in self._stale and self._engine is not None:
            arr = self._engine.get(_DEVICE_ARRAYS[name])[0]
            self._host[name] = arr
            self._stale.discard(name)
        elif name == "weights":
            self._materialise_weights()
        value = self._host.get(name)
        if _is_device_tensor(value):           # assigned as a GPU tensor and not yet handed to an engine
            return value.detach().cpu().numpy().astype(self.dtype, copy=False)
        return value

    def _set_dev(self, name, value):
        if name == "phase":
            self._flush_populate()            # the results of the last optimize() describe the phase it ended on
        self._host[name] = value
        self._stale.discard(name)
        if name == "weights":
            self._weights_reset = False
        if value is not None and name in ("phase", "weights", "phase_ff"):
            self._upload.add(name)

    def _materialise_weights(self):
        """Host copy of freshly reset weights (target with NaN -> 0, :608-614), built only when somebody looks:
        the device derives its own from the target it already holds (hgs_reset_weights)."""
        if self.__dict__.get("_weights_reset") and self._host.get("weights") is None and self.target is not None:
            w = np.array(self.target, copy=True)
            np.nan_to_num(w, copy=False, nan=0)
            self._host["weights"] = w

    phase = property(lambda s: s._get_dev("phase"), lambda s, v: s._set_dev("phase", v))
    weights = property(lambda s: s._get_dev("weights"), lambda s, v: s._set_dev("weights", v))
    amp_ff = property(lambda s: s._get_dev("amp_ff"), lambda s, v: s._set_dev("amp_ff", v))
    phase_ff = property(lambda s: s._get_dev("phase_ff"), lambda s, v: s._set_dev("phase_ff", v))
    farfield = property(lambda s: s._get_dev("farfield"), lambda s, v: s._set_dev("farfield", v))

    @property
    def nearfield(self):
        """Padded complex nearfield amp*exp(i*phase) (the reference's buffer of _build_nearfield)."""
        nf = np.zeros(self.shape, dtype=self.dtype_complex)
        i0, i1, i2, i3 = toolbox.unpad(self.shape, self.slm_shape)
        ph = self.phase if self.propagation_kernel is None else self.phase + self.propagation_kernel
        nf[i0:i1, i2:i3] = self.amp * np.exp(1j * ph)
        return nf

    def _n_spots(self):
        return 0

    def _get_engine(self):
        """Create the engine on first use and push every pending host array."""
        e = self._engine
        if e is None:
            e = self._engine = Engine(self.shape, self.slm_shape, self.dtype, batch=1, n_spots=self._n_spots())
            for opt, val in self.engine_options.items():
                e.set_option(opt, val)
            if np.isscalar(self.amp):
                e.set(L.AMP_SCALAR, np.array([self.amp], dtype=self.dtype))
            else:
                e.set(L.AMP, self.amp)
            if self.propagation_kernel is not None:
                e.set(L.PROP_KERNEL, self.propagation_kernel)
            self._upload_target(e)
            self._upload.add("phase")
            if self._weights_reset and self._host.get("weights") is None:
                e.reset_weights()                     # built on the device from the target just uploaded
                self._upload.discard("weights")
            else:
                self._upload.add("weights")
            if self._host.get("phase_ff") is not None:
                self._upload.add("phase_ff")
            self._engine_setup(e)
        for name in list(self._upload):
            value = self._host.get(name)
            if _is_device_tensor(value):
                # already on a GPU: device to device, and from here on the engine's copy is the one that counts
                e.set_tensor(_DEVICE_ARRAYS[name], value.reshape((1,) + tuple(value.shape[-2:])))
                self._stale.add(name)          # (a read downloads the engine's copy and replaces the tensor kept here)
            elif value is not None:
                e.set(_DEVICE_ARRAYS[name], value)
            self._upload.discard(name)
        return e

    def _engine_setup(self, e):
        pass

    def _upload_target(self, e):
        """The target's way to the device; SpotHologram sends its spot list instead of the raster."""
        e.set(L.TARGET, self.target)

    # ---- reset helpers (_hologram.py:442-614) -----------------------------------------------------------
    def reset(self, reset_phase=True, reset_flags=False):
        """``Hologram.reset`` (:442-478): iteration counter, history and weights start over; the phase only on request
        (or when there is none); the flags only on request.  The engine, if any, survives."""
        if reset_phase or (self._host.get("phase") is None and "phase" not in self._stale):
            self.reset_phase()
        self.reset_weights()
        self.iter, self.stats = 0, dict(method=[], flags={}, stats={})
        if reset_flags:
            self.flags = dict(method="")
        self._host.update(amp_ff=None, phase_ff=None, farfield=np.zeros(self.shape, dtype=self.dtype_complex))
        self._stale -= {"amp_ff", "phase_ff", "farfield"}
        self._upload.discard("phase_ff")
        self._populate_pending = False
        if self._engine is not None:
            # weights from its target, phase_ff / farfield / amp_ff back to "None" (hgs_reset); a phase only the device
            # holds (reset_phase False after an optimize()) simply stays there
            self._engine.reset()

    def _release_engine(self):
        """Download every device-fresh array (the engine may not be able to serve all of them), then destroy it."""
        if self._engine is None:
            return
        self._flush_populate()
        for name in list(self._stale):
            try:
                self._get_dev(name)
            except L.HgsError:                  # e.g. a farfield the loop has consumed since: nothing to keep
                self._host[name] = None
        self._stale.clear()
        self._engine.close()
        self._engine = None

    def _get_random_phase(self):
        return np.random.default_rng().uniform(-np.pi, np.pi, size=self.slm_shape).astype(self.dtype)

    def _get_target_moments_knm_norm(self):
        """Centre and standard deviation of the target in knm pixels / shape (_hologram.py:480-500)."""
        center, std = toolbox.image_center_and_std(self.target, nansum=True)
        shape = np.flip(self.shape).astype(float)
        return center / shape, std / shape

    def _get_quadratic_initial_phase(self, scaling=1):
        """
        Lens + blaze that roughly spreads the source over the target (_hologram.py:502-527): the blaze steers to
        the target's centroid, the lens matches its standard deviation to that of the source amplitude.
        Needs an array-valued ``amp`` (the reference's image moments cannot take the scalar form either).
        """
        if np.isscalar(self.amp) or np.ndim(self.amp) != 2:
            raise ValueError("quadratic_phase needs an array-valued source amplitude (amp= or an SLM)")
        _, std_amp = toolbox.image_center_and_std(self.amp)
        slm_shape = np.flip(self.slm_shape).astype(float)
        std_amp = std_amp / slm_shape
        center_knm_norm, std_knm_norm = self._get_target_moments_knm_norm()
        h, w = self.slm_shape
        x = ((np.arange(w, dtype=float) - float(w - 1) / 2).astype(self.dtype) / w).reshape(1, w)
        y = ((np.arange(h, dtype=float) - float(h - 1) / 2).astype(self.dtype) / h).reshape(h, 1)
        grid = (np.broadcast_to(x, (h, w)).astype(self.dtype), np.broadcast_to(y, (h, w)).astype(self.dtype))
        with np.errstate(divide="ignore"):
            f = np.reciprocal(scaling * slm_shape * std_knm_norm / std_amp)
        return np.array(toolbox.blaze(grid, slm_shape * center_knm_norm) + toolbox.lens(grid, f), dtype=self.dtype)

    def reset_phase(self, custom_phase=None, random_phase=None, quadratic_phase=None):
        """``Hologram.reset_phase`` (:536-601): a given phase verbatim; otherwise ``quadratic_phase`` x the lens + blaze
        guess plus ``random_phase`` x uniform noise, each taken from the flags when not passed (defaults: off, 1)."""
        if _is_device_tensor(custom_phase):
            if tuple(custom_phase.shape) != tuple(self.slm_shape):
                raise ValueError(f"Reset phase of shape {tuple(custom_phase.shape)} is not of slm_shape {self.slm_shape}")
            self.phase = custom_phase.detach().clone()          # stays on the GPU; uploaded device to device
            return
        if custom_phase is not None:
            given = np.array(custom_phase, dtype=self.dtype)
            if tuple(given.shape) != tuple(self.slm_shape):
                raise ValueError(f"Reset phase of shape {given.shape} is not of slm_shape {self.slm_shape}")
            self.phase = given.copy()
            return
        quadratic = self.flags.get("quadratic_phase", False) if quadratic_phase is None else quadratic_phase
        noise = self.flags.get("random_phase", 1) if random_phase is None else random_phase
        total = np.zeros(self.slm_shape, dtype=self.dtype)
        if quadratic:
            total += self._get_quadratic_initial_phase(quadratic)
        if noise:
            total += noise * self._get_random_phase()
        self.phase = total

    def reset_weights(self):
        self._host["weights"] = None          # = target with NaN -> 0; see _materialise_weights
        self._stale.discard("weights")
        self._upload.discard("weights")
        self._weights_reset = True
        if self._engine is not None:
            self._engine.reset_weights()

    # ---- padding rule (_hologram.py:616-725) -----------------------------------------------------------
    @staticmethod
    def get_padded_shape(slm_shape, padding_order=1, square_padding=True, precision=np.inf,
                         precision_basis="kxy"):
        """
        Computational shape for an SLM (``_hologram.py:616-725``): per axis the larger of (a) the ``padding_order``-th
        power of two above the SLM shape (order 0: the SLM shape itself) and (b), with a finite ``precision``, the power
        of two whose DFT grid resolves ``precision`` in ``precision_basis`` (needs the SLM's pitch, hence an SLM or
        FourierSLM object; "ij" needs the FourierSLM's calibration); squared up on request.
        """
        fourier, slm = None, None
        if hasattr(slm_shape, "slm") and hasattr(slm_shape, "cam"):
            fourier, slm = slm_shape, slm_shape.slm
        elif hasattr(slm_shape, "shape"):
            slm = slm_shape
            if precision_basis == "ij":
                raise ValueError("Must pass a CameraSLM object under slm_shape to use the 'ij' precision_basis!")
        base = tuple(slm.shape) if slm is not None else tuple(slm_shape)

        floor = base                                   # (b): nothing asked for
        if np.isfinite(precision):
            if slm is None:
                raise ValueError("Must pass a CameraSLM object under slm_shape to implement precision calculations!")
            if precision <= 0:
                raise ValueError("Precision passed to get_padded_shape() must be positive.")
            sampling = 1 / np.amin(slm.pitch)
            extent = np.amax(fourier.kxyslm_to_ijcam([sampling, sampling])) if precision_basis == "ij" else sampling
            side = 1 << int(np.ceil(np.log2(extent / precision)))
            floor = (side, side)
        padded = base if padding_order <= 0 else tuple(
            int(2 ** (np.ceil(np.log2(v)) + padding_order - 1)) for v in base)
        shape = tuple(int(max(u, v)) for u, v in zip(floor, padded))
        return (max(shape),) * 2 if square_padding else shape

    # ---- target / accessors (_hologram.py:741-931) ---------------------------------------------------------
    def _set_target(self, new_target, reset_weights=False):
        if not reset_weights:
            self._materialise_weights()        # weights derived from the OLD target must not follow the new one
            self._weights_reset = False
        if new_target is None or (hasattr(new_target, "__len__") and len(new_target) == 0):
            self.target = np.zeros(self.shape, dtype=self.dtype)
        else:
            self.target = np.array(new_target, dtype=self.dtype)
            np.abs(self.target, out=self.target)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                self.target *= 1 / _norm(self.target)
        self._target_changed(reset_weights)

    def _target_changed(self, reset_weights):
        """A new ``self.target`` goes to the engine (if one exists); the weights follow it on request."""
        if self._engine is not None:
            self._upload_target(self._engine)
        if reset_weights:
            self.reset_weights()

    def set_target(self, new_target, reset_weights=False):
        self._set_target(new_target, reset_weights)

    def get_phase(self, include_propagation=False):
        """The mask for the SLM (:786-806): ``phase + pi`` -- or, with ``include_propagation`` and a kernel, the phase at
        the kernel's depth (without the pi, as in the reference)."""
        with_kernel = include_propagation and self.propagation_kernel is not None
        return self.phase + (self.propagation_kernel if with_kernel else np.pi)

    def get_amp(self):
        amplitude = self.amp
        return amplitude

    def set_weights(self, new_weights):
        if tuple(np.shape(new_weights)) != tuple(self.target.shape):
            raise ValueError(f"New weights {np.shape(new_weights)} do not match target shape {self.target.shape}")
        self.weights = np.array(new_weights, dtype=self.dtype)

    def get_weights(self):
        return self.weights

    def get_farfield(self, shape=None, propagation_kernel=None, affine=None, get=True):
        """
        Complex DFT farfield of the current phase at an arbitrary ``shape``, optionally at another depth
        (``propagation_kernel``) and resampled by ``affine`` (_hologram.py:853-931; the call SimulatedCamera makes per
        frame, hardware/cameras/simulated.py:370).  One engine per requested shape is kept alive; a call costs two kernels
        plus what the caller asks to move:

        * the phase goes engine -> engine on the device when this hologram's engine holds it (``hgs_copy_phase``);
        * source amplitude and kernel are re-sent only when they are other objects than last time; no kernel = none sent;
        * ``get=False`` returns the field as a torch tensor on the GPU (the reference: a CuPy array), ``get=True`` a NumPy
          array -- also for ``get=False`` in a process that has not imported torch.  The affine step is the reference's host call (``scipy.ndimage.affine_transform``, order 3, constant 0).
        """
        self._flush_populate()
        shape = self.shape if shape is None else self.slm_shape if len(shape) == 1 else shape
        shape = (int(shape[0]), int(shape[1]))
        for fallback in (self.propagation_kernel, 0):
            propagation_kernel = fallback if propagation_kernel is None else propagation_kernel
        e, slot = self._side_engine(shape, with_kernel=False)
        if np.isscalar(propagation_kernel):
            if propagation_kernel == 0:
                if slot["kernel"] is not False:
                    e.clear_propagation_kernel()
                    slot["kernel"] = False
            else:
                e.set(L.PROP_KERNEL, np.full(self.slm_shape, propagation_kernel, dtype=self.dtype))
                slot["kernel"] = None
        elif slot["kernel"] != _fingerprint(propagation_kernel):
            kern = np.ascontiguousarray(propagation_kernel, dtype=self.dtype)
            if kern.shape != tuple(self.slm_shape):
                raise ValueError(f"propagation_kernel must have the SLM shape {tuple(self.slm_shape)}")
            e.set(L.PROP_KERNEL, kern)
            slot["kernel"] = _fingerprint(propagation_kernel)
        if self._engine is not None:
            e.copy_phase_from(self._get_engine())            # (pending host edits of the phase go up first)
        elif _is_device_tensor(self._host.get("phase")):
            e.set_tensor(L.PHASE, self._host["phase"])
        else:
            e.set(L.PHASE, self.phase)
        refresh = shape == tuple(self.shape) and self._host.get("amp_ff") is not None
        e.nearfield2farfield(store_phase_ff=refresh)
        if refresh:
            self.amp_ff = e.get(L.AMP_FF)[0]
            self.phase_ff = e.get(L.PHASE_FF)[0]
        # get=False: the field stays on the GPU as a torch tensor where the process works with torch (the reference returns a
        # CuPy array when CuPy is its backend) and is the NumPy array otherwise, as in the reference without CuPy -- a
        # NumPy-only process must not import torch here: its HIP runtime would come up after libhgs.so's (_lib docstring)
        on_gpu = not get and "torch" in sys.modules
        if on_gpu and affine is None:
            return e.get_tensor(L.FARFIELD)[0]
        ff = e.get(L.FARFIELD)[0]
        if affine is not None:
            from scipy.ndimage import affine_transform
            ff = affine_transform(input=ff, matrix=affine["M"], offset=affine["b"], output_shape=shape, order=3,
                                  mode="constant", cval=0)
        if on_gpu:
            import torch
            return torch.from_numpy(ff).cuda()
        return ff

    def _side_engine(self, shape, with_kernel=True):
        """
        The engine kept per DFT shape for transforms beside the loop (get_farfield; phase_ff inside a callback), with this
        hologram's source amplitude -- re-sent only when it is another object than last time -- and, ``with_kernel``, its
        own propagation kernel.  Returns the engine (``with_kernel``) or (engine, slot) for get_farfield's kernel logic.
        """
        shape = (int(shape[0]), int(shape[1]))
        engines = self.__dict__.setdefault("_ff_engines", {})
        slot = engines.get(shape)
        if slot is None:
            slot = engines[shape] = {"engine": Engine(shape, self.slm_shape, self.dtype, batch=1), "amp": None, "kernel": None}
        e = slot["engine"]
        fp = _fingerprint(self.amp)
        if slot["amp"] != fp:
            if np.isscalar(self.amp):
                e.set(L.AMP_SCALAR, np.array([self.amp], dtype=self.dtype))
            else:
                e.set(L.AMP, self.amp)
            slot["amp"] = fp
        if not with_kernel:
            return e, slot
        kern = self.propagation_kernel
        if kern is None:
            if slot["kernel"] is not False:
                e.clear_propagation_kernel()
                slot["kernel"] = False
        elif slot["kernel"] != _fingerprint(kern):
            e.set(L.PROP_KERNEL, np.ascontiguousarray(kern, dtype=self.dtype))
            slot["kernel"] = _fingerprint(kern)
        return e

    def refresh_farfield_inputs(self):
        """Forget the device copies of the source amplitude and propagation kernel the per-shape transform engines hold
        (get_farfield re-sends them only when the arrays look changed -- see _fingerprint): after an in-place edit that the
        sample cannot see."""
        for slot in self.__dict__.get("_ff_engines", {}).values():
            slot["amp"] = slot["kernel"] = None

    # ---- statistics bookkeeping (_stats.py:118-223) ---------------------------------------------------------
    @staticmethod
    def _raw_stats(feedback_amp, target_amp):
        """The two arrays ``raw_stats=True`` adds to a statistics group (_stats.py:104-114): the normalised feedback
        power and its ratio to the normalised target power (NaN where the target is zero or NaN)."""
        fp = np.square(np.asarray(feedback_amp, dtype=float))
        fp = fp * (1 / np.sum(fp))
        tp = np.square(np.asarray(target_amp, dtype=float))
        tp = tp * (1 / np.nansum(tp))
        mask = np.logical_and(tp != 0, np.logical_not(np.isnan(tp)))
        ratio = np.full_like(tp, np.nan)
        ratio[mask] = fp[mask] / tp[mask]
        return {"raw_pwr": fp, "raw_pwr_ratio": ratio}

    def _calculate_stats_computational(self, stats, stat_groups=[]):
        if "computational" in stat_groups:
            stats["computational"] = self._get_engine().stats(0)[0]
            if self.flags.get("raw_stats", False):      # host arrays, as the reference stores them (off the fast path)
                stats["computational"].update(self._raw_stats(self.amp_ff, self.target))

    def _update_stats_dictionary(self, stats):
        """
        History bookkeeping of one iteration (``_stats.py:118-223``): the method, every flag and every statistic of
        ``stats`` (``{group: {name: value}}``) land in slot ``self.iter`` of their lists.  One iteration is a range of
        length one of :meth:`_update_stats_batch`, which holds the rules (list creation, NaN padding, the reference's
        choice of statistic names).
        """
        self._update_stats_batch(1, [self.flags.get("fixed_phase")], [stats], list(stats))
        if self.flags.get("raw_stats", False):
            frames = self.stats.setdefault("raw_farfield", [])
            _history_put(frames, self.iter, [np.array(self.farfield, copy=True)])

    def _update_stats_batch(self, n, fixed_hist, per_iter, groups):
        """
        The history after ``n`` iterations starting at ``self.iter``, written range by range (twenty iterations of
        per-entry bookkeeping cost as much as their kernels on a small grid).  Over such a range the flags are constant
        except ``fixed_phase`` (``fixed_hist[k]``); the statistics of iteration k are ``per_iter[k][group]`` for the
        ``groups`` computed (``per_iter`` None: none).  Rules, as the reference's per-iteration writer leaves them:

        * ``stats["method"]`` grows with "" up to the range, every other list with NaN;
        * a list that does not exist yet is created as long as the method list (so it also covers a history that is
          longer than ``self.iter + n``, e.g. after ``iter`` was lowered by hand); an existing one only grows;
        * flags that were recorded once but are no longer in ``self.flags`` keep their list in step (NaN);
        * every group holds the same statistic names: those computed now plus those of the FIRST group already in
          the history (quirk of ``_stats.py:168-178``).
        """
        first = self.iter
        book = self.stats
        _history_put(book["method"], first, [self.flags["method"]] * n, fill="")
        depth = len(book["method"])

        def column(table, key):
            col = table.get(key)
            if col is None:
                col = table[key] = [np.nan] * depth
            return col

        for name in set(self.flags) | set(book["flags"]):
            col = column(book["flags"], name)
            if name not in self.flags:
                _history_put(col, first + n, [])
            elif name == "fixed_phase":
                _history_put(col, first, list(fixed_hist[:n]))
            else:
                _history_put(col, first, [self.flags[name]] * n)

        fresh = list(groups) if per_iter is not None else []
        tables = book["stats"]
        if not fresh and not tables:
            return
        names = set()
        for g in fresh:
            names |= set(per_iter[0][g])
        if tables:
            names |= set(next(iter(tables.values())))
        for g in list(tables) + [g for g in fresh if g not in tables]:
            table = tables.setdefault(g, {})
            for name in names:
                col = column(table, name)
                if g in fresh and name in per_iter[0][g]:
                    _history_put(col, first, [per_iter[k][g][name] for k in range(n)])
                else:
                    _history_put(col, first + n, [])

    def _update_stats(self, stat_groups=[]):
        stats = {}
        self._calculate_stats_computational(stats, stat_groups)
        self._update_stats_dictionary(stats)

    # ---- optimize (_hologram.py:1076-1424) --------------------------------------------------------------------
    def optimize(self, method="GS", maxiter=20, verbose=True, callback=None, feedback=None,
                 stat_groups=[], **kwargs):
        name = kwargs.pop("name", None)
        self._update_flags(method, verbose, feedback, stat_groups, **kwargs)
        iterations = range(maxiter)
        if verbose and maxiter > 1 and tqdm is not None:
            iterations = tqdm(iterations, desc=name)
        if "GS" in method:
            self.optimize_gs(iterations, callback)
        elif "CG" in method:
            raise NotImplementedError(
                "'CG' (torch autograd, experimental in the reference) is outside the GS/WGS hot path of this build")
        else:
            raise ValueError(f"Unsupported optimization method '{method}'")

    def _update_flags(self, method, verbose, feedback, stat_groups, **kwargs):
        """
        ``self.flags`` for this call (``_hologram.py:1370-1424``).  Precedence, weakest first: the method's defaults
        (only where the hologram has no value yet -- flags persist between calls), ``fixed_phase = False`` likewise, the
        caller's keyword flags, then ``stat_groups`` and ``feedback``, each checked against FEEDBACK_OPTIONS right
        before it is stored (so a rejected name leaves the earlier updates in place, as in the reference).
        """
        defaults = ALGORITHM_DEFAULTS.get(method)
        if defaults is None:
            raise ValueError("Unrecognized method '{}'.\nValid methods include {}".format(method, list(ALGORITHM_DEFAULTS)))
        fl = self.flags
        fl["method"] = method
        for name, value in {**defaults, "fixed_phase": False}.items():
            fl.setdefault(name, value)
        fl.update(kwargs)
        checked = (("stat_groups", "Statistics group", stat_groups, stat_groups),
                   ("feedback", "Feedback", feedback, () if feedback is None else (feedback,)))
        for key, what, value, names in checked:
            for name in names:
                if name not in FEEDBACK_OPTIONS:
                    raise ValueError("{} '{}' not recognized as a feedback option.\nValid options: {}".format(
                        what, name, FEEDBACK_OPTIONS))
            if value is not None:
                fl[key] = value
        if verbose > 1:          # (:1411-1424) the flags this method reads, before the progress bar starts
            import pprint
            print(f"Optimizing with '{method}' using the following method-specific flags:")
            pprint.pprint({name: fl[name] for name in fl if name in defaults})
            print("", end="", flush=True)

    # ---- the loop (_hologram.py:1427-1493) ------------------------------------------------------------------------
    def _false_run(self, skip_last=False):
        """
        Trailing run of recorded False entries of stats["flags"]["fixed_phase"] (:1574-1577).
        NaN placeholders (iterations recorded before the flag existed) end the run: `not nan` is False.
        """
        hist = self.stats["flags"].get("fixed_phase", [])
        # only WGS-Kim consults the run, and only against fix_phase_iteration: counting past it would make every call
        # walk the whole history (thousands of entries in a re-optimisation loop)
        method = self.flags.get("method")
        cap = (int(self.flags.get("fix_phase_iteration", 0) or 0) + 1 if method == "WGS-Kim" else
               1 if method is not None else len(hist))
        run = 0
        for i in range(len(hist) - (2 if skip_last else 1), -1, -1):
            v = hist[i]
            if (isinstance(v, float) and np.isnan(v)) or v or run >= cap:
                break
            run += 1
        return run

    # the target is a plain host array; assigning it invalidates what was derived from it
    @property
    def target(self):
        return self.__dict__.get("_target")

    @target.setter
    def target(self, value):
        self.__dict__["_target"] = value
        self.__dict__["_mraf_flag"] = None

    def _mraf_enabled(self):
        """_mraf_helper_routines :1498: NaN anywhere in the target.  Evaluated once per assigned target (a
        16.7 M-pixel host reduction costs more than forty fused iterations); in-place edits of ``target``
        need ``set_target`` to reach the device anyway."""
        flag = self.__dict__.get("_mraf_flag")
        if flag is None:
            flag = self.__dict__["_mraf_flag"] = bool(np.isnan(np.sum(self.target)))
        return flag

    def _spot_window(self):
        return 3

    def _make_step(self, skip_last=False, efficiency_group=None):
        return make_step(self.flags, self.iter, false_run=self._false_run(skip_last),
                         mraf_enabled=self._mraf_enabled(), spot_window=self._spot_window(),
                         efficiency_group=efficiency_group)

    def _mark_device_fresh(self, names):
        for n in names:
            self._stale.add(n)
            self._upload.discard(n)

    def _device_stat_groups(self):
        """(groups, width, spot_xy) for Engine.iterate_stats: the groups _update_stats computes on the device."""
        return [g for g in ("computational",) if g in self.flags["stat_groups"]], 1, None

    def _efficiency_group(self):
        """
        WGS-Kim fixed by efficiency (:1560-1569) decides on ``stats[groups[-1]]["efficiency"][iter]``, the group
        that entered ``stats["stats"]`` last.  Returns None when that gate is off, the group's index in
        Engine.STAT_GROUPS when the engine can decide inside hgs_iterate_stats (exactly one requested group, computed
        on the device, and it is -- or will become -- the last key), and -1 when the host has to (no statistics:
        the reference's ValueError; several groups: their order is the iteration order of a Python set).
        """
        fl = self.flags
        if not ("Kim" in fl["method"] and fl.get("fix_phase_efficiency", None) is not None):
            return None
        requested = list(fl["stat_groups"])
        groups = self._device_stat_groups()[0] if len(requested) > 0 else []
        if len(requested) != 1 or len(groups) != 1:
            return -1
        keys = list(self.stats["stats"].keys())
        if groups[0] in keys and keys[-1] != groups[0]:
            return -1
        return Engine.STAT_GROUPS.index(groups[0])

    def _callback_loop_ok(self):
        """
        A callback runs against the device-resident loop -- one fused engine call per iteration, the arrays it may look at
        materialised only when it does -- unless something needs the general operators' intermediate arrays on the host
        every iteration: ``raw_stats``, an efficiency-gated Kim fixing the host has to decide, or the caller's own
        ``HGS_OPT_FORCE_STEPWISE`` (``engine_options``), which keeps the materialising loop.
        """
        fl = self.flags
        if fl.get("raw_stats", False) or self._efficiency_group() == -1:
            return False
        return not self.engine_options.get(L.OPT_FORCE_STEPWISE, 0) and os.environ.get("HGS_CALLBACK_STEPWISE") != "1"

    def _midloop_materialise(self, name):
        """
        What a callback sees mid-iteration (_hologram.py:1465-1477): ``farfield`` and ``amp_ff`` of the CURRENT phase -- one
        forward transform from the phase the engine holds, run the first time either is read; it leaves the fused state
        alone -- and ``phase_ff`` as the previous body left it: atan2 of the farfield THAT body started from (zero where an
        MRAF target is zero), unless the phase was fixed.  The fused kernels never store that array; the engine keeps the
        phase the body started from instead (HGS_OPT_KEEP_PREV_PHASE) and its farfield phase is formed here on demand.
        """
        state = self._midloop
        # A phase the callback has just assigned is NOT what these arrays describe: the reference formed them before it
        # called the callback (:1465-1477), and the body overwrites such a phase anyway (optimize_gs discards the pending
        # upload after the callback).  Hold the upload back, so that the result does not depend on whether -- or in which
        # order -- the callback looked at the farfield.
        held = "phase" in self._upload
        self._upload.discard("phase")
        try:
            self._midloop_materialise_held(name, state)
        finally:
            if held:
                self._upload.add("phase")

    def _midloop_materialise_held(self, name, state):
        e = self._get_engine()
        if self.__dict__.get("_populate_pending"):
            # first invocation of this call and nobody has looked at the previous call's results yet: its trailing transform
            # is exactly what the callback is entitled to see (farfield and amp_ff of the current phase, phase_ff as
            # _populate_results leaves it)
            self._flush_populate()
            state["ff"] = state["pff"] = True
            return
        if name in ("farfield", "amp_ff"):
            if not state.get("ff"):
                e.nearfield2farfield(store_phase_ff=False)
                self._mark_device_fresh(["farfield", "amp_ff"])
                state["ff"] = True
        elif not state.get("pff"):
            state["pff"] = True
            if "phase_ff" in self._upload:                     # assigned by this callback: that is what it reads back
                return
            prev = e.get_prev_phase() if state["bodies"] > 0 else None
            if prev is not None:
                side = self._side_engine(self.shape)
                side.set(L.PHASE, prev)
                side.nearfield2farfield(store_phase_ff=True)
                pf = side.get(L.PHASE_FF)[0]
                if self._mraf_enabled():
                    pf[self.target == 0] = 0                   # the zero region is cleared before the arctan2 (:1613-1636)
                self._host["phase_ff"] = pf
                self._stale.discard("phase_ff")
                self._upload.discard("phase_ff")
            elif state["bodies"] > 0:
                # the body ran the general operators (Bluestein shapes, compressed engines, MRAF with zero_weights, spot
                # feedback on a dense target): they store HGS_PHASE_FF itself, so the engine's copy is the one to show -- as
                # the host-driven loop marked it after every constraint step.  An engine that holds none (a body that
                # never formed it) leaves the host copy as it is.
                held = e.try_get(L.PHASE_FF)
                if held is not None:
                    self._host["phase_ff"] = held[0]
                    self._stale.discard("phase_ff")
                    self._upload.discard("phase_ff")

    def _device_loop_ok(self, callback):
        """
        True when the whole loop can run inside one engine call: no callback, no raw farfield capture, and an
        efficiency-gated Kim fixing only where the engine can take the decision (see _efficiency_group).  Requested
        statistics are computed on the device in the same pass (hgs_iterate_stats).
        """
        fl = self.flags
        return not (callback is not None or fl.get("raw_stats", False) or self._efficiency_group() == -1)

    _BAR_FIRST_CHUNK = 8        # iterations of the first engine call under a progress bar
    _BAR_PERIOD = 0.1           # seconds of device work per later call (a bar refreshes at 10 Hz)

    def optimize_gs(self, iterations, callback):
        if self._populate_pending:
            # Nobody looked at the results of the previous call.  Its trailing transform matters to this loop only
            # through phase_ff, which a fixed phase (WGS-Kim after fixing, "GS" after such a run: quirk A4) reads;
            # a callback may look at any of it.  Otherwise this call's own trailing transform replaces all of it.
            if self.flags.get("fixed_phase", False) or (callback is not None and not self._callback_loop_ok()):
                self._flush_populate()
            elif callback is None:
                self._populate_pending = False
                self._stale -= {"farfield", "amp_ff", "phase_ff"}
            # (a callback against the device-resident loop: still pending -- run if its first invocation looks)
        e = self._get_engine()
        self._pre_loop_checks()
        n_total = len(iterations)
        if self._device_loop_ok(callback):
            # device-resident loop: flags history (and statistics) are replayed on the host afterwards
            bar = iterations if (tqdm is not None and not isinstance(iterations, range)) else None
            done = 0
            groups, width, xy = self._device_stat_groups() if len(self.flags["stat_groups"]) > 0 else ([], 1, None)
            eg = self._efficiency_group()
            # Without a progress bar the loop is ONE engine call.  With one, the calls are sized by time, not by count: each
            # ends in a stream sync (a bar that runs ahead of the device is no bar), so a call must be long enough for the
            # sync not to matter -- start with a few iterations, then aim at _BAR_PERIOD seconds per call
            chunk = n_total if bar is None else min(n_total, self._BAR_FIRST_CHUNK)
            while done < n_total:
                n = min(chunk, n_total - done)
                st = self._make_step(efficiency_group=eg)
                t0 = time.perf_counter()
                if groups:
                    hist, per_iter = e.iterate_stats(st, n, groups, width, xy)
                else:
                    hist, per_iter = e.iterate(st, n), None
                self._update_stats_batch(n, hist, None if per_iter is None else
                                         [{g: per_iter[k][g][0] for g in groups} for k in range(n)], groups)
                self.iter += n
                self.flags["fixed_phase"] = bool(st.fixed_phase)
                done += n
                if bar is not None:
                    e.sync()
                    per_iteration = max((time.perf_counter() - t0) / n, 1e-7)
                    chunk = max(n, int(self._BAR_PERIOD / per_iteration))
                    bar.update(n)
            if bar is not None:
                bar.close()
            self._mark_device_fresh(["phase", "weights"])
        elif self._callback_loop_ok():
            # a callback against the device-resident loop: one fused engine call per iteration (the last launch of a call
            # leaves G of the next body behind, so this costs a call's overhead, not a transform); what the callback may
            # read is materialised when it does (_midloop_materialise)
            groups, width, xy = self._device_stat_groups() if len(self.flags["stat_groups"]) > 0 else ([], 1, None)
            eg = self._efficiency_group()
            e.set_option(L.OPT_KEEP_PREV_PHASE, 1)
            bodies = 0
            try:
                for _ in iterations:
                    self._get_engine()                       # push what the caller changed since the last body
                    self._midloop = {"bodies": bodies}
                    try:
                        stop = callback(self)
                    finally:
                        self._midloop = None
                    if stop:
                        break
                    # a phase assigned inside the callback is overwritten by this body's own result, as in the reference
                    # (:1483-1487: _farfield2nearfield extracts the phase from the farfield the body started with)
                    if "phase" in self._upload:
                        self._upload.discard("phase")
                        self._stale.add("phase")
                    e = self._get_engine()
                    st = self._make_step(efficiency_group=eg)
                    if groups:
                        hist, per_iter = e.iterate_stats(st, 1, groups, width, xy)
                    else:
                        hist, per_iter = e.iterate(st, 1), None
                    self._update_stats_batch(1, hist, None if per_iter is None else [{g: per_iter[0][g][0] for g in groups}], groups)
                    self.iter += 1
                    bodies += 1
                    if self._populate_pending:               # the previous call's trailing transform: nobody looked, now stale
                        self._populate_pending = False
                        self._stale -= {"farfield", "amp_ff", "phase_ff"}
                    self.flags["fixed_phase"] = bool(st.fixed_phase)
                    self._mark_device_fresh(["phase", "weights"])
                    self._stale -= {"farfield", "amp_ff"}
            finally:
                if self._engine is not None:
                    self._engine.set_option(L.OPT_KEEP_PREV_PHASE, 0)
        else:
            for _ in iterations:
                self._get_engine()                       # push user edits made inside callbacks
                e.nearfield2farfield(store_phase_ff=False)
                self._mark_device_fresh(["farfield", "amp_ff"])
                if callback is not None:
                    if callback(self):
                        break
                    self._get_engine()
                self._update_stats(self.flags["stat_groups"])
                self._constraint_step(e)
                e.farfield2nearfield()
                self._mark_device_fresh(["phase"])
                self.iter += 1
        # _populate_results (:934-949) runs when somebody asks for farfield / amp_ff / phase_ff, when the phase is
        # replaced, or ahead of the next call that depends on it -- not before optimize() returns
        self._populate_pending = True

    def _flush_populate(self):
        if self.__dict__.get("_populate_pending"):
            self._populate_pending = False
            if self._engine is not None:
                self._populate_results()

    def _constraint_step(self, e):
        """_gs_farfield_routines (:1550-1661) of the general path, after _update_stats of this iteration."""
        # the engine re-counts the history entry _update_stats just appended
        st = self._make_step(skip_last=True)
        if self._kim_efficiency_gate():
            # fix_phase_efficiency reached (:1560-1569): fix (and store the phase) right now
            st.fix_phase_iteration = 1
            st.false_run = 0
        e.farfield_constraint(st)
        self.flags["fixed_phase"] = bool(st.fixed_phase)
        self._mark_device_fresh(["weights", "phase_ff", "farfield"])

    def _pre_loop_checks(self):
        fb = self.flags.get("feedback", "computational")
        if fb in ("experimental", "experimental_spot"):
            raise NotImplementedError(f"Feedback '{fb}' needs camera hardware and is outside this build")
        if fb in ("computational_spot", "external_spot") and self._n_spots() == 0 and "WGS" in self.flags["method"]:
            raise ValueError(f"Feedback '{fb}' is specific to SpotHologram")
        # Reference quirk A12: with NaN in the target, a phase that is still flagged as fixed and no stored
        # phase_ff -- reset(reset_flags=False) after a WGS-Kim run -- the MRAF branch evaluates exp(1j * None)
        # (_hologram.py:1643: no "or self.phase_ff is None" as at :1601) and raises TypeError in the first iteration, unless
        # that iteration is a WGS one past iteration 0 (Kim stores the phase first, :1582; the other rules clear the flag,
        # :1585).  Same condition, a clear message, nothing touched yet.
        fl = self.flags
        if (fl.get("fixed_phase", False) and "phase_ff" not in self._stale and self._host.get("phase_ff") is None
                and not ("WGS" in fl["method"] and self.iter > 0) and self._mraf_enabled()):
            raise RuntimeError("fixed_phase is set but there is no stored phase_ff, and the target holds NaN (MRAF): the reference "
                               "fails here with a TypeError (_hologram.py:1643).  Clear the flag (reset(reset_flags=True) or "
                               "flags['fixed_phase'] = False) or run an iteration that stores the phase first.")

    def _kim_efficiency_gate(self):
        """fix_phase_efficiency branch of _gs_farfield_routines (:1560-1569), evaluated on the host."""
        fl = self.flags
        if not ("WGS" in fl["method"] and self.iter > 0 and "Kim" in fl["method"]):
            return False
        if fl.get("fix_phase_efficiency", None) is None or fl["fixed_phase"]:
            return False
        stats = self.stats["stats"]
        if len(stats) == 0:
            raise ValueError("Must track statistics to fix phase based on efficiency!")
        eff = stats[tuple(stats.keys())[-1]]["efficiency"][self.iter]
        return bool(eff > fl["fix_phase_efficiency"])

    def _populate_results(self):
        """_hologram.py:934-949: one more forward transform, amp_ff and phase_ff of the final phase."""
        e = self._get_engine()
        e.nearfield2farfield(store_phase_ff=True)
        self._mark_device_fresh(["farfield", "amp_ff", "phase_ff"])

    # mempool helpers of the reference have no meaning here
    @staticmethod
    def set_mempool_limit(device=0, size=None, fraction=None):
        raise ValueError("Cannot set mempool: the HIP engine owns its device memory explicitly.")

    @staticmethod
    def get_mempool_limit(device=0):
        raise ValueError("Cannot get mempool: the HIP engine owns its device memory explicitly.")

    _norm = staticmethod(_norm)


class FeedbackHologram(Hologram):
    """
    Reference: ``FeedbackHologram`` (_feedback.py:5-411).  On the optimize() path: the constructor plumbing (shape / amp
    from a cameraslm or SLM), camera-basis targets (``target_ij``, :meth:`update_target`, :meth:`ijcam_to_knmslm` --
    host-side set-up through the Fourier calibration) and the "computational" weight update.  Acquiring camera frames
    (:meth:`measure`, the "experimental" feedback modes) needs hardware and raises.
    """

    def __init__(self, shape, target_ij=None, cameraslm=None, null_region=None,
                 null_region_radius_frac=None, **kwargs):
        # `cameraslm` may be a FourierSLM (kept: its calibrations are used later) or a bare SLM (only its shape and
        # source amplitude are read, then it is dropped); without either, `amp` / `slm_shape` come from the keywords
        self.cameraslm = cameraslm
        geometry = None
        if cameraslm is not None:
            slm = getattr(cameraslm, "slm", cameraslm)
            if not (hasattr(slm, "_get_source_amplitude") and hasattr(slm, "shape")):
                raise ValueError("Expected a CameraSLM or SLM to be passed to cameraslm.")
            if slm is cameraslm:
                self.cameraslm = None
            amp, geometry = slm._get_source_amplitude(), slm.shape
        else:
            amp = kwargs.pop("amp", None)
        kwargs.setdefault("slm_shape", geometry)
        super().__init__(target=shape, amp=amp, **kwargs)

        self.img_ij = self.img_knm = None
        self.target_ij = None if target_ij is None else np.asarray(target_ij).astype(self.dtype)
        self._cam_points = None
        if self._has_fourier(self.cameraslm):
            self._cam_points = self._camera_outline_knm()
            if target_ij is not None:
                self.update_target(target_ij, null_region, null_region_radius_frac, reset_weights=True)

    @staticmethod
    def _has_fourier(cameraslm):
        return cameraslm is not None and "fourier" in getattr(cameraslm, "calibrations", {})

    def _camera_outline_knm(self):
        """The sensor's corner pixels (closed loop, five points) in "knm": where the camera looks in the computational
        grid (_feedback.py:108-125; the reference keeps it for plotting)."""
        h, w = self.cameraslm.cam.shape
        loop_ij = np.array([[0, 0, w - 1, w - 1, 0], [0, h - 1, h - 1, 0, 0]], dtype=float)
        return toolbox.convert_vector(self.cameraslm.ijcam_to_kxyslm(loop_ij), "kxy", "knm", self.cameraslm.slm, self.shape)

    def _knm_to_ij_affine(self):
        """
        ``(A, c)`` with ``ij = A @ knm + c`` (column vectors, x first): the scaling "knm" -> "kxy" of this hologram's
        grid followed by the Fourier calibration ``ij = M (kxy - a) + b``.
        """
        slm = self.cameraslm.slm
        step = np.ravel(toolbox.convert_vector((1, 1), "knm", "kxy", slm, self.shape)
                        - toolbox.convert_vector((0, 0), "knm", "kxy", slm, self.shape))
        scale = np.diag(step)
        centre = np.array([[self.shape[1] / 2], [self.shape[0] / 2]], dtype=float)
        cal = self.cameraslm.calibrations["fourier"]
        M = np.array(cal["M"], dtype=float)
        shift = np.array(cal["b"], dtype=float).reshape(2, 1)
        if "a" in cal:
            shift = shift - M @ np.array(cal["a"], dtype=float).reshape(2, 1)
        return M @ scale, M @ (scale @ -centre) + shift

    def ijcam_to_knmslm(self, img, out=None, blur_ij=None, order=3):
        """
        A camera image resampled onto the computational grid (_feedback.py:141-233): amplitude |img| (after an optional
        Gaussian blur of ``blur_ij`` camera pixels, default the ``"blur_ij"`` flag or none) interpolated at the camera
        position of every "knm" pixel with splines of ``order``, NaN where the camera does not look, unit L2 norm over the
        defined pixels.  Host-side set-up (scipy), like the reference without CuPy.
        """
        from scipy import ndimage
        if self.cameraslm is None:
            raise RuntimeError("Cannot use ijcam_to_knmslm without the calibrations in a cameraslm.")
        if not self._has_fourier(self.cameraslm):
            raise RuntimeError("ijcam_to_knmslm requires a Fourier calibration.")
        A, c = self._knm_to_ij_affine()
        swap = np.array([[0.0, 1.0], [1.0, 0.0]])          # array axes are (row, column) = (y, x)
        width = self.flags.get("blur_ij", 0) if blur_ij is None else blur_ij
        if width > 0:
            img = ndimage.gaussian_filter(img, (width, width), output=img, truncate=2)
        source = np.abs(np.array(img, dtype=self.dtype))
        grid = ndimage.affine_transform(source, swap @ A @ swap, offset=np.ravel(swap @ c), output_shape=self.shape,
                                        order=order, output=out, mode="constant", cval=np.nan)
        np.abs(grid, out=grid)
        power = _norm(grid)
        if power == 0:
            raise ValueError("No power in hologram. Maybe target_ij is out of range of knm space? Check transformations.")
        grid *= 1 / power
        return grid

    def update_target(self, new_target_ij, null_region=None, null_region_radius_frac=None, reset_weights=False):
        """
        A new camera-basis target (_feedback.py:277-329): resampled to "knm" with nearest-neighbour lookup (no NaN
        bleeding).  Pixels the camera cannot see are undefined; they become zero (dark) inside the null region and stay
        NaN (free, MRAF noise region) elsewhere.  The null region is ``null_region`` plus everything beyond
        ``null_region_radius_frac`` of the grid's half-extent -- but with the fraction absent or >= 1 the reference zeroes
        EVERY undefined pixel and ignores ``null_region``; kept.
        """
        self.target_ij = np.asarray(new_target_ij).astype(self.dtype)
        raster = self.ijcam_to_knmslm(new_target_ij, order=0)
        unseen = np.isnan(raster)
        fraction = 1 if null_region_radius_frac is None else null_region_radius_frac
        if fraction >= 1:
            raster[unseen] = 0
        else:
            if null_region is None:
                null_region = np.zeros(self.shape, dtype=bool)
            rows, cols = np.shape(null_region)
            u, v = np.meshgrid(np.linspace(-1, 1, cols), np.linspace(-1, 1, rows))
            null_region[u * u + v * v > fraction ** 2] = True          # (the caller's array is extended, as in the reference)
            raster[unseen & np.asarray(null_region, dtype=bool)] = 0
        if not reset_weights:
            self._materialise_weights()
            self._weights_reset = False
        self.target = raster
        if self._engine is not None:
            self._upload_target(self._engine)
        if reset_weights:
            self.reset_weights()

    def measure(self, basis="ij"):
        raise NotImplementedError("measure() needs camera hardware and is outside this build")


class SpotHologram(FeedbackHologram):
    """
    Optical focus arrays, one DFT pixel per spot (reference: ``SpotHologram``,
    _spots.py:1020-1697).  Feedback modes on the GPU: "computational" (pixel-wise),
    "computational_spot" (w x w window integration around each spot) and "external_spot".
    """

    def __init__(self, shape, spot_vectors, basis="kxy", spot_amp=None, cameraslm=None,
                 null_vectors=None, null_radius=None, null_region=None,
                 null_region_radius_frac=None, **kwargs):
        """
        Constructor of the reference (_spots.py:1160-1373), in five steps: amplitudes; the spots in all three bases
        (``_resolve_bases``); the null points and region in ``"knm"`` (``_resolve_nulls``); integration widths and bounds
        (``_integration_widths``, ``_check_bounds``); then the ``FeedbackHologram`` set-up and the target raster.
        """
        vectors = toolbox.format_2vectors(spot_vectors)
        count = vectors.shape[1]
        if spot_amp is None:
            self.spot_amp = np.full(count, 1.0 / np.sqrt(count))
        else:
            self.spot_amp = np.ravel(spot_amp)
            if len(self.spot_amp) != count:
                raise ValueError("spot_amp must have the same length as the provided spots.")
        self.external_spot_amp = np.copy(self.spot_amp)

        self._resolve_bases(vectors, "knm" if basis is None else basis, cameraslm, shape)
        self._resolve_nulls(null_vectors, null_radius, null_region, "knm" if basis is None else basis, cameraslm, shape)
        self._integration_widths(cameraslm, shape)
        self._check_bounds(cameraslm, shape)

        super().__init__(shape, target_ij=None, cameraslm=cameraslm, **kwargs)

        if basis == "ij" and null_region is not None:
            # a region drawn on the camera: resampled like a target (nearest neighbour); what the camera cannot see is
            # NaN there and counts as null too (_spots.py:1352-1357)
            self.null_region_knm = self.ijcam_to_knmslm(null_region, order=0) != 0
        if null_region_radius_frac is not None:
            self._null_outside_radius(null_region_radius_frac)
        self.set_target(reset_weights=True)

    def _resolve_bases(self, vectors, basis, cameraslm, shape):
        """``spot_knm`` / ``spot_kxy`` / ``spot_ij`` from vectors given in ``basis``; what cannot be derived (no SLM
        for angles, no Fourier calibration for camera pixels) stays None -- except that "kxy" and "ij" input REQUIRE it."""
        have_cal = self._has_fourier(cameraslm)
        if basis == "knm":
            knm = vectors
            kxy = None if cameraslm is None else toolbox.convert_vector(knm, "knm", "kxy", cameraslm, shape)
            ij = cameraslm.kxyslm_to_ijcam(kxy) if have_cal else None
        elif basis == "kxy":
            assert cameraslm is not None, "We need a cameraslm to interpret kxy."
            kxy = vectors
            ij = cameraslm.kxyslm_to_ijcam(kxy) if have_cal else None
            knm = toolbox.convert_vector(kxy, "kxy", "knm", cameraslm, shape)
        elif basis == "ij":
            assert cameraslm is not None, "We need an cameraslm to interpret ij."
            assert have_cal, "We need a fourier-calibrated cameraslm to interpret ij."
            ij = vectors
            kxy = cameraslm.ijcam_to_kxyslm(ij)
            knm = toolbox.convert_vector(ij, "ij", "knm", cameraslm, shape)
        else:
            raise Exception("Unrecognized basis for spots '{}'.".format(basis))
        self.spot_knm, self.spot_kxy, self.spot_ij = knm, kxy, ij

    def _resolve_nulls(self, null_vectors, null_radius, null_region, basis, cameraslm, shape):
        """Null points and their radius in "knm" (converted when the spots came in another basis; the region is a "knm"
        mask either way).  A missing radius is a quarter of the closest approach among nulls and spots; radii are whole
        pixels, rounded up (_spots.py:1341-1352)."""
        self.null_region_knm = null_region
        self.null_knm = self.null_radius_knm = None
        if null_vectors is None:
            if basis == "knm":
                self.null_radius_knm = null_radius
        else:
            points = toolbox.format_2vectors(null_vectors)
            if basis == "knm":
                self.null_knm, radius = points, null_radius
            else:
                self.null_knm = toolbox.convert_vector(points, basis, "knm", cameraslm, shape)
                radius = None if null_radius is None else toolbox.convert_radius(null_radius, basis, "knm", cameraslm, shape)
            if radius is None:
                radius = toolbox.smallest_distance(np.hstack((self.null_knm, self.spot_knm))) / 4
            self.null_radius_knm = int(np.ceil(radius))

    _MIN_WINDOW = 3

    @classmethod
    def _odd_window(cls, psf, vectors):
        """Odd integration width: ten point-spread radii, at least _MIN_WINDOW, at most 2/3 of the closest spot pair."""
        ceiling = max(toolbox.smallest_distance(vectors) / 1.5, cls._MIN_WINDOW)
        return int(2 * np.floor(np.clip(10 * psf, cls._MIN_WINDOW, ceiling) / 2) + 1)

    def _integration_widths(self, cameraslm, shape):
        """``spot_integration_width_knm`` / ``_ij`` (_spots.py:1270-1306) from the SLM's diffraction-limited spot radius,
        where an SLM that can state one is at hand (otherwise the minimum width)."""
        slm = getattr(cameraslm, "slm", cameraslm)
        psf = {"knm": 0, "ij": 0}
        if slm is not None and hasattr(slm, "get_spot_radius_kxy"):
            radius_kxy = np.mean(slm.get_spot_radius_kxy())
            wanted = [("knm", slm)] + ([("ij", cameraslm)] if self.spot_ij is not None else [])
            for name, converter in wanted:
                value = toolbox.convert_radius(radius_kxy, "kxy", name, converter, shape)
                psf[name] = 0 if np.isnan(value) else value
        self.spot_integration_width_knm = self._odd_window(psf["knm"], self.spot_knm)
        self.spot_integration_width_ij = None if self.spot_ij is None else self._odd_window(psf["ij"], self.spot_ij)

    def _check_bounds(self, cameraslm, shape):
        """Spots must sit on the computational grid, and -- when a camera is known -- at least half an integration
        window inside its sensor (_spots.py:1308-1339)."""
        x, y = self.spot_knm[0], self.spot_knm[1]
        if np.any((x < 0) | (y < 0) | (x >= shape[1]) | (y >= shape[0])):
            raise ValueError("Spots outside SLM computational space bounds!\nSpots:\n{}\nBounds: {}".format(self.spot_knm, shape))
        cam = getattr(cameraslm, "cam", None)
        if self.spot_ij is not None and cam is not None:
            margin = self.spot_integration_width_ij / 2
            i, j = self.spot_ij[0], self.spot_ij[1]
            if np.any((i < margin) | (j < margin) | (i >= cam.shape[1] - margin) | (j >= cam.shape[0] - margin)):
                raise ValueError("Spots outside camera bounds!\nSpots:\n{}\nBounds: {}".format(self.spot_ij, cam.shape))

    def _null_outside_radius(self, fraction):
        """``null_region_radius_frac`` (_spots.py:1361-1371): everything farther than ``fraction`` of the half-extent
        from the centre joins the null region (normalised coordinates in [-1, 1] per axis; the reference lays the two
        axes out in the order that only square shapes support, kept)."""
        if self.null_region_knm is None:
            self.null_region_knm = np.zeros(self.shape, dtype=bool)
        rows, cols = self.null_region_knm.shape
        u, v = np.meshgrid(np.linspace(-1, 1, rows), np.linspace(-1, 1, cols))
        self.null_region_knm[u * u + v * v > fraction ** 2] = True

    def __len__(self):
        return self.spot_knm.shape[1]

    def _n_spots(self):
        return self.spot_knm.shape[1]

    @staticmethod
    def make_rectangular_array(shape, array_shape, array_pitch, array_center=None, basis="knm",
                               orientation_check=False, **kwargs):
        """
        A rectangular lattice of spots (_spots.py:1387-1488): ``array_shape`` = (columns, rows) spots ``array_pitch`` apart
        around ``array_center`` (default: the zero order of ``basis``), listed row by row; ``orientation_check`` drops the
        last two so that the pattern shows which way is up.
        """
        def pair(value, cast=lambda v: v):
            return (cast(value), cast(value)) if isinstance(value, toolbox.REAL_TYPES) else value

        (nx, ny), (px, py) = pair(array_shape, int), pair(array_pitch)
        if array_center is None:
            if basis == "ij":
                cameraslm = kwargs.get("cameraslm", None)
                assert cameraslm is not None, "We need an cameraslm to interpret ij."
                array_center = toolbox.convert_vector((0, 0), "kxy", "ij", cameraslm)
            else:
                array_center = {"knm": (shape[1] / 2.0, shape[0] / 2.0), "kxy": (0, 0)}.get(basis)
        xs = (np.arange(nx) - (nx - 1) / 2.0) * px + array_center[0]
        ys = (np.arange(ny) - (ny - 1) / 2.0) * py + array_center[1]
        lattice = np.vstack([axis.ravel() for axis in np.meshgrid(xs, ys)])
        if orientation_check and lattice.shape[1] > 2:
            lattice = lattice[:, :-2]
        return SpotHologram(shape, lattice, basis=basis, spot_amp=None, **kwargs)

    def _set_target_spots(self, reset_weights=False):
        """
        The target raster of the spot list (_spots.py:1490-1546): ``spot_amp`` at the rounded "knm" positions, zero
        elsewhere -- or, with null points, NaN (free) elsewhere except the null region and a disk of the null radius around
        every null point AND every spot (zero); unit norm.  Also refreshes the rounded positions in the other bases.
        """
        nearest = self.spot_knm_rounded = np.rint(self.spot_knm).astype(int)
        self.spot_kxy_rounded = self.spot_ij_rounded = None
        if self.cameraslm is not None:
            self.spot_kxy_rounded = toolbox.convert_vector(nearest, "knm", "kxy", self.cameraslm.slm, self.shape)
            if self._has_fourier(self.cameraslm):
                self.spot_ij_rounded = self.cameraslm.kxyslm_to_ijcam(self.spot_kxy_rounded)
        if self.null_knm is None:
            raster = np.zeros(self.shape, dtype=self.dtype)
        else:
            raster = np.full(self.shape, np.nan, dtype=self.dtype)
            if self.null_region_knm is not None:
                raster[np.asarray(self.null_region_knm, dtype=bool)] = 0
            diameter = int(2 * self.null_radius_knm + 1)
            for x, y in np.rint(np.hstack((self.null_knm, self.spot_knm))).T:
                toolbox.imprint_disk_zero(raster, x, y, diameter)
        raster[nearest[1], nearest[0]] = self.spot_amp
        # (without null points the raster holds no NaN: nansum's NaN-free copy of it -- a third of this call at 4096^2 -- is the
        #  array itself, and np.sum over it gives nansum's bits)
        raster /= _norm(raster) if self.null_knm is not None else np.sqrt(np.sum(np.square(raster)))
        if not reset_weights:
            self._materialise_weights()
            self._weights_reset = False
        self.target = raster
        self._spot_raster = True           # the raster IS the spot list (cleared by any other assignment of .target)
        if self._engine is not None:
            self._engine_setup(self._engine)
        self._target_changed(reset_weights)

    def set_target(self, reset_weights=False, plot=False):
        self._set_target_spots(reset_weights)

    # assigning ``target`` by hand (or through the base class's _set_target) makes it an arbitrary raster again
    @property
    def target(self):
        return self.__dict__.get("_target")

    @target.setter
    def target(self, value):
        Hologram.target.fset(self, value)
        self._spot_raster = False

    def _sparse_target(self):
        """True when the target is exactly what _set_target_spots laid out without null points: zeros and the spot list."""
        return self.__dict__.get("_spot_raster", False) and self.null_knm is None

    def _upload_target(self, e):
        """One value per spot instead of the padded raster (hgs_set_array_sparse) -- unless null points give the target
        a NaN background, or somebody replaced the raster (then it goes up whole, like any Hologram's)."""
        if not self._sparse_target():
            return super()._upload_target(e)
        ky, kx = self.spot_knm_rounded[1], self.spot_knm_rounded[0]
        e.set_sparse(L.TARGET, self.spot_knm_rounded, self.target[ky, kx])

    def _engine_setup(self, e):
        e.set(L.SPOT_INDEX, self.spot_knm_rounded)
        e.set(L.SPOT_AMP, self.spot_amp)
        e.set(L.EXTERNAL_AMP, self.external_spot_amp)

    def _spot_window(self):
        return self.spot_integration_width_knm

    def _mraf_enabled(self):
        """NaN in the target (_mraf_helper_routines :1498) without a 67 MB reduction: the raster is zeros, the spot
        amplitudes and -- only with null points -- a NaN background."""
        if not self._sparse_target():
            return super()._mraf_enabled()
        return bool(np.isnan(np.sum(self.target[self.spot_knm_rounded[1], self.spot_knm_rounded[0]])))

    def _pre_loop_checks(self):
        if self.flags.get("feedback") == "experimental":
            warnings.warn("SpotHologram feedback 'experimental' is interpreted as 'experimental_spot'")
            self.flags["feedback"] = "experimental_spot"
        super()._pre_loop_checks()
        if self.flags.get("feedback") == "external_spot":
            self._engine.set(L.EXTERNAL_AMP, self.external_spot_amp)

    def _device_stat_groups(self):
        groups = [g for g in ("computational", "computational_spot") if g in self.flags["stat_groups"]]
        if tuple(self.shape) == tuple(self.slm_shape):
            return groups, 1, self.spot_knm_rounded
        return groups, self.spot_integration_width_knm, self.spot_knm

    def _calculate_stats_computational_spot(self, stats, stat_groups=[]):
        """_spots.py:1626-1679."""
        if "computational_spot" in stat_groups:
            e = self._get_engine()
            if tuple(self.shape) == tuple(self.slm_shape):
                width, vectors = 1, self.spot_knm_rounded
            else:
                width, vectors = self.spot_integration_width_knm, self.spot_knm
            stats["computational_spot"] = e.stats(1, width, vectors)[0]
            if self.flags.get("raw_stats", False):
                v = np.floor(np.asarray(vectors, dtype=float)).astype(int)       # analysis.take floors (quirk A18)
                off = np.floor(np.arange(width) - (width - 1) / 2.0).astype(int)
                ox, oy = np.meshgrid(off, off)
                pwr = np.square(self.amp_ff)[oy.ravel()[None, :] + v[1][:, None], ox.ravel()[None, :] + v[0][:, None]]
                fb = np.sqrt(np.sum(pwr.astype(float), axis=-1))
                stats["computational_spot"].update(self._raw_stats(fb, self.spot_amp))

    def _update_stats(self, stat_groups=[]):
        stats = {}
        self._calculate_stats_computational(stats, stat_groups)
        self._calculate_stats_computational_spot(stats, stat_groups)
        self._update_stats_dictionary(stats)


__all__ = ["Hologram", "FeedbackHologram", "SpotHologram", "ALGORITHM_DEFAULTS", "ALGORITHM_INDEX",
           "FEEDBACK_OPTIONS"]


class CompressedSpotHologram(FeedbackHologram):
    """
    Kernel-based focal arrays: N free-floating spots, each with its own Zernike phase kernel, no
    padded DFT grid (reference: ``CompressedSpotHologram``, _spots.py:178-1018).  The farfield is the
    N-vector of spot amplitudes; both directions of the non-uniform DFT are regenerated on the fly on
    the GPU (no kernel cache, no 256-spot batching).

    ``cameraslm`` is duck-typed: it needs ``.slm`` with ``shape``, ``grid`` (x/lambda meshgrids),
    ``pitch``, ``get_source_zernike_scaling()`` and ``_get_source_amplitude()``
    (``slmsuite_amd.hardware.SimpleFourierSLM`` is a minimal stand-in).
    """

    def __init__(self, spot_vectors, basis="kxy", spot_amp=None, cameraslm=None, cuda=False, **kwargs):
        if cameraslm is None:
            raise ValueError("cameraslm must be passed.")
        vectors = toolbox.format_vectors(spot_vectors)
        rank, count = vectors.shape
        self.spot_amp = np.full(count, 1.0 / np.sqrt(count)) if spot_amp is None else np.array(spot_amp)
        if self.spot_amp.size != count:
            raise ValueError(f"spot_amp (length {self.spot_amp.size}) must have the same length as the provided spots ({count}).")

        units = self._parse_basis(basis, rank)
        self._resolve_vectors(vectors, units, cameraslm)
        slm = getattr(cameraslm, "slm", cameraslm)
        if np.any(np.abs(self.spot_kxy[:2]) > 1.1 * (0.5 / np.min(slm.pitch))):      # 10 % beyond the Nyquist angle
            raise ValueError("Spots laterally outside the bounds of the farfield")
        self._camera_geometry(cameraslm)

        # the base classes work on a padded grid; until the N-vector target exists they see a placeholder (_set_target)
        self._compressed_ready = False
        FeedbackHologram.__init__(self, None, None, cameraslm, **kwargs)
        self.shape = self.slm_shape                      # no DFT grid: the "shape" of this hologram is the SLM's
        self.set_target(self.spot_amp, reset_weights=True)
        # quirk: the reference ends its constructor with reset() (_spots.py:495), which replaces any
        # phase passed by the caller with a fresh random one; reproduce, then honour reset_phase().
        self.reset()
        self.external_spot_amp, self.cuda = np.ones(self.target.shape), False
        # pixel coordinates in the units the Zernike polynomials are defined on (unit disk = the source radius)
        stretch = slm.get_source_zernike_scaling()
        stretch = (stretch, stretch) if np.isscalar(stretch) else tuple(stretch)
        self._xg, self._yg = (np.asarray(axis, dtype=float) * k for axis, k in zip(toolbox.process_grid(slm), stretch))
        self._spot_zernike_cached = None
        self._compressed_ready = True

    def _parse_basis(self, basis, rank):
        """
        ``zernike_basis`` (ANSI indices of the ``rank`` coefficient rows) and where in it the Cartesian terms sit.  A string
        names the units of the vectors and implies the default basis [2, 1, 4, 3, 5, ...]; a list of indices IS the basis
        and the vectors are Zernike coefficients.  Returns the units.  (_spots.py:366-396)
        """
        if isinstance(basis, str):
            self.zernike_basis, units = toolbox.zernike_indices_parse(None, rank), basis
        else:
            self.zernike_basis, units = np.ravel(basis), "zernike"
            if len(self.zernike_basis) != rank:
                raise ValueError(f"zernike_basis (length {len(self.zernike_basis)}) must have the same "
                                 f"dimension as the provided spots ({rank}).")
            if 0 in self.zernike_basis:
                warnings.warn("Found ANSI index '0' (Zernike piston) in the zernike_basis; this is not necessary.")
        where = {int(j): k for k, j in reversed(list(enumerate(self.zernike_basis)))}       # first occurrence wins
        if 2 not in where or 1 not in where:
            raise ValueError("Compressed basis must include x, y (Zernike ANSI indices 2, 1)")
        self.zernike_basis_cartesian = np.array([where[j] for j in (2, 1, 4) if j in where])
        return units

    def _resolve_vectors(self, vectors, units, cameraslm):
        """``spot_zernike`` (what the kernels are built from) and ``spot_kxy`` (tilts and, if present, focus)."""
        if units == "zernike":
            self.spot_zernike = np.array(vectors, dtype=float)
            _, self.spot_kxy = toolbox.convert_vector_zernike(vectors[self.zernike_basis_cartesian], "zernike", cameraslm)
            return
        if units == "ij":          # camera pixels (and pixel depth): through the Fourier calibration
            if not self._has_fourier(cameraslm):
                raise RuntimeError("Fourier calibration must exist to interpret the 'ij' basis.")
            vectors, units = cameraslm.ijcam_to_kxyslm(vectors), "kxy"
        if units not in ("kxy", "norm"):
            raise NotImplementedError(f"basis '{units}' is not supported; use 'kxy', 'ij' or Zernike indices")
        self.spot_zernike, self.spot_kxy = toolbox.convert_vector_zernike(vectors, "kxy", cameraslm)

    def _camera_geometry(self, cameraslm):
        """Where the spots land on the camera, with a Fourier calibration at hand (_spots.py:433-483): ``spot_ij``, the odd
        integration width (two point-spread radii, at least 3, at most 2/3 of the closest pair) and the sensor bounds."""
        self.spot_ij = self.spot_integration_width_ij = None
        cam = getattr(cameraslm, "cam", None)
        if not self._has_fourier(cameraslm) or cam is None:
            return
        self.spot_ij = cameraslm.kxyslm_to_ijcam(self.spot_kxy)
        psf = toolbox.convert_radius(np.mean(cameraslm.slm.get_spot_radius_kxy()), "kxy", "ij", cameraslm)
        psf = 0 if np.isnan(psf) else psf
        ceiling = max(toolbox.smallest_distance(self.spot_ij[:2]) / 1.5, 3)
        if psf > ceiling:
            warnings.warn("The expected camera spot point-spread-function is too large. Clipping to a smaller one.")
        width = self.spot_integration_width_ij = int(2 * np.floor(np.clip(2 * psf, 3, ceiling) / 2) + 1)
        i, j = self.spot_ij[0], self.spot_ij[1]
        if np.any((i < width / 2) | (j < width / 2) | (i >= cam.shape[1] - width / 2) | (j >= cam.shape[0] - width / 2)):
            raise ValueError("Spots outside camera bounds!\nSpots:\n{}\nBounds: {}".format(self.spot_ij, cam.shape))

    def __len__(self):
        return self.spot_amp.size

    def _n_spots(self):
        return len(self)

    def get_padded_shape(self, *args, **kwargs):
        raise NameError("CompressedSpotHologram does not use a DFT grid and does not need padding.")

    # the base-class target plumbing works on the padded grid; here the target is the N-vector
    def _set_target(self, new_target, reset_weights=False):
        if not getattr(self, "_compressed_ready", False) and (new_target is None or np.size(new_target) == 0
                                                               or np.ndim(new_target) != 1):
            self.target = np.zeros((len(self),), dtype=self.dtype)   # placeholder during base construction
            return
        self.set_target(new_target, reset_weights)

    def set_target(self, new_target=None, reset_weights=False):
        """_spots.py:917-947."""
        if new_target is None:
            target = np.array(self.spot_amp, dtype=self.dtype)
        else:
            new_target = np.squeeze(np.ravel(new_target))
            if new_target.shape != (len(self),):
                raise ValueError("Target must be of appropriate shape. Initialize a new Hologram if a "
                                 "different shape is desired.")
            target = np.array(new_target, dtype=self.dtype)
            self.spot_amp = np.array(new_target, dtype=self.dtype)
        np.abs(target, out=target)
        target *= 1 / _norm(target)
        if not reset_weights:
            self._materialise_weights()
            self._weights_reset = False
        self.target = target
        if self._engine is not None:
            self._engine.set(L.TARGET, self.target)
        if reset_weights:
            self.reset_weights()

    def reset(self, reset_phase=True, reset_flags=False):
        super().reset(reset_phase=reset_phase, reset_flags=reset_flags)
        self._host["farfield"] = np.zeros(self.target.shape, dtype=self.dtype_complex)

    @property
    def nearfield(self):
        ph = self.phase if self.propagation_kernel is None else self.phase + self.propagation_kernel
        return (self.amp * np.exp(1j * ph)).astype(self.dtype_complex)

    def _check_spot_zernike_change(self):
        """_spots.py:638-650: whole-array comparison against the cached copy."""
        changed = self._spot_zernike_cached is None or np.any(self._spot_zernike_cached != self.spot_zernike)
        if changed:
            self._spot_zernike_cached = self.spot_zernike.copy()
        return changed

    def _push_kernels(self, e):
        terms, w = toolbox.zernike_monomial_weights(self.zernike_basis, self.spot_zernike)
        if terms.shape[0] != e.n_monomials:
            raise ValueError("the monomial set of the kernels changed; build a new hologram")
        e.set(L.MONOMIALS, terms)
        e.set(L.SPOT_COEFF, w)

    def _get_engine(self):
        e = self._engine
        if e is None:
            terms, _ = toolbox.zernike_monomial_weights(self.zernike_basis, self.spot_zernike)
            e = self._engine = Engine(self.slm_shape, self.slm_shape, self.dtype, batch=1, n_spots=len(self),
                                      kind=1, n_monomials=terms.shape[0])
            for opt, val in self.engine_options.items():
                e.set_option(opt, val)
            if np.isscalar(self.amp):
                e.set(L.AMP_SCALAR, np.array([self.amp], dtype=self.dtype))
            else:
                e.set(L.AMP, self.amp)
            if self.propagation_kernel is not None:
                e.set(L.PROP_KERNEL, self.propagation_kernel)
            e.set(L.XGRID, self._xg)
            e.set(L.YGRID, self._yg)
            e.set(L.TARGET, self.target)
            self._spot_zernike_cached = None
            self._materialise_weights()
            self._upload |= {"phase", "weights"}
            if self._host.get("phase_ff") is not None:
                self._upload.add("phase_ff")
        if self._check_spot_zernike_change():
            self._push_kernels(e)
        for name in list(self._upload):
            if self._host.get(name) is not None:
                e.set(_DEVICE_ARRAYS[name], self._host[name])
            self._upload.discard(name)
        return e

    def _make_step(self, skip_last=False, efficiency_group=None):
        fl = dict(self.flags)
        if fl.get("feedback") in ("computational", "computational_spot"):
            fl["feedback"] = "computational"          # the N-vector rule on amp_ff
        return make_step(fl, self.iter, false_run=self._false_run(skip_last),
                         mraf_enabled=self._mraf_enabled(), spot_window=1)

    def _pre_loop_checks(self):
        fb = self.flags.get("feedback", "computational")
        if fb == "computational":                      # _spots.py:957-959
            self.flags["feedback"] = "computational_spot"
        if fb == "experimental":
            warnings.warn("CompressedSpotHologram feedback 'experimental' is interpreted as 'experimental_spot'")
            self.flags["feedback"] = "experimental_spot"
        if self.flags["feedback"] in ("experimental", "experimental_spot"):
            raise NotImplementedError("experimental feedback needs camera hardware and is outside this build")
        if self.flags["feedback"] == "external_spot":
            self._engine.set(L.EXTERNAL_AMP, np.asarray(self.external_spot_amp, dtype=float))

    def _device_stat_groups(self):
        # the reference records no computational statistics for this class (_spots.py:1004-1018)
        return [], 1, None

    def _update_stats(self, stat_groups=[]):
        """_spots.py:1004-1018: the computational_spot group is disabled in the reference."""
        self._update_stats_dictionary({})

    def get_farfield(self, *args, **kwargs):
        raise NotImplementedError("CompressedSpotHologram has no DFT-grid farfield; see .farfield for the spots")


class MultiplaneHologram(Hologram):
    """
    Several holograms -- other planes of focus, other point sets -- optimised together into one
    phase mask (reference: ``MultiplaneHologram``, _multiplane.py:8-289).  The children keep their own
    engines, targets, weights, flags history and statistics; per iteration each child transforms the
    shared phase forward and applies its own constraint, then the engine sums the children's
    complex nearfields on the GPU (``hgs_multiplane_farfield2nearfield``) into the new common phase.

    ``weights`` (one per child, L2-normalised) redistribute power between the children.
    """

    def __init__(self, holograms, weights=None):
        self.holograms = list(holograms)
        if len(self.holograms) == 0:
            raise ValueError("Multiplane hologram must be provided child holograms")
        for h in self.holograms:
            if "MultiplaneHologram" in str(type(h)):
                raise ValueError("Multiplane hologram recursion is not supported.")
            if "Hologram" not in str(type(h)):
                raise ValueError(f"Multiplane hologram must be provided child holograms, not {type(h)}")
        h0 = self.holograms[0]
        for h in self.holograms[1:]:
            if tuple(h.slm_shape) != tuple(h0.slm_shape) or h.dtype != h0.dtype:
                raise ValueError("All child holograms must share one SLM shape and precision")
        self._mp_ready = False
        # (the reference needs array-valued child amplitudes here; a scalar amp = uniform is accepted too)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")             # the fake target has the (non power-of-two) SLM shape
            super().__init__(target=h0.slm_shape,      # the parent has a fake target (:63-70)
                             amp=None if np.isscalar(h0.amp) else h0.amp, phase=h0.phase,
                             slm_shape=h0.slm_shape, dtype=h0.dtype)
        self.target = None
        self._mp_ready = True
        # the children point to the same data (:73-76)
        for h in self.holograms:
            h.amp = self.amp
            h._release_engine()                         # a new amplitude needs a fresh engine (state comes home first)
        self.phase = self._host["phase"]
        if weights is None:
            weights = np.ones(len(self), dtype=self.dtype)
        self.weights = np.array(weights, dtype=self.dtype)
        if self.weights.shape != (len(self),):
            raise ValueError("Expected one weight per child hologram")
        self.weights = self.weights / _norm(self.weights)

    def __len__(self):
        return len(self.holograms)

    # ---- the shared phase ---------------------------------------------------------------------------
    def _get_dev(self, name):
        if name == "phase" and "phase" in self._stale:
            self._host["phase"] = self.holograms[0].phase
            self._stale.discard("phase")
        return self._host.get(name)

    def _set_dev(self, name, value):
        self._host[name] = value
        self._stale.discard(name)
        if name == "phase" and value is not None and getattr(self, "_mp_ready", False):
            for h in self.holograms:
                h.phase = value

    phase = property(lambda s: s._get_dev("phase"), lambda s, v: s._set_dev("phase", v))
    weights = property(lambda s: s._get_dev("weights"), lambda s, v: s._set_dev("weights", v))

    def _get_engine(self):
        raise RuntimeError("MultiplaneHologram has no engine of its own; its children do")

    # ---- meta functionality (_multiplane.py:174-289) -----------------------------------------------------
    def _update_flags(self, method, verbose, feedback, stat_groups, **kwargs):
        super()._update_flags(method, verbose, feedback, stat_groups, **kwargs)
        for h in self.holograms:
            h.flags.update(self.flags)

    def reset(self, reset_phase=True, reset_flags=False):
        """The parent keeps the iteration counter, history and flags; every child starts over on the shared phase."""
        if not getattr(self, "_mp_ready", False):          # (still inside Hologram.__init__)
            return super().reset(reset_phase, reset_flags)
        if reset_phase or self._host.get("phase") is None:
            self.reset_phase()
        self.iter, self.stats = 0, dict(method=[], flags={}, stats={})
        if reset_flags:
            self.flags = dict(method="")
        for child in self.holograms:
            child.reset(reset_phase=False, reset_flags=reset_flags)
            child.phase = self._host["phase"]

    def reset_weights(self):
        if getattr(self, "_mp_ready", False):
            for h in self.holograms:
                h.reset_weights()
        else:
            super().reset_weights()

    def _update_stats(self, stat_groups=[]):
        for h in self.holograms:
            h._update_stats(stat_groups)

    def set_target(self, *args, **kwargs):
        raise RuntimeError("Do not use MultiplaneHologram.set_target(). "
                           "Instead, update the targets of the children holograms directly.")

    def get_farfield(self, *args, **kwargs):
        raise NotImplementedError("MultiplaneHologram has no farfield of its own; ask a child hologram")

    def optimize_gs(self, iterations, callback):
        """The loop of optimize_gs (:1465-1490) with the overloads of _multiplane.py:245-289."""
        for h in self.holograms:
            h._pre_loop_checks()
        for _ in iterations:
            # (A) every child populates its own farfield from the shared phase (:245-249)
            for h in self.holograms:
                h._get_engine().nearfield2farfield(store_phase_ff=False)
                h._mark_device_fresh(["farfield", "amp_ff"])
                h.iter = self.iter
            if callback is not None:
                if callback(self):
                    break
            self._update_stats(self.flags["stat_groups"])
            # (B) each child's own constraint and weight update (:281-283)
            for h in self.holograms:
                h._constraint_step(h._get_engine())
            # (C) weighted sum of the complex nearfields -> common phase (:251-276)
            Engine.multiplane_farfield2nearfield([h._get_engine() for h in self.holograms], self.weights)
            for h in self.holograms:
                h._mark_device_fresh(["phase"])
                h.iter = self.iter
            self._stale.add("phase")
            self.iter += 1
        self._populate_results()

    def _populate_results(self):
        """:934-949 with the overloaded _nearfield2farfield: the children's farfield and amp_ff are refreshed."""
        for h in self.holograms:
            h._get_engine().nearfield2farfield(store_phase_ff=False)
            h._mark_device_fresh(["farfield", "amp_ff"])
            h.iter = self.iter


__all__ += ["CompressedSpotHologram", "MultiplaneHologram"]
