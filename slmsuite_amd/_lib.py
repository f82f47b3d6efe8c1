"""
ctypes binding of libhgs.so (the C ABI declared in include/hgs.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C slmsuite_amd/csrc``.
There is deliberately NO fallback: if the shared library or a gfx950 device is missing, using
the engine raises -- the product path never computes on the CPU.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HGS_LIB", os.path.join(_HERE, "libhgs.so"))   # HGS_LIB: A/B builds only

HGS_OK, HGS_ERR_ARG, HGS_ERR_DEVICE, HGS_ERR_STATE, HGS_ERR_UNSUPPORTED = 0, -1, -2, -3, -4
# hgs_set_option
OPT_SPARSE_COLUMNS, OPT_FORCE_STEPWISE, OPT_TILE_KERNEL, OPT_SEPARABLE, OPT_SEPARABLE_MIN_SPOTS, OPT_ROCTX = 1, 2, 3, 4, 5, 6
OPT_RUN_KERNELS = 7
OPT_KEEP_PREV_PHASE = 8

# array selectors (include/hgs.h)
(PHASE, AMP, AMP_SCALAR, PROP_KERNEL, TARGET, WEIGHTS, PHASE_FF, FARFIELD, AMP_FF, SPOT_INDEX,
 SPOT_AMP, EXTERNAL_AMP, ZERO_WEIGHTS, XGRID, YGRID, MONOMIALS, SPOT_COEFF) = range(17)
PHASE_PREV = 17
FB_PIXEL, FB_SPOT_WINDOW, FB_EXTERNAL = 0, 1, 2
K_NAMES = ("row", "col_fused", "col_fwd", "col_inv", "elementwise")


class hgs_config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("device", "pad_h", "pad_w", "slm_h", "slm_w", "real_bytes", "batch", "n_spots", "kind", "n_monomials")]


class hgs_step(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("method", "feedback", "iter", "fixed_phase", "fix_phase_iteration", "false_run",
                 "mraf_enabled", "has_mraf_factor", "zero_mode", "spot_window", "efficiency_group", "reserved")] + \
               [(n, C.c_double) for n in
                ("feedback_exponent", "feedback_factor", "mraf_factor", "zero_factor", "fix_phase_efficiency")]


_lib = None


class HgsError(RuntimeError):
    pass


def _torch_runtime_first():
    """
    PyTorch-ROCm wheels carry their own copy of the HIP runtime (torch/lib/libamdhip64.so) next to the system one
    libhgs.so links (libamdhip64.so.7).  Both can live in one process -- the batch driver hands torch tensors to
    hgs_get_array_device -- but only when torch's copy has opened the GPU FIRST: after the system runtime did,
    ``torch.cuda`` reports "No HIP GPUs are available" (measured on the MI355X box, round 3).

    So a process that ALREADY imported torch gets torch's runtime brought up before libhgs.so is loaded.  A process that
    has not is left alone: a pure NumPy caller pays neither the import (seconds) nor a HIP context in a parent that may
    still fork workers.  Load order for mixed use: ``import torch`` before the first engine (slmsuite_amd.batch's
    distributed path and bench.py do); ``HGS_TORCH_INIT=1`` forces the old behaviour, ``HGS_SKIP_TORCH_INIT=1`` skips it.
    """
    import sys
    if os.environ.get("HGS_SKIP_TORCH_INIT") == "1":
        return
    if "torch" not in sys.modules and os.environ.get("HGS_TORCH_INIT") != "1":
        return
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:       # torch present but unusable: the engine does not need it
        pass


def load():
    """Load libhgs.so once and declare prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HgsError(
            f"{LIB_PATH} not found: build the HIP engine first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C slmsuite_amd/csrc). "
            "slmsuite_amd has no CPU fallback.")
    _torch_runtime_first()
    lib = C.CDLL(LIB_PATH)
    P = C.POINTER
    eng = C.c_void_p
    protos = {
        "hgs_create": (C.c_int, [P(hgs_config), P(eng)]),
        "hgs_destroy": (C.c_int, [eng]),
        "hgs_set_array": (C.c_int, [eng, C.c_int, C.c_void_p, C.c_size_t]),
        "hgs_get_array": (C.c_int, [eng, C.c_int, C.c_void_p, C.c_size_t]),
        "hgs_get_array_device": (C.c_int, [eng, C.c_int, C.c_void_p, C.c_size_t]),
        "hgs_set_array_device": (C.c_int, [eng, C.c_int, C.c_void_p, C.c_size_t]),
        "hgs_copy_phase": (C.c_int, [eng, eng]),
        "hgs_reset_weights": (C.c_int, [eng]),
        "hgs_reset": (C.c_int, [eng]),
        "hgs_set_array_sparse": (C.c_int, [eng, C.c_int, P(C.c_int32), C.c_void_p, C.c_int32]),
        "hgs_nearfield2farfield": (C.c_int, [eng, C.c_int]),
        "hgs_farfield_constraint": (C.c_int, [eng, P(hgs_step)]),
        "hgs_farfield2nearfield": (C.c_int, [eng]),
        "hgs_iterate": (C.c_int, [eng, P(hgs_step), C.c_int, P(C.c_uint8)]),
        "hgs_stats": (C.c_int, [eng, C.c_int, C.c_int, P(C.c_double), P(C.c_double)]),
        "hgs_multiplane_farfield2nearfield": (C.c_int, [P(eng), P(C.c_double), C.c_int]),
        "hgs_iterate_stats": (C.c_int, [eng, P(hgs_step), C.c_int, P(C.c_uint8), C.c_int, C.c_int,
                                        P(C.c_double), P(C.c_double)]),
        "hgs_sync": (C.c_int, [eng]),
        "hgs_set_option": (C.c_int, [eng, C.c_int, C.c_int]),
        "hgs_profile_enable": (C.c_int, [eng, C.c_int]),
        "hgs_profile_read": (C.c_int, [eng, P(C.c_double)]),
        "hgs_iterate_timed": (C.c_int, [eng, P(hgs_step), C.c_int, P(C.c_double)]),
        "hgs_dispatch_read": (C.c_int, [eng, C.c_char_p, C.c_size_t, P(C.c_size_t)]),
        "hgs_last_error": (C.c_char_p, []),
        "hgs_version": (C.c_char_p, []),
    }
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


EXPORTS = ("hgs_create", "hgs_destroy", "hgs_set_array", "hgs_get_array", "hgs_get_array_device", "hgs_set_array_device", "hgs_copy_phase",
           "hgs_reset_weights", "hgs_reset", "hgs_set_array_sparse", "hgs_nearfield2farfield", "hgs_farfield_constraint",
           "hgs_farfield2nearfield", "hgs_iterate", "hgs_iterate_stats", "hgs_stats", "hgs_multiplane_farfield2nearfield", "hgs_set_option", "hgs_sync", "hgs_profile_enable",
           "hgs_profile_read", "hgs_iterate_timed", "hgs_dispatch_read", "hgs_last_error", "hgs_version")


def check(code):
    """Map engine status codes onto the reference's exception types (ValueError / RuntimeError)."""
    if code == HGS_OK:
        return
    msg = load().hgs_last_error().decode(errors="replace")
    if code == HGS_ERR_ARG:
        raise ValueError(msg)
    if code == HGS_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise HgsError(msg)
