"""Rebuild the exact inputs of each golden fixture from its meta (shared by CPU and GPU tests)."""
import numpy as np

from slmsuite_amd import synth


def mraf_target(seed, n=128, dtype=np.float32):
    t = np.zeros((n, n), dtype=dtype)
    a, b = n // 8, n - n // 8
    t[a:b, a:b] = np.nan
    a, b = n // 4, n - n // 4
    t[a:b, a:b] = synth.random_target(seed, (b - a, b - a), 0.2, 1.0, dtype=dtype)
    return t


def hologram_inputs(meta):
    """kwargs for a Hologram-like constructor (target, amp, phase, slm_shape, dtype, propagation_kernel)."""
    dt = np.dtype(meta["dtype"]).type
    shape, slm, seed = tuple(meta["shape"]), tuple(meta["slm_shape"]), meta["seed"]
    if meta["kind"] == "mraf":
        target = mraf_target(seed, shape[0], dt)
        amp = kernel = None
    else:
        target = synth.random_target(seed, shape, dtype=dt)
        amp = synth.gaussian_amp(slm, dtype=dt) if meta.get("amp") == "gaussian" else None
        kernel = None
        if meta.get("kernel_seed") is not None:
            kernel = (0.3 * synth.seed_phase(meta["kernel_seed"], slm)).astype(dt)
    return dict(target=target, amp=amp, phase=synth.seed_phase(seed, slm, dtype=dt),
                slm_shape=slm, dtype=dt, propagation_kernel=kernel)


def spot_external_amp(meta, spot_amp):
    return spot_amp * (1 + 0.2 * (synth.uniform01(meta["seed"], (len(spot_amp),), 5) - 0.5))
