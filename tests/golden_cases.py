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


def spot_null_region(shape):
    """The blanket null region of the spotnull_* fixtures (tools/make_golden.py spot_null_region)."""
    m = np.zeros(shape, dtype=bool)
    m[:24, :] = m[-24:, :] = m[:, :24] = m[:, -24:] = True
    m[150:170, 40:90] = True
    return m


def spot_null_ctor(meta, gold):
    """Constructor keywords (null_vectors, null_radius, null_region, null_region_radius_frac) of a spotnull fixture."""
    kw = dict(null_vectors=np.array(gold["null_vectors"], dtype=float))
    for k, v in meta["ctor"].items():
        kw[k] = spot_null_region(tuple(meta["shape"])) if v == "spot_null_region" else v
    return kw


# ---- MultiplaneHologram fixtures (tools/make_golden.py gen_multiplane_cases) -----------------------------
MULTIPLANE_SLM = (48, 80)
MULTIPLANE_WEIGHTS = (1.0, 2.0, 0.5)


def multiplane_children(Hologram, SpotHologram, make_spots=None, dtype=np.float32):
    """
    The three children of the multiplane fixtures, built from the given classes (reference, product
    or oracle adapters): a dense image at another depth (propagation kernel), a spot array on a larger
    grid, a dense image on a non-square grid.  All share the 48x80 SLM and the seed phase 700.
    """
    slm = MULTIPLANE_SLM
    dt = np.dtype(dtype).type
    phase0 = synth.seed_phase(700, slm, dtype=dt)
    amp = synth.gaussian_amp(slm, dtype=dt)
    kern = (0.3 * synth.seed_phase(701, slm)).astype(dt)
    h1 = Hologram(synth.random_target(702, (128, 128), dtype=dt), amp=amp.copy(), phase=phase0.copy(),
                  slm_shape=slm, dtype=dt, propagation_kernel=kern)
    if make_spots is None:
        h2 = SpotHologram.make_rectangular_array((256, 256), array_shape=(6, 6), array_pitch=(20, 20), basis="knm",
                                                 slm_shape=slm, amp=amp.copy(), phase=phase0.copy(), dtype=dt)
    else:
        h2 = make_spots((256, 256), (6, 6), (20, 20), slm, amp.copy(), phase0.copy(), dt)
    h3 = Hologram(synth.random_target(703, (64, 128), dtype=dt), amp=amp.copy(), phase=phase0.copy(),
                  slm_shape=slm, dtype=dt)
    return [h1, h2, h3]
