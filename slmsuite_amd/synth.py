"""
Deterministic synthetic inputs (seed phases, random targets) for parity tests and benchmarks.

A counter-based generator (splitmix64 finaliser over ``seed * 2**32 + index``) written with plain
uint64 NumPy arithmetic, so the same seed gives bit-identical arrays on any NumPy version and on
the GPU box -- the reference's own default phase comes from an unseeded RNG
(_hologram.py:529-534), which is useless for parity, so every parity run passes ``phase=``.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return x ^ (x >> np.uint64(31))


def uniform01(seed, shape, stream=0):
    """float64 uniform in [0, 1) with 53 random bits, element i = hash(seed, stream, i)."""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        base = _splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream))
        idx = np.arange(n, dtype=np.uint64)
        bits = _splitmix64(idx ^ base)
    return ((bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))).reshape(shape)


def seed_phase(seed, slm_shape, dtype=np.float32):
    """Uniform phase in [-pi, pi)."""
    return ((uniform01(seed, slm_shape, stream=1) * 2.0 - 1.0) * np.pi).astype(dtype)


def random_target(seed, shape, lo=0.0, hi=1.0, dtype=np.float32):
    """Uniform(lo, hi) amplitude image."""
    return (lo + (hi - lo) * uniform01(seed, shape, stream=2)).astype(dtype)


def gaussian_amp(slm_shape, frac=0.35, dtype=np.float32):
    """Smooth non-uniform source amplitude (Gaussian beam, 1/e^2 radius = frac * min(shape))."""
    h, w = slm_shape
    y = (np.arange(h) - (h - 1) / 2.0)[:, None]
    x = (np.arange(w) - (w - 1) / 2.0)[None, :]
    r0 = frac * min(h, w)
    return np.exp(-(x * x + y * y) / (r0 * r0)).astype(dtype)


def random_pixels_target(seed, shape, n, dtype=np.float32):
    """``n`` distinct unit pixels (test_algorithms.py:86-119 style target)."""
    u = uniform01(seed, (shape[0] * shape[1],), stream=3)
    idx = np.argsort(u, kind="stable")[:n]
    t = np.zeros(shape[0] * shape[1], dtype=dtype)
    t[idx] = 1
    return t.reshape(shape)
