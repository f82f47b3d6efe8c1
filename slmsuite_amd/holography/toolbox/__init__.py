"""
Host-side helpers the hologram classes need at set-up time (index arithmetic and unit conversion).
They mirror the parts of ``slmsuite.holography.toolbox`` / ``analysis`` that sit on the optimize()
path; none of this runs per iteration.
"""
import numpy as np

REAL_TYPES = (int, float, np.integer, np.floating)


def format_shape(shape):
    """toolbox/__init__.py:1601-1618: a 2-tuple of positive ints."""
    shape = tuple(np.squeeze(np.asarray(shape)).tolist()) if not isinstance(shape, tuple) else shape
    if len(shape) != 2:
        raise ValueError(f"Expected a 2-tuple shape, got {shape}")
    for d in shape:
        if not isinstance(d, (int, np.integer)) or d <= 0:
            raise ValueError(f"Expected positive integer dimensions, got {shape}")
    return tuple(int(d) for d in shape)


def pad(matrix, shape):
    """Centred zero padding.  toolbox/__init__.py:1621-1662."""
    if shape is None:
        return matrix
    shape = format_shape(shape)
    dh = (shape[0] - matrix.shape[0]) / 2.0
    dw = (shape[1] - matrix.shape[1]) / 2.0
    if not (dh >= 0 and dw >= 0):
        raise ValueError(f"Shape {tuple(matrix.shape)} is too large to pad to shape {shape}")
    return np.pad(matrix, [(int(np.floor(dh)), int(np.ceil(dh))), (int(np.floor(dw)), int(np.ceil(dw)))],
                  mode="constant", constant_values=0)


def unpad(matrix, shape):
    """
    Centred crop, or the four slicing integers when ``matrix`` is itself a shape.
    toolbox/__init__.py:1665-1719.
    """
    mshape = np.shape(matrix)
    return_args = False
    if len(mshape) == 1 or np.prod(mshape) == 2:
        mshape = format_shape(tuple(int(x) for x in np.ravel(matrix)))
        return_args = True
    if shape is None:
        return (0, mshape[0], 0, mshape[1]) if return_args else matrix
    shape = format_shape(shape)
    dh = (shape[0] - mshape[0]) / 2.0
    dw = (shape[1] - mshape[1]) / 2.0
    if not (dh <= 0 and dw <= 0):
        raise ValueError(f"Shape {tuple(mshape)} is too small to unpad to shape {shape}")
    b, t = int(np.floor(-dh)), int(mshape[0] - np.ceil(-dh))
    l, r = int(np.floor(-dw)), int(mshape[1] - np.ceil(-dw))
    if return_args:
        return (b, t, l, r)
    return matrix[b:t, l:r]


def format_2vectors(vectors):
    """An array of 2-vectors as float ndarray of shape (2, N).  toolbox/__init__.py:939-960."""
    v = np.array(vectors, dtype=float, copy=True)
    if v.ndim == 1:
        v = v.reshape(-1, 1)
    if v.ndim != 2:
        v = np.squeeze(v)
        if v.ndim != 2:
            raise ValueError(f"Expected a (2, N) array of vectors, got shape {np.shape(vectors)}")
    if v.shape[0] != 2:
        if v.shape[1] == 2:
            v = v.T
        elif v.shape[0] > 2:
            v = v[:2, :]
        else:
            raise ValueError(f"Expected a (2, N) array of vectors, got shape {np.shape(vectors)}")
    return v


def smallest_distance(vectors, metric="chebyshev"):
    """Minimum pairwise distance (inf for < 2 points).  toolbox/__init__.py:1127-1250."""
    v = format_2vectors(vectors)
    n = v.shape[1]
    if n < 2:
        return np.inf
    if n > 400:
        from scipy.spatial import cKDTree
        p = {"chebyshev": np.inf, "euclidean": 2, "cityblock": 1}[metric]
        d, _ = cKDTree(v.T).query(v.T, k=2, p=p)
        return float(np.min(d[:, 1]))
    best = np.inf
    for i in range(n - 1):
        diff = np.abs(v[:, i + 1:] - v[:, i:i + 1])
        if metric == "chebyshev":
            d = np.max(diff, axis=0)
        elif metric == "euclidean":
            d = np.sqrt(np.sum(diff * diff, axis=0))
        else:
            d = np.sum(diff, axis=0)
        best = min(best, float(np.min(d)))
    return best


def convert_vector(vector, from_units="norm", to_units="norm", hardware=None, shape=None):
    """
    k-space unit conversion for the units the hologram constructors use
    (norm/kxy/rad/mrad/deg/knm/freq, and ij through a calibrated cameraslm).
    toolbox/__init__.py:91-400.  ``hardware`` is duck-typed (``pitch``, ``shape``, ``pitch_um``,
    ``wav_um`` on the SLM; ``slm``/``cam``/``kxyslm_to_ijcam``/``ijcam_to_kxyslm`` on a cameraslm).
    """
    v = format_2vectors(vector)
    if from_units == to_units:
        return v
    if hasattr(hardware, "slm") and hasattr(hardware, "cam"):
        cameraslm, slm = hardware, hardware.slm
    else:
        cameraslm, slm = None, hardware
    knm_conv = shape_xy = None
    if "knm" in (from_units, to_units):
        pitch = np.nan if slm is None else format_2vectors(slm.pitch)
        shp = np.array(slm.shape if shape is None else format_shape(tuple(int(s) for s in shape)), dtype=float)
        shape_xy = format_2vectors(np.flip(np.squeeze(shp)))
        knm_conv = pitch * shape_xy
    angle = {"norm": 1.0, "kxy": 1.0, "rad": 1.0, "mrad": 1e-3, "deg": np.pi / 180}
    if from_units in angle:
        rad = v * angle[from_units]
    elif from_units == "knm":
        rad = (v - shape_xy / 2.0) / knm_conv
    elif from_units == "freq":
        rad = v * slm.wav_um / format_2vectors(slm.pitch_um)
    elif from_units == "ij":
        if cameraslm is None:
            raise ValueError("a calibrated cameraslm is required for 'ij' units")
        rad = cameraslm.ijcam_to_kxyslm(v)
    else:
        raise NotImplementedError(f"unit '{from_units}' is outside the optimize() path of this build")
    if to_units in angle:
        return rad / angle[to_units]
    if to_units == "knm":
        return rad * knm_conv + shape_xy / 2.0
    if to_units == "freq":
        return rad * format_2vectors(slm.pitch_um) / slm.wav_um
    if to_units == "ij":
        if cameraslm is None:
            raise ValueError("a calibrated cameraslm is required for 'ij' units")
        return cameraslm.kxyslm_to_ijcam(rad)
    raise NotImplementedError(f"unit '{to_units}' is outside the optimize() path of this build")


def convert_radius(radius, from_units="norm", to_units="norm", hardware=None, shape=None):
    """Mean scaling of a small circle under convert_vector (toolbox.convert_radius)."""
    v0 = convert_vector((0, 0), from_units, to_units, hardware, shape)
    vx = convert_vector((radius, 0), from_units, to_units, hardware, shape)
    vy = convert_vector((0, radius), from_units, to_units, hardware, shape)
    return float(np.mean([np.linalg.norm(vx - v0), np.linalg.norm(vy - v0)]))


def take_indices(vectors, size):
    """Integer window offsets of analysis.take (floor of centred coordinates; analysis/__init__.py:133-143)."""
    v = np.floor(format_2vectors(vectors)).astype(int)
    off = np.floor(np.arange(int(size)) - (int(size) - 1) / 2.0).astype(int)
    return v, off


def imprint_disk_zero(matrix, cx, cy, w):
    """
    toolbox.imprint(matrix, (cx, w, cy, w), 0, centered=True, circular=True): set the pixels of a
    centred w-wide disk to zero (clipped to the array).  Used by SpotHologram for null points
    (_spots.py:1531-1538).
    """
    h_, w_ = matrix.shape
    r = w / 2.0
    x0, y0 = int(cx - w // 2), int(cy - w // 2)
    for yy in range(max(0, y0), min(h_, y0 + w)):
        for xx in range(max(0, x0), min(w_, x0 + w)):
            if (xx - cx) ** 2 + (yy - cy) ** 2 <= r * r:
                matrix[yy, xx] = 0
    return matrix
