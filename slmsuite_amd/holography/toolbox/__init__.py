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


def _pair_distance(diff, metric):
    if metric == "chebyshev":
        return np.max(diff, axis=0)
    if metric == "euclidean":
        return np.sqrt(np.sum(diff * diff, axis=0))
    if metric == "cityblock":
        return np.sum(diff, axis=0)
    raise ValueError(f"smallest_distance: unknown metric {metric!r}")


def smallest_distance(vectors, metric="chebyshev"):
    """
    Minimum pairwise distance (inf for < 2 points).  toolbox/__init__.py:1127-1250.

    A sweep over the points sorted by x: point i against point i + k for k = 1, 2, ... until no pair that far apart in the
    order is closer in x than the best distance found (all three metrics are >= |dx|).  Exact, a few vectorised passes for
    the usual spot arrays -- and no ``scipy.spatial`` import, which was 180 ms of a process' first ``SpotHologram``
    (tools/first_use_probe.py).  Point sets on which the sweep would degenerate (thousands of points sharing one x) go to
    the KD-tree.
    """
    v = format_2vectors(vectors)
    n = v.shape[1]
    if n < 2:
        return np.inf
    _pair_distance(np.zeros((2, 1)), metric)                 # (unknown metric: fail before any work)
    # (the axis along which fewer points coincide leads the order)
    lead = 0 if np.unique(v[0]).size >= np.unique(v[1]).size else 1
    order = np.lexsort((v[1 - lead], v[lead]))
    x, y = v[lead, order], v[1 - lead, order]
    best = np.inf
    for k in range(1, n):
        dx = x[k:] - x[:-k]                                  # >= 0: sorted
        if k > 4096:
            from scipy.spatial import cKDTree
            p = {"chebyshev": np.inf, "euclidean": 2, "cityblock": 1}[metric]
            d, _ = cKDTree(v.T).query(v.T, k=2, p=p)
            return float(np.min(d[:, 1]))
        if float(np.min(dx)) >= best:
            break
        near = dx < best
        d = _pair_distance(np.stack((dx[near], np.abs(y[k:] - y[:-k])[near])), metric)
        if d.size:
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
    Zero a w-wide disk around (cx, cy), as SpotHologram does for null points and spots
    (_spots.py:1531-1538 through toolbox.imprint(..., centered=True, circular=True)).

    Follows the reference at the array edge too (window_slice, toolbox/__init__.py:499-528): the w x w
    bounding box is clamped into [0, n - 1] per axis before the disk is laid out, its upper bound stays
    exclusive, and the disk is centred (w - 1) // 2 pixels from the CLAMPED lower corner -- so a disk that
    crosses the low edge moves inwards and one that crosses the high edge loses the last row / column.
    """
    n_y, n_x = matrix.shape
    half = (w - 1) // 2
    lo_x = min(max(int(cx - (w - 2) / 2), 0), n_x - 1)
    lo_y = min(max(int(cy - (w - 2) / 2), 0), n_y - 1)
    hi_x = min(max(int(cx - (w - 2) / 2) + int(w), 0), n_x - 1)
    hi_y = min(max(int(cy - (w - 2) / 2) + int(w), 0), n_y - 1)
    if hi_x <= lo_x or hi_y <= lo_y:
        return matrix
    dx = np.arange(lo_x, hi_x, dtype=float) - (lo_x + half)
    dy = np.arange(lo_y, hi_y, dtype=float) - (lo_y + half)
    inside = np.add.outer(dy * dy, dx * dx) <= (w * w) / 4.0
    block = matrix[lo_y:hi_y, lo_x:hi_x]
    block[inside] = 0
    return matrix


# ---- Zernike machinery used by CompressedSpotHologram (toolbox/phase.py) -------------------------------
def zernike_ansi_to_radial(j):
    """ANSI index -> (n, l)  (phase.zernike_convert_index, phase.py:570-680)."""
    j = int(j)
    n = int(np.ceil((-3 + np.sqrt(9 + 8 * j)) / 2))
    return n, 2 * j - n * (n + 2)


def zernike_cartesian(j):
    """
    {(px, py): integer coefficient} of the real Zernike polynomial of ANSI index j, normalised to
    +-1 at the pupil edge (Z1 = y, Z2 = x, Z4 = 2x^2 + 2y^2 - 1), as phase._zernike_coefficients
    (phase.py:1357-1442) caches them.  R_n^|l|(r) expanded with (x+iy)^|l| and (x^2+y^2)^k.
    """
    from math import comb, factorial
    n, l = zernike_ansi_to_radial(j)
    al = abs(l)
    out = {}
    for s in range((n - al) // 2 + 1):
        rc = (-1) ** s * factorial(n - s) // (factorial(s) * factorial((n + al) // 2 - s) * factorial((n - al) // 2 - s))
        k = (n - 2 * s - al) // 2
        for t in range(k + 1):
            for q in range(al + 1):
                if (l >= 0 and q % 2 == 1) or (l < 0 and q % 2 == 0):
                    continue
                key = (al - q + 2 * (k - t), q + 2 * t)
                out[key] = out.get(key, 0) + rc * comb(k, t) * comb(al, q) * (-1) ** (q // 2)
    return {k: v for k, v in out.items() if v != 0}


def zernike_indices_parse(indices=None, D=None):
    """Default bases [2,1], [2,1,4], [2,1,4,3,5,...]  (phase._zernike_indices_parse, phase.py:923-961)."""
    if indices is None and D is None:
        raise ValueError("Either dimension or indices must be defined.")
    if indices is None:
        # tilts first (x before y), then focus, then ANSI order; a prefix of that for D = 2 .. 4
        ladder = [2, 1, 4, 3] + list(range(5, D + 1))
        indices = ladder[:D] if D >= 2 else ladder
    chosen = np.ravel(np.array(indices))
    if D is not None and len(chosen) != D:
        raise ValueError(f"Expected data (dimension {D}) to have common size with indices (length {len(chosen)}).")
    return chosen


def zernike_monomial_weights(indices, weights):
    """
    (terms [M,2] int, monomial weights [M,N]) such that sum_d weights[d,n] Z_indices[d](x, y) =
    sum_m out[m,n] x^terms[m,0] y^terms[m,1]; monomials in ascending Cantor order
    (phase._zernike_get_cantor, phase.py:850-920).  Negative (special) indices are not polynomials: they
    are appended after the monomials as pseudo-terms (index, 0) carrying their own weights (:909-918); the
    only one the kernels know is -1, the vortex plate  w * atan2(y, x)  for w > 0 (phase.py:1783-1790).
    """
    a = np.asarray(weights, dtype=float)
    if a.ndim == 1:
        a = a.reshape(-1, 1)
    acc = {}
    special = []
    for d, idx in enumerate(np.ravel(indices)):
        if int(idx) < 0:
            if int(idx) != -1:
                raise ValueError(f"Unrecognized terms {(int(idx), 0)} for index {d}.")     # phase.py:1792
            special.append(((int(idx), 0), a[d]))
            continue
        for key, c in zernike_cartesian(int(idx)).items():
            acc[key] = acc.get(key, 0) + c * a[d]
    keys = sorted(acc, key=lambda k: (k[0] + k[1]) * (k[0] + k[1] + 1) // 2 + k[1])
    terms = [k for k in keys] + [k for k, _ in special]
    rows = [acc[k] for k in keys] + [w for _, w in special]
    return (np.array(terms, dtype=np.int32).reshape(-1, 2),
            np.array(rows, dtype=float).reshape(len(terms), -1))


def image_center_and_std(image, nansum=False):
    """
    (centre [x, y], standard deviation [x, y]) of one image in pixels on the CENTRED pixel grid
    (x - (w - 1)/2): first moments and square roots of the second central moments of the image
    normalised to unit sum, in float64 (analysis.image_positions / image_variances,
    analysis/__init__.py:646-790, with the image_moment pixel-grid branch :531-545).
    """
    img = np.array(image, dtype=float)
    if img.ndim != 2:
        raise ValueError("moments need a 2-D image (the reference cannot unpack a scalar amplitude either)")
    total = (np.nansum if nansum else np.sum)(img)
    img = np.zeros_like(img) if total == 0 else img / total
    add = np.nansum if nansum else np.sum
    h, w = img.shape
    x = (np.arange(w) - float(w - 1) / 2).reshape(1, w)
    y = (np.arange(h) - float(h - 1) / 2).reshape(h, 1)
    cx, cy = add(img * x), add(img * y)
    vx, vy = add(img * np.square(x - cx)), add(img * np.square(y - cy))
    return np.array([cx, cy]), np.sqrt(np.array([vx, vy]))


def blaze(grid, vector):
    """2 pi (kx x + ky y) on normalised grids (toolbox.phase.blaze, phase.py:20-75, same branch structure)."""
    x_grid, y_grid = process_grid(grid)
    vector = np.asarray(vector, dtype=np.float64)      # float64 scalars promote the products to float64 (NEP 50)
    # a vanishing component contributes no term at all (not a term of zeros: the result keeps the other term's rounding)
    ramps = [(2 * np.pi * k) * axis for k, axis in ((vector[0], x_grid), (vector[1], y_grid)) if k != 0]
    return ramps[0] + ramps[1] if len(ramps) == 2 else ramps[0] if ramps else np.zeros_like(x_grid)


def lens(grid, f):
    """pi (x^2 / fx + y^2 / fy) (toolbox.phase.lens, phase.py:397-452); scalar f = isotropic."""
    x_grid, y_grid = process_grid(grid)
    f = np.squeeze(np.array([f, f] if np.isscalar(f) else f, dtype=np.float64))   # float64 scalars: the sum is
    if f.size != 2:                                                               # formed in float64 (NEP 50)
        raise ValueError(f"Expected two terms in focal list. Found {f}.")
    if np.any(f == 0):
        raise ValueError(f"Cannot interpret a focal length of zero. Found {f}.")
    x_curved, y_curved = bool(np.isfinite(f[0])), bool(np.isfinite(f[1]))
    if not y_curved:                   # (flat in y: the reference's x-only branch is unreachable, :447 -- no lens at all)
        return np.zeros_like(x_grid)
    bowl = (np.pi / f[1]) * np.square(y_grid)
    return (np.pi / f[0]) * np.square(x_grid) + bowl if x_curved else bowl


def process_grid(grid):
    """(x_grid, y_grid) from a cameraslm, an SLM or a pair of arrays (toolbox._process_grid)."""
    for attribute in ("slm", "grid"):          # FourierSLM -> its SLM -> the SLM's coordinate arrays
        if attribute == "grid" or hasattr(grid, "cam"):
            grid = getattr(grid, attribute, grid)
    return grid[0], grid[1]


def format_vectors(vectors, expected_dimension=None):
    """(D, N) float array from tuples / row vectors (toolbox.format_vectors with handle_dimension='pass')."""
    v = np.array(vectors, dtype=float, copy=True)
    if v.ndim == 1:
        v = v.reshape(-1, 1)
    if v.ndim != 2:
        raise ValueError(f"Expected a (D, N) array of vectors, got shape {np.shape(vectors)}")
    if expected_dimension is not None and v.shape[0] != expected_dimension:
        raise ValueError(f"Expected vectors of dimension {expected_dimension}")
    return v


def convert_vector_zernike(vector, from_units, hardware):
    """
    ("zernike" coefficients [D,N], "kxy" [D,N]) of 2- or 3-vectors given in "kxy"/"norm" or "zernike"
    units: xy scale 2*pi / slm.get_source_zernike_scaling(), depth (8 pi)/scale^2
    (toolbox.convert_vector, toolbox/__init__.py:318-392).
    """
    slm = hardware.slm if hasattr(hardware, "slm") else hardware
    v = format_vectors(vector)
    zs = 2 * np.pi * np.reciprocal(slm.get_source_zernike_scaling())
    if from_units in ("kxy", "norm", "rad"):
        kxy = v.copy()
        zern = v.copy()
        zern[:2] = v[:2] * zs
        if v.shape[0] > 2:
            zern[2] = v[2] * ((zs * zs) / (8 * np.pi))
    elif from_units == "zernike":
        zern = v.copy()
        kxy = v.copy()
        kxy[:2] = v[:2] / zs
        if v.shape[0] > 2:
            kxy[2] = v[2] * ((8 * np.pi) / (zs * zs))
    else:
        raise NotImplementedError(f"unit '{from_units}' needs camera calibration hardware")
    return zern, kxy
