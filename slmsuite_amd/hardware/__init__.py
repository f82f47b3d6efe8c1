"""
Minimal duck-typed stand-ins for the hardware objects the hologram constructors read
(``slm.shape / pitch / grid / _get_source_amplitude() / get_source_zernike_scaling()``,
``cameraslm.slm``).  Real ``slmsuite.hardware`` objects can be passed instead.
Device drivers, cameras and calibration are out of scope.
"""
import numpy as np


class SimpleSLM:
    """Geometry of an SLM: ``shape`` (h, w), pixel ``pitch_um`` and wavelength ``wav_um``."""

    def __init__(self, shape, pitch_um=(8.0, 8.0), wav_um=0.78, source_radius=None):
        self.shape = (int(shape[0]), int(shape[1]))
        self.pitch_um = np.array([pitch_um, pitch_um] if np.isscalar(pitch_um) else pitch_um, dtype=float)
        self.wav_um = float(wav_um)
        self.pitch = self.pitch_um / self.wav_um                       # slm.py:196-201
        h, w = self.shape
        xs = (np.arange(w) - (w - 1) / 2.0) * self.pitch[0]            # centred, units of wavelengths
        ys = (np.arange(h) - (h - 1) / 2.0) * self.pitch[1]
        self.grid = np.meshgrid(xs, ys)
        # radius (in wavelengths) of the source amplitude; default: a quarter of the shorter side, which
        # is what the reference's SimulatedSLM reports for its uniform source
        self._radius = float(source_radius) if source_radius is not None else \
            min(w * self.pitch[0], h * self.pitch[1]) / 4.0
        self.source = {"amplitude_radius": self._radius}

    def _get_source_amplitude(self):
        return np.ones(self.shape)

    def get_source_radius(self):
        return self.source["amplitude_radius"]

    def get_spot_radius_kxy(self):
        """Approximate farfield spot radius in "kxy" (slm.py:1355-1390)."""
        from ..holography import toolbox
        rad_pix = self.source["amplitude_radius"] / np.mean(self.pitch)
        rad_freq = np.reciprocal(rad_pix)
        return float(np.mean(toolbox.convert_vector([rad_freq, rad_freq], from_units="freq", to_units="kxy",
                                                    hardware=self, shape=self.shape)))

    def get_source_zernike_scaling(self):
        return np.reciprocal(2 * self.source["amplitude_radius"])       # slm.py:1205-1213


class SimpleCamera:
    """Geometry of a camera: ``shape`` (h, w) and pixel ``pitch_um``; no acquisition."""

    def __init__(self, shape, pitch_um=(4.0, 4.0)):
        self.shape = (int(shape[0]), int(shape[1]))
        self.pitch_um = np.array([pitch_um, pitch_um] if np.isscalar(pitch_um) else pitch_um, dtype=float)


class SimpleFourierSLM:
    """
    An SLM, optionally with a camera geometry and an *analytic* Fourier calibration -- enough for every
    computational feedback mode, the ``"ij"`` spot basis and the hologram-side work of
    ``FourierSLM.fourier_grid_project`` (cameraslms.py:1088-1155).  No image acquisition.
    """

    def __init__(self, slm, cam=None, mag=1.0):
        self.slm = slm
        self.cam = cam
        self.mag = mag
        self.calibrations = {}

    def fourier_calibrate_analytic(self, M, b):
        """cameraslms.py:1157-1194: ij = M (kxy - a) + b with a = 0."""
        M = np.squeeze(np.asarray(M, dtype=float))
        if M.shape != (2, 2):
            raise ValueError("Expected a 2x2 matrix for M.")
        self.calibrations["fourier"] = {"M": M, "b": np.asarray(b, dtype=float).reshape(2, 1),
                                        "a": np.zeros((2, 1))}
        return self.calibrations["fourier"]

    def _fourier(self):
        if "fourier" not in self.calibrations:
            raise RuntimeError("Fourier calibration must exist to be used.")
        return self.calibrations["fourier"]

    _METRES = {"m": 1.0, "cm": 1e-2, "mm": 1e-3, "um": 1e-6, "nm": 1e-9}

    def get_effective_focal_length(self, units="norm"):
        """Scalar focal length of the train between SLM and camera from the calibration's magnification
        (cameraslms.py:1436-1487): sqrt|det M| camera pixels per radian, in pixels ("ij"), wavelengths ("norm") or metres."""
        pixels = np.sqrt(np.abs(np.linalg.det(self._fourier()["M"])))
        if units == "ij":
            return pixels
        pitch_um = None if self.cam is None else getattr(self.cam, "pitch_um", None)
        if pitch_um is None:
            import warnings
            warnings.warn(f"cam.pitch_um must be set to use units '{units}'")
            return np.nan
        if units == "norm":
            return pixels * np.array(pitch_um) / self.slm.wav_um
        if units in self._METRES:
            return pixels * np.array(pitch_um) * 1e-6 / self._METRES[units]
        raise ValueError(f"Unit '{units}' not recognized as a length.")

    def _depth_gain(self):
        """Camera-pixel depth per unit of focal power x_z (cameraslms.py:1221-1237): wav * f_eff^2 / cam pitch."""
        f_eff = np.mean(self.get_effective_focal_length("norm"))
        pitch_um = None if self.cam is None else getattr(self.cam, "pitch_um", None)
        return self.slm.wav_um * f_eff * f_eff / (np.nan if pitch_um is None else np.mean(pitch_um))

    @staticmethod
    def _columns(v):
        """Vectors as columns, 2 (lateral) or 3 (lateral + depth) rows."""
        v = np.asarray(v, dtype=float)
        if v.ndim == 1:
            v = v.reshape(-1, 1)
        if v.shape[0] not in (2, 3):
            raise ValueError(f"Expected 2- or 3-vectors as columns, got an array of shape {v.shape}")
        return v

    def kxyslm_to_ijcam(self, kxy):
        """cameraslms.py:1240-1294: ij = M (kxy - a) + b on the lateral rows; a depth row (focal power) scales to camera-pixel depth."""
        c, v = self._fourier(), self._columns(kxy)
        ij = np.matmul(c["M"], v[:2] - c["a"]) + c["b"]
        return ij if v.shape[0] == 2 else np.vstack((ij, v[2:] * self._depth_gain()))

    def ijcam_to_kxyslm(self, ij):
        """cameraslms.py:1296-1354: the inverse map; a depth row in camera pixels becomes focal power."""
        c, v = self._fourier(), self._columns(ij)
        kxy = np.matmul(np.linalg.inv(c["M"]), v[:2] - c["b"]) + c["a"]
        return kxy if v.shape[0] == 2 else np.vstack((kxy, v[2:] / self._depth_gain()))

    def fourier_grid_project(self, array_shape=10, array_pitch=10, array_center=None, **kwargs):
        """
        cameraslms.py:1088-1155: a ``"knm"`` grid of spots on the smallest square power-of-two pad,
        with the orientation check, optimised (default 10 iterations); the phase is written to the
        SLM when the SLM object can take one.  Returns the optimised hologram.
        """
        import warnings
        from ..holography.algorithms import SpotHologram
        from ..holography.toolbox import format_2vectors
        if not np.all(np.isclose(array_pitch, np.rint(array_pitch))):
            warnings.warn("array_pitch is non-integer")
        shape = SpotHologram.get_padded_shape(self, padding_order=1, square_padding=True)
        hologram = SpotHologram.make_rectangular_array(
            shape, array_shape=array_shape, array_pitch=array_pitch,
            array_center=None if array_center is None else (
                format_2vectors(array_center) + format_2vectors((shape[1] / 2.0, shape[0] / 2.0))),
            basis="knm", orientation_check=True, cameraslm=self)
        if "maxiter" not in kwargs:
            kwargs["maxiter"] = 10
        for key in kwargs:
            if key not in ("method", "maxiter", "verbose", "callback", "feedback", "stat_groups", "name",
                           "fixed_phase", "raw_stats", "blur_ij"):
                warnings.warn(f"Unexpected argument '{key}' passed to fourier_grid_project(). This may be ignored.")
        hologram.optimize(**kwargs)
        if hasattr(self.slm, "set_phase"):
            self.slm.set_phase(hologram.get_phase(), settle=True)
        return hologram
