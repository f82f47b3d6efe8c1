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

    def get_source_zernike_scaling(self):
        return np.reciprocal(2 * self.source["amplitude_radius"])       # slm.py:1205-1213


class SimpleFourierSLM:
    """An SLM without camera calibration, enough for the computational feedback modes."""

    def __init__(self, slm):
        self.slm = slm
        self.calibrations = {}
