"""`nif.demo` counterpart (reference nif/demo/__init__.py): the bundled point-wise datasets as PointWiseData objects.

The two travelling-wave tables of the reference (nif/demo/dataset/*.npz: 10 times x 200 positions, float32) are the closed form
    u(t, x) = exp(-1000 s^2) sin(omega s),  s = x - 0.2 - 0.006 t,   omega = 4 / 400
sampled on t = 0, 10, .., 90 and x = 0, 0.005, .., 0.995 (tests/test_host_logic.py pins the generator against copies of the
reference's files); they are generated here instead of shipped.  The cylinder-flow table is not part of the reference tree
(.MISSING_LARGE_BLOBS): `CylinderFlow(path)` loads a user-provided file of the same layout."""
import numpy as np

from .data import PointWiseData, traveling_wave

__all__ = ["TravelingWave", "TravelingWaveHighFreq", "CylinderFlow"]


def _lattice(omega):
    t = np.repeat(np.arange(10, dtype=np.float64) * 10.0, 200)
    x = np.tile((np.arange(200, dtype=np.float64) * 0.005).astype(np.float32).astype(np.float64), 10)
    return np.stack([t, x, traveling_wave(t, x, omega)], axis=1).astype(np.float32)


class TravelingWave(PointWiseData):
    """traveling_wave.py:8-36: standard-normalised (t, x, u) table, omega = 4"""

    def __init__(self):
        data = _lattice(4.0)
        super(TravelingWave, self).__init__(data[:, [0]], data[:, [1]], data[:, [2]])
        self.data, self.mean, self.std = self.standard_normalize(self.data_raw)


class TravelingWaveHighFreq(PointWiseData):
    """traveling_wave_high_freq.py:8-41: min-max-normalised table, omega = 400"""

    def __init__(self):
        data = _lattice(400.0)
        super(TravelingWaveHighFreq, self).__init__(data[:, [0]], data[:, [1]], data[:, [2]])
        self.data, self.mean, self.std = self.minmax_normalize(self.data_raw, n_para=self.n_p, n_x=self.n_x, n_target=1)


class CylinderFlow(PointWiseData):
    """cylinderflow.py:8-38: columns (t, x, y, u, v, area weight), area-weighted min-max normalisation"""

    def __init__(self, path=None):
        if path is None:
            raise FileNotFoundError("the cylinder-flow table is not bundled (it is not in the reference tree either): "
                                    "CylinderFlow(path='cylinderflow.npz') with the reference's layout, key 'data'")
        data = np.load(path)["data"]
        super(CylinderFlow, self).__init__(data[:, [0]], data[:, [1, 2]], data[:, [3, 4]], data[:, -1:])
        self.data, self.mean, self.std, self.sample_weight = self.minmax_normalize(
            self.data_raw, n_para=self.n_p, n_x=self.n_x, n_target=2, area_weighted=True)
