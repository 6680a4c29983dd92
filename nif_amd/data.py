"""Point-wise data helpers on the host side of the path: the reference's normalisers
(nif/data/point_wise_data.py:50-114, NumPy) and a closed-form generator of the bundled
travelling-wave datasets (nif/demo/dataset/*.npz: u = exp(-1000 s^2) sin(omega s),
s = x - 0.2 - 0.006 t; omega = 4 / 400) used for synthetic benchmark batches."""
import numpy as np


class PointWiseData(object):
    """Container with the reference's column convention [parameters | coordinates | outputs (| weight)]
    (point_wise_data.py:16-48)."""

    def __init__(self, parameter_data, x_data, u_data, sample_weight=None):
        cols = [parameter_data, x_data, u_data] + ([sample_weight] if sample_weight is not None else [])
        self.data_raw = np.hstack(cols)
        self.data = None
        self.sample_weight = None
        self.n_p = parameter_data.shape[-1]
        self.n_x = x_data.shape[-1]
        self.n_o = u_data.shape[-1]

    @property
    def parameter(self):
        return self.data[:, : self.n_p]

    @property
    def x(self):
        return self.data[:, self.n_p: self.n_p + self.n_x]

    @property
    def u(self):
        return self.data[:, self.n_p + self.n_x: self.n_p + self.n_x + self.n_o]

    @staticmethod
    def standard_normalize(raw_data, area_weighted=False):
        """point_wise_data.py:50-78"""
        mean = raw_data.mean(axis=0)
        std = raw_data.std(axis=0)
        if area_weighted:
            mean[-1] = 0.0
            std[-1] = np.mean(raw_data[:, -1])
            nd = (raw_data - mean) / std
            return nd[:, :-1], mean, std, nd[:, -1]
        return (raw_data - mean) / std, mean, std

    @staticmethod
    def minmax_normalize(raw_data, n_para, n_x, n_target, area_weighted=False):
        """point_wise_data.py:80-114"""
        mean = raw_data.mean(axis=0)
        std = raw_data.std(axis=0)
        for i in range(n_para + n_x):
            lo, hi = np.min(raw_data[:, i]), np.max(raw_data[:, i])
            mean[i] = 0.5 * (lo + hi)
            std[i] = 0.5 * (hi - lo)
        for j in range(n_para + n_x, n_para + n_x + n_target):
            std[j] = np.max(np.abs(raw_data[:, j]))
        if area_weighted:
            mean[-1] = 0.0
            std[-1] = np.mean(raw_data[:, -1])
            nd = (raw_data - mean) / std
            return nd[:, :-1], mean, std, nd[:, -1]
        return (raw_data - mean) / std, mean, std


def traveling_wave(t, x, omega=4.0):
    s = x - 0.2 - 0.006 * t
    return np.exp(-1000.0 * s * s) * np.sin(omega * s)


def synthetic_wave_batch(n_points, seed=0, omega=4.0):
    """(inputs [N,2] = normalised (t, x), targets [N,1]) float32: t~U[0,90], x~U[0,1)."""
    rng = np.random.default_rng(seed)
    t = rng.uniform(0.0, 90.0, size=n_points)
    x = rng.uniform(0.0, 1.0, size=n_points)
    raw = np.stack([t, x, traveling_wave(t, x, omega)], axis=1)
    if omega <= 10:
        data, _, _ = PointWiseData.standard_normalize(raw)
    else:
        data, _, _ = PointWiseData.minmax_normalize(raw, 1, 1, 1)
    return np.ascontiguousarray(data[:, :2], dtype=np.float32), np.ascontiguousarray(data[:, 2:3], dtype=np.float32)
