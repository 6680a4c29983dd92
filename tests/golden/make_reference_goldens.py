"""Generates tests/golden/normalisers.npz by EXECUTING the reference's NumPy-only normalisers
(/root/reference/nif/data/point_wise_data.py:50-114, loaded by file path because `import nif`
needs TensorFlow) on the reference's bundled dataset, and copies the two bundled .npz
datasets (data, not source) next to it.  Run in the build container only:

    python tests/golden/make_reference_goldens.py
"""
import importlib.util
import os
import shutil

import numpy as np

REF = "/root/reference/nif"
HERE = os.path.dirname(os.path.abspath(__file__))

spec = importlib.util.spec_from_file_location("pwd_ref", os.path.join(REF, "data", "point_wise_data.py"))
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
PWD = mod.PointWiseData

for fn in ("traveling_wave.npz", "traveling_wave_high_freq.npz"):
    shutil.copyfile(os.path.join(REF, "demo", "dataset", fn), os.path.join(HERE, fn))
    os.chmod(os.path.join(HERE, fn), 0o644)

raw = np.load(os.path.join(HERE, "traveling_wave_high_freq.npz"))["data"].astype(np.float64)
sd, sm, ss = PWD.standard_normalize(raw.copy())
md, mm, ms = PWD.minmax_normalize(raw.copy(), 1, 1, 1)
np.savez_compressed(os.path.join(HERE, "normalisers.npz"), raw=raw, std_data=sd, std_mean=sm, std_std=ss,
                    mm_data=md, mm_mean=mm, mm_std=ms)
print("wrote goldens")
