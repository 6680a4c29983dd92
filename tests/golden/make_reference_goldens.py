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
# area-weighted variants (a trailing cell-area column becomes the sample weight) and the container's column convention
area = np.random.default_rng(0).uniform(0.5, 1.5, size=(raw.shape[0], 1))
raw_a = np.hstack([raw, area])
asd, asm, ass, asw = PWD.standard_normalize(raw_a.copy(), area_weighted=True)
amd, amm, ams, amw = PWD.minmax_normalize(raw_a.copy(), 1, 1, 1, area_weighted=True)
obj = PWD(raw[:, :1], raw[:, 1:2], raw[:, 2:3], area)
obj.data = asd
np.savez_compressed(os.path.join(HERE, "normalisers.npz"), raw=raw, std_data=sd, std_mean=sm, std_std=ss,
                    mm_data=md, mm_mean=mm, mm_std=ms, area=area, astd_data=asd, astd_mean=asm, astd_std=ass,
                    astd_w=asw, amm_data=amd, amm_mean=amm, amm_std=ams, amm_w=amw, obj_data_raw=obj.data_raw,
                    obj_dims=np.array([obj.n_p, obj.n_x, obj.n_o]), obj_parameter=obj.parameter, obj_x=obj.x, obj_u=obj.u)
print("wrote goldens")
