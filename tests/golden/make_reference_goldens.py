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

# ---- recorded outputs of the reference's own tutorial notebooks (stored cell outputs = what the reference printed when its
# author ran it): the JacobianLayer / HessianLayer shape contract of tutorial 4 and the shard-streaming log of tutorial 5
# (number of files cut from 10^6 points at 10^4 per file, Keras steps per file at batch 128, the per-file losses)
import json
import re

TUT = "/root/reference/tutorial"


def _outputs(nb_path):
    nb = json.load(open(nb_path))
    out = []
    for c in nb["cells"]:
        if c["cell_type"] != "code":
            continue
        txt = ""
        for o in c.get("outputs", []):
            if "text" in o:
                txt += "".join(o["text"])
        out.append(("".join(c["source"]), txt))
    return out


rec = {}
for src, txt in _outputs(os.path.join(TUT, "4_get_gradients_by_wrapping_model_with_layer.ipynb")):
    for key, pat in (("y_shape", r"f\(x\) shape =\s*\(([^)]*)\)"), ("dydx_shape", r"df\(x\)/dx shape =\s*\(([^)]*)\)"),
                     ("d2ydx2_shape", r"d2f\(x\)/dx2 shape =\s*\(([^)]*)\)")):
        m = re.search(pat, txt)
        if m:
            rec[key] = [int(v) for v in m.group(1).split(",") if v.strip()]
    m = re.search(r"x_index = (\[[^\]]*\])\s*y_index = (\[[^\]]*\])", src)
    if m:
        rec["x_index"], rec["y_index"] = json.loads(m.group(1)), json.loads(m.group(2))
for src, txt in _outputs(os.path.join(TUT, "5_large_scale_training_on_tensorflow_record_data.ipynb")):
    m = re.search(r"total number of TFR files =\s*(\d+)", txt)
    if m:
        rec["n_files"] = int(m.group(1))
        a = re.search(r"num_pts_per_file=([0-9.e+]+)", src)
        rec["num_pts_per_file"] = int(float(a.group(1)))
    if "model.fit(batch_dataset" in src:
        rec["steps_per_file"] = sorted({int(a) for a, b in re.findall(r"(\d+)/(\d+) \[=+\]", txt)})
        rec["file_losses"] = [float(v) for v in re.findall(r"loss: ([0-9.]+)", txt)]
        rec["batch_size"] = int(re.search(r"batch_size = (\d+)", src).group(1))
    m = re.search(r"np.random.uniform\(0,1,\((\d+),(\d+)\)\)", src)
    if m:
        rec["table_shape"] = [int(m.group(1)), int(m.group(2))]
    m = re.search(r"TFRDataset\(n_feature=(\d+), n_target=(\d+)\)", src)
    if m:
        rec["n_feature"], rec["n_target"] = int(m.group(1)), int(m.group(2))
    m = re.search(r"^epoch = (\d+)", src, flags=re.M)
    if m:
        rec["epoch"] = int(m.group(1))
json.dump(rec, open(os.path.join(HERE, "tutorial_logs.json"), "w"), indent=1, sort_keys=True)
print("wrote tutorial_logs.json:", {k: (v if not isinstance(v, list) or len(v) < 8 else "%d values" % len(v)) for k, v in rec.items()})
