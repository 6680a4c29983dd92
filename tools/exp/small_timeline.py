"""s_memtime stamps of thread 0 / workgroup 0 of k_small at its phase boundaries (needs -DNIF_TIMELINE on k_small.hip:
python tools/build_variant.py tls "-DNIF_TIMELINE" k_small.hip; NIF_LIB=nif_amd/libnif_hip_tls.so python tools/exp/small_timeline.py)"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import nif_amd
from oracle import nif_oracle as O
cs = {"input_dim": 1, "output_dim": 1, "units": 32, "nlayers": 2, "activation": "swish"}
cp = {"input_dim": 1, "latent_dim": 1, "units": 32, "nlayers": 2, "activation": "swish"}
nif_amd.set_seed(0)
m = nif_amd.NIF(cs, cp); model = m.build(); e = m._engine
x, y = O.synthetic_wave_batch(512, seed=0)
for _ in range(5):
    e.loss_and_grad(x, y)
e.lib.nif_debug_timeline(e.ctx, None, 0)
for _ in range(3):
    e.loss_and_grad(x, y)
buf = (C.c_int64 * 4096)()
e.lib.nif_debug_timeline(e.ctx, buf, 2048)
t = np.array(buf[:10])
t[1] = t[0]      # (r6 final form: ONE gather builds the LDS images, stamp 1 is gone)
names = ["-", "LDS images: one gather through the index map", "ParameterNet forward + latent", "ShapeNet forward", "last layer + loss",
         "ShapeNet adjoint", "ParameterNet adjoint", "loss partial + barrier", "gradient entries (22 tensors)"]
print("k_small, 512 points of configs[0]'s net, workgroup 0 / thread 0, s_memtime ticks (shader clocks):")
for i, nm in enumerate(names):
    if nm != "-":
        print("  %-46s %6d ticks" % (nm, t[i + 1] - t[i]))
print("  total %d ticks" % (t[9] - t[0]))
