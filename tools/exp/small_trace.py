import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, nif_amd
from oracle import nif_oracle as O
cs = {"input_dim": 1, "output_dim": 1, "units": 32, "nlayers": 2, "activation": "swish"}
cp = {"input_dim": 1, "latent_dim": 1, "units": 32, "nlayers": 2, "activation": "swish"}
nif_amd.set_seed(0)
m = nif_amd.NIF(cs, cp); model = m.build()
x, y = O.synthetic_wave_batch(10000, seed=0)
model.compile(nif_amd.Adam(1e-3), "mse")
model.fit(x, y, epochs=3, batch_size=512, shuffle=True, verbose=0)
