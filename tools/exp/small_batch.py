"""per-step cost of small batches (BASELINE configs[0]: 10k points, batch 512): fit() wall time and the raw C-ABI step loop"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import nif_amd
from oracle import nif_oracle as O
cs = {"input_dim": 1, "output_dim": 1, "units": 32, "nlayers": 2, "activation": "swish"}
cp = {"input_dim": 1, "latent_dim": 1, "units": 32, "nlayers": 2, "activation": "swish"}
nif_amd.set_seed(0)
m = nif_amd.NIF(cs, cp); model = m.build()
x, y = O.synthetic_wave_batch(10000, seed=0)
model.compile(nif_amd.Adam(1e-3), "mse")
for graph in (True, False):
  nif_amd.Model._graph_epochs = graph
  print("epochs captured into a hipGraph:" if graph else "eager launches:")
  for bs in (512, 2048, 10000):
    model.fit(x, y, epochs=2, batch_size=bs, shuffle=True, verbose=0)
    t0 = time.perf_counter(); ep = 50
    model.fit(x, y, epochs=ep, batch_size=bs, shuffle=True, verbose=0)
    dt = time.perf_counter() - t0
    nsteps = ep * ((10000 + bs - 1) // bs)
    print("batch %5d: %7.1f us/step (%d steps, fit wall %.3f s)" % (bs, dt / nsteps * 1e6, nsteps, dt))
