"""N>1 path on CPU: world_size-2 gloo process group exercising nif_amd.distributed (shard bounds, the
scalar and buffer all-reduces) with the oracle standing in for the per-shard HIP compute: the SUM of the
per-shard [grad | loss] buffers, each pre-scaled by 1/B_global, must equal the full-batch gradient and loss
(this is exactly what the GPU path all-reduces over RCCL)."""
import os
import socket

import numpy as np
import pytest

from oracle import nif_oracle as O
from tests.cfgs import ALL_SMALL


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, outdir):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank),
                       "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank)})
    from nif_amd import distributed as dist
    r, w = dist.init("gloo")
    assert (r, w) == (rank, world) and dist.is_initialized() and dist.world_size() == world and dist.rank() == rank
    kind, cs, cp = ALL_SMALL[name]
    spec = O.Spec(kind, cs, cp)
    rng = np.random.default_rng(0)  # same data and weights on every rank
    ws = O.init_weights(spec, rng)
    B = 37  # not divisible by the world size: shards are 19 / 18
    x = rng.uniform(-1, 1, size=(B, spec.pi + spec.si))
    y = rng.uniform(-1, 1, size=(B, spec.so))
    sw = rng.uniform(0.5, 1.5, size=(B,))
    lo, hi = dist.shard_bounds(B, world, rank)
    bg = dist.all_reduce_scalar_sum(hi - lo)
    assert bg == B
    loss, grads = O.loss_and_grad(spec, ws, x[lo:hi], y[lo:hi], sw[lo:hi], batch_global=bg)
    buf = np.concatenate([O.flatten(grads), [loss]]).astype(np.float32)   # the [grad | loss] buffer
    dist.all_reduce_host(buf)
    np.save(os.path.join(outdir, "rank%d.npy" % rank), buf)
    dist.shutdown()


@pytest.mark.parametrize("name", ["ms_plain", "nif_swish"])
def test_two_rank_gradient_allreduce_equals_full_batch(name, tmp_path):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, name, str(tmp_path)), nprocs=2, join=True)
    b0 = np.load(tmp_path / "rank0.npy")
    b1 = np.load(tmp_path / "rank1.npy")
    assert np.array_equal(b0, b1)  # every rank ends with the same buffer -> identical Adam update
    kind, cs, cp = ALL_SMALL[name]
    spec = O.Spec(kind, cs, cp)
    rng = np.random.default_rng(0)
    ws = O.init_weights(spec, rng)
    x = rng.uniform(-1, 1, size=(37, spec.pi + spec.si))
    y = rng.uniform(-1, 1, size=(37, spec.so))
    sw = rng.uniform(0.5, 1.5, size=(37,))
    loss, grads = O.loss_and_grad(spec, ws, x, y, sw)
    ref = np.concatenate([O.flatten(grads), [loss]])
    assert np.allclose(b0, ref, rtol=2e-6, atol=1e-7)


def test_shard_bounds_partition_rows():
    from nif_amd import distributed as dist
    for n in (1, 7, 64, 1000003):
        for world in (1, 2, 3, 8):
            cuts = [dist.shard_bounds(n, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_defaults():
    from nif_amd import distributed as dist
    assert dist.world_size() == 1 and dist.rank() == 0 and dist.local_device() == 0
