"""GPU tests of the data path around the training step: resident table + device-side shuffle in
Model.fit, the sharded dataset streamed through the double-buffered copy stream, chunked predict."""
import numpy as np
import pytest

from oracle import nif_oracle as O

pytestmark = pytest.mark.gpu


def _model(seed=2):
    import nif_amd
    from tests.test_gpu_parity import _cfg
    kind, cs, cp = _cfg("NIFMultiScale", 64, 2, 32, 2, 1, 1, 1, 1, p_act="swish")
    nif_amd.set_seed(seed)
    m = nif_amd.NIFMultiScale(cs, cp)
    model = m.build()
    model.compile(nif_amd.Adam(1e-3), "mse")
    return nif_amd, m, model, O.Spec(kind, cs, cp)


def test_fit_device_shuffle_follows_the_oracle_with_the_same_permutations():
    nif_amd, m, model, spec = _model()
    x, y = nif_amd.data.synthetic_wave_batch(700, seed=4)
    sw = np.random.default_rng(1).uniform(0.5, 1.5, 700).astype(np.float32)
    ws = [w.astype(np.float64) for w in model.get_weights()]
    model._shuffle_seed = 13
    h = model.fit(x, y, sample_weight=sw, epochs=3, batch_size=256, shuffle=True, verbose=0)
    rng = np.random.default_rng(13)
    th = O.flatten(ws); mm = np.zeros_like(th); vv = np.zeros_like(th); t = 0
    f32 = lambda a: float(np.float32(a))
    losses = []
    for _ in range(3):
        perm = rng.permutation(700)
        xs, ys, ss = x[perm].astype(np.float64), y[perm].astype(np.float64), sw[perm].astype(np.float64)
        tot = 0.0
        for b0 in range(0, 700, 256):
            l, g = O.loss_and_grad(spec, O.unflatten(spec, th), xs[b0:b0 + 256], ys[b0:b0 + 256], ss[b0:b0 + 256])
            t += 1
            th, mm, vv = O.adam_step(th, O.flatten(g), mm, vv, t, lr=f32(1e-3), b1=f32(0.9), b2=f32(0.999), eps=f32(1e-7))
            tot += l * min(256, 700 - b0)
        losses.append(tot / 700)
    assert np.allclose(h.history["loss"], losses, rtol=2e-4), (h.history["loss"], losses)
    assert np.abs(O.flatten(model.get_weights()) - th).max() < 0.05 * 1e-3 * t


def test_streamed_shard_dataset_trains_like_the_host_staged_one(tmp_path):
    """README.md:155-178 workflow on .npz shards; files i+1 streams to HBM (pinned staging, copy stream) while file i trains"""
    from nif_amd.data import NPZShardDataset
    nif_amd, m, model, spec = _model(seed=6)
    x, y = nif_amd.data.synthetic_wave_batch(21000, seed=8)
    w = np.random.default_rng(2).uniform(0.5, 1.5, (21000, 1)).astype(np.float32)
    np.savez(tmp_path / "big.npz", data=np.hstack([x, y, w]))
    fh = NPZShardDataset(n_feature=2, n_target=1, area_weight=True)
    assert fh.create_from_npz(4096, str(tmp_path / "big.npz"), "data", str(tmp_path / "shards"), "wave", seed=0) == 6
    model._shuffle_seed = 3
    n = 0
    for batch_file in fh.get_tfr_meta_dataset(str(tmp_path / "shards"), epoch=2, model=model):
        h = model.fit(fh.gen_dataset_from_batch_file(batch_file, 512), epochs=1, verbose=0)
        assert np.isfinite(h.history["loss"][0])
        n += 1
    assert n == 12
    _, m2, model2, _ = _model(seed=6)
    model2._shuffle_seed = 3
    for batch_file in fh.get_meta_dataset(str(tmp_path / "shards"), epoch=2):
        model2.fit(fh.gen_dataset_from_batch_file(batch_file, 512), epochs=1, verbose=0)
    assert np.array_equal(O.flatten(model.get_weights()), O.flatten(model2.get_weights()))
    # and the same as fitting the concatenated file tables in memory, file by file
    _, m3, model3, _ = _model(seed=6)
    model3._shuffle_seed = 3
    for _ in range(2):
        for sh in fh.get_meta_dataset(str(tmp_path / "shards"), epoch=1):
            xs, ys, ws_ = sh.load_host()
            model3.fit(xs, ys, sample_weight=ws_, batch_size=512, epochs=1, verbose=0)
    assert np.array_equal(O.flatten(model.get_weights()), O.flatten(model3.get_weights()))
    assert model.evaluate(x, y) < 1.3


def test_predict_is_chunked_through_the_staging_buffers():
    nif_amd, m, model, spec = _model()
    x, _ = nif_amd.data.synthetic_wave_batch(5000, seed=1)
    u = model.predict(x)
    model._PREDICT_CHUNK = 1024
    assert np.array_equal(model.predict(x), u)
    assert np.array_equal(model.predict(x, batch_size=2048), u)
    lr = m.model_p_to_lr().predict(x[:, :1])
    w = m.model_lr_to_w().predict(lr[:600])
    sub = m.model_x_to_u_given_w()
    sub._PREDICT_CHUNK = 256
    assert np.array_equal(sub.predict([x[:600, 1:], w]), m.model_x_to_u_given_w()._run([x[:600, 1:], w]))


@pytest.mark.parametrize("shuffle", [False, True])
def test_fit_10k_points_batch_512_graph_epochs_equal_eager_epochs(shuffle):
    """BASELINE configs[0] at its own size: tutorial/1's NIF (ParameterNet 2x32, ShapeNet 2x32), 10 000 (t; x) points, batch 512 = 20
    steps per epoch (19 full + one of 272).  fit() records such an epoch ONCE into a hipGraph and replays it (nif_graph_*: Adam's
    bias correction from a device-side iteration counter); the trajectory must be the eager one's, epoch by epoch, and the first
    steps the oracle's."""
    import nif_amd
    cs = {"input_dim": 1, "output_dim": 1, "units": 32, "nlayers": 2, "activation": "swish"}
    cp = {"input_dim": 1, "latent_dim": 1, "units": 32, "nlayers": 2, "activation": "swish"}
    x, y = O.synthetic_wave_batch(10000, seed=0)
    runs = {}
    for graph in (True, False):
        nif_amd.set_seed(4)
        m = nif_amd.NIF(cs, cp); model = m.build()
        model._graph_epochs = graph
        model._shuffle_seed = 11
        model.compile(nif_amd.Adam(1e-3), "mse")
        w0 = [w.copy() for w in model.get_weights()]
        h = model.fit(x, y, epochs=4, batch_size=512, shuffle=shuffle, verbose=0)
        _, _, step = m._engine.get_opt_state()
        assert step == 4 * 20
        h2 = model.fit(x, y, epochs=2, batch_size=512, shuffle=shuffle, verbose=0)       # a second fit captures its own graph
        runs[graph] = (h.history["loss"] + h2.history["loss"], O.flatten(model.get_weights()), w0)
    lg, wg, _ = runs[True]
    le, we, w0 = runs[False]
    assert np.allclose(lg, le, rtol=2e-6), (lg, le)
    assert np.abs(wg - we).max() < 2e-6, np.abs(wg - we).max()          # (lr_t in fp64 on the device vs the host: the same float almost always)
    assert lg[-1] < lg[0]
    if not shuffle:     # the first epoch against the oracle's trajectory
        spec = O.Spec("NIF", cs, cp)
        th = O.flatten([w.astype(np.float64) for w in w0]); mm = np.zeros_like(th); vv = np.zeros_like(th)
        f32 = lambda a: float(np.float32(a))
        tot = 0.0
        for t in range(1, 21):
            lo, hi = (t - 1) * 512, min(10000, t * 512)
            l_, g_ = O.loss_and_grad(spec, O.unflatten(spec, th), x[lo:hi].astype(np.float64), y[lo:hi].astype(np.float64))
            tot += l_ * (hi - lo)
            th, mm, vv = O.adam_step(th, O.flatten(g_), mm, vv, t, lr=f32(1e-3), b1=f32(0.9), b2=f32(0.999), eps=f32(1e-7))
        assert abs(lg[0] - tot / 10000) < 2e-3 * (tot / 10000), (lg[0], tot / 10000)


def test_fit_with_compiled_metrics_on_the_device_path():
    """compile(metrics=[...]) through the real engine: device shuffle, a partial last batch, sample weights (which the unweighted
    metrics ignore) -- the epoch's 'mse' metric is the oracle's unweighted mean over the same permutations with the pre-update
    weights, training itself is untouched (same losses as without metrics), evaluate() returns [loss, mse, mae]"""
    nif_amd, m, model, spec = _model()
    x, y = nif_amd.data.synthetic_wave_batch(700, seed=4)
    sw = np.random.default_rng(1).uniform(0.5, 1.5, 700).astype(np.float32)
    ws0 = model.get_weights()
    model._shuffle_seed = 13
    h_plain = model.fit(x, y, sample_weight=sw, epochs=2, batch_size=256, shuffle=True, verbose=0)
    model.set_weights(ws0)
    model.compile(nif_amd.Adam(1e-3), "mse", metrics=["mse", "mae"])
    model._shuffle_seed = 13
    h = model.fit(x, y, sample_weight=sw, epochs=2, batch_size=256, shuffle=True, verbose=0)
    assert np.allclose(h.history["loss"], h_plain.history["loss"], rtol=1e-6)
    rng = np.random.default_rng(13)
    th = O.flatten([w.astype(np.float64) for w in ws0]); mm = np.zeros_like(th); vv = np.zeros_like(th); t = 0
    f32 = lambda a: float(np.float32(a))
    mse, mae = [], []
    for _ in range(2):
        perm = rng.permutation(700)
        xs, ys, ss = x[perm].astype(np.float64), y[perm].astype(np.float64), sw[perm].astype(np.float64)
        s2 = s1 = 0.0
        for b0 in range(0, 700, 256):
            u = O.forward(spec, O.unflatten(spec, th), xs[b0:b0 + 256])
            s2 += ((u - ys[b0:b0 + 256]) ** 2).sum(); s1 += np.abs(u - ys[b0:b0 + 256]).sum()
            l, g = O.loss_and_grad(spec, O.unflatten(spec, th), xs[b0:b0 + 256], ys[b0:b0 + 256], ss[b0:b0 + 256])
            t += 1
            th, mm, vv = O.adam_step(th, O.flatten(g), mm, vv, t, lr=f32(1e-3), b1=f32(0.9), b2=f32(0.999), eps=f32(1e-7))
        mse.append(s2 / 700); mae.append(s1 / 700)
    assert np.allclose(h.history["mse"], mse, rtol=2e-4) and np.allclose(h.history["mae"], mae, rtol=2e-4), (h.history, mse, mae)
    ev = model.evaluate(x, y)
    u = model.predict(x)
    assert np.allclose(ev[1:], [np.mean((u - y) ** 2), np.mean(np.abs(u - y))], rtol=1e-5) and abs(ev[0] - ev[1]) < 1e-5 * ev[0]
