"""ConvLSTM2D family on the GPU (satellite_computervision_amd/lstm_tools.py, csrc/convlstm.hip) against the NumPy float64 oracle
(oracle/convlstm.py: Keras ConvLSTM2D cell as the reference calls it, utils/model_tools.py:666-920; parity unpinned, cross-checked
against torch autograd in tests/test_oracle_cpu.py)."""
import os
import numpy as np
import pytest
import torch

from oracle import convlstm as CL
from oracle import keras_ops as K
from oracle import losses as OL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def lt():
    from satellite_computervision_amd import lstm_tools
    assert torch.cuda.is_available()
    return lstm_tools


def rel(got, ref):
    return np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)


def cosine(a, b):
    return float((a * b).sum() / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
@pytest.mark.parametrize('rec_act', ['hard_sigmoid', 'sigmoid'])
@pytest.mark.parametrize('dil,rs', [(1, True), (3, False)])
def test_convlstm2d_layer_forward_and_bptt(lt, dtype, rec_act, dil, rs, monkeypatch):
    """ONE ConvLSTM2D layer (utils/model_tools.py:690-700 / 710-720): output sequence or last state, and through BPTT the gradients of
    the input, kernel, recurrent kernel and bias -- with a gradient arriving through return_state as well (build_lstm_layers2, :737)."""
    from satellite_computervision_amd import ops
    from satellite_computervision_amd._lib import BF16, F32
    monkeypatch.setattr(lt, 'RECURRENT_ACTIVATION', rec_act)
    f32 = dtype == 'float32'
    code, td = (F32, torch.float32) if f32 else (BF16, torch.bfloat16)
    rng = np.random.default_rng(3)
    B, T, H, W, Cc, F = 2, 4, 16, 16, 6, 16
    p = CL.convlstm_init(rng, Cc, F)
    p['bias'] = p['bias'] + 0.1 * rng.standard_normal(4 * F)
    p = {k: v.astype(np.float32).astype(np.float64) for k, v in p.items()}
    x = rng.standard_normal((B, T, H, W, Cc)).astype(np.float32).astype(np.float64)
    if not f32:          # values the device stores exactly
        x = torch.tensor(x, dtype=torch.float32).to(td).double().numpy()
    pq = p if f32 else {k: (torch.tensor(v, dtype=torch.float32).to(td).double().numpy() if k != 'bias' else v) for k, v in p.items()}
    out_ref, cache = CL.convlstm_forward(x, pq, dil, None, rec_act, rs)
    dout = rng.standard_normal(out_ref.shape)
    dhl = rng.standard_normal((B, H, W, F))
    if not f32:
        dout = torch.tensor(dout, dtype=torch.float32).to(td).double().numpy()
        dhl = torch.tensor(dhl, dtype=torch.float32).to(td).double().numpy()
    dx_ref, g_ref = CL.convlstm_backward(dout, cache, dh_last=dhl)

    P = lt._Params()
    layer = lt.ConvLSTM2D(P, np.random.default_rng(0), 'l', Cc, F, dil, rs)
    P.build()
    for k, v in p.items():
        P.p('l/' + k).copy_(torch.tensor(v, dtype=torch.float32).cuda())
    xt, _ = lt._ingest_seq(x.astype(np.float32), 16, code)
    out, stats, cnt = layer.forward(lt.Act(xt, Cc), T, B, True, code)
    got = out.t.float().cpu().numpy()[..., :F]
    got = got.reshape(T, B, H, W, F).transpose(1, 0, 2, 3, 4) if rs else got
    assert rel(got, out_ref) < (2e-5 if f32 else 2e-2)
    # BatchNorm statistics of the stored output
    s = stats.sum(0).cpu().numpy()
    np.testing.assert_allclose(s[0, :F], got.reshape(-1, F).sum(0), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(s[1, :F], (got.reshape(-1, F) ** 2).sum(0), rtol=1e-4, atol=1e-3)
    assert cnt == (T if rs else 1) * B * H * W
    Fp = out.t.shape[-1]
    d_tm = dout.transpose(1, 0, 2, 3, 4).reshape(T * B, H, W, F) if rs else dout
    dd = torch.zeros(d_tm.shape[0], H, W, Fp, dtype=td, device='cuda')
    dd[..., :F] = torch.tensor(d_tm, dtype=torch.float32).to(td).cuda()
    ds = torch.zeros(B, H, W, Fp, dtype=td, device='cuda')
    ds[..., :F] = torch.tensor(dhl, dtype=torch.float32).to(td).cuda()
    dx = layer.backward(dd, dstate_h=ds, need_dx=True)
    dxg = dx.float().cpu().numpy()[..., :Cc].reshape(T, B, H, W, Cc).transpose(1, 0, 2, 3, 4)
    tol = 5e-5 if f32 else 3e-2
    assert rel(dxg, dx_ref) < tol, rel(dxg, dx_ref)
    for k in ('kernel', 'recurrent_kernel', 'bias'):
        g = P.g('l/' + k).cpu().numpy().astype(np.float64)
        assert rel(g, g_ref[k]) < tol, (k, rel(g, g_ref[k]))
        assert cosine(g, g_ref[k]) > (0.99999 if f32 else 0.999), k


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
def test_get_lstm_model_training_step_matches_oracle(lt, dtype):
    """get_lstm_model (utils/model_tools.py:773-808: build_lstm_layers -> Conv2D(n_classes, 1x1) -> ReLU(max_value = 2)) end to end:
    prediction, mse_4d loss (:142-166) and every gradient of one training step against the float64 oracle."""
    from satellite_computervision_amd import model_tools as mt
    f32 = dtype == 'float32'
    mt.set_compute_dtype(dtype)
    try:
        B, T, H, W, Cc, ncls = 2, 3, 16, 16, 6, 4
        o = CL.LSTMLayersOracle(Cc, ncls, filters=64, rec_act=lt.RECURRENT_ACTIVATION, seed=5)
        m = lt.get_lstm_model(Cc, ncls, T)
        assert m.compute_dtype == dtype
        names = {'l1': 'conv_lstm', 'l2': 'dilated_conv_lstm', 'bn1': 'batch_norm', 'bn2': 'batch_norm2', 'dense': 'conv2d'}
        w = {}
        for lk, lv in o.p.items():
            for pk, pv in lv.items():
                lv[pk] = pv.astype(np.float32).astype(np.float64)
                w[f'{names[lk]}/{pk}'] = lv[pk]
        m.set_weights_dict(w)
        rng = np.random.default_rng(8)
        x = rng.random((B, T, H, W, Cc)).astype(np.float32)
        y = rng.random((B, H, W, ncls)).astype(np.float32) * 1.5
        out_ref = o.forward(x.astype(np.float64))
        loss_ref, dout = OL.mse_4d(y.astype(np.float64), out_ref)
        g_ref = o.backward(dout)
        m.compile(optimizer=mt.Adam(0.0), loss=mt.mse_4d)
        loss = m.train_on_batch(x, y)
        np.testing.assert_allclose(loss, loss_ref, rtol=1e-4 if f32 else 3e-2)
        key = {'l1.kernel': 'conv_lstm/kernel', 'l1.recurrent_kernel': 'conv_lstm/recurrent_kernel', 'l1.bias': 'conv_lstm/bias',
               'l2.kernel': 'dilated_conv_lstm/kernel', 'l2.recurrent_kernel': 'dilated_conv_lstm/recurrent_kernel', 'l2.bias': 'dilated_conv_lstm/bias',
               'bn1.gamma': 'batch_norm/gamma', 'bn1.beta': 'batch_norm/beta', 'bn2.gamma': 'batch_norm2/gamma', 'bn2.beta': 'batch_norm2/beta',
               'dense.kernel': 'conv2d/kernel', 'dense.bias': 'conv2d/bias'}
        for ok, dk in key.items():
            g = m.P.g(dk).cpu().numpy().astype(np.float64).reshape(g_ref[ok].shape)
            c = cosine(g, g_ref[ok])
            assert c > (0.9999 if f32 else 0.98), (ok, c, rel(g, g_ref[ok]))
            if f32:
                assert rel(g, g_ref[ok]) < 2e-3, (ok, rel(g, g_ref[ok]))
        # inference mode (Model.predict): each BatchNormalization uses its MOVING statistics.  Give both sides the same non-trivial moving
        # mean / variance (the step above used learning rate 0: the weights are still the oracle's) and compare the predictions
        mv = {'bn1': (rng.standard_normal(64) * 0.05, 0.5 + rng.random(64)), 'bn2': (rng.standard_normal(64) * 0.05, 0.5 + rng.random(64))}
        m.set_weights_dict({'batch_norm/moving_mean': mv['bn1'][0], 'batch_norm/moving_var': mv['bn1'][1],
                            'batch_norm2/moving_mean': mv['bn2'][0], 'batch_norm2/moving_var': mv['bn2'][1]})
        mv64 = {k: (v[0].astype(np.float32).astype(np.float64), v[1].astype(np.float32).astype(np.float64)) for k, v in mv.items()}
        pred_ref = o.forward_infer(x.astype(np.float64), mv64)
        pred = m.predict(x)
        assert pred.shape == (B, H, W, ncls) and pred.min() >= 0 and pred.max() <= 2.0
        err = np.abs(pred - pred_ref).max()
        assert err < (2e-4 if f32 else 6e-2), err
        assert rel(pred, pred_ref) < (1e-4 if f32 else 2e-2), rel(pred, pred_ref)
    finally:
        mt.set_compute_dtype('bfloat16')


def test_hybrid_model_forward_and_training_step(lt):
    """get_hybrid_model (utils/model_tools.py:874-920), fp32: the fusion head -- Conv2D(relu) on each branch, nearest resize of the LSTM
    map to the U-Net grid, concat [lstm, unet], Conv2D(softmax) -- against the oracle ON THE DEVICE'S OWN branch features (the two
    branches are covered by test_full_unet_* and the tests above), then a training step that must move every parameter group."""
    from satellite_computervision_amd import model_tools as mt
    mt.set_compute_dtype('float32')
    try:
        mt.reset_uids(); mt.set_seed(4)
        ncls = 3
        m = lt.get_hybrid_model((48, 48, 4), (3, 8, 8, 6), ncls, filters=[32, 64], factors=[3, 2])
        rng = np.random.default_rng(2)
        xu = rng.random((2, 48, 48, 4)).astype(np.float32)
        xl = rng.random((2, 3, 8, 8, 6)).astype(np.float32)
        lab = rng.integers(0, ncls, (2, 48, 48))
        y = np.eye(ncls, dtype=np.float32)[lab]
        probs = m.predict([xu, xl])
        assert probs.shape == (2, 48, 48, ncls)
        np.testing.assert_allclose(probs.sum(-1), 1.0, atol=1e-5)
        # oracle of the fusion head on the device's branch outputs
        zu = m._zu.cpu().numpy().astype(np.float64)                                  # U-Net dense pre-activation (n, 48, 48, k)
        zl = m.lstm_dense.ctx['out'].cpu().numpy().astype(np.float64)               # LSTM dense output, after its ReLU (n, 8, 8, k)
        up, _ = CL.resize_nearest(zl, 48, 48)
        cat = np.concatenate([up, np.maximum(zu, 0)], -1)
        w = m.get_weights_dict()
        logits = K.conv2d_same(cat, w['probabilities/kernel'].astype(np.float64), w['probabilities/bias'].astype(np.float64))
        np.testing.assert_allclose(probs, K.softmax(logits), atol=2e-6)
        m.compile(optimizer=mt.Adam(1e-3), loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, [1.0] * ncls))
        w0 = m.get_weights_dict()
        u0 = m.unet.runtime.pflat.clone()
        l0 = m.train_on_batch([xu, xl], y)
        for _ in range(15):
            l1 = m.train_on_batch([xu, xl], y)
        assert np.isfinite(l1) and l1 < l0, (l0, l1)
        w1 = m.get_weights_dict()
        for k in ('conv_lstm/kernel', 'conv_lstm/recurrent_kernel', 'dilated_conv_lstm/kernel', 'batch_norm/gamma', 'lstm_dense/kernel', 'probabilities/kernel'):
            assert np.abs(w1[k] - w0[k]).max() > 0, k
        assert float((m.unet.runtime.pflat - u0).abs().max()) > 0
        # gradient check of the fusion head's two data gradients against the oracle (finite structure: softmax cross-entropy)
        m.P.grad.zero_()
        pr, _ = m._forward([xu, xl], True)
        loss, dlog = m._loss_grad(pr, y, 'softmax')
        dzl, dau = m.fusion.backward(dlog, need_dx=(True, True))
        zu = m._zu.cpu().numpy().astype(np.float64)
        zl = m.lstm_dense.ctx['out'].cpu().numpy().astype(np.float64)
        up, idx = CL.resize_nearest(zl, 48, 48)
        cat = np.concatenate([up, np.maximum(zu, 0)], -1)
        wk = m.P.p('probabilities/kernel').cpu().numpy().astype(np.float64)
        dcat, dk_ref, db_ref = K.conv2d_same_bwd(cat, wk, dlog.cpu().numpy().astype(np.float64))
        np.testing.assert_allclose(m.P.g('probabilities/kernel').cpu().numpy(), dk_ref, rtol=2e-3, atol=1e-6)
        np.testing.assert_allclose(m.P.g('probabilities/bias').cpu().numpy(), db_ref, rtol=2e-3, atol=1e-6)
        np.testing.assert_allclose(dau.cpu().numpy(), dcat[..., ncls:], rtol=1e-4, atol=1e-8)
        np.testing.assert_allclose(dzl.cpu().numpy(), CL.resize_nearest_bwd(dcat[..., :ncls], idx, 8, 8), rtol=1e-4, atol=1e-8)
    finally:
        mt.set_compute_dtype('bfloat16')


def test_lstm_autoencoder_matches_oracle(lt):
    """get_lstm_autoencoder (utils/model_tools.py:810-872) with its build_lstm_layers2 encoder (:719-771; ReLU(state_h + BN(h2))), fp32:
    both outputs and every gradient of one training step (mse_4d on both outputs) against the float64 oracle."""
    from satellite_computervision_amd import model_tools as mt
    mt.set_compute_dtype('float32')
    try:
        B, T, H, W, Cc, ncls = 2, 3, 16, 16, 6, 6
        o = CL.LSTMAutoencoderOracle(Cc, T, ncls, rec_act=lt.RECURRENT_ACTIVATION, seed=7)
        m = lt.get_lstm_autoencoder(Cc, T, ncls)
        names = {'l1': 'conv_lstm', 'l2': 'dilated_conv_lstm', 'dec': 'lstm_decoder', 'bn1': 'batch_norm', 'bn2': 'batch_norm2',
                 'temporal': 'temporal_dense', 'single': 'single_dense'}
        w = {}
        for lk, lv in o.p.items():
            for pk, pv in lv.items():
                lv[pk] = pv.astype(np.float32).astype(np.float64)
                w[f'{names[lk]}/{pk}'] = lv[pk]
        m.set_weights_dict(w)
        rng = np.random.default_rng(11)
        x = rng.random((B, T, H, W, Cc)).astype(np.float32)
        sc = rng.standard_normal((B, H, W, 2)).astype(np.float32)
        t_ref, s_ref = o.forward(x.astype(np.float64), sc.astype(np.float64))
        m.compile(optimizer=mt.Adam(0.0), loss=mt.mse_4d)
        ty = np.flip(x, axis=1).copy()                       # LSTMAutoencoderGenerator: the reversed input sequence (utils/processing.py:1034)
        sy = rng.random((B, H, W, ncls)).astype(np.float32)
        l1, d1 = OL.mse_4d(ty.astype(np.float64), t_ref)
        l2, d2 = OL.mse_4d(sy.astype(np.float64), s_ref)
        g_ref = o.backward(d1, d2)
        loss = m.train_on_batch([x, sc], [ty, sy])
        np.testing.assert_allclose(loss, l1 + l2, rtol=1e-4)
        for ok, gv in g_ref.items():
            pre, pk = ok.split('.', 1)
            g = m.P.g(f'{names[pre]}/{pk}').cpu().numpy().astype(np.float64).reshape(gv.shape)
            assert cosine(g, gv) > 0.9999 and rel(g, gv) < 3e-3, (ok, cosine(g, gv), rel(g, gv))
        tp, sp = m.predict([x, sc])
        assert tp.shape == (B, T, H, W, ncls) and sp.shape == (B, H, W, ncls) and np.isfinite(tp).all()
    finally:
        mt.set_compute_dtype('bfloat16')


def test_hybrid_generator_feeds_the_hybrid_model(lt):
    """HybridDataGenerator (utils/processing.py:1051-1187) -> get_hybrid_model.fit: NAIP tiles through the device pipeline, the Sentinel-2
    sequence through the NumPy helpers, one-hot labels; one epoch runs and the loss is finite."""
    import random
    from satellite_computervision_amd import model_tools as mt, processing as P
    mt.set_compute_dtype('float32')
    try:
        rng = np.random.default_rng(3)
        n = 4
        naip = [rng.integers(0, 255, (4, 52, 52)).astype(np.uint8) for _ in range(n)]
        s2 = [(rng.random((4, 6, 10, 10)) * 9000).astype(np.float32) for _ in range(n)]
        lab = [rng.integers(0, 3, (1, 52, 52)).astype(np.uint8) for _ in range(n)]
        random.seed(1)
        gen = P.HybridDataGenerator(s2files=s2, lstm_dim=(3, 8, 8, 6), unet_dim=(48, 48), labelfiles=lab, naipfiles=naip, batch_size=2, n_channels=4,
                                    n_classes=3, to_fit=True, shuffle=False, lc_transitions=[], lu_transitions=[])
        (xu, xl), y = gen[0]
        assert tuple(xu.shape) == (2, 48, 48, 4) and xl.shape == (2, 3, 8, 8, 6) and tuple(y.shape) == (2, 48, 48, 3)
        assert float(y.sum()) == 2 * 48 * 48 and xl.max() <= 1.1
        mt.reset_uids(); mt.set_seed(2)
        m = lt.get_hybrid_model((48, 48, 4), (3, 8, 8, 6), 3, filters=[32, 64], factors=[3, 2], compile_model=True, optim=mt.Adam(1e-3),
                                loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, [1.0, 1.0, 1.0]))
        h = m.fit(gen, epochs=2)
        assert len(h.history['loss']) == 2 and np.isfinite(h.history['loss']).all()
        pred_gen = P.HybridDataGenerator(s2files=s2, lstm_dim=(3, 8, 8, 6), unet_dim=(48, 48), naipfiles=naip, batch_size=2, n_channels=4, n_classes=3,
                                         to_fit=False, shuffle=False)
        xu, xl = pred_gen[1]
        pr = m.predict([xu, xl])
        assert pr.shape == (2, 48, 48, 3)
    finally:
        mt.set_compute_dtype('bfloat16')


def test_hierarchical_model_matches_oracle(lt):
    """get_hierarchical_model (utils/model_tools.py:1016-1060), fp32: the three softmax outputs and every gradient of one training step
    (weighted categorical cross-entropy on each head, summed) against the float64 oracle -- the atrous trunk's residual sums, a middle
    block feeding both its successor and a head, the last block feeding two heads, and the LSTM branch through the nearest resize."""
    from satellite_computervision_amd import model_tools as mt
    mt.set_compute_dtype('float32')
    try:
        ncls, nacnn, nsub, nf, depth = 3, 4, 5, 16, 3
        B, H, W, Cc, T, hh, ww, lc = 2, 24, 24, 4, 2, 8, 8, 6
        o = CL.HierarchicalOracle(ncls, nacnn, nsub, Cc, lc, nf, depth, rec_act=lt.RECURRENT_ACTIVATION, seed=3)
        m = lt.get_hierarchical_model(ncls, nacnn, nsub, (H, W, Cc), (T, hh, ww, lc), nf, depth)
        w = {}
        for lk, lv in o.p.items():
            for pk, pv in lv.items():
                lv[pk] = pv.astype(np.float32).astype(np.float64)
                w[f'{lk}/{pk}'] = lv[pk]
        lnames = {'l1': 'conv_lstm', 'l2': 'dilated_conv_lstm', 'bn1': 'batch_norm', 'bn2': 'batch_norm2'}
        for lk, lv in o.lstm.p.items():
            if lk == 'dense':
                continue
            for pk, pv in lv.items():
                lv[pk] = pv.astype(np.float32).astype(np.float64)
                w[f'{lnames[lk]}/{pk}'] = lv[pk]
        m.set_weights_dict(w)
        rng = np.random.default_rng(6)
        xa = rng.random((B, H, W, Cc)).astype(np.float32)
        xl = rng.random((B, T, hh, ww, lc)).astype(np.float32)
        refs = o.forward(xa.astype(np.float64), xl.astype(np.float64))
        ys = [np.eye(k, dtype=np.float32)[rng.integers(0, k, (B, H, W))] for k in (nsub, nacnn, ncls)]
        m.compile(optimizer=mt.Adam(0.0), loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, [1.0] * 8))
        # (one weight vector per head: the compiled loss carries max(n) weights, each head uses its first n)
        losses, dls = [], []
        for r, y, k in zip(refs, ys, (nsub, nacnn, ncls)):
            l, dp, _ = OL.weighted_categorical_crossentropy(y.astype(np.float64), r, [1.0] * k)
            losses.append(l)
            dls.append(K.softmax_bwd(r, dp))
        g_ref = o.backward(dls)
        m._loss.weights = None
        # per-head weights: run the step with unit weights of the right length per head
        import satellite_computervision_amd.lstm_tools as LT
        orig = LT._SeqModelBase._loss_grad

        def per_head(self, out, y_true, activation):
            self._loss.weights = np.ones(out.shape[-1], np.float32)
            return orig(self, out, y_true, activation)
        LT._SeqModelBase._loss_grad = per_head
        try:
            loss = m.train_on_batch([xa, xl], ys)
        finally:
            LT._SeqModelBase._loss_grad = orig
        np.testing.assert_allclose(loss, sum(losses), rtol=1e-4)
        outs = m.predict([xa, xl])          # inference-mode BatchNorm: shapes and normalisation only
        assert [o_.shape for o_ in outs] == [r.shape for r in refs]
        for ok, gv in g_ref.items():
            if ok == 'lstm.input':
                continue
            if ok.startswith('lstm.'):
                pre, pk = ok[5:].split('.', 1)
                dk = f'{lnames[pre]}/{pk}'
            else:
                pre, pk = ok.split('.', 1)
                dk = f'{pre}/{pk}'
            g = m.P.g(dk).cpu().numpy().astype(np.float64).reshape(gv.shape)
            assert cosine(g, gv) > 0.9999 and rel(g, gv) < 5e-3, (ok, cosine(g, gv), rel(g, gv))
    finally:
        mt.set_compute_dtype('bfloat16')


def test_sequence_models_save_load_evaluate(lt, tmp_path):
    """Model.save_weights / load_weights / evaluate as the reference's training scripts call them (utils/model_tools.py:1162-1196):
    a second model built from the same arguments reproduces the first one's predictions bit for bit after load_weights (the hybrid
    includes its U-Net branch), and evaluate() equals the loss train_on_batch reports with a zero learning rate."""
    from satellite_computervision_amd import model_tools as mt
    mt.set_compute_dtype('float32')
    try:
        rng = np.random.default_rng(3)
        # ---- get_lstm_model
        mt.reset_uids(); mt.set_seed(1)
        a = lt.get_lstm_model(4, 3, 3)
        x = rng.random((3, 3, 16, 16, 4)).astype(np.float32)
        y = rng.random((3, 16, 16, 3)).astype(np.float32)
        a.compile(optimizer=mt.Adam(1e-3), loss=mt.mse_4d)
        for _ in range(3):
            a.train_on_batch(x, y)
        a.save_weights(str(tmp_path / 'lstm'))
        mt.reset_uids(); mt.set_seed(99)
        b = lt.get_lstm_model(4, 3, 3)
        assert not np.array_equal(a.predict(x), b.predict(x))
        b.load_weights(str(tmp_path / 'lstm'))
        np.testing.assert_array_equal(a.predict(x), b.predict(x))
        b.compile(optimizer=mt.Adam(0.0), loss=mt.mse_4d)
        ev = b.evaluate(x, y)
        assert np.isfinite(ev) and ev > 0
        # (evaluate runs the inference graph -- moving statistics --, train_on_batch the training graph: equal only up to BatchNorm's mode)
        # ---- get_hybrid_model
        mt.reset_uids(); mt.set_seed(2)
        h1 = lt.get_hybrid_model((48, 48, 4), (3, 8, 8, 4), 3, filters=[32, 64], factors=[3, 2])
        xu = rng.random((2, 48, 48, 4)).astype(np.float32)
        xl = rng.random((2, 3, 8, 8, 4)).astype(np.float32)
        yy = np.eye(3, dtype=np.float32)[rng.integers(0, 3, (2, 48, 48))]
        h1.compile(optimizer=mt.Adam(1e-3), loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, [1.0] * 3))
        for _ in range(2):
            h1.train_on_batch([xu, xl], yy)
        h1.save_weights(str(tmp_path / 'hybrid.npz'))
        mt.reset_uids(); mt.set_seed(77)
        h2 = lt.get_hybrid_model((48, 48, 4), (3, 8, 8, 4), 3, filters=[32, 64], factors=[3, 2])
        h2.load_weights(str(tmp_path / 'hybrid.npz'))
        np.testing.assert_array_equal(h1.predict([xu, xl]), h2.predict([xu, xl]))
        h2.compile(optimizer=mt.Adam(0.0), loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, [1.0] * 3))
        ev = h2.evaluate([xu, xl], yy)
        assert np.isfinite(ev) and ev > 0
        with pytest.raises(ValueError):
            a.load_weights(str(tmp_path / 'hybrid.npz'))          # a file without this model's variables
    finally:
        mt.set_compute_dtype('bfloat16')


def test_lstm_training_step_graph_replay_matches_eager(lt):
    """SATCV_LSTM_GRAPH: after two eager steps the third is captured and later ones replay.  Eight steps of get_lstm_model with and
    without replay from the same seed and batches: equal loss curves and parameters (up to the summation order of the kernels a capture
    selects: split-K slabs are never allocated inside a capture), learning-rate changes between replays take effect."""
    from satellite_computervision_amd import model_tools as mt
    mt.set_compute_dtype('float32')
    try:
        rng = np.random.default_rng(11)
        xs = [rng.random((4, 3, 16, 16, 4)).astype(np.float32) for _ in range(8)]
        ys = [rng.random((4, 16, 16, 3)).astype(np.float32) for _ in range(8)]

        def run(flag):
            os.environ['SATCV_LSTM_GRAPH'] = flag
            mt.reset_uids(); mt.set_seed(5)
            m = lt.get_lstm_model(4, 3, 3)
            opt = mt.Adam(1e-3)
            m.compile(optimizer=opt, loss=mt.mse_4d)
            losses = []
            for i, (x, y) in enumerate(zip(xs, ys)):
                if i == 6:
                    opt._lr = 1e-4                      # a ReduceLROnPlateau-style change between replays
                losses.append(m.train_on_batch(x, y))
            return m, losses

        try:
            me, le = run('0')
            mg, lg = run('1')
        finally:
            os.environ.pop('SATCV_LSTM_GRAPH', None)
        assert any('g' in st for st in mg._graphs.values()), 'no step was captured'
        assert not any('g' in st for st in me._graphs.values())
        np.testing.assert_allclose(lg, le, rtol=2e-4)
        we, wg = me.get_weights_dict(), mg.get_weights_dict()
        for k in we:
            assert np.abs(we[k] - wg[k]).max() < 3e-4, (k, np.abs(we[k] - wg[k]).max())
        assert le[-1] < le[0]
    finally:
        mt.set_compute_dtype('bfloat16')


def test_lstm_graph_capture_after_an_interleaved_predict_repacks_weights(lt):
    """train x2 -> predict -> train ...: the eager predict() repacks the operand images, so the step that gets CAPTURED finds them fresh.  The
    captured graph must hold the repack launch all the same (_Params.prepare_capture) -- otherwise every replay multiplies with the images of
    capture time (the loss would stop falling: ~30 % off by the eighth step).  Eight steps with a predict() before the third, with and without replay: the
    same losses and parameters up to what Adam's first steps make of the kernels' different summation orders (a capture never selects split-K)."""
    from satellite_computervision_amd import model_tools as mt
    mt.set_compute_dtype('float32')
    try:
        rng = np.random.default_rng(23)
        xs = [rng.random((4, 3, 16, 16, 4)).astype(np.float32) for _ in range(8)]
        ys = [rng.random((4, 16, 16, 3)).astype(np.float32) for _ in range(8)]

        def run(flag):
            os.environ['SATCV_LSTM_GRAPH'] = flag
            mt.reset_uids(); mt.set_seed(7)
            m = lt.get_lstm_model(4, 3, 3)
            m.compile(optimizer=mt.Adam(1e-3), loss=mt.mse_4d)
            losses = []
            for i, (x, y) in enumerate(zip(xs, ys)):
                if i in (2, 5):
                    m.predict(x)                        # fresh images when step 3 is captured; an eager repack between replays
                losses.append(m.train_on_batch(x, y))
            return m, losses

        try:
            me, le = run('0')
            mg, lg = run('1')
        finally:
            os.environ.pop('SATCV_LSTM_GRAPH', None)
        assert any('g' in st for st in mg._graphs.values()), 'no step was captured'
        np.testing.assert_allclose(lg, le, rtol=5e-3)
        assert le[-1] < 0.9 * le[2] and lg[-1] < 0.9 * lg[2]
        we, wg = me.get_weights_dict(), mg.get_weights_dict()
        for k in we:
            assert np.abs(we[k] - wg[k]).max() < 3e-3, (k, np.abs(we[k] - wg[k]).max())
    finally:
        mt.set_compute_dtype('bfloat16')


def test_lstm_model_dropout_matches_oracle_given_the_mask(lt):
    """build_lstm_layers(dropout=rate): layers.Dropout between the two ConvLSTM2D layers (utils/model_tools.py:699-700).  The device
    draws the mask; the oracle receives that mask and must reproduce loss and every gradient (fp32); inference ignores dropout."""
    from satellite_computervision_amd import model_tools as mt
    mt.set_compute_dtype('float32')
    try:
        B, T, H, W, Cc, ncls, rate = 2, 3, 16, 16, 6, 4, 0.3
        o = CL.LSTMLayersOracle(Cc, ncls, filters=64, rec_act=lt.RECURRENT_ACTIVATION, seed=5)
        mt.reset_uids(); mt.set_seed(3)
        m = lt.get_lstm_model(Cc, ncls, T, dropout=rate)
        mt.reset_uids(); mt.set_seed(3)
        m0 = lt.get_lstm_model(Cc, ncls, T)
        names = {'l1': 'conv_lstm', 'l2': 'dilated_conv_lstm', 'bn1': 'batch_norm', 'bn2': 'batch_norm2', 'dense': 'conv2d'}
        w = {}
        for lk, lv in o.p.items():
            for pk, pv in lv.items():
                lv[pk] = pv.astype(np.float32).astype(np.float64)
                w[f'{names[lk]}/{pk}'] = lv[pk]
        m.set_weights_dict(w); m0.set_weights_dict(w)
        rng = np.random.default_rng(8)
        x = rng.random((B, T, H, W, Cc)).astype(np.float32)
        y = rng.random((B, H, W, ncls)).astype(np.float32) * 1.5
        np.testing.assert_array_equal(m.predict(x), m0.predict(x))              # inference: dropout is the identity
        m.compile(optimizer=mt.Adam(0.0), loss=mt.mse_4d)
        loss = m.train_on_batch(x, y)
        mask = m.layers_.drop.mask.cpu().numpy().astype(np.float64).reshape(T, B, H, W, 64).transpose(1, 0, 2, 3, 4)
        vals = np.unique(mask)
        assert len(vals) == 2 and vals[0] == 0.0 and abs(vals[1] - 1.0 / (1.0 - rate)) < 1e-6
        assert abs((mask > 0).mean() - (1.0 - rate)) < 0.02
        out_ref = o.forward(x.astype(np.float64), mask1=mask)
        loss_ref, dout = OL.mse_4d(y.astype(np.float64), out_ref)
        g_ref = o.backward(dout)
        np.testing.assert_allclose(loss, loss_ref, rtol=1e-4)
        key = {'l1.kernel': 'conv_lstm/kernel', 'l1.recurrent_kernel': 'conv_lstm/recurrent_kernel', 'l1.bias': 'conv_lstm/bias',
               'l2.kernel': 'dilated_conv_lstm/kernel', 'l2.recurrent_kernel': 'dilated_conv_lstm/recurrent_kernel', 'l2.bias': 'dilated_conv_lstm/bias',
               'bn1.gamma': 'batch_norm/gamma', 'bn1.beta': 'batch_norm/beta', 'bn2.gamma': 'batch_norm2/gamma', 'bn2.beta': 'batch_norm2/beta',
               'dense.kernel': 'conv2d/kernel', 'dense.bias': 'conv2d/bias'}
        for ok, dk in key.items():
            g = m.P.g(dk).cpu().numpy().astype(np.float64).reshape(g_ref[ok].shape)
            assert cosine(g, g_ref[ok]) > 0.9999 and rel(g, g_ref[ok]) < 2e-3, (ok, cosine(g, g_ref[ok]), rel(g, g_ref[ok]))
        # a second step draws a different mask; steps with dropout are never replayed from a graph
        m.train_on_batch(x, y); m.train_on_batch(x, y); m.train_on_batch(x, y)
        mask2 = m.layers_.drop.mask.cpu().numpy().reshape(T, B, H, W, 64).transpose(1, 0, 2, 3, 4)
        assert (mask2 != mask).mean() > 0.2
        assert not any('g' in st for st in m._graphs.values())
    finally:
        mt.set_compute_dtype('bfloat16')


def test_hybrid_model_with_dropout_trains(lt):
    """get_hybrid_model(dropout=rate) (utils/model_tools.py:897-905): SpatialDropout2D inside the U-Net, on the U-Net output and on the
    LSTM output, Dropout inside the LSTM stack.  Training moves every parameter group with a finite, falling loss; inference is
    deterministic and normalised."""
    from satellite_computervision_amd import model_tools as mt
    mt.set_compute_dtype('float32')
    try:
        mt.reset_uids(); mt.set_seed(4)
        ncls = 3
        m = lt.get_hybrid_model((48, 48, 4), (3, 8, 8, 6), ncls, filters=[32, 64], factors=[3, 2], dropout=0.2)
        rng = np.random.default_rng(2)
        xu = rng.random((2, 48, 48, 4)).astype(np.float32)
        xl = rng.random((2, 3, 8, 8, 6)).astype(np.float32)
        y = np.eye(ncls, dtype=np.float32)[(xu[..., 0] > 0.5).astype(np.int64) + (xu[..., 1] > 0.7)]
        p1, p2 = m.predict([xu, xl]), m.predict([xu, xl])
        np.testing.assert_array_equal(p1, p2)
        np.testing.assert_allclose(p1.sum(-1), 1.0, atol=1e-5)
        m.compile(optimizer=mt.Adam(2e-3), loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, [1.0] * ncls))
        w0, u0 = m.get_weights_dict(), m.unet.runtime.pflat.clone()
        losses = [m.train_on_batch([xu, xl], y) for _ in range(25)]
        assert np.isfinite(losses).all() and np.mean(losses[-5:]) < np.mean(losses[:5])
        w1 = m.get_weights_dict()
        for k in ('conv_lstm/kernel', 'dilated_conv_lstm/recurrent_kernel', 'lstm_dense/kernel', 'probabilities/kernel'):
            assert np.abs(w1[k] - w0[k]).max() > 0, k
        assert (m.unet.runtime.pflat != u0).any()
        sm = m.lstm_drop.mask.cpu().numpy()
        assert sm.shape == (2, 64) and set(np.unique(sm).round(4)) <= {0.0, 1.25}
    finally:
        mt.set_compute_dtype('bfloat16')


@pytest.mark.parametrize('which', ['autoencoder', 'hierarchical'])
def test_multi_output_sequence_models_graph_replay_matches_eager(lt, which):
    """get_lstm_autoencoder / get_hierarchical_model: training steps replayed from a captured graph (third step on) against the eager
    tape -- same seed and batches, six steps: equal losses and parameters up to the kernels' summation order."""
    from satellite_computervision_amd import model_tools as mt
    mt.set_compute_dtype('float32')
    try:
        rng = np.random.default_rng(21)
        if which == 'autoencoder':
            B, T, H, W, Cc, ncls = 2, 3, 16, 16, 6, 6
            xs = [[rng.random((B, T, H, W, Cc)).astype(np.float32), rng.standard_normal((B, H, W, 2)).astype(np.float32)] for _ in range(6)]
            ys = [[np.flip(x[0], axis=1).copy(), rng.random((B, H, W, ncls)).astype(np.float32)] for x in xs]
            build = lambda: lt.get_lstm_autoencoder(Cc, T, ncls)
            loss = mt.mse_4d
        else:
            ncls, nacnn, nsub, nf, depth = 3, 4, 5, 16, 3
            B, H, W, Cc, T, hh, ww, lc = 2, 24, 24, 4, 2, 8, 8, 6
            xs = [[rng.random((B, H, W, Cc)).astype(np.float32), rng.random((B, T, hh, ww, lc)).astype(np.float32)] for _ in range(6)]
            ys = [[np.eye(k, dtype=np.float32)[rng.integers(0, k, (B, H, W))] for k in (nsub, nacnn, ncls)] for _ in range(6)]
            build = lambda: lt.get_hierarchical_model(ncls, nacnn, nsub, (H, W, Cc), (T, hh, ww, lc), nf, depth)
            loss = lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, [1.0] * 8)

        def run(flag):
            os.environ['SATCV_LSTM_GRAPH'] = flag
            mt.reset_uids(); mt.set_seed(9)
            m = build()
            m.compile(optimizer=mt.Adam(1e-3), loss=loss)
            return m, [m.train_on_batch(x, y) for x, y in zip(xs, ys)]
        try:
            me, le = run('0')
            mg, lg = run('1')
        finally:
            os.environ.pop('SATCV_LSTM_GRAPH', None)
        assert any('g' in st for st in mg._graphs.values()), 'no step was captured'
        np.testing.assert_allclose(lg, le, rtol=5e-4)
        we, wg = me.get_weights_dict(), mg.get_weights_dict()
        for k in we:
            assert np.abs(we[k] - wg[k]).max() < 5e-4, (k, np.abs(we[k] - wg[k]).max())
    finally:
        mt.set_compute_dtype('bfloat16')


def test_sequence_models_keras_surface(lt, tmp_path):
    """compile(metrics=) / fit(validation_data=, callbacks=) / evaluate -> list aligned with metrics_names / load_weights(by_name, skip_mismatch)
    of the ConvLSTM2D family as the reference's call sites use them (utils/model_tools.py:1162-1176, notebooks/UNET_G4G_2019_solar.ipynb:1206-1275),
    and: arguments the implementation does not act on are refused, not swallowed."""
    import json
    from satellite_computervision_amd import model_tools as mt
    mt.set_compute_dtype('float32')
    try:
        B, T, H, W, C, ncls = 4, 3, 16, 16, 5, 3
        m = lt.get_lstm_model(C, ncls, T)
        rng = np.random.default_rng(1)
        x = rng.random((B, T, H, W, C)).astype(np.float32)
        y = np.eye(ncls, dtype=np.float32)[rng.integers(0, ncls, (B, H, W))]
        with pytest.raises(ValueError):
            m.compile(optimizer=mt.Adam(1e-3), loss=mt.mse_4d, metrics=['no_such_metric'])
        with pytest.raises(TypeError):
            m.compile(optimizer=mt.Adam(1e-3), loss=mt.mse_4d, run_eagerly=True)
        m.compile(optimizer=mt.Adam(1e-3), loss=mt.mse_4d, metrics=['accuracy', mt.MeanIoU(num_classes=ncls)])
        assert m.metrics_names == ['loss', 'accuracy', 'mean_io_u']
        ev = m.evaluate(x, y, batch_size=2)
        assert isinstance(ev, list) and len(ev) == 3 and all(np.isfinite(ev)) and 0 <= ev[1] <= 1 and 0 <= ev[2] <= 1
        # accuracy against the host argmax of predict()
        acc = float((m.predict(x).argmax(-1) == y.argmax(-1)).mean())
        assert abs(ev[1] - acc) < 1e-6
        ck = mt.ModelCheckpoint(str(tmp_path / 'best.npz'), monitor='val_loss', save_best_only=True, save_weights_only=True)
        tb = mt.TensorBoard(log_dir=str(tmp_path / 'tb'))
        h = m.fit(x, y, batch_size=2, epochs=3, validation_data=(x, y), callbacks=[ck, tb])
        assert len(h.history['loss']) == 3 and len(h.history['val_loss']) == 3 and len(h.history['val_mean_io_u']) == 3
        assert h.history['loss'][-1] < h.history['loss'][0]
        assert os.path.exists(tmp_path / 'best.npz') and ck.best == min(h.history['val_loss'])
        lines = [json.loads(l) for l in open(tmp_path / 'tb' / 'scalars.jsonl')]
        assert len(lines) == 3 and 'val_accuracy' in lines[0]
        # generator batches with a third element: None weights pass, real weights are refused
        assert np.isfinite(m.fit([(x[:2], y[:2], None), (x[2:], y[2:], None)], epochs=1).history['loss'][0])
        with pytest.raises(NotImplementedError):
            m.fit([(x[:2], y[:2], np.ones(2, np.float32))], epochs=1)
        with pytest.raises(NotImplementedError):
            m.train_on_batch(x, y, sample_weight=np.ones(B, np.float32))
        with pytest.raises(TypeError):
            m.fit(x, y, epochs=1, workers=4)
        with pytest.raises(TypeError):
            m.evaluate(x, y, callbacks=[])
        # weights: by_name is implied by the container; skip_mismatch skips what does not fit instead of raising
        m2 = lt.get_lstm_model(C, ncls + 1, T)
        with pytest.raises(ValueError):
            m2.load_weights(str(tmp_path / 'best.npz'))
        skipped = m2.load_weights(str(tmp_path / 'best.npz'), by_name=True, skip_mismatch=True)
        assert sorted(skipped) == ['conv2d/bias', 'conv2d/kernel']
        m3 = lt.get_lstm_model(C, ncls, T)
        m3.load_weights(str(tmp_path / 'best.npz'), by_name=True)
        m3.compile(optimizer=mt.Adam(0.0), loss=mt.mse_4d)
        assert isinstance(m3.evaluate(x, y), float)
        with pytest.raises(NotImplementedError):
            m.save_weights(str(tmp_path / 'w.h5'))
        with pytest.raises(NotImplementedError):
            m.load_weights(str(tmp_path / 'w.hdf5'))
        # multi-output model: evaluate lists the total, the per-output losses and the per-output metrics
        hm = lt.get_hierarchical_model(3, 4, 2, (16, 16, 4), (T, 8, 8, C), 16, 3)
        hm.compile(optimizer=mt.Adam(1e-3), loss=mt.mse_4d, metrics=['accuracy'])
        assert hm.metrics_names == ['loss', 'sub_probs_loss', 'acnn_probs_loss', 'lstm_probs_loss', 'sub_probs_accuracy', 'acnn_probs_accuracy', 'lstm_probs_accuracy']
    finally:
        mt.set_compute_dtype('bfloat16')
