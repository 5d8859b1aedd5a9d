"""CPU checks of the oracle itself: pinned tiling vectors, Appendix-D known answers,
and the NumPy restatement vs an independent PyTorch-CPU implementation."""
import os
import numpy as np
import pytest
import torch

from oracle import keras_ops as K
from oracle import losses as OL
from oracle import tiling as OT
from oracle import torch_unet as TU
from oracle.unet import UNetOracle, aspp_param_specs, aspp_forward

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


# ------------------------------------------------------------------ pinned: tiling
def test_tiling_matches_reference_fixtures():
    z = np.load(os.path.join(GOLD, 'tiling_reference.npz'))
    for key in z['cases']:
        _, h, w, c, buff, kernel = str(key).split('_')
        got = OT.generate_chip_indices(np.zeros((int(h), int(w), int(c)), np.float32), int(buff), int(kernel))
        assert np.array_equal(np.asarray(got, np.int64).reshape(-1, 2), z[str(key)]), key
    arr = z['pc_arr']
    tmpl = OT.predict_chips(arr, [tuple(i) for i in z['pc_idx']], np.zeros(arr.shape[:2]),
                            lambda x: 2.0 * x[..., :1], 32, 16)
    assert np.array_equal(tmpl, z['pc_template'])
    chips = OT.extract_chips(z['ec_arr'], 16, 32)
    assert np.array_equal(np.stack(chips), z['ec_chips'])


def test_tiling_appendix_d_known_answers():
    idx = OT.generate_chip_indices(np.zeros((1024, 1024, 4)), 128, 256)
    assert idx == [(y, x) for y in (64, 320, 576) for x in (64, 320, 576)]
    assert OT.generate_chip_indices(np.zeros((384, 384, 4)), 128, 256) == []
    assert OT.generate_chip_indices(np.zeros((640, 640, 4)), 128, 256) == [(64, 64)]
    idx = OT.generate_chip_indices(np.zeros((1000, 1300, 4)), 128, 256)
    assert idx == [(y, x) for y in (64, 320, 576) for x in (64, 320, 576, 832)]
    arr = np.random.default_rng(0).random((1024, 1024, 4))
    t = OT.predict_chips(arr, OT.generate_chip_indices(arr), np.zeros((1024, 1024)), lambda x: 2 * x[..., :1])
    assert np.array_equal(t[64:832, 64:832], 2 * arr[64:832, 64:832, 0])
    t[64:832, 64:832] = 0
    assert not t.any()


# ------------------------------------------------- unpinned: numpy vs torch-CPU ops
def _t(x):
    return torch.tensor(x, dtype=torch.float64)


@pytest.mark.parametrize('k,d', [(3, 1), (1, 1), (3, 3), (3, 6)])
def test_conv_fwd_bwd_vs_torch(k, d):
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 14, 15, 5)); w = rng.standard_normal((k, k, 5, 7)); b = rng.standard_normal(7)
    dy = rng.standard_normal((2, 14, 15, 7))
    y = K.conv2d_same(x, w, b, d)
    dx, dw, db = K.conv2d_same_bwd(x, w, dy, d)
    xt, wt, bt = _t(x).requires_grad_(), _t(w).requires_grad_(), _t(b).requires_grad_()
    yt = TU._conv(xt.permute(0, 3, 1, 2), wt, bt, d).permute(0, 2, 3, 1)
    yt.backward(_t(dy))
    np.testing.assert_allclose(y, yt.detach().numpy(), rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(dx, xt.grad.numpy(), rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(dw, wt.grad.numpy(), rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(db, bt.grad.numpy(), rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize('s', [2, 3])
def test_convT_vs_torch(s):
    rng = np.random.default_rng(2)
    x = rng.standard_normal((2, 5, 6, 4)); k = rng.standard_normal((s, s, 3, 4)); b = rng.standard_normal(3)
    dy = rng.standard_normal((2, 5 * s, 6 * s, 3))
    y = K.conv2d_transpose_ks(x, k, b)
    dx, dk, db = K.conv2d_transpose_ks_bwd(x, k, dy)
    xt, kt, bt = _t(x).requires_grad_(), _t(k).requires_grad_(), _t(b).requires_grad_()
    yt = torch.nn.functional.conv_transpose2d(xt.permute(0, 3, 1, 2), kt.permute(3, 2, 0, 1), bt, stride=s).permute(0, 2, 3, 1)
    yt.backward(_t(dy))
    np.testing.assert_allclose(y, yt.detach().numpy(), atol=1e-10)
    np.testing.assert_allclose(dx, xt.grad.numpy(), atol=1e-10)
    np.testing.assert_allclose(dk, kt.grad.numpy(), atol=1e-10)
    np.testing.assert_allclose(db, bt.grad.numpy(), atol=1e-10)


def test_bn_pool_vs_torch():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((3, 8, 9, 6)); g = rng.standard_normal(6); b = rng.standard_normal(6)
    dy = rng.standard_normal(x.shape)
    y, mean, var = K.batchnorm_train(x, g, b)
    dx, dg, db = K.batchnorm_train_bwd(x, g, mean, var, dy)
    xt, gt, bt = _t(x).requires_grad_(), _t(g).requires_grad_(), _t(b).requires_grad_()
    yt = torch.nn.functional.batch_norm(xt.permute(0, 3, 1, 2), None, None, gt, bt, True, 0.0, 1e-3).permute(0, 2, 3, 1)
    yt.backward(_t(dy))
    np.testing.assert_allclose(y, yt.detach().numpy(), atol=1e-10)
    np.testing.assert_allclose(dx, xt.grad.numpy(), atol=1e-9)
    np.testing.assert_allclose(dg, gt.grad.numpy(), atol=1e-9)
    np.testing.assert_allclose(db, bt.grad.numpy(), atol=1e-9)
    for f in (2, 3):
        xp = rng.standard_normal((2, 7, 9, 4))
        dyp = rng.standard_normal((2, 7 // f, 9 // f, 4))
        xt = _t(xp).requires_grad_()
        yt = torch.nn.functional.max_pool2d(xt.permute(0, 3, 1, 2), f, f).permute(0, 2, 3, 1)
        yt.backward(_t(dyp))
        np.testing.assert_allclose(K.maxpool(xp, f), yt.detach().numpy())
        np.testing.assert_allclose(K.maxpool_bwd(xp, f, dyp), xt.grad.numpy())


def test_losses_vs_torch_autograd():
    rng = np.random.default_rng(4)
    logits = rng.standard_normal((2, 6, 5, 3))
    p = K.softmax(logits)
    lab = rng.integers(0, 3, (2, 6, 5))
    t = np.eye(3)[lab]
    pt = _t(p).requires_grad_()
    TU.weighted_cce_mean(_t(t), pt, [1.0, 20.0, 3.0]).backward()
    l, g, _ = OL.weighted_categorical_crossentropy(t, p, [1.0, 20.0, 3.0])
    np.testing.assert_allclose(g, pt.grad.numpy(), atol=1e-12)
    pt = _t(p).requires_grad_()
    lt = TU.weighted_bce_mean(_t(t), pt, 5.0); lt.backward()
    l, g = OL.weighted_bce(t, p, 5.0)
    np.testing.assert_allclose(l, lt.item(), rtol=1e-12)
    np.testing.assert_allclose(g, pt.grad.numpy(), atol=1e-12)
    # logits form
    xt = _t(logits).requires_grad_()
    lt = torch.nn.functional.binary_cross_entropy_with_logits(xt, _t(t), pos_weight=torch.tensor(5.0, dtype=torch.float64)); lt.backward()
    l, g = OL.weighted_bce(t, logits, 5.0, logits=True)
    np.testing.assert_allclose(l, lt.item(), rtol=1e-10)
    np.testing.assert_allclose(g, xt.grad.numpy(), atol=1e-12)
    # dice / iou / mse
    for gw in ([0.3, 0.5, 0.2], None):
        pt = _t(p).requires_grad_()
        tt = _t(t).reshape(2, 30, 3); pp = pt.reshape(2, 30, 3)
        if gw:
            w = torch.tensor(gw, dtype=torch.float64).view(1, 3)
        else:
            cnt = tt.sum(1); w = 1.0 / cnt ** 2; w = torch.where(torch.isfinite(w), w, torch.tensor(1e-6, dtype=torch.float64))
        num = (w * (tt * pp).sum(1)).sum(-1); den = (w * (tt + pp).sum(1)).sum(-1)
        lt = (1 - 2 * num / den).mean(); lt.backward()
        l, g = OL.gen_dice(t, p, global_weights=gw)
        np.testing.assert_allclose(l, lt.item(), rtol=1e-12)
        np.testing.assert_allclose(g, pt.grad.numpy(), atol=1e-12)
    pt = _t(p).requires_grad_(); tt = _t(t)
    lt = 1 - (tt * pt).sum() / (tt + (1 - tt) * pt).sum(); lt.backward()
    l, g = OL.iou_loss(t, p)
    np.testing.assert_allclose(l, lt.item(), rtol=1e-12)
    np.testing.assert_allclose(g, pt.grad.numpy(), atol=1e-12)
    tn = rng.standard_normal(p.shape); tn[0, 0, 0, 0] = np.nan
    l, g = OL.mse_4d(tn, p)
    fin = np.isfinite(tn)
    np.testing.assert_allclose(l, ((p - tn)[fin] ** 2).mean())
    assert g[0, 0, 0, 0] == 0


def test_unet_oracle_vs_torch_autograd_tiny():
    """Whole graph (as coded: single conv per level), fwd + all grads + one Keras-Adam step."""
    filters, factors = [4, 8], [2, 2]
    o = UNetOracle(2, 4, filters, factors, dtype=np.float64, seed=5)
    rng = np.random.default_rng(6)
    for n, _, kind in o.specs:                      # non-trivial BN params
        if kind in ('gamma', 'beta', 'bias'):
            o.params[n] = o.params[n] + 0.3 * rng.standard_normal(o.params[n].shape)
    x = rng.random((3, 16, 16, 4))
    lab = (rng.random((3, 16, 16)) < 0.3).astype(np.int64)
    t = np.eye(2)[lab]
    p0 = {k: v.copy() for k, v in o.params.items()}
    probs, classes = o.forward(x, training=True)
    loss, dprobs, _ = OL.weighted_categorical_crossentropy(t, probs, [1.0, 20.0])
    grads = o.backward(dprobs)

    tp = TU.params_to_torch(p0, torch.float64)
    xt = _t(x).requires_grad_()
    pr, cl = TU.unet_forward(tp, xt, filters, factors, training=True)
    lt = TU.weighted_cce_mean(_t(t), pr, [1.0, 20.0]); lt.backward()
    np.testing.assert_allclose(probs, pr.detach().numpy(), atol=1e-10)
    assert np.array_equal(classes, cl.numpy())
    np.testing.assert_allclose(loss, lt.item(), rtol=1e-10)
    for n in o.trainable:
        np.testing.assert_allclose(grads[n], tp[n].grad.numpy(), atol=1e-8, err_msg=n)
    np.testing.assert_allclose(grads['input'], xt.grad.numpy(), atol=1e-8)
    # moving stats: encoder/center updated twice (Q2), decoder once
    cnt = x.shape[0] * x.shape[1] * x.shape[2]                          # dec0 runs at full resolution: N*H*W values per channel
    np.testing.assert_allclose(o.params['dec0.bn1.moving_var'],         # TF 2.x fused BatchNorm: Bessel-corrected variance in the moving average
                               0.99 + 0.01 * o.cache['dec0.conv1'][2][1] * cnt / (cnt - 1), atol=1e-12)
    mean0 = o.cache['enc0.conv'][2][0]
    np.testing.assert_allclose(o.params['enc0.bn.moving_mean'], mean0 * (1 - 0.99 ** 2), atol=1e-12)
    # inference-mode forward agrees too
    pi, ci = o.forward(x, training=False)
    tp2 = TU.params_to_torch(o.params, torch.float64, requires_grad=False)
    pr2, cl2 = TU.unet_forward(tp2, _t(x), filters, factors, training=False)
    np.testing.assert_allclose(pi, pr2.numpy(), atol=1e-10)
    # Adam step
    g_t = {n: tp[n].grad for n in o.trainable}
    m = {n: torch.zeros_like(tp[n]) for n in o.trainable}; v = {n: torch.zeros_like(tp[n]) for n in o.trainable}
    TU.keras_adam_(tp, g_t, m, v, 1)
    o.adam_step(grads)
    for n in ('enc0.conv.kernel', 'dec1.up.kernel', 'dec0.bn0.gamma', 'probs.kernel'):
        np.testing.assert_allclose(o.params[n], tp[n].detach().numpy(), atol=1e-9, err_msg=n)


def test_param_count_matches_survey():
    o = UNetOracle(2, 4, dtype=np.float32)
    total = sum(v.size for v in o.params.values())
    train = sum(o.params[n].size for n in o.trainable)
    assert total == 18536898 and train == 18524930
    o13 = UNetOracle(2, 13, dtype=np.float32)
    assert sum(v.size for v in o13.params.values()) == 18539490


def test_aspp_shapes():
    rng = np.random.default_rng(0)
    params = {}
    for n, shape, kind in aspp_param_specs(8, 16):
        params[n] = rng.standard_normal(shape) * 0.1 if kind in ('kernel', 'bias', 'beta', 'mm') else np.ones(shape)
    y = aspp_forward(params, rng.standard_normal((1, 32, 32, 8)))
    assert y.shape == (1, 32, 32, 16) and (y >= 0).all()


def test_input_pipeline_restatement_matches_reference_array_tools():
    """oracle/input_pipeline.py against outputs of the REAL utils/array_tools.py (generated by
    tests/golden/make_reference_fixtures.py): 16 flip/rot90 cases, the colour augmentation with the recorded random draws,
    merge_classes rule order."""
    from oracle import input_pipeline as ip
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'array_tools_reference.npz'))
    x = z['morph_in']
    for v in (0, 1):
        for h in (0, 1):
            for r in range(4):
                assert np.array_equal(ip.aug_array_morph(x, bool(v), bool(h), r), z[f'morph_{v}{h}{r}'])
    for seed in (0, 1):
        c, b = (float(v_) for v_ in z[f'color_mul_{seed}'])          # python floats, as random.uniform returns
        got = ip.aug_array_color(z['color_in'], c, b)
        np.testing.assert_array_equal(got, z[f'color_out_{seed}'])
    lab = np.arange(256).reshape(1, 1, 16, 16)
    lc = [(12, 3), (11, 3), (10, 3), (9, 8), (255, 0)]
    assert np.array_equal(ip.merge_classes(lab, lc, lab), z['merge_lc_default'])
    lut = ip.merge_lut(lc)
    assert np.array_equal(np.where(lut[lab] >= 0, lut[lab], lab), z['merge_lc_default'])
    dup = [(5, 1), (1, 7), (5, 2)]
    lut = ip.merge_lut(dup)
    assert np.array_equal(np.where(lut[lab] >= 0, lut[lab], lab), z['merge_dup_rule'])


def test_convlstm_oracle_matches_torch_autograd():
    """oracle/convlstm.py (Keras ConvLSTM2D cell, hand-written BPTT) against an independent torch-CPU autograd restatement: both
    recurrent activations, linear and tanh cell activations, dilated input convolution, sequence / last-state outputs, and a gradient
    entering through return_state."""
    import torch
    from oracle import convlstm as CL
    torch.set_default_dtype(torch.float64)
    try:
        def conv(x, k, b, d):
            y = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), k.permute(3, 2, 0, 1), b, padding=d * (k.shape[0] - 1) // 2, dilation=d)
            return y.permute(0, 2, 3, 1)
        for rk in ('hard_sigmoid', 'sigmoid'):
            for act in (None, 'tanh'):
                for dil, rs in ((1, True), (3, False)):
                    rng = np.random.default_rng(0)
                    B, T, H, W, Cc, F = 2, 3, 7, 6, 5, 4
                    p = CL.convlstm_init(rng, Cc, F)
                    p['bias'] = p['bias'] + 0.1 * rng.standard_normal(4 * F)
                    x = rng.standard_normal((B, T, H, W, Cc))
                    out, cache = CL.convlstm_forward(x, p, dil, act, rk, rs)
                    dout, dhl = rng.standard_normal(out.shape), rng.standard_normal((B, H, W, F))
                    dx, g = CL.convlstm_backward(dout, cache, dh_last=dhl)
                    tp = {k: torch.tensor(v, requires_grad=True) for k, v in p.items()}
                    tx = torch.tensor(x, requires_grad=True)
                    h, c, hs = torch.zeros(B, H, W, F), torch.zeros(B, H, W, F), []
                    ra = (lambda z: torch.clamp(0.2 * z + 0.5, 0, 1)) if rk == 'hard_sigmoid' else torch.sigmoid
                    aa = (lambda z: z) if act is None else torch.tanh
                    for t in range(T):
                        z = conv(tx[:, t], tp['kernel'], tp['bias'], dil) + conv(h, tp['recurrent_kernel'], None, 1)
                        zi, zf, zc, zo = z.split(F, dim=-1)
                        c = ra(zf) * c + ra(zi) * aa(zc)
                        h = ra(zo) * aa(c)
                        hs.append(h)
                    o_t = torch.stack(hs, 1) if rs else h
                    np.testing.assert_allclose(o_t.detach().numpy(), out, atol=1e-12)
                    ((o_t * torch.tensor(dout)).sum() + (h * torch.tensor(dhl)).sum()).backward()
                    for k in p:
                        np.testing.assert_allclose(tp[k].grad.numpy(), g[k], atol=1e-10)
                    np.testing.assert_allclose(tx.grad.numpy(), dx, atol=1e-10)
        # the nearest-neighbour resize and its adjoint
        a = rng.standard_normal((2, 4, 5, 3))
        up, idx = CL.resize_nearest(a, 11, 9)
        g = rng.standard_normal(up.shape)
        assert abs((up * g).sum() - (a * CL.resize_nearest_bwd(g, idx, 4, 5)).sum()) < 1e-9
    finally:
        torch.set_default_dtype(torch.float32)


def test_round_bf16_matches_torch_and_store_dtype_oracle_stays_close():
    """round 4: oracle/unet.py::round_bf16 is torch's round-to-nearest-even bfloat16 conversion; UNetOracle(store_dtype='bfloat16')
    -- the oracle the bf16 chain tests compare against -- stores bf16-representable tensors and stays within the storage rounding of the
    unrounded chain on a tiny net (forward 2e-2; gradients: cosine > 0.8 -- storage rounding through BatchNorm backward passes at 512 values
    per channel is NOT small, which is why the device is compared with THIS oracle and not with the unrounded one: DESIGN.md section 4)."""
    from oracle.unet import round_bf16
    rng = np.random.default_rng(0)
    v = np.concatenate([rng.standard_normal(4096) * 10.0 ** rng.integers(-6, 6, 4096), [0.0, -0.0, 1.0, 1.00390625, 1.0078125, 3.0e38]])
    want = torch.tensor(v, dtype=torch.float32).to(torch.bfloat16).to(torch.float64).numpy()
    np.testing.assert_array_equal(round_bf16(v.astype(np.float64)), want)
    filters, factors = [8, 16], [2, 2]
    x = rng.random((2, 16, 16, 4))
    t = np.eye(2)[(rng.random((2, 16, 16)) < 0.3).astype(np.int64)]
    outs = {}
    for sd in (None, 'bfloat16'):
        o = UNetOracle(2, 4, filters, factors, dtype=np.float64, seed=5, store_dtype=sd)
        probs, _ = o.forward(x, training=True)
        _, dprobs, _ = OL.weighted_categorical_crossentropy(t, probs, [1.0, 20.0])
        outs[sd] = (probs, o.backward(dprobs), o)
    o_b = outs['bfloat16'][2]
    y = o_b.cache['enc0.conv'][0] if isinstance(o_b.cache['enc0.conv'], (tuple, list)) else None
    if y is not None and isinstance(y, np.ndarray):
        np.testing.assert_array_equal(round_bf16(y), y)                   # a stored tensor is bf16-representable
    np.testing.assert_allclose(outs['bfloat16'][0], outs[None][0], atol=2e-2)
    for n in outs[None][2].trainable:
        a, b = outs[None][1][n].ravel(), outs['bfloat16'][1][n].ravel()
        if np.linalg.norm(a) > 1e-12:
            assert a @ b / (np.linalg.norm(a) * np.linalg.norm(b)) > 0.8, n


def test_lstm_layers_oracle_dropout_mask_gradients_by_finite_differences():
    """round 4: LSTMLayersOracle.forward(x, mask1) -- the layers.Dropout between the two ConvLSTM2D layers as a given mask
    (utils/model_tools.py:699-700): analytic gradients of a scalar loss against central differences."""
    from oracle import convlstm as CL
    rng = np.random.default_rng(3)
    B, T, H, W, Cc, ncls = 1, 2, 5, 5, 3, 2
    o = CL.LSTMLayersOracle(Cc, ncls, filters=8, rec_act='sigmoid', seed=2)
    x = rng.random((B, T, H, W, Cc))
    mask = (rng.random((B, T, H, W, 8)) > 0.3) / 0.7
    wgt = rng.standard_normal((B, H, W, ncls))

    def loss():
        return float((o.forward(x, mask1=mask) * wgt).sum())
    base = loss()
    g = o.backward(wgt)
    assert np.isfinite(base)
    for key, (lk, pk) in {'l1.kernel': ('l1', 'kernel'), 'l2.recurrent_kernel': ('l2', 'recurrent_kernel'), 'bn1.gamma': ('bn1', 'gamma'),
                          'dense.kernel': ('dense', 'kernel')}.items():
        arr = o.p[lk][pk]
        for _ in range(3):
            idx = tuple(rng.integers(0, s) for s in arr.shape)
            old = arr[idx]
            arr[idx] = old + 1e-5; lp = loss()
            arr[idx] = old - 1e-5; lm = loss()
            arr[idx] = old
            fd = (lp - lm) / 2e-5
            assert abs(fd - g[key][idx]) <= 2e-5 * max(1.0, abs(fd)), (key, idx, fd, g[key][idx])
