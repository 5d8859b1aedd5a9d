"""Graph-level parity on the GPU: the Keras-style surface (get_unet_model / predict / fit /
predict_chips) through the C ABI against the NumPy oracle on the same weights and tiles."""
import os
import numpy as np
import pytest
import torch

from oracle import losses as OL
from oracle import tiling as OT
from oracle.unet import UNetOracle

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def mt():
    from satellite_computervision_amd import model_tools
    assert torch.cuda.is_available()
    return model_tools


def build_pair(mt, dtype, nclasses, nchannels, filters, factors, seed=3, perturb=True, store_dtype=None):
    """same weights in the oracle (float64) and in the device model."""
    o = UNetOracle(nclasses, nchannels, filters, factors, dtype=np.float64, seed=seed, store_dtype=store_dtype)
    rng = np.random.default_rng(seed + 1)
    if perturb:
        for n, _, kind in o.specs:
            if kind in ('gamma', 'beta', 'bias', 'mm'):
                o.params[n] = o.params[n] + 0.2 * rng.standard_normal(o.params[n].shape)
            if kind == 'mv':
                o.params[n] = o.params[n] * (0.5 + rng.random(o.params[n].shape))
    for n in o.params:                       # values exactly representable in fp32
        o.params[n] = o.params[n].astype(np.float32).astype(np.float64)
    mt.reset_uids()
    m = mt.get_unet_model(nclasses, nchannels, filters, factors)
    m.compute_dtype = dtype
    names = mt.structural_names(m)
    m.set_weights_dict({names[k]: v for k, v in o.params.items()})
    return o, m, names


def consumers_of(m, t):
    return [nd for nd in m.nodes if any(i is t for i in nd.inputs)]


def iou(a, b, cls=1):
    inter = np.logical_and(a == cls, b == cls).sum()
    union = np.logical_or(a == cls, b == cls).sum()
    return inter / union if union else 1.0


def balance_head(o, x):
    """shift the class-1 head bias so that about half of the pixels are class 1 (a random-weight
    network otherwise predicts a single class and the mask comparison is vacuous)."""
    p, _ = o.forward(x[:1], training=False)
    med = np.median(np.log(p[..., 1] / p[..., 0]))
    o.params['probs.bias'][1] -= np.float32(med)


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
def test_tiny_unet_predict_and_train_step(mt, dtype):
    filters, factors = [32, 64], [2, 2]
    o, m, names = build_pair(mt, dtype, 2, 4, filters, factors)
    rng = np.random.default_rng(0)
    x = rng.random((3, 32, 32, 4)).astype(np.float32)
    lab = (rng.random((3, 32, 32)) < 0.3).astype(np.int64)
    t = np.eye(2)[lab].astype(np.float32)
    f32 = dtype == 'float32'
    # ---- inference-mode forward (moving statistics)
    p_ref, c_ref = o.forward(x, training=False)
    probs, classes = m.predict(x)
    assert probs.shape == (3, 32, 32, 2) and probs.dtype == np.float32
    assert classes.shape == (3, 32, 32) and classes.dtype == np.int32
    np.testing.assert_allclose(probs, p_ref, atol=2e-5 if f32 else 3e-2)
    margin = np.abs(p_ref[..., 0] - p_ref[..., 1])
    ok = margin > (1e-4 if f32 else 8e-2)
    assert np.array_equal(classes[ok], c_ref[ok])
    # ---- one training step: loss, every gradient, Adam update, BN moving statistics
    m.compile(optimizer=mt.Adam(9e-4), loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, [1.0, 20.0]),
              metrics=['categorical_accuracy', mt.MeanIoU(2)])
    p0 = {k: v.copy() for k, v in o.params.items()}
    pr, _ = o.forward(x, training=True)
    loss_ref, dprobs, _ = OL.weighted_categorical_crossentropy(t.astype(np.float64), pr, [1.0, 20.0])
    g_ref = o.backward(dprobs)
    loss = m.train_on_batch(x, t)
    np.testing.assert_allclose(loss, loss_ref, rtol=2e-5 if f32 else 3e-2)
    rt = m.runtime
    torch.cuda.synchronize()
    worst = {}
    for n in o.trainable:
        g = rt.get_grad(names[n]).cpu().numpy().astype(np.float64)
        scale = max(np.abs(g_ref[n]).max(), 1e-3 * max(np.abs(v).max() for v in g_ref.values() if isinstance(v, np.ndarray)))
        if n.endswith('.bias') and not n.startswith('probs'):
            # bias in front of a BatchNorm: the exact gradient is 0; both sides hold rounding noise
            assert np.abs(g).max() < (1e-4 if f32 else 5e-2) * max(np.abs(g_ref['probs.kernel']).max(), 1.0)
            continue
        err = np.abs(g - g_ref[n]).max() / scale
        worst[n] = err
        if f32:
            # exact up to fp32 rounding for the bulk of the elements; a ReLU whose pre-activation is ~1e-7 may flip
            # between fp32 and the float64 oracle and perturb its neighbourhood (measured: tools/diag_chain.py), so the
            # criterion is bulk exactness + tight global agreement rather than a max-norm bound
            l2 = np.linalg.norm(g - g_ref[n]) / max(np.linalg.norm(g_ref[n]), 1e-30)
            med = np.median(np.abs(g - g_ref[n])) / scale
            assert l2 < 1e-2, f'grad {n}: relL2 {l2:.3e} median {med:.3e} max {err:.3e}'
        else:
            # bf16 storage of activations and gradients: on this deliberately ill-conditioned case
            # (random weights, saturated loss ~10, class weight 20, random labels) BatchNorm's mean
            # removal cancels most of the incoming gradient, so the 2^-9 storage rounding is amplified
            # ~40x per the same mechanism that amplifies fp32's 6e-8 to 2e-6 (measured).
            # The direction must still agree with the float64 oracle.
            cos = (g * g_ref[n]).sum() / (np.linalg.norm(g) * np.linalg.norm(g_ref[n]))
            assert cos > 0.9 and err < 0.6, f'grad {n}: cos {cos:.4f} err {err:.3e}'
    # parameters after the Keras-Adam step (fp32 mode only: Adam's sign-like first step amplifies noise)
    o.adam_step(g_ref, lr=9e-4)
    if f32:
        for n in ('enc0.conv.kernel', 'dec1.conv1.kernel', 'dec0.up.kernel', 'dec0.bn0.gamma', 'probs.kernel', 'center.bn.beta'):
            got = rt.get_param(names[n]).cpu().numpy()
            ref = o.params[n]
            big = np.abs(g_ref[n]) > 1e-3 * np.abs(g_ref[n]).max()      # where the update direction is well conditioned
            np.testing.assert_allclose(got[big], ref[big], atol=2e-5, err_msg=n)
    for bnname, upd in (('enc0.bn', 2), ('center.bn', 2), ('dec0.bn0', 1), ('dec1.bn2', 1)):
        got = rt.get_param(names[bnname + '.moving_mean']).cpu().numpy()
        np.testing.assert_allclose(got, o.params[bnname + '.moving_mean'], atol=2e-5 if f32 else 5e-3, err_msg=bnname)
        got = rt.get_param(names[bnname + '.moving_var']).cpu().numpy()
        np.testing.assert_allclose(got, o.params[bnname + '.moving_var'], rtol=1e-4 if f32 else 2e-2, err_msg=bnname)


def test_full_unet_forward_256_fp32_mask_exact(mt):
    """BASELINE config 1/2 tile shape: get_unet_model(2, 4), 256x256x4, fp32 storage.
    Bit-exact argmax mask on margin-filtered pixels; probs within 1e-4."""
    o, m, names = build_pair(mt, 'float32', 2, 4, [32, 64, 128, 256, 512], [2, 2, 2, 2, 2], seed=11)
    rng = np.random.default_rng(1)
    x = (rng.beta(2, 5, (2, 256, 256, 4))).astype(np.float32)
    balance_head(o, x)
    m.set_weights_dict({names['probs.bias']: o.params['probs.bias']})
    p_ref, c_ref = o.forward(x, training=False)
    assert 0.2 < c_ref.mean() < 0.8
    probs, classes = m.predict(x, batch_size=2)
    np.testing.assert_allclose(probs, p_ref, atol=1e-4)
    ok = np.abs(p_ref[..., 0] - p_ref[..., 1]) > 1e-3
    assert ok.mean() > 0.9
    assert np.array_equal(classes[ok], c_ref[ok])
    assert abs(iou(classes, c_ref) - 1.0) < 1e-3


def test_full_unet_forward_256_bf16_iou(mt):
    """bf16 storage: per-pixel IoU of the class mask within 1e-3 of the oracle's on the same tiles."""
    o, m, names = build_pair(mt, 'bfloat16', 2, 4, [32, 64, 128, 256, 512], [2, 2, 2, 2, 2], seed=11)
    rng = np.random.default_rng(1)
    x = (rng.beta(2, 5, (2, 256, 256, 4))).astype(np.float32)
    balance_head(o, x)
    m.set_weights_dict({names['probs.bias']: o.params['probs.bias']})
    p_ref, c_ref = o.forward(x, training=False)
    assert 0.2 < c_ref.mean() < 0.8
    probs, classes = m.predict(x, batch_size=2)
    assert np.abs(probs - p_ref).max() < 0.08
    # random weights balanced at the median put many pixels on the decision boundary; the masks must
    # agree wherever the oracle's margin exceeds the bf16 noise, and overall on > 97 % of the pixels
    ok = np.abs(p_ref[..., 0] - p_ref[..., 1]) > 0.1
    assert np.array_equal(classes[ok], c_ref[ok])
    assert (classes == c_ref).mean() > 0.97
    assert abs(iou(classes, c_ref) - 1.0) < 0.05


def test_trained_model_bf16_iou_within_1e3(mt):
    """North-star parity target: on a TRAINED model the per-pixel IoU of the bf16 device mask is within
    1e-3 of the reference-semantics (float64 oracle) mask's IoU, both scored against the ground truth."""
    mt.reset_uids(); mt.set_seed(1)
    filters, factors = [32, 64], [2, 2]
    m = mt.get_unet_model(2, 4, filters=filters, factors=factors)
    m.compute_dtype = 'float32'
    rng = np.random.default_rng(7)

    def make(n):
        # spatially coherent tiles: low-frequency fields (8x8 noise, bilinearly upsampled) + pixel noise
        lo = torch.tensor(rng.random((n, 4, 8, 8)), dtype=torch.float32)
        x = torch.nn.functional.interpolate(lo, size=(64, 64), mode='bilinear', align_corners=False)
        x = (x.permute(0, 2, 3, 1).numpy() + 0.05 * rng.standard_normal((n, 64, 64, 4))).astype(np.float32)
        lab = (x[..., 0] + x[..., 3] > 1.0).astype(np.int64)
        return x, lab
    x, lab = make(32)
    m.compile(optimizer=mt.Adam(2e-3), loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, [1.0, 1.0]))
    m.fit(x, np.eye(2, dtype=np.float32)[lab], batch_size=8, epochs=100, verbose=0)
    w = m.get_weights_dict()
    xt, labt = make(8)
    names = mt.structural_names(m)
    o = UNetOracle(2, 4, filters, factors, dtype=np.float64)
    for k in o.params:
        o.params[k] = w[names[k]].astype(np.float64)
    p_ref, c_ref = o.forward(xt, training=False)
    iou_ref = iou(c_ref, labt)
    assert iou_ref > 0.85, iou_ref                    # the model has actually learned the task
    mt.reset_uids()
    mb = mt.get_unet_model(2, 4, filters=filters, factors=factors)
    mb.compute_dtype = 'bfloat16'
    mb.set_weights_dict({names[k]: w[names[k]] for k in o.params})
    _, c_bf = mb.predict(xt)
    _, c_f32 = m.predict(xt)
    # fp32 mode: the mask is the oracle's except possibly on pixels whose two probabilities tie to 1e-4 (the trained weights differ
    # from run to run -- float atomics in the bias / head gradients -- so such a pixel turns up in some runs; one pixel is 1e-4 of IoU)
    sure = np.abs(p_ref[..., 0] - p_ref[..., 1]) > 1e-4
    assert np.array_equal(c_f32[sure], c_ref[sure]) and (c_f32 != c_ref).sum() <= 3
    assert abs(iou(c_f32, labt) - iou_ref) < 4e-4
    assert abs(iou(c_bf, labt) - iou_ref) < 1e-3, (iou(c_bf, labt), iou_ref)


def _synthetic_step_batch(n, ch, seed=4):
    rng = np.random.default_rng(seed)
    x = rng.beta(2, 5, (n, 256, 256, ch)).astype(np.float32)
    lab = np.zeros((n, 256, 256), np.int64)
    for i in range(n):
        for _ in range(6):
            hh, ww = rng.integers(16, 96, 2)
            y0, x0 = rng.integers(0, 256 - hh), rng.integers(0, 256 - ww)
            lab[i, y0:y0 + hh, x0:x0 + ww] = 1
    return x, np.eye(2, dtype=np.float32)[lab]


def _grad_report(o, rt, names, g_ref):
    tot = np.sqrt(sum(np.linalg.norm(g_ref[k]) ** 2 for k in o.trainable if not (k.endswith('.bias') and not k.startswith('probs'))))
    report = []
    for k in o.trainable:
        if k.endswith('.bias') and not k.startswith('probs'):
            continue                          # bias in front of a training-mode BatchNorm: exact 0 here, rounding noise in the oracle
        g = rt.get_grad(names[k]).cpu().numpy().astype(np.float64)
        nr = np.linalg.norm(g_ref[k])
        l2 = np.linalg.norm(g - g_ref[k]) / max(nr, 1e-30)
        cos = (g * g_ref[k]).sum() / max(np.linalg.norm(g) * nr, 1e-30)
        report.append((k, l2, cos, nr / tot))
        assert np.isfinite(g).all(), k
    if os.environ.get('SATCV_TEST_VERBOSE'):
        for k, l2, cos, share in report:
            print(f'   {k:28s} relL2 {l2:.3e} cos {cos:.5f} share {share:.4f}')
    return report


def _layer_local_check(m, plan, rt, node, n, f32):
    """layer-local parity of ONE 3x3 conv_batch_act node's data- and weight-gradient launches on the device's own tensors: with the device's
    stored dy (and its activated input, rounded to the storage type as the loader rounds it) the float64 oracle's conv backward must
    reproduce the device's dW (fp32, exact products) and dx (stored type) -- no propagated noise in this comparison"""
    from oracle import keras_ops as K
    td = torch.float32 if f32 else torch.bfloat16
    lay, cx = node.layer, plan.node_ctx[id(node)]
    r = cx['r']
    fused_bwd = 'dy:' + lay.name not in plan.dbg
    if not fused_bwd:
        dy = plan.dbg['dy:' + lay.name].double().cpu().numpy()
    else:
        # thin layers (csrc/conv_bwd_fused.hip): dy never reaches memory -- rebuild it from what the launch read (g, the raw conv
        # output, the BatchNorm coefficients), rounded to the storage type as the kernel rounds it before its two products
        pz = plan.dbg['dyparts:' + lay.name]
        gt, goff, gld = pz['g']
        gq = gt.double().reshape(-1, gld)[:, goff:goff + pz['cout']].reshape(n, cx['r'].h, cx['r'].w, pz['cout'])
        if 'head' in pz:                # the block under the head: g was never stored, the launch formed it from the logit gradients
            dl, wname = pz['head']
            wh = rt.get_param(wname).reshape(pz['cout'], -1)
            gq = (dl.reshape(-1, wh.shape[1]) @ wh.t()).to(td).double().reshape(gq.shape)
        yq = pz['y'].double().reshape(-1, pz['ldy'])[:, pz['yoff']:pz['yoff'] + pz['cout']].reshape(gq.shape)
        af = {k_: pz['aff'][k_].double()[pz['aoff']:pz['aoff'] + pz['cout']] for k_ in ('scale', 'shift', 'mean', 'rstd')}
        if 'dp' in pz:                  # encoder blocks: + MaxPooling2D's gradient routed to the first maximum of every 2 x 2 window
            dpt, _, dpld = pz['dp']
            dpq = dpt.double().reshape(gq.shape[0], gq.shape[1] // 2, gq.shape[2] // 2, dpld)[..., :pz['cout']]
            am = pz['amax'].long()
            gq = gq.clone()
            for sub in range(4):
                gq[:, sub // 2::2, sub % 2::2, :] += torch.where(am == sub, dpq, torch.zeros_like(dpq))
        c1, c2 = pz['coef'].double()[0], pz['coef'].double()[1]
        gm = gq if pz['linear'] else torch.where(yq * af['scale'] + af['shift'] > 0, gq, torch.zeros_like(gq))
        dy = (af['scale'] * (gm - c1 - (yq - af['mean']) * af['rstd'] * c2)).to(td).double().cpu().numpy()
    a = torch.cat([t_ for t_, _ in r.srcs], -1).double()
    if r.affine is not None:
        a = a * r.affine['scale'].double() + r.affine['shift'].double()
        if r.relu:
            a = a.clamp_min(0)
        a = a.to(td).double()                   # the staged tile is rounded to the storage type
    a = a.cpu().numpy()
    kern = rt.get_param(lay.name + '/kernel').double().cpu().numpy()
    cin = kern.shape[2]
    kq = torch.tensor(kern, dtype=torch.float32).to(td).double().numpy()       # packed operand image
    dx_ref, dk_ref, _ = K.conv2d_same_bwd(a[..., :cin], kq, dy, 1)
    dk = rt.get_grad(lay.name + '/kernel').double().cpu().numpy()
    e = np.abs(dk - dk_ref).max() / max(np.abs(dk_ref).max(), 1e-30)
    # (fused layers: the kernel's dy and the rebuilt one can differ by one bf16 rounding on a few elements)
    assert e < (2e-4 if f32 else (4e-3 if fused_bwd else 2e-3)), f'local wgrad {lay.name} {kern.shape}: {e:.3e}'
    key = 'dx:' + lay.name
    if key in plan.dbg and len(consumers_of(m, node.inputs[0])) == 1:
        dx = plan.dbg[key].double().cpu().numpy()[..., :cin]
        e = np.abs(dx - dx_ref).max() / max(np.abs(dx_ref).max(), 1e-30)
        assert e < (2e-5 if f32 else 1.2e-2), f'local dgrad {lay.name} {kern.shape}: {e:.3e}'


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
def test_full_unet_training_step_matches_oracle(mt, dtype):
    """One training step of the BENCHMARKED network -- get_unet_model(2, 4) with its default filters [32 .. 512] + 1024-channel
    centre (utils/model_tools.py:394-415), 256x256x4 tiles -- against the float64 oracle: loss and EVERY gradient, so the deep
    (256..1024-channel) data- and weight-gradient instantiations that bench.py times are under parity, not only the thin ones.
    Criteria: fp32 relative L2 < 1e-2 per tensor (bulk exact; isolated ReLU-mask flips).  bf16: against the float64
    oracle that rounds every STORED tensor to bf16 where the device stores it (oracle/unet.py store_dtype: same values in, exact sums)
    every gradient tensor must agree to cosine >= 0.90 (>= 0.99 on the last decoder level; why not tighter: see the assertion); the
    drift against the unrounded float64 chain (accumulated storage rounding through 30 BatchNorm backward passes at batch 2: ~0.78 at
    the encoder) is reported, not asserted."""
    f32 = dtype == 'float32'
    o, m, names = build_pair(mt, dtype, 2, 4, [32, 64, 128, 256, 512], [2, 2, 2, 2, 2], seed=17, perturb=False,
                             store_dtype=None if f32 else 'bfloat16')
    n = 2
    x, t = _synthetic_step_batch(n, 4)
    m.compile(optimizer=mt.Adam(0.0), loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, [1.0, 20.0]))
    pr, _ = o.forward(x, training=True)
    loss_ref, dprobs, _ = OL.weighted_categorical_crossentropy(t.astype(np.float64), pr, [1.0, 20.0])
    g_ref = o.backward(dprobs)
    loss = m.train_on_batch(x, t)
    np.testing.assert_allclose(loss, loss_ref, rtol=2e-5 if f32 else 2e-3)
    rt = m.runtime
    torch.cuda.synchronize()
    report = _grad_report(o, rt, names, g_ref)
    if not f32:
        print('bf16 step, cosine per tensor: ' + ', '.join(f'{r_[0]} {r_[2]:.4f}' for r_ in report))
    for k, l2, cos, share in report:
        if f32:
            assert l2 < 1e-2, f'grad {k}: relL2 {l2:.3e} cos {cos:.6f}'
        else:
            # same stored values on both sides.  What is left is NOT storage rounding but the conditioning of the chain: a BatchNorm
            # backward output is orthogonal to 1 and to xhat by construction, so the weight gradient below it is a small residual
            # (N cov(a, dy)) and a relative error of 6e-4 in one of the two per-channel means -- the level the device's fp32 partial
            # sums reach (dec0.bn2.beta: 5.9e-4) -- shows up as ~4 % in that weight gradient (measured: 4.5e-2 at dec0.conv2.kernel with
            # every layer-local kernel check below at 2e-3).  The error compounds by ~0.3-0.5 % of cosine per BatchNorm layer towards
            # the encoder: 0.999 (dec0) ... 0.947 (enc0), against 0.78 for the unrounded chain.  DESIGN.md section 4.
            assert cos >= 0.90, f'grad {k}: cos {cos:.4f} relL2 {l2:.3e} vs the bf16-storage oracle'
            if k.startswith(('probs', 'dec0.')):
                assert cos >= (0.995 if k.startswith(('probs', 'dec0.conv2', 'dec0.bn2')) else 0.98), f'grad {k}: cos {cos:.4f}'
            elif k.startswith(('dec1.', 'dec2.', 'dec3.')):
                # per decoder level, from the measured table this test prints (round 6, 4 bands: dec1 0.973-0.993, dec2 0.962-0.977,
                # dec3 0.950-0.968; dec4 / centre / encoder 0.939-0.960 keep the blanket bound) with ~2 points of margin (the 13-band network sits ~1 point lower) for the summation
                # orders of other tile choices -- a mis-scaled slab in one layer would cost that layer tens of points
                lim = {'dec1': 0.955, 'dec2': 0.94, 'dec3': 0.925}[k[:4]]
                assert cos >= lim, f'grad {k}: cos {cos:.4f} (decoder level {k[3]}: >= {lim})'
    if not f32:
        # reported drift against the UNROUNDED float64 chain
        o2 = UNetOracle(2, 4, [32, 64, 128, 256, 512], [2, 2, 2, 2, 2], dtype=np.float64, seed=17)
        o2.params = {k: v.copy() for k, v in o.params.items()}
        pr2, _ = o2.forward(x, training=True)
        _, dp2, _ = OL.weighted_categorical_crossentropy(t.astype(np.float64), pr2, [1.0, 20.0])
        rep2 = _grad_report(o2, rt, names, o2.backward(dp2))
        wk, _, wcos, _ = min(rep2, key=lambda r_: r_[2])
        print(f'bf16 step vs the unrounded float64 chain: lowest cosine {wcos:.3f} ({wk}); vs the bf16-storage oracle: '
              f'{min(r_[2] for r_ in report):.4f}')
        assert wcos > 0.5
    worst = sorted(report, key=lambda r: -r[1])[:4]
    print(f'full-size {dtype} step: loss {loss:.6f} (oracle {loss_ref:.6f}); worst relL2 ' + ', '.join(f'{k} {l2:.2e}' for k, l2, _, _ in worst))
    if f32:
        # the deep layers specifically (1024 / 512-channel kernels)
        for k in ('center.conv.kernel', 'dec4.conv1.kernel', 'dec4.conv2.kernel', 'dec4.up.kernel', 'enc4.conv.kernel', 'dec3.conv1.kernel'):
            l2 = [r for r in report if r[0] == k][0][1]
            assert l2 < 1e-2, (k, l2)
    # ---- layer-local parity of EVERY 3x3 data- and weight-gradient launch on the device's own full-size tensors (_layer_local_check)
    plan = m._head_plan(n, 256, 256, True)
    checked = 0
    for node in m.nodes:
        if node.op != 'cba' or node.attrs['k'] != 3:
            continue
        _layer_local_check(m, plan, rt, node, n, f32)
        checked += 1
    assert checked == 16


def test_fused_backward_launches_agree_with_the_separate_ones(mt):
    """bf16 training step of a three-level U-Net with every backward fusion on (thin layers: BatchNorm apply + data gradient + weight
    gradient in one launch with the sums of the layer below; encoder BatchNorm sums formed by the concat-BN apply pass and by the next
    block's data gradient; reduce passes in data-gradient / head epilogues) against the same step with one launch per pass: every
    gradient agrees to the bf16 noise of two differently rounded chains (cos >= 0.999), and the fused launches did run."""
    rng = np.random.default_rng(4)
    n, hw = 3, 64
    x = rng.random((n, hw, hw, 4)).astype(np.float32)
    y = np.eye(2, dtype=np.float32)[(rng.random((n, hw, hw)) < 0.3).astype(np.int64)]

    def run(fuse):
        mt.reset_uids(); mt.set_seed(3)
        m = mt.get_unet_model(2, 4, filters=[32, 64, 128], factors=[2, 2, 2])
        m.compute_dtype = 'bfloat16'
        m.fuse_thin_bwd = m.fuse_pool_bwd = m.fuse_pool_bn_sums = m.fuse_dgrad_bn_bwd = m.fuse_head_bn_bwd = fuse
        m.compile(optimizer=mt.Adam(1e-3), loss=lambda a, b: mt.weighted_categorical_crossentropy(a, b, [1.0, 3.0]))
        plan = m.train_step_device(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
        torch.cuda.synchronize()
        rt_ = m.runtime
        inv = {v: k for k, v in mt.structural_names(m).items()}
        g = {inv.get(p.name, p.name): rt_.get_grad(p.name).double().cpu().numpy().ravel() for p in m.param_specs if p.name in rt_.offsets}
        return g, [getattr(f, 'label', '') or '' for f in plan.bwd]
    g0, lab0 = run(False)
    g1, lab1 = run(True)
    assert not any('fused' in l or 'sums' in l or 'bnred' in l for l in lab0)
    assert sum('bwd_fused' in l for l in lab1) == 5 and sum('pooled' in l for l in lab1) == 2 and sum('+poolsums' in l for l in lab1) >= 2 and sum('+skipsums' in l for l in lab1) >= 2
    assert sum('bn_bwd_reduce' in l for l in lab1) < sum('bn_bwd_reduce' in l for l in lab0) - 5
    for k in g0:
        a, b = g0[k], g1[k]
        if np.abs(a).max() == 0:
            assert np.abs(b).max() == 0, k
            continue
        cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))
        assert cos >= 0.999, f'{k}: cos {cos:.5f}'


def test_trained_five_level_model_bf16_iou_within_1e3(mt):
    """North-star parity target on the FULL-DEPTH network: get_unet_model(2, 4) with the default five levels, trained on the GPU
    (fp32 storage) on a synthetic rectangles task, then scored on held-out 256x256 tiles -- the bf16 device mask's IoU is within
    1e-3 of the float64 oracle's on the same weights and tiles, and the fp32 device mask equals the oracle's on every pixel
    whose class margin exceeds 1e-4."""
    mt.reset_uids(); mt.set_seed(2)
    m = mt.get_unet_model(2, 4)
    m.compute_dtype = 'float32'
    rng = np.random.default_rng(9)

    def make(n):
        # bright rectangles on a smooth background + pixel noise; label = rectangle
        lo = torch.tensor(rng.random((n, 4, 8, 8)), dtype=torch.float32)
        x = torch.nn.functional.interpolate(lo, size=(256, 256), mode='bilinear', align_corners=False).permute(0, 2, 3, 1).numpy() * 0.5
        lab = np.zeros((n, 256, 256), np.int64)
        for i in range(n):
            for _ in range(4):
                hh, ww = rng.integers(24, 96, 2)
                y0, x0 = rng.integers(0, 256 - hh), rng.integers(0, 256 - ww)
                lab[i, y0:y0 + hh, x0:x0 + ww] = 1
        x = x + 0.35 * lab[..., None] * np.array([1.0, 0.6, 0.8, 1.2], np.float32) + 0.05 * rng.standard_normal((n, 256, 256, 4))
        return x.astype(np.float32), lab
    x, lab = make(32)
    m.compile(optimizer=mt.Adam(1e-3), loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, [1.0, 2.0]))
    m.fit(x, np.eye(2, dtype=np.float32)[lab], batch_size=8, epochs=60, verbose=0)          # 240 steps
    w = m.get_weights_dict()
    xt, labt = make(4)
    names = mt.structural_names(m)
    o = UNetOracle(2, 4, dtype=np.float64)
    for k in o.params:
        o.params[k] = w[names[k]].astype(np.float64)
    p_ref, c_ref = o.forward(xt, training=False)
    iou_ref = iou(c_ref, labt)
    assert iou_ref > 0.8, iou_ref                    # the model has learned the task
    _, c_f32 = m.predict(xt, batch_size=4)
    sure = np.abs(p_ref[..., 0] - p_ref[..., 1]) > 1e-4
    assert np.array_equal(c_f32[sure], c_ref[sure]) and (c_f32 != c_ref).sum() <= 8
    mt.reset_uids()
    mb = mt.get_unet_model(2, 4)
    mb.compute_dtype = 'bfloat16'
    mb.set_weights_dict({names[k]: w[names[k]] for k in o.params})
    _, c_bf = mb.predict(xt, batch_size=4)
    d = abs(iou(c_bf, labt) - iou_ref)
    print(f'five-level trained model: IoU oracle {iou_ref:.5f}, bf16 {iou(c_bf, labt):.5f}, fp32 {iou(c_f32, labt):.5f}; pixels differing bf16 {(c_bf != c_ref).sum()}')
    assert d <= 1e-3, (iou(c_bf, labt), iou_ref)
    assert abs(iou(c_f32, labt) - iou_ref) < 2e-4


def test_predict_chips_matches_reference_semantics(mt):
    """sliding-window stitching through the device model == the restated reference loop fed with
    the same model's per-chip outputs; index lists equal the reference-generated fixtures."""
    from satellite_computervision_amd import prediction_tools as pt
    z = np.load(os.path.join(GOLD, 'tiling_reference.npz'))
    for key in z['cases']:
        _, h, w, c, buff, kernel = str(key).split('_')
        got = pt.generate_chip_indices(np.zeros((int(h), int(w), int(c)), np.float32), int(buff), int(kernel))
        assert np.array_equal(np.asarray(got, np.int64).reshape(-1, 2), z[str(key)])
    assert np.array_equal(np.stack(pt.extract_chips(z['ec_arr'], 16, 32)), z['ec_chips'])

    o, m, _ = build_pair(mt, 'float32', 2, 4, [32, 64], [2, 2], seed=5)
    rng = np.random.default_rng(2)
    arr = rng.random((200, 264, 4)).astype(np.float32)
    idx = pt.generate_chip_indices(arr, 32, 64)
    assert len(idx) == 6
    got = pt.predict_chips(arr, idx, np.zeros(arr.shape[:2]), m, kernel=64, buff=32, batch_size=4)
    ref = OT.predict_chips(arr, idx, np.zeros(arr.shape[:2]), lambda chip: m.predict(chip)[0], kernel=64, buff=32)
    np.testing.assert_allclose(got, ref, atol=1e-6)
    ref_o = OT.predict_chips(arr, idx, np.zeros(arr.shape[:2]), lambda chip: o.forward(chip)[0], kernel=64, buff=32)
    np.testing.assert_allclose(got, ref_o, atol=5e-5)
    assert got[:16].max() == 0 and got[:, :16].max() == 0          # never-predicted border stays 0


def test_fit_reduces_loss_and_evaluate(mt, tmp_path):
    mt.reset_uids(); mt.set_seed(0)
    m = mt.get_unet_model(2, 4, filters=[32, 64], factors=[2, 2])
    rng = np.random.default_rng(3)
    x = rng.random((16, 32, 32, 4)).astype(np.float32)
    lab = (x[..., 0] + x[..., 3] > 1.0).astype(np.int64)          # learnable from the pixels
    y = np.eye(2, dtype=np.float32)[lab]
    m.compile(optimizer=mt.Adam(2e-3), loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, [1.0, 1.0]),
              metrics=['categorical_accuracy', mt.MeanIoU(2)])
    assert m.metrics_names == ['loss', 'categorical_accuracy', 'mean_io_u']
    ck = mt.ModelCheckpoint(str(tmp_path / 'best.npz'), monitor='val_mean_io_u', save_best_only=True, mode='max')
    # Keras BN momentum 0.99: the moving statistics need a few hundred steps before inference-mode
    # evaluation is meaningful
    hist = m.fit(x, y, batch_size=8, epochs=150, validation_data=(x, y), callbacks=[ck], verbose=0)
    assert hist.history['loss'][-1] < 0.6 * hist.history['loss'][0]
    ev = m.evaluate(x, y, batch_size=8)
    assert len(ev) == 3 and ev[1] > 0.8
    assert os.path.exists(tmp_path / 'best.npz') and ck.best > 0.5
    # save / load round trip reproduces predictions exactly
    m.save(str(tmp_path / 'model.npz'))
    m2 = mt.load_model(str(tmp_path / 'model.npz'))
    a, b = m.predict(x[:4]), m2.predict(x[:4])
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # optimizer.learning_rate get/set and layer freezing (retrain_model surface)
    m.optimizer.learning_rate = 1e-4
    assert abs(float(m.optimizer.learning_rate.numpy()) - 1e-4) < 1e-9
    for layer in m.layers[:-1]:
        layer.trainable = False
    w0 = m.get_weights_dict()
    m.fit(x, y, batch_size=8, epochs=1, verbose=0)
    w1 = m.get_weights_dict()
    changed = [k for k in w0 if not np.array_equal(w0[k], w1[k]) and 'moving' not in k]
    assert changed and all(k.startswith('probs/') for k in changed)


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
def test_aspp_block_forward_backward(mt, dtype):
    """DilatedSpatialPyramidPooling (rates 3/6/12 + 1x1, concat, 1x1) inside a trainable graph: predictions, loss and
    every gradient against the PyTorch-CPU restatement (autograd)."""
    from oracle import torch_unet as TU
    mt.reset_uids(); mt.set_seed(2)
    inp = mt.Input([None, None, 4])
    pooled, enc = mt.encoder_block(32, name='encoder_0')(inp)
    a = mt.DilatedSpatialPyramidPooling(64)(pooled)
    d = mt.decoder_block(a, enc, 32)
    probs_t = mt._Head(2, 'softmax', None, 'probs')(d)
    m = mt.Model(inp, [probs_t, mt._classes(probs_t, 'classes')])
    m.compute_dtype = dtype
    cbas = [nd for nd in m.nodes if nd.op == 'cba']
    names = ['enc', 'aspp.cba', 'aspp.cba3_3', 'aspp.cba3_6', 'aspp.cba3_12', 'aspp.cba3', 'conv1', 'conv2']
    ref_of = {}
    for nm, nd in zip(names, cbas):
        ref_of[nd.layer.name + '/kernel'] = nm + '.kernel'; ref_of[nd.layer.name + '/bias'] = nm + '.bias'
        bn = nd.attrs['owner'].bn_layer.name
        for s_, r_ in (('gamma', 'gamma'), ('beta', 'beta'), ('moving_mean', 'moving_mean'), ('moving_var', 'moving_var')):
            ref_of[f'{bn}/{s_}'] = f'{nm}.bn.{r_}'
    up = [nd for nd in m.nodes if nd.op == 'convT'][0]
    ref_of[up.layer.name + '/kernel'] = 'up.kernel'; ref_of[up.layer.name + '/bias'] = 'up.bias'
    cat = [nd for nd in m.nodes if nd.op == 'concat_bn_relu'][0]
    for s_ in ('gamma', 'beta', 'moving_mean', 'moving_var'):
        ref_of[f'{cat.layer.name}/{s_}'] = f'bn0.{s_}'
    ref_of['probs/kernel'] = 'probs.kernel'; ref_of['probs/bias'] = 'probs.bias'
    rng = np.random.default_rng(9)
    w = {}
    for ps in m.param_specs:
        if ps.kind == 'kernel':
            fan = np.prod(ps.shape[:3])
            w[ps.name] = (rng.standard_normal(ps.shape) * np.sqrt(2.0 / fan)).astype(np.float32)
        elif ps.kind == 'moving_var':
            w[ps.name] = (0.5 + rng.random(ps.shape)).astype(np.float32)
        elif ps.kind == 'gamma':
            w[ps.name] = (1 + 0.2 * rng.standard_normal(ps.shape)).astype(np.float32)
        else:
            w[ps.name] = (0.2 * rng.standard_normal(ps.shape)).astype(np.float32)
    assert set(w) == set(ref_of)
    m.set_weights_dict(w)
    tp = TU.params_to_torch({ref_of[k]: v for k, v in w.items()}, torch.float64)
    x = rng.random((2, 48, 48, 4)).astype(np.float32)
    t = np.eye(2, dtype=np.float32)[(rng.random((2, 48, 48)) < 0.4).astype(np.int64)]
    f32 = dtype == 'float32'
    # inference
    with torch.no_grad():
        p_ref, c_ref = TU.aspp_net_forward(tp, torch.tensor(x, dtype=torch.float64), training=False)
    probs, classes = m.predict(x)
    np.testing.assert_allclose(probs, p_ref.numpy(), atol=3e-5 if f32 else 4e-2)
    # training step
    m.compile(optimizer=mt.Adam(1e-3), loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, [1.0, 3.0]))
    pr, _ = TU.aspp_net_forward(tp, torch.tensor(x, dtype=torch.float64), training=True)
    lt = TU.weighted_cce_mean(torch.tensor(t, dtype=torch.float64), pr, [1.0, 3.0]); lt.backward()
    loss = m.train_on_batch(x, t)
    np.testing.assert_allclose(loss, lt.item(), rtol=3e-5 if f32 else 3e-2)
    rt = m.runtime
    bad = []
    for k, rname in ref_of.items():
        if 'moving' in k or (k.endswith('/bias') and not k.startswith('probs')):
            continue
        g = rt.get_grad(k).cpu().numpy().astype(np.float64)
        r = tp[rname].grad.numpy()
        err = np.abs(g - r).max() / max(np.abs(r).max(), 1e-6)
        cos = (g * r).sum() / (np.linalg.norm(g) * np.linalg.norm(r))
        l2 = np.linalg.norm(g - r) / max(np.linalg.norm(r), 1e-30)
        med = np.median(np.abs(g - r)) / max(np.abs(r).max(), 1e-6)
        # fp32: bulk exact (median) and tight global agreement; isolated ReLU flips vs the float64 reference are tolerated
        if (f32 and (l2 > 2e-2 or cos < 0.9999)) or (not f32 and cos < 0.9):
            bad.append(f'{rname}: relL2 {l2:.2e} median {med:.2e} relmax {err:.2e} cos {cos:.5f}')
    assert not bad, '\n'.join(bad)


def test_dropout_training_matches_oracle_given_masks(mt):
    """get_unet_model(dropout=r): SpatialDropout2D after pool0, Dropout after the centre, SpatialDropout2D inside dec0 and in
    front of the head (utils/model_tools.py:350-351, 362-363, 375, 401-402).  The device masks are read back and handed to
    the oracle: loss and gradients must then agree; at inference dropout is the identity."""
    filters, factors = [32, 64], [2, 2]
    o, _, _ = build_pair(mt, 'float32', 2, 4, filters, factors, seed=21)
    mt.reset_uids()
    m = mt.get_unet_model(2, 4, filters, factors, dropout=0.25)
    m.compute_dtype = 'float32'
    mt.reset_uids()
    names = mt.structural_names(mt.get_unet_model(2, 4, filters, factors))        # same layer names (dropout has no weights)
    m.set_weights_dict({names[k]: v for k, v in o.params.items()})
    rng = np.random.default_rng(5)
    x = rng.random((4, 32, 32, 4)).astype(np.float32)
    t = np.eye(2, dtype=np.float32)[(rng.random((4, 32, 32)) < 0.4).astype(np.int64)]
    p_ref, _ = o.forward(x, training=False)
    np.testing.assert_allclose(m.predict(x)[0], p_ref, atol=3e-5)                 # identity at inference
    m.compile(optimizer=mt.Adam(1e-3), loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, [1.0, 2.0]))
    loss = m.train_on_batch(x, t)
    plan = m.runtime.plan(4, 32, 32, True)
    assert len(plan.dropouts) == 4
    dm = [d['mask'].cpu().numpy().astype(np.float64) for d in plan.dropouts]
    keep = 1.0 / 0.75
    for mk in dm:
        assert set(np.unique(mk.round(6))) <= {0.0, round(keep, 6)}
    assert 0.6 < (dm[1] > 0).mean() < 0.9                                        # element-wise mask: keep rate ~0.75
    masks = {'pool0': dm[0].reshape(4, 1, 1, -1), 'center': dm[1].reshape(4, 8, 8, -1), 'dec0': dm[2].reshape(4, 1, 1, -1),
             'final': dm[3].reshape(4, 1, 1, -1)}
    pr, _ = o.forward(x, training=True, masks=masks)
    loss_ref, dprobs, _ = OL.weighted_categorical_crossentropy(t.astype(np.float64), pr, [1.0, 2.0])
    g_ref = o.backward(dprobs)
    np.testing.assert_allclose(loss, loss_ref, rtol=3e-5)
    rt = m.runtime
    for k in ('probs.kernel', 'dec0.conv1.kernel', 'dec0.bn0.gamma', 'dec1.up.kernel', 'center.conv.kernel', 'enc1.conv.kernel', 'enc0.conv.kernel',
              'enc0.bn.beta'):
        g = rt.get_grad(names[k]).cpu().numpy().astype(np.float64)
        l2 = np.linalg.norm(g - g_ref[k]) / np.linalg.norm(g_ref[k])
        assert l2 < 1e-2, f'{k}: relL2 {l2:.3e}'
    # a second step draws different masks
    m.train_on_batch(x, t)
    assert not np.array_equal(plan.dropouts[0]['mask'].cpu().numpy(), dm[0]) or not np.array_equal(plan.dropouts[1]['mask'].cpu().numpy(), dm[1])


def test_config4_13_band_tiles_and_config5_1024_scene(mt):
    """BASELINE configs 4 and 5 as parity cases: 13-band Sentinel-2 tiles (stored padded to 16 channels), and a
    1024x1024 scene both one-shot (fully convolutional) and through the 384^2-chip sliding window of
    utils/prediction_tools.py:87-156 (9 chips, centre 256^2 kept, border never predicted)."""
    from satellite_computervision_amd import prediction_tools as pt
    filters, factors = [32, 64], [2, 2]
    o, m, _ = build_pair(mt, 'float32', 2, 13, filters, factors, seed=31)
    rng = np.random.default_rng(4)
    x = rng.random((2, 64, 64, 13)).astype(np.float32)
    p_ref, c_ref = o.forward(x, training=False)
    probs, classes = m.predict(x)
    np.testing.assert_allclose(probs, p_ref, atol=3e-5)
    ok = np.abs(p_ref[..., 0] - p_ref[..., 1]) > 1e-3
    assert np.array_equal(classes[ok], c_ref[ok])

    o4, m4, _ = build_pair(mt, 'float32', 2, 4, filters, factors, seed=32)
    scene = rng.random((1024, 1024, 4)).astype(np.float32)
    one_shot = m4.predict(scene[None])[0][0]                      # (1024,1024,2)
    ref_one = o4.forward(scene[None], training=False)[0][0]
    np.testing.assert_allclose(one_shot, ref_one, atol=5e-5)
    idx = pt.generate_chip_indices(scene, buff=128, kernel=256)
    assert idx == [(y, x_) for y in (64, 320, 576) for x_ in (64, 320, 576)]
    got = pt.predict_chips(scene, idx, np.zeros(scene.shape[:2]), m4, kernel=256, buff=128, batch_size=9)
    ref = OT.predict_chips(scene, idx, np.zeros(scene.shape[:2]), lambda chip: o4.forward(chip, training=False)[0], kernel=256, buff=128)
    np.testing.assert_allclose(got, ref, atol=5e-5)
    assert not got[:64].any() and not got[832:].any() and not got[:, :64].any() and not got[:, 832:].any()


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
def test_config4_13_band_five_level_training_step(mt, dtype):
    """BASELINE configs[3]: get_unet_model(2, 13) (utils/model_tools.py:394-415 with nchannels = 13: all Sentinel-2 bands,
    utils/ee_tools.py:100), the five-level graph bench.py --channels 13 times, 256 x 256 tiles, ONE training step against the float64
    oracle: loss and every gradient -- in particular enc0.conv.kernel, whose 13 real input channels live in 16 stored ones (the
    padded channels must neither feed the forward sum nor appear in the gradient).  bf16: the storage-rounding oracle, cosine >= 0.90
    per tensor (>= 0.99 on the last decoder level), plus the forward mask's IoU against the oracle's."""
    f32 = dtype == 'float32'
    o, m, names = build_pair(mt, dtype, 2, 13, [32, 64, 128, 256, 512], [2, 2, 2, 2, 2], seed=23, perturb=False,
                             store_dtype=None if f32 else 'bfloat16')
    n = 2
    x, t = _synthetic_step_batch(n, 13, seed=9)
    m.compile(optimizer=mt.Adam(0.0), loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, [1.0, 20.0]))
    pr, _ = o.forward(x, training=True)
    loss_ref, dprobs, _ = OL.weighted_categorical_crossentropy(t.astype(np.float64), pr, [1.0, 20.0])
    g_ref = o.backward(dprobs)
    loss = m.train_on_batch(x, t)
    np.testing.assert_allclose(loss, loss_ref, rtol=2e-5 if f32 else 2e-3)
    rt = m.runtime
    torch.cuda.synchronize()
    gk = rt.get_grad(names['enc0.conv.kernel']).cpu().numpy()
    assert gk.shape == (3, 3, 13, 32) and np.abs(gk).max() > 0
    for k, l2, cos, share in _grad_report(o, rt, names, g_ref):
        if f32:
            assert l2 < 1e-2, f'grad {k}: relL2 {l2:.3e} cos {cos:.6f}'
        else:
            assert cos >= 0.90, f'grad {k}: cos {cos:.4f} relL2 {l2:.3e}'       # (see test_full_unet_training_step_matches_oracle)
            if k.startswith(('probs', 'dec0.')):
                assert cos >= (0.995 if k.startswith(('probs', 'dec0.conv2', 'dec0.bn2')) else 0.98), f'grad {k}: cos {cos:.4f}'
            elif k.startswith(('dec1.', 'dec2.', 'dec3.')):
                # per decoder level, from the measured table this test prints (round 6, 4 bands: dec1 0.973-0.993, dec2 0.962-0.977,
                # dec3 0.950-0.968; dec4 / centre / encoder 0.939-0.960 keep the blanket bound) with ~2 points of margin (the 13-band network sits ~1 point lower) for the summation
                # orders of other tile choices -- a mis-scaled slab in one layer would cost that layer tens of points
                lim = {'dec1': 0.955, 'dec2': 0.94, 'dec3': 0.925}[k[:4]]
                assert cos >= lim, f'grad {k}: cos {cos:.4f} (decoder level {k[3]}: >= {lim})'
    # forward of the trained-mode statistics' moving averages: inference mask against the oracle (bit-exact beyond the margin)
    balance_head(o, x)
    m.set_weights_dict({names['probs.bias']: o.params['probs.bias']})
    for k in o.params:                      # the training step above moved the BatchNorm moving statistics on both sides
        if k.endswith(('moving_mean', 'moving_var')):
            m.set_weights_dict({names[k]: o.params[k]})
    p_ref, c_ref = o.forward(x, training=False)
    probs, classes = m.predict(x, batch_size=2)
    ok = np.abs(p_ref[..., 0] - p_ref[..., 1]) > (1e-3 if f32 else 0.1)
    assert np.array_equal(classes[ok], c_ref[ok])
    if f32:
        np.testing.assert_allclose(probs, p_ref, atol=5e-5)
    else:
        assert (classes == c_ref).mean() > 0.97 and abs(iou(classes, c_ref) - 1.0) < 0.05


@pytest.mark.parametrize('dtype,size,batch', [('float32', 64, 2), ('bfloat16', 128, 2), ('bfloat16', 512, 1)])
def test_deeplabv3_resnet50_inference(mt, dtype, size, batch):
    """BASELINE config 3 (build-defined): DeepLab-v3 ResNet-50 (OS16) + the reference's ASPP on NAIP-like 4-band
    tiles, inference -- including the configuration as BASELINE.json names it (512 x 512 x 4, batch 1, bf16).  Self-consistency
    against the PyTorch-CPU float64 restatement of the same graph and weights."""
    from oracle import torch_unet as TU
    mt.reset_uids(); mt.set_seed(3)
    m = mt.get_deeplabv3_model(3, 4)
    m.compute_dtype = dtype
    rng = np.random.default_rng(13)
    w = {}
    for ps in m.param_specs:
        if ps.kind == 'kernel':
            w[ps.name] = (rng.standard_normal(ps.shape) * np.sqrt(1.0 / np.prod(ps.shape[:3]))).astype(np.float32)
        elif ps.kind == 'moving_var':
            w[ps.name] = (0.5 + rng.random(ps.shape)).astype(np.float32)
        elif ps.kind == 'gamma':
            w[ps.name] = (1 + 0.1 * rng.standard_normal(ps.shape)).astype(np.float32)
        else:
            w[ps.name] = (0.1 * rng.standard_normal(ps.shape)).astype(np.float32)
    m.set_weights_dict(w)
    plist = []
    for nd in m.nodes:
        if nd.op == 'cba':
            bn = nd.attrs['owner'].bn_layer.name
            plist.append({'kernel': torch.tensor(w[nd.layer.name + '/kernel'], dtype=torch.float64), 'bias': torch.tensor(w[nd.layer.name + '/bias'], dtype=torch.float64),
                          **{k: torch.tensor(w[f'{bn}/{k}'], dtype=torch.float64) for k in ('gamma', 'beta', 'moving_mean', 'moving_var')}})
    head = (torch.tensor(w['logits/kernel'], dtype=torch.float64), torch.tensor(w['logits/bias'], dtype=torch.float64))
    x = (rng.integers(0, 256, (batch, size, size, 4)) / 255.0).astype(np.float32)        # NAIP uint8 / 255 (utils/processing.py:601)
    with torch.no_grad():
        p_ref, c_ref = TU.deeplab_forward(plist, head, torch.tensor(x, dtype=torch.float64))
    probs, classes = m.predict(x)
    assert probs.shape == (batch, size, size, 3) and classes.shape == (batch, size, size) and classes.dtype == np.int32
    p_ref = p_ref.numpy()
    np.testing.assert_allclose(probs, p_ref, atol=2e-4 if dtype == 'float32' else 6e-2)
    srt = np.sort(p_ref, -1)
    ok = (srt[..., -1] - srt[..., -2]) > (1e-3 if dtype == 'float32' else 0.15)
    assert np.array_equal(classes[ok], c_ref.numpy()[ok])
    # hipGraph replay of this launch-bound plan (Model._replay_graph): first call eager, second call captures + replays, later calls
    # replay -- every one bit-identical to the eager launch list, also after the resident input tensor's CONTENTS have changed
    xd = torch.from_numpy(x).cuda()
    os.environ['SATCV_INFER_GRAPH'] = '0'
    try:
        eager = [t_.clone() for t_ in m.predict_on_device(xd)]
        x2 = torch.from_numpy(np.ascontiguousarray(x[:, ::-1])).cuda()
        eager2 = [t_.clone() for t_ in m.predict_on_device(x2)]
    finally:
        os.environ['SATCV_INFER_GRAPH'] = '1'
    plan = m._infer_plan(batch, size, size)
    for _ in range(3):
        got = [t_.clone() for t_ in m.predict_on_device(xd)]
        assert all(torch.equal(a_, b_) for a_, b_ in zip(got, eager))
    graphs = [v for v in plan.__dict__.get('_graphs', {}).values() if v not in ('warm', 'off')]
    assert len(graphs) == 1, plan.__dict__.get('_graphs')
    xd.copy_(x2)
    got2 = [t_.clone() for t_ in m.predict_on_device(xd)]
    assert all(torch.equal(a_, b_) for a_, b_ in zip(got2, eager2))
    with pytest.raises(NotImplementedError):
        m.compile(optimizer=mt.Adam(), loss=lambda a, b: mt.weighted_categorical_crossentropy(a, b, [1, 1, 1]))
        m.train_on_batch(x, np.zeros((batch, size, size, 3), np.float32))


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
def test_siamese_unet_shared_weights_forward_backward(mt, dtype):
    """make_siamese_unet (utils/model_tools.py:576-663): shared-weight encoder and ASPP applied to two dates, concatenated
    skips, sigmoid head + threshold.  Predictions, weighted-BCE loss and the ACCUMULATED gradients of the shared layers
    against the PyTorch-CPU restatement (autograd)."""
    from oracle import torch_unet as TU
    filters, factors = [32, 64], [2, 2]
    mt.reset_uids(); mt.set_seed(4)
    m = mt.make_siamese_unet(4, filters, factors, class_thresh=0.4)
    m.compute_dtype = dtype
    assert len(m.inputs) == 2 and [t.name for t in m.outputs] == ['probs', 'classes']
    conv_of = {'enc0': 'conv2d', 'enc1': 'conv2d_2', 'aspp.cba': 'conv2d_4', 'aspp.cba3': 'conv2d_6', 'aspp.cba3_3': 'conv2d_7', 'aspp.cba3_6': 'conv2d_8',
               'aspp.cba3_12': 'conv2d_9', 'dec1.conv1': 'conv2d_10', 'dec1.conv2': 'conv2d_11', 'dec0.conv1': 'conv2d_12', 'dec0.conv2': 'conv2d_13'}
    ref_of = {'probs/kernel': 'probs.kernel', 'probs/bias': 'probs.bias', 'conv2d_transpose/kernel': 'dec1.up.kernel', 'conv2d_transpose/bias': 'dec1.up.bias',
              'conv2d_transpose_1/kernel': 'dec0.up.kernel', 'conv2d_transpose_1/bias': 'dec0.up.bias'}
    for rn, cn in conv_of.items():
        ref_of[cn + '/kernel'] = rn + '.kernel'; ref_of[cn + '/bias'] = rn + '.bias'
        idx = cn.split('_')[1] if '_' in cn else '0'
        idx = {'10': '11', '11': '12', '12': '14', '13': '15'}.get(idx, idx)          # decoder convs follow a concat BN
        bn = 'batch_normalization' if idx == '0' else f'batch_normalization_{idx}'
        for s_ in ('gamma', 'beta', 'moving_mean', 'moving_var'):
            ref_of[f'{bn}/{s_}'] = f'{rn}.bn.{s_}'
    for bn, rn in (('batch_normalization_10', 'dec1.bn0'), ('batch_normalization_13', 'dec0.bn0')):
        for s_ in ('gamma', 'beta', 'moving_mean', 'moving_var'):
            ref_of[f'{bn}/{s_}'] = f'{rn}.{s_}'
    assert set(ref_of) == {ps.name for ps in m.param_specs}, set(ref_of) ^ {ps.name for ps in m.param_specs}
    rng = np.random.default_rng(17)
    w = {}
    for ps in m.param_specs:
        if ps.kind == 'kernel':
            w[ps.name] = (rng.standard_normal(ps.shape) * np.sqrt(2.0 / np.prod(ps.shape[:3]))).astype(np.float32)
        elif ps.kind == 'moving_var':
            w[ps.name] = (0.5 + rng.random(ps.shape)).astype(np.float32)
        elif ps.kind == 'gamma':
            w[ps.name] = (1 + 0.2 * rng.standard_normal(ps.shape)).astype(np.float32)
        else:
            w[ps.name] = (0.2 * rng.standard_normal(ps.shape)).astype(np.float32)
    m.set_weights_dict(w)
    tp = TU.params_to_torch({ref_of[k]: v for k, v in w.items()}, torch.float64)
    xa = rng.random((2, 48, 48, 4)).astype(np.float32)
    xb = rng.random((2, 48, 48, 4)).astype(np.float32)
    t = (rng.random((2, 48, 48, 1)) < 0.3).astype(np.float32)
    f32 = dtype == 'float32'
    with torch.no_grad():
        p_ref = TU.siamese_forward(tp, torch.tensor(xa, dtype=torch.float64), torch.tensor(xb, dtype=torch.float64), filters, factors).numpy()
    probs, classes = m.predict([xa, xb])
    assert probs.shape == (2, 48, 48, 1) and classes.shape == (2, 48, 48, 1) and classes.dtype == np.int32
    np.testing.assert_allclose(probs, p_ref, atol=3e-5 if f32 else 5e-2)
    ok = np.abs(p_ref - 0.4) > (1e-3 if f32 else 0.1)
    assert np.array_equal(classes[ok], (p_ref > 0.4).astype(np.int32)[ok])
    m.compile(optimizer=mt.Adam(1e-3), loss=lambda yt, yp: mt.weighted_bce(yt, yp, 5.0))
    pr = TU.siamese_forward(tp, torch.tensor(xa, dtype=torch.float64), torch.tensor(xb, dtype=torch.float64), filters, factors, training=True)
    lt = TU.weighted_bce_mean(torch.tensor(t, dtype=torch.float64), pr, 5.0); lt.backward()
    loss = m.train_on_batch([xa, xb], t)
    np.testing.assert_allclose(loss, lt.item(), rtol=3e-5 if f32 else 3e-2)
    rt = m.runtime
    bad = []
    for k, rname in ref_of.items():
        if 'moving' in k or (k.endswith('/bias') and not k.startswith('probs')):
            continue
        g = rt.get_grad(k).cpu().numpy().astype(np.float64)
        r = tp[rname].grad.numpy()
        cos = (g * r).sum() / (np.linalg.norm(g) * np.linalg.norm(r))
        l2 = np.linalg.norm(g - r) / max(np.linalg.norm(r), 1e-30)
        # (a ReLU whose pre-activation is ~1e-7 can flip with the summation order of the kernel and moves the gradients downstream of it
        #  by a per cent or so: relative L2 2e-2 / cosine 0.9995)
        if (f32 and (l2 > 2e-2 or cos < 0.9995)) or (not f32 and cos < 0.9):
            bad.append(f'{rname}: relL2 {l2:.2e} cos {cos:.5f}')
    assert not bad, '\\n'.join(bad)


def test_tfrecord_patches_to_mosaic_through_the_device_model(mt, tmp_path):
    """EE export path of the reference (utils/prediction_tools.py:159-373): GZIP TFRecord patches -> make_pred_dataset ->
    Model.predict(dataset, steps) -> mixer.json mosaic, equal to predicting the stacked patches directly."""
    import json
    from satellite_computervision_amd import tfrecord_io as tio
    rng = np.random.default_rng(4)
    kernel, buff, feats = [32, 32], [32, 32], ['B2', 'B3', 'B4', 'B8']
    H = kernel[0] + buff[0]
    tiles = [{k: rng.random((H, H)).astype(np.float32) for k in feats} for _ in range(6)]
    path = str(tmp_path / 'patches.tfrecord.gz')
    with tio.TFRecordWriter(path, compression='GZIP') as w:
        for t in tiles:
            w.write(tio.encode_example({k: v.reshape(-1) for k, v in t.items()}))
    (tmp_path / 'mixer.json').write_text(json.dumps({'totalPatches': 6, 'patchesPerRow': 3}))
    mt.reset_uids(); mt.set_seed(2)
    m = mt.get_unet_model(2, 4, filters=[32, 64], factors=[2, 2])
    m.compute_dtype = 'float32'
    ds = tio.make_pred_dataset([path], feats, kernel, buff, moments=[(0, 1)] * 4)
    mosaic = tio.make_array_predictions(ds, m, str(tmp_path / 'mixer.json'), kernel, buff)
    assert mosaic.shape == (2 * 32, 3 * 32, 3)
    x = np.stack([np.stack([t[k] for k in feats], axis=-1) for t in tiles]) / (1.0 + 1e-8)
    probs, classes = m.predict(x.astype(np.float32), batch_size=6)
    full = np.concatenate([probs, classes[..., None].astype(np.float32)], axis=3)[:, 16:48, 16:48, :]
    ref = np.concatenate([np.concatenate(list(full[r * 3:(r + 1) * 3]), axis=1) for r in range(2)], axis=0)
    np.testing.assert_allclose(mosaic[..., :2], ref[..., :2], atol=1e-6)
    assert np.array_equal(mosaic[..., 2], ref[..., 2])


def test_load_weights_by_name_accepts_keras_variable_names(mt, tmp_path):
    """the .npz a TensorFlow host writes with tools/keras_to_npz.py: '<layer>/<variable>' keys, `moving_variance`, extra
    layers and mismatching shapes skipped under by_name / skip_mismatch (utils/model_tools.py:1162)."""
    mt.reset_uids(); mt.set_seed(3)
    a = mt.get_unet_model(2, 4, filters=[32, 64], factors=[2, 2])
    w = a.get_weights_dict()
    keras = {(k[:-len('moving_var')] + 'moving_variance' if k.endswith('/moving_var') else k): v for k, v in w.items()}
    keras['some_other_layer/kernel'] = np.zeros((3, 3, 4, 4), np.float32)
    keras['probs/kernel'] = np.zeros((1, 1, 32, 5), np.float32)                    # wrong class count: skipped
    np.savez(tmp_path / 'from_keras.npz', **keras)
    mt.reset_uids(); mt.set_seed(99)
    b = mt.get_unet_model(2, 4, filters=[32, 64], factors=[2, 2])
    b.load_weights(str(tmp_path / 'from_keras.npz'), by_name=True, skip_mismatch=True)
    wb = b.get_weights_dict()
    for k in w:
        if k == 'probs/kernel':
            assert not np.array_equal(wb[k], w[k])
        else:
            assert np.array_equal(wb[k], w[k]), k
    with pytest.raises((KeyError, ValueError)):
        b.load_weights(str(tmp_path / 'from_keras.npz'))


@pytest.mark.parametrize('dtype', ['bfloat16', 'float32'])
def test_full_size_batch_invariance_property(mt, dtype):
    """BASELINE full size (batch 64 of 256x256x4 through the 18.5 M-parameter U-Net): inference is per-tile work, so the batch
    of 64 must give BIT-IDENTICAL probabilities and masks to eight batches of 8 and to single tiles -- a size-independent
    check of the multi-image tiling (several whole images per workgroup at the deep levels), tile-edge and XCD-remap logic."""
    mt.reset_uids(); mt.set_seed(21)
    m = mt.get_unet_model(2, 4)
    m.compute_dtype = dtype
    rng = np.random.default_rng(8)
    x = rng.beta(2, 5, (64, 256, 256, 4)).astype(np.float32)
    p64, c64 = m.predict(x, batch_size=64)
    p8, c8 = m.predict(x, batch_size=8)
    assert np.array_equal(p64, p8) and np.array_equal(c64, c8)
    p1, c1 = m.predict(x[37:38], batch_size=1)
    assert np.array_equal(p64[37:38], p1) and np.array_equal(c64[37:38], c1)
    assert np.isfinite(p64).all() and np.allclose(p64.sum(-1), 1.0, atol=1e-5)
    # spatial sanity at full size: a vertically flipped tile is a different input, but a tile repeated in the batch is not
    x2 = x.copy(); x2[5] = x[11]
    p2, _ = m.predict(x2, batch_size=64)
    assert np.array_equal(p2[5], p64[11])


def test_full_size_training_step_permutation_property(mt):
    """BASELINE full size, one fp32 training step at batch 64: the batch is a set -- permuting the tiles must leave the loss and
    every gradient unchanged up to the summation order (BN statistics are sums over the batch; weight gradients are sums
    over pixels).  Size-independent check of the split-K slabs, the statistics rows and the multi-image tiles at full scale."""
    mt.reset_uids(); mt.set_seed(5)
    m = mt.get_unet_model(2, 4)
    m.compute_dtype = 'float32'
    m.compile(optimizer=mt.Adam(0.0), loss=lambda t, p: mt.weighted_categorical_crossentropy(t, p, [1.0, 20.0]))
    rng = np.random.default_rng(12)
    x = rng.beta(2, 5, (64, 256, 256, 4)).astype(np.float32)
    lab = (rng.random((64, 256, 256)) < 0.05).astype(np.int64)
    y = np.eye(2, dtype=np.float32)[lab]
    l1 = m.train_on_batch(x, y)
    g1 = m.runtime.gflat.clone()
    m.train_on_batch(x, y)
    rel_same = ((g1 - m.runtime.gflat).norm() / g1.norm()).item()        # run-to-run (was 2e-3 with fp32 statistics rows)
    perm = rng.permutation(64)
    l2 = m.train_on_batch(x[perm], y[perm])
    g2 = m.runtime.gflat.clone()
    assert abs(l1 - l2) < 1e-5 * abs(l1)
    rel = ((g1 - g2).norm() / g1.norm()).item()
    print(f'full-size gradient repeatability: same batch {rel_same:.2e}, permuted batch {rel:.2e}')
    # re-ordered fp32 sums flip isolated ReLU masks whose pre-activation is ~1e-7, which perturbs the gradients downstream of
    # them; a wrong tile mapping or a dropped slab would show up as O(1)
    assert rel < 1e-2 and rel_same < 2e-5, (rel, rel_same)         # same batch: reproducible to fp32 rounding (double statistics rows)
    assert torch.isfinite(g1).all() and g1.abs().max() > 0


def test_deferred_batched_slab_sums_agree_with_the_per_layer_sums(mt, monkeypatch):
    """round 5: the slab sums of the weight-gradient launches run deferred, several layers per launch (satcv_reduce_slabs_batched, engine.Plan);
    SATCV_DEFER_REDUCE=0 keeps one sum launch per layer.  Same slabs, another (fixed) summation order: every gradient of a four-level bf16
    step agrees to fp32 rounding, and the deferred form is bit-reproducible."""
    def grads(defer):
        monkeypatch.setenv('SATCV_DEFER_REDUCE', defer)
        mt.reset_uids(); mt.set_seed(9)
        m = mt.get_unet_model(2, 4, [32, 64, 128, 256], [2, 2, 2, 2])
        m.compute_dtype = 'bfloat16'
        m.compile(optimizer=mt.Adam(0.0), loss=lambda t, p: mt.weighted_categorical_crossentropy(t, p, [1.0, 5.0]))
        rng = np.random.default_rng(4)
        x = rng.random((4, 64, 96, 4)).astype(np.float32)
        y = np.eye(2, dtype=np.float32)[(rng.random((4, 64, 96)) < 0.3).astype(np.int64)]
        m.train_on_batch(x, y)
        g = m.runtime.gflat.clone()
        m.train_on_batch(x, y)
        assert torch.equal(g, m.runtime.gflat)
        labels = [getattr(s, 'label', '') for s in m.runtime.plan(4, 64, 96, True).bwd]
        return g, sum('wgrad_reduce_batched' in l for l in labels)
    g1, n1 = grads('1')
    g0, n0 = grads('0')
    assert n1 >= 1 and n0 == 0
    assert ((g1 - g0).norm() / g0.norm()).item() < 1e-5
    assert (g1 - g0).abs().max().item() <= 1e-4 * g0.abs().max().item()


def test_round6_scheduling_switches_leave_the_numbers_alone(mt, monkeypatch):
    """round 6, both measured as nulls and off by default: (1) SATCV_EARLY_OPT -- Adam + operand repack of the parameters whose
    gradients are final once the deep encoder blocks are done run on the weight-gradient stream inside the backward pass (satcv_adam_step_part, the
    step counter bumped by the last part): elementwise, so three steps must give BIT-identical parameters and packed images; (2) SATCV_REDUCE_STREAM --
    every side-stream weight gradient's slab sum on a third stream behind events, a workspace per layer (the batched sum kernel: another fixed order)."""
    from satellite_computervision_amd import engine as E
    rng = np.random.default_rng(12)
    xs = [rng.random((2, 64, 64, 4)).astype(np.float32) for _ in range(3)]
    ys = [np.eye(2, dtype=np.float32)[(rng.random((2, 64, 64)) < 0.3).astype(np.int64)] for _ in range(3)]

    def run(early, red3):
        monkeypatch.setattr(E, 'EARLY_OPT', early)
        monkeypatch.setenv('SATCV_REDUCE_STREAM', '1' if red3 else '0')
        mt.reset_uids(); mt.set_seed(21)
        m = mt.get_unet_model(2, 4)
        m.compute_dtype = 'bfloat16'
        m.compile(optimizer=mt.Adam(1e-3), loss=lambda t, p: mt.weighted_categorical_crossentropy(t, p, [1.0, 5.0]))
        losses, gfirst = [], None
        for x, y in zip(xs, ys):
            losses.append(m.train_on_batch(x, y))
            if gfirst is None:
                gfirst = m.runtime.gflat.clone()          # (the first step's gradient: before any parameter has moved)
        torch.cuda.synchronize()
        plan = m.runtime.plan(2, 64, 64, True)
        labels = [getattr(s_, 'label', '') or '' for s_ in plan.bwd]
        packed = torch.cat([pk['fwd'].float().flatten() for pk in m.runtime.packed.values()])
        return m.runtime.pflat.clone(), m.runtime.gflat.clone(), packed, losses, labels, plan, gfirst
    p0, g0, k0, l0, lab0, _, f0 = run(False, False)
    p1, g1, k1, l1, lab1, plan1, _ = run(True, False)
    assert any('early optimizer step' in l for l in lab1) and not any('early optimizer step' in l for l in lab0)
    assert plan1.eo_done and 0 < plan1.eo_lo < 0.05 * p1.numel()
    assert torch.equal(p0, p1) and torch.equal(g0, g1) and torch.equal(k0, k1)
    np.testing.assert_allclose(l1, l0, rtol=1e-6)                  # (the loss scalar is a float atomic sum: last bit)
    p2, g2, k2, l2, _, plan2, f2 = run(False, True)
    assert plan2.rstream is not None
    # the first step's gradients: the same slabs summed in another fixed order (fp32 rounding).  Later steps are not compared element by
    # element -- Adam's first updates are +-lr whatever the gradient's size, so last-bit differences move weights by 2 lr and the bf16
    # BatchNorm chain amplifies that -- only the loss curve
    assert ((f2 - f0).norm() / f0.norm()).item() < 1e-5
    assert (f2 - f0).abs().max().item() <= 1e-4 * f0.abs().max().item()
    np.testing.assert_allclose(l2, l0, rtol=2e-2)


@pytest.mark.parametrize('channels', [4, 13])
def test_timed_configuration_bf16_batch64_training_step_properties(mt, channels):
    """The configuration bench.py times (BASELINE configs[1]; configs[3] with 13 bands): ONE bf16 training step of get_unet_model(2, C) at
    batch 64 of 256 x 256 tiles -- the only size at which the 16x16x32 deep tile (csrc/conv_igemm_m16.hip, wave roles), the LDS-DMA weight
    gradient on 128 workgroups, the multi-image deep tiles and the fused thin-layer backward all run as benchmarked.  Size-independent
    properties: (1) two identical steps give BIT-identical gradients and loss sums (fixed summation orders everywhere), (2) the batch is a
    set: a permutation of the tiles leaves the loss and every gradient unchanged up to the re-ordered sums (measured relL2 3.5e-2 with the
    32x32x16 tiles, 3.7e-2 with the 16x16x32 tile: re-ordered BatchNorm sums move statistics in the last bit, which flips bf16 storage
    roundings of every tensor below, and BatchNorm's mean removal amplifies those -- DESIGN.md section 4; the fp32 step of the test above
    holds 1e-2.  Bound 8e-2 on the flat gradient, 0.3 per kernel: a wrong tile mapping or a dropped slab would be O(1) in its layer),
    (3) every gradient is finite and non-zero, the loss equals the mean over the tiles of the per-tile losses of the same step's
    probabilities within bf16 tolerance."""
    mt.reset_uids(); mt.set_seed(31)
    m = mt.get_unet_model(2, channels)
    m.compute_dtype = 'bfloat16'
    m.compile(optimizer=mt.Adam(0.0), loss=lambda t, p: mt.weighted_categorical_crossentropy(t, p, [1.0, 20.0]))
    rng = np.random.default_rng(77 + channels)
    x = rng.beta(2, 5, (64, 256, 256, channels)).astype(np.float32)
    lab = (rng.random((64, 256, 256)) < 0.05).astype(np.int64)
    y = np.eye(2, dtype=np.float32)[lab]
    l1 = m.train_on_batch(x, y)
    g1 = m.runtime.gflat.clone()
    if channels == 4:
        # layer-local data- / weight-gradient parity AT THE TIMED BATCH for the six deepest 3x3 layers (8 x 8 ... 32 x 32 maps, 256 ... 1024 channels):
        # the 128-workgroup LDS-DMA weight gradient, the multi-image deep tiles and the persistent 16x16x32 kernel run in THIS form only at batch 64
        # (the full-network check of test_full_unet_training_step_matches_oracle runs at batch 2).  float64 oracle on the device's own tensors.
        torch.cuda.synchronize()
        plan64 = m._head_plan(64, 256, 256, True)
        deep = [nd for nd in m.nodes if nd.op == 'cba' and nd.attrs['k'] == 3 and plan64.node_ctx[id(nd)]['r'].h <= 32]
        deep = sorted(deep, key=lambda nd: (plan64.node_ctx[id(nd)]['r'].h, -plan64.node_ctx[id(nd)]['cout']))[:6]
        assert len(deep) == 6 and all('dy:' + nd.layer.name in plan64.dbg for nd in deep)
        for nd in deep:
            _layer_local_check(m, plan64, m.runtime, nd, 64, False)
    l1b = m.train_on_batch(x, y)
    g1b = m.runtime.gflat.clone()
    assert torch.equal(g1, g1b), 'two identical bf16 steps at batch 64 must give bit-identical gradients'
    assert abs(l1 - l1b) <= 1e-6 * abs(l1)
    assert torch.isfinite(g1).all() and np.isfinite(l1)
    perm = rng.permutation(64)
    l2 = m.train_on_batch(x[perm], y[perm])
    g2 = m.runtime.gflat.clone()
    assert abs(l1 - l2) < 2e-4 * abs(l1), (l1, l2)
    rel = ((g1 - g2).norm() / g1.norm()).item()
    print(f'bf16 batch-64 gradient repeatability ({channels} bands): permuted batch relL2 {rel:.2e}')
    assert rel < 8e-2, rel
    # per-layer: every kernel gradient non-zero and permutation-stable (a layer served by a wrong tile map would stand out here even if
    # the flat norm hid it)
    for name, off in m.runtime.offsets.items():
        if not name.endswith('/kernel'):
            continue
        sz = m.runtime.get_grad(name).numel()
        a, b = g1[off:off + sz], g2[off:off + sz]
        assert a.abs().max() > 0, name
        assert ((a - b).norm() / a.norm()).item() < 0.3, name


@pytest.mark.parametrize('n,h,w', [(1, 4, 4), (5, 12, 20), (3, 40, 72), (2, 100, 36), (7, 8, 264)])
def test_ragged_shapes_forward_and_gradients(mt, n, h, w):
    """tile sizes that are not multiples of any kernel tile (partial tiles in x and y, several images per workgroup, a 1x1
    deepest level, more pixels per row than one tile): fp32 predictions and every gradient against the float64 oracle."""
    filters, factors = [32, 64], [2, 2]
    o, m, names = build_pair(mt, 'float32', 2, 4, filters, factors, seed=n + h)
    rng = np.random.default_rng(h * w)
    x = rng.random((n, h, w, 4)).astype(np.float32)
    lab = (rng.random((n, h, w)) < 0.4).astype(np.int64)
    t = np.eye(2)[lab].astype(np.float32)
    p_ref, c_ref = o.forward(x, training=False)
    probs, classes = m.predict(x, batch_size=n)
    np.testing.assert_allclose(probs, p_ref, atol=3e-5)
    ok = np.abs(p_ref[..., 0] - p_ref[..., 1]) > 1e-4
    assert np.array_equal(classes[ok], c_ref[ok])
    if n * h * w < 64:
        return                                        # BN batch statistics of a handful of pixels: forward only
    m.compile(optimizer=mt.Adam(1e-3), loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, [1.0, 3.0]))
    pr, _ = o.forward(x, training=True)
    loss_ref, dprobs, _ = OL.weighted_categorical_crossentropy(t.astype(np.float64), pr, [1.0, 3.0])
    g_ref = o.backward(dprobs)
    loss = m.train_on_batch(x, t)
    np.testing.assert_allclose(loss, loss_ref, rtol=3e-5)
    for k in o.trainable:
        if k.endswith('.bias') and not k.startswith('probs'):
            continue
        g = m.runtime.get_grad(names[k]).cpu().numpy().astype(np.float64)
        l2 = np.linalg.norm(g - g_ref[k]) / max(np.linalg.norm(g_ref[k]), 1e-30)
        assert l2 < 2e-2, f'{k}: relL2 {l2:.2e} at {(n, h, w)}'
    with pytest.raises(ValueError):
        m.predict(rng.random((1, h + 2, w, 4)).astype(np.float32))      # not divisible by the down-sampling


@pytest.mark.parametrize('filters,factors,ncls,nch,hw', [([32, 64, 128], [2, 2, 2], 8, 1, (24, 40)), ([32, 64], [3, 3], 3, 5, (18, 27)),
                                                          ([32, 32, 64, 64, 128], [2, 2, 2, 2, 2], 2, 17, (32, 64)), ([64, 32], [2, 4], 4, 3, (16, 48))])
def test_model_variants_forward_and_gradients(mt, filters, factors, ncls, nch, hw):
    """get_unet_model's free parameters (utils/model_tools.py:394): class count, band count (not a multiple of the channel
    padding), depth, pool / up-sampling factors other than 2, non-monotone filter lists -- fp32 against the float64 oracle."""
    o, m, names = build_pair(mt, 'float32', ncls, nch, filters, factors, seed=ncls + nch)
    rng = np.random.default_rng(ncls * 7 + nch)
    n, (h, w) = 4, hw
    x = rng.random((n, h, w, nch)).astype(np.float32)
    lab = rng.integers(0, ncls, (n, h, w))
    t = np.eye(ncls)[lab].astype(np.float32)
    p_ref, c_ref = o.forward(x, training=False)
    probs, classes = m.predict(x)
    np.testing.assert_allclose(probs, p_ref, atol=3e-5)
    srt = np.sort(p_ref, axis=-1)
    ok = (srt[..., -1] - srt[..., -2]) > 1e-4
    assert np.array_equal(classes[ok], c_ref[ok])
    wts = list(np.linspace(1.0, 2.0, ncls))
    m.compile(optimizer=mt.Adam(1e-3), loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, wts))
    pr, _ = o.forward(x, training=True)
    loss_ref, dprobs, _ = OL.weighted_categorical_crossentropy(t.astype(np.float64), pr, wts)
    g_ref = o.backward(dprobs)
    loss = m.train_on_batch(x, t)
    np.testing.assert_allclose(loss, loss_ref, rtol=3e-5)
    for k in o.trainable:
        if k.endswith('.bias') and not k.startswith('probs'):
            continue
        g = m.runtime.get_grad(names[k]).cpu().numpy().astype(np.float64)
        l2 = np.linalg.norm(g - g_ref[k]) / max(np.linalg.norm(g_ref[k]), 1e-30)
        assert l2 < 2e-2, f'{k}: relL2 {l2:.2e}'


def test_training_trajectory_matches_oracle_over_steps(mt):
    """twelve fp32 optimisation steps (forward, loss, backward, Keras-Adam, BN moving statistics incl. the reference's double
    update) on the device vs the float64 oracle doing the same twelve steps: loss per step and final parameters."""
    filters, factors = [32, 64], [2, 2]
    o, m, names = build_pair(mt, 'float32', 2, 4, filters, factors, seed=9)
    rng = np.random.default_rng(33)
    x = rng.random((4, 32, 32, 4)).astype(np.float32)
    lab = (x[..., 0] + x[..., 2] > 1.0).astype(np.int64)
    t = np.eye(2)[lab].astype(np.float32)
    m.compile(optimizer=mt.Adam(2e-3), loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, [1.0, 2.0]))
    p_start = {k: v.copy() for k, v in o.params.items()}
    dev_losses, ref_losses = [], []
    for step in range(1, 13):
        pr, _ = o.forward(x, training=True)
        loss_ref, dprobs, _ = OL.weighted_categorical_crossentropy(t.astype(np.float64), pr, [1.0, 2.0])
        g = o.backward(dprobs)
        o.adam_step(g, lr=2e-3)
        ref_losses.append(float(loss_ref))
        dev_losses.append(m.train_on_batch(x, t))
    np.testing.assert_allclose(dev_losses, ref_losses, rtol=5e-3)
    assert ref_losses[-1] < ref_losses[0]
    w = m.get_weights_dict()
    for k in o.params:
        if k.endswith('.bias') and not k.startswith('probs'):
            continue                                   # ~zero gradients: Adam steps of rounding-noise sign (see test_dp_gpu.py)
        # Adam moves an element whose gradient is rounding noise by +-lr per step, so single elements may differ by up to
        # 12*lr; the tensors as a whole must agree
        ref, got = o.params[k], w[names[k]].astype(np.float64)
        if k.endswith('moving_mean') or k.endswith('moving_var'):
            continue                                   # batch statistics of weights that differ by that noise (step 1 is checked exactly
                                                       # in test_tiny_unet_predict_and_train_step)
        upd = np.linalg.norm(ref - p_start[k])
        assert np.linalg.norm(got - ref) < 0.5 * max(upd, 1e-9), f'{k}: {np.linalg.norm(got - ref):.3e} vs update {upd:.3e}'


@pytest.mark.parametrize('variant', ['acnn', 'acnn2'])
@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
def test_atrous_cnn_family_forward_backward(mt, variant, dtype):
    """get_acnn_model / get_acnn_model2 (utils/model_tools.py:922-1014): residual sums ReLU(BN(conv) + shortcut), dilation-3
    convolutions, a single softmax output, Keras layer names as coded -- including build_acnn_layers feeding each block's first
    Conv2D with the previous Conv2D's un-normalised output.  Predictions, loss and every gradient against the PyTorch-CPU
    restatement (autograd, float64)."""
    from oracle import torch_unet as TU
    depth, nf, ncls = 3, 16, 3
    mt.reset_uids(); mt.set_seed(9)
    if variant == 'acnn':
        m = mt.get_acnn_model(ncls, nf, 4, depth)
        fwd = lambda p, x, training=False: TU.acnn_forward(p, x, depth, training)
        assert m.output_names == ['probabilities']
        names = {ps.name for ps in m.param_specs}
        assert 'BN_1_2/gamma' not in names and 'BN_2_2/gamma' in names and 'Conv2D_1_2/kernel' in names     # dead BN layers own no weights
        with pytest.raises(NameError):
            mt.get_acnn_model(ncls, nf, 4, 1)
    else:
        m = mt.get_acnn_model2(ncls, 4, nfilters=nf, depth=depth)
        fwd = lambda p, x, training=False: TU.acnn2_forward(p, x, depth, training)
        assert m.output_names == ['probs'] and m.count_params() == (9 * 4 * nf + nf + 4 * nf) + 5 * (9 * nf * nf + nf + 4 * nf) + nf * ncls + ncls
    m.compute_dtype = dtype
    rng = np.random.default_rng(23)
    w = {}
    for ps in m.param_specs:
        if ps.kind == 'kernel':
            w[ps.name] = (rng.standard_normal(ps.shape) * np.sqrt(2.0 / np.prod(ps.shape[:3]))).astype(np.float32)
        elif ps.kind == 'moving_var':
            w[ps.name] = (0.5 + rng.random(ps.shape)).astype(np.float32)
        elif ps.kind == 'gamma':
            w[ps.name] = (1 + 0.2 * rng.standard_normal(ps.shape)).astype(np.float32)
        else:
            w[ps.name] = (0.2 * rng.standard_normal(ps.shape)).astype(np.float32)
    m.set_weights_dict(w)
    tp = TU.params_to_torch(w, torch.float64)
    x = rng.random((2, 40, 36, 4)).astype(np.float32)
    lab = rng.integers(0, ncls, (2, 40, 36))
    t = np.eye(ncls, dtype=np.float32)[lab]
    f32 = dtype == 'float32'
    with torch.no_grad():
        p_ref = fwd(tp, torch.tensor(x, dtype=torch.float64)).numpy()
    probs = m.predict(x)
    assert isinstance(probs, np.ndarray) and probs.shape == (2, 40, 36, ncls)
    np.testing.assert_allclose(probs, p_ref, atol=3e-5 if f32 else 6e-2)
    wts = [1.0, 2.0, 3.0]
    m.compile(optimizer=mt.Adam(1e-3), loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, wts))
    pr = fwd(tp, torch.tensor(x, dtype=torch.float64), training=True)
    lt = TU.weighted_cce_mean(torch.tensor(t, dtype=torch.float64), pr, wts); lt.backward()
    loss = m.train_on_batch(x, t)
    np.testing.assert_allclose(loss, lt.item(), rtol=3e-5 if f32 else 3e-2)
    rt = m.runtime
    bad = []
    for k in w:
        if 'moving' in k:
            continue
        r = tp[k].grad.numpy()
        g = rt.get_grad(k).cpu().numpy().astype(np.float64)
        if np.linalg.norm(r) < 1e-9:                      # bias of a Conv2D followed by BatchNormalization: zero in exact arithmetic
            assert np.linalg.norm(g) < (1e-4 if f32 else 5e-2), k
            continue
        cos = (g * r).sum() / (np.linalg.norm(g) * np.linalg.norm(r))
        l2 = np.linalg.norm(g - r) / max(np.linalg.norm(r), 1e-30)
        if (f32 and (l2 > 2e-2 or cos < 0.9999)) or (not f32 and cos < 0.9):
            bad.append(f'{k}: relL2 {l2:.2e} cos {cos:.5f}')
    assert not bad, '\n'.join(bad)


def test_autoencoder_linear_head_mse_training(mt):
    """get_autoencoder (utils/model_tools.py:496-531): U-Net with a linear 1x1 'continuous' output trained with mse_4d (mean over the
    finite elements, :142-166).  Loss and gradients against autograd of the PyTorch-CPU U-Net restatement with the head left linear."""
    from oracle import torch_unet as TU
    import torch.nn.functional as F
    filters, factors = [32, 64], [2, 2]
    mt.reset_uids(); mt.set_seed(12)
    m = mt.get_autoencoder(4, optim=mt.Adam(1e-3), loss=mt.mse_4d, filters=filters, factors=factors)
    m.compute_dtype = 'float32'
    assert m.output_names == ['continuous']
    names = mt.structural_names(m)
    w = m.get_weights_dict()
    rng = np.random.default_rng(3)
    for k in w:
        if k.endswith('/bias') or k.endswith('/beta'):
            w[k] = (0.1 * rng.standard_normal(w[k].shape)).astype(np.float32)
    m.set_weights_dict(w)
    x = rng.random((2, 32, 32, 4)).astype(np.float32)
    t = rng.standard_normal((2, 32, 32, 1)).astype(np.float32)
    t[0, 3, 4, 0] = np.nan                                   # ignored element (mse_4d averages over the finite ones)
    tp = TU.params_to_torch({rn: w[kn] for rn, kn in names.items()}, torch.float64)

    def fwd(training):
        # unet_forward ends in softmax; recompute its last feature map and apply the linear head
        xx = torch.tensor(x, dtype=torch.float64).permute(0, 3, 1, 2)
        skips, h = [], xx
        for i in range(2):
            a = F.relu(TU._bn(TU._conv(h, tp[f'enc{i}.conv.kernel'], tp[f'enc{i}.conv.bias']), tp, f'enc{i}.bn', training))
            skips.append(a); h = F.max_pool2d(a, 2, 2)
        h = F.relu(TU._bn(TU._conv(h, tp['center.conv.kernel'], tp['center.conv.bias']), tp, 'center.bn', training))
        for j in (1, 0):
            up = F.conv_transpose2d(h, tp[f'dec{j}.up.kernel'].permute(3, 2, 0, 1), tp[f'dec{j}.up.bias'], stride=2)
            a0 = F.relu(TU._bn(torch.cat([skips[j], up], dim=1), tp, f'dec{j}.bn0', training))
            a1 = F.relu(TU._bn(TU._conv(a0, tp[f'dec{j}.conv1.kernel'], tp[f'dec{j}.conv1.bias']), tp, f'dec{j}.bn1', training))
            h = F.relu(TU._bn(TU._conv(a1, tp[f'dec{j}.conv2.kernel'], tp[f'dec{j}.conv2.bias']), tp, f'dec{j}.bn2', training))
        return TU._conv(h, tp['probs.kernel'], tp['probs.bias']).permute(0, 2, 3, 1)
    with torch.no_grad():
        p_ref = fwd(False).numpy()
    pred = m.predict(x)
    if isinstance(pred, list):
        pred = pred[0]
    np.testing.assert_allclose(pred, p_ref, atol=3e-5)
    out = fwd(True)
    tt = torch.tensor(t, dtype=torch.float64)
    fin = torch.isfinite(tt)
    lt = ((out - torch.nan_to_num(tt)) ** 2)[fin].mean(); lt.backward()
    loss = m.train_on_batch(x, t)
    np.testing.assert_allclose(loss, lt.item(), rtol=3e-5)
    rt = m.runtime
    for rn in ('probs.kernel', 'probs.bias', 'dec0.conv2.kernel', 'enc0.conv.kernel', 'center.bn.gamma'):
        g = rt.get_grad(names[rn]).cpu().numpy().astype(np.float64)
        r = tp[rn].grad.numpy()
        assert np.linalg.norm(g - r) / np.linalg.norm(r) < 2e-3, rn


def test_predict_chunk_cached_model_files(mt, tmp_path, capsys):
    """predict_chunk (utils/model_tools.py:1271-1304): (C, H, W) chunk -> squeezed probabilities of the saved model, optional weights
    overlay, model cached between chunks; remote URLs are refused loudly."""
    mt.reset_uids(); mt.set_seed(3)
    m = mt.get_unet_model(2, 4, filters=[32, 64], factors=[2, 2])
    path = str(tmp_path / 'model.h5')
    m.save(path)
    rng = np.random.default_rng(1)
    chunk = rng.random((4, 64, 48)).astype(np.float32)
    want = m.predict(np.moveaxis(chunk, 0, -1)[None])[0][0]
    got = mt.predict_chunk(chunk, path)
    assert got.shape == (64, 48, 2) and np.array_equal(got, want)
    assert 'input shape (4, 64, 48)' in capsys.readouterr().out
    assert mt.get_blob_model(h5_url='file://' + path) is mt.get_blob_model(hdf5_url=path)           # cached, not rebuilt per chunk
    w = m.get_weights_dict()
    w['probs/bias'] = w['probs/bias'] + np.array([2.0, -2.0], np.float32)
    m.set_weights_dict(w)
    wpath = str(tmp_path / 'weights.hdf5')
    m.save_weights(wpath)
    got2 = mt.predict_chunk(chunk, path, weights_blob_url=wpath)
    assert np.array_equal(got2, m.predict(np.moveaxis(chunk, 0, -1)[None])[0][0]) and not np.array_equal(got2, got)
    with pytest.raises(RuntimeError):
        mt.predict_chunk(chunk, 'https://account.blob.core.windows.net/models/model.h5?sig=x')
    assert mt.get_blob_model() is None


def test_fit_and_evaluate_from_tfrecord_datasets(mt, tmp_path):
    """The notebook's training path (notebooks/UNET_G4G_2019_solar.ipynb:1189-1277): get_training_dataset / get_eval_dataset over
    GZIP TFRecords -> Model.fit(x=training, steps_per_epoch, validation_data, validation_steps, callbacks=[TensorBoard])."""
    import glob
    from satellite_computervision_amd import tfrecord_io as tio
    rng = np.random.default_rng(5)
    H, bands = 32, ['B2', 'B3', 'B4', 'B8']
    path = str(tmp_path / 'train.tfrecord.gz')
    with tio.TFRecordWriter(path, compression='GZIP') as w:
        for i in range(12):
            d = {b: (rng.random((H, H)) * 3000).astype(np.float32) for b in bands}
            d['landcover'] = (d['B8'] > 1500).astype(np.float32)
            w.write(tio.encode_example({k: v.reshape(-1) for k, v in d.items()}))
    ft = {k: tio.FixedLenFeature([H, H]) for k in bands + ['landcover']}
    tio.set_seed(0)
    training = tio.get_training_dataset([path], ft, bands, {'landcover': 2}, buff=8, batch=4, moments=[(0, 3000)] * 4)
    evaluation = tio.get_eval_dataset([path], ft, bands, {'landcover': 2}, moments=[(0, 3000)] * 4)
    mt.reset_uids(); mt.set_seed(2)
    m = mt.get_unet_model(2, 4, filters=[32, 64], factors=[2, 2])
    m.compile(optimizer=mt.Adam(2e-3), loss=lambda t, p: mt.weighted_categorical_crossentropy(t, p, [1.0, 1.0]), metrics=[mt.MeanIoU(2)])
    tb = mt.TensorBoard(log_dir=str(tmp_path / 'logs'))
    h = m.fit(x=training, epochs=3, steps_per_epoch=6, validation_data=evaluation, validation_steps=12, callbacks=[tb], verbose=0)
    assert len(h.history['loss']) == 3 and h.history['loss'][-1] < h.history['loss'][0]
    assert 'val_loss' in h.history and 'val_mean_io_u' in h.history
    ev = [tio.decode_event(r) for r in tio.read_records(glob.glob(str(tmp_path / 'logs' / 'train' / 'events.out.tfevents.*'))[0])]
    assert [e['step'] for e in ev[1:]] == [0, 1, 2] and abs(ev[-1]['scalars']['epoch_loss'] - h.history['loss'][-1]) < 1e-6
    res = m.evaluate(evaluation, steps=12, verbose=0)
    assert len(res) == len(m.metrics_names) == 2


def test_input_validation_messages(mt):
    """Keras-style ValueErrors instead of kernel-level failures: empty arrays, missing batch axis, wrong band count, sizes the pooling
    pyramid does not divide, x / y length mismatch; an empty chip list leaves the template untouched."""
    from satellite_computervision_amd import prediction_tools as pt
    mt.reset_uids(); mt.set_seed(0)
    m = mt.get_unet_model(2, 4, filters=[32, 64], factors=[2, 2])
    m.compile(optimizer=mt.Adam(1e-3), loss=lambda t, p: mt.weighted_bce(t, p, 2.0))
    for bad, msg in ((np.zeros((0, 64, 64, 4), np.float32), 'non-empty'), (np.zeros((64, 64, 4), np.float32), '4-D'),
                     (np.zeros((1, 64, 64, 3), np.float32), '4 channels'), (np.zeros((1, 62, 64, 4), np.float32), 'divisible')):
        with pytest.raises(ValueError, match=msg):
            m.predict(bad)
    with pytest.raises(ValueError, match='non-empty'):
        m.predict(iter([]))
    with pytest.raises(ValueError, match='non-empty'):
        m.fit(np.zeros((0, 64, 64, 4), np.float32), np.zeros((0, 64, 64, 2), np.float32), epochs=1, verbose=0)
    with pytest.raises(ValueError, match='different numbers'):
        m.fit(np.zeros((4, 64, 64, 4), np.float32), np.zeros((3, 64, 64, 2), np.float32), epochs=1, verbose=0)
    t = pt.predict_chips(np.zeros((100, 100, 4), np.float32), [], np.ones((100, 100)), m, 32, 16)
    assert np.array_equal(t, np.ones((100, 100)))
    s = mt.make_siamese_unet(4, [32, 64], [2, 2])
    with pytest.raises(ValueError, match='2 input'):
        s.predict(np.zeros((1, 64, 64, 4), np.float32))


def test_keras_hdf5_weight_files_load_into_the_model(mt, tmp_path):
    """load_weights / load_model / predict_chunk on REAL HDF5 files in the Keras layout (written by h5py, tests/golden/make_h5_fixtures.py):
    positional (Keras default) and by_name loading give the same model as setting the arrays directly; shape / count mismatches are
    reported like Keras does."""
    import os
    from oracle.unet import UNetOracle
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    z = np.load(os.path.join(gold, 'keras_unet_weights.npz'))
    arrays = [z[k] for k in sorted(z.files)]
    mt.reset_uids(); mt.set_seed(1)
    ref = mt.get_unet_model(2, 4, filters=[16, 32], factors=[2, 2])
    ref.compute_dtype = 'float32'
    ref.set_weights_dict({ps.name: a for ps, a in zip(ref.param_specs, arrays)})
    x = np.random.default_rng(0).random((2, 32, 32, 4)).astype(np.float32)
    want = ref.predict(x)
    # independent check of the loaded numbers: the float64 oracle with the same arrays
    names = mt.structural_names(ref)
    o = UNetOracle(2, 4, [16, 32], [2, 2], dtype=np.float64)
    w = ref.get_weights_dict()
    for k in o.params:
        o.params[k] = w[names[k]].astype(np.float64)
    p_ref, _ = o.forward(x, training=False)
    np.testing.assert_allclose(want[0], p_ref, atol=3e-5)
    for kw in (dict(), dict(by_name=True), dict(by_name=True, skip_mismatch=True)):
        for fname in ('keras_unet_weights.h5', 'keras_unet_model.h5'):
            mt.reset_uids(); mt.set_seed(7)
            m = mt.get_unet_model(2, 4, filters=[16, 32], factors=[2, 2])
            m.compute_dtype = 'float32'
            m.load_weights(os.path.join(gold, fname), **kw)
            got = m.predict(x)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    chunk = np.moveaxis(x[0], -1, 0)
    mt.set_compute_dtype('float32')              # load_model builds the runtime: the precision is the module default at that moment
    try:
        lm = mt.load_model(os.path.join(gold, 'keras_unet_model.h5'), custom_objects={'weighted_bce': None})
        assert lm._builder['filters'] == [16, 32] and lm._builder['factors'] == [2, 2] and np.array_equal(lm.predict(x)[0], want[0])
        assert np.array_equal(mt.predict_chunk(chunk, os.path.join(gold, 'keras_unet_model.h5')), want[0][0])
    finally:
        mt.set_compute_dtype('bfloat16')
    mt.reset_uids()
    other = mt.get_unet_model(3, 4, filters=[16, 32], factors=[2, 2])
    with pytest.raises(ValueError, match='does not match'):
        other.load_weights(os.path.join(gold, 'keras_unet_weights.h5'))
    other.load_weights(os.path.join(gold, 'keras_unet_weights.h5'), by_name=True, skip_mismatch=True)       # head skipped, the rest loaded
    assert np.array_equal(other.get_weights_dict()['conv2d/kernel'], arrays[0])
    with pytest.raises(ValueError, match='skip_mismatch'):
        other.load_weights(os.path.join(gold, 'keras_unet_weights.h5'), skip_mismatch=True)
    mt.reset_uids()
    deeper = mt.get_unet_model(2, 4, filters=[16, 32, 64], factors=[2, 2, 2])
    with pytest.raises(ValueError, match='weight arrays'):
        deeper.load_weights(os.path.join(gold, 'keras_unet_weights.h5'))


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
def test_sixteen_filter_unet_inference_and_training(mt, dtype):
    """filters=[16, 32]: 16-channel transposed convolutions (a depth-to-space tile then spans several sub-pixel positions) and a decoder
    conv over concat([16-channel skip, 16-channel up]) -- inference AND one training step (loss, every gradient) against the NumPy
    oracle (the round-1 weight-gradient kernel refused concatenations whose first part was not a multiple of 32 channels)."""
    from oracle.unet import UNetOracle
    filters, factors = [16, 32], [2, 2]
    mt.reset_uids(); mt.set_seed(11)
    m = mt.get_unet_model(2, 4, filters=filters, factors=factors)
    m.compute_dtype = dtype
    names = mt.structural_names(m)
    o = UNetOracle(2, 4, filters, factors, dtype=np.float64)
    w = m.get_weights_dict()
    for k in o.params:
        o.params[k] = w[names[k]].astype(np.float64)
    rng = np.random.default_rng(4)
    x = rng.random((3, 32, 48, 4)).astype(np.float32)
    t = np.eye(2, dtype=np.float32)[(rng.random((3, 32, 48)) < 0.3).astype(int)]
    p_ref, _ = o.forward(x, training=False)
    np.testing.assert_allclose(m.predict(x)[0], p_ref, atol=3e-5 if dtype == 'float32' else 6e-2)
    m.compile(optimizer=mt.Adam(0.0), loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, [1.0, 3.0]))
    pr, _ = o.forward(x, training=True)
    loss_ref, dprobs, _ = OL.weighted_categorical_crossentropy(t.astype(np.float64), pr, [1.0, 3.0])
    g_ref = o.backward(dprobs)
    loss = m.train_on_batch(x, t)
    f32 = dtype == 'float32'
    np.testing.assert_allclose(loss, loss_ref, rtol=2e-5 if f32 else 3e-2)
    rt = m.runtime
    for k in o.trainable:
        if k.endswith('.bias') and not k.startswith('probs'):
            continue
        g = rt.get_grad(names[k]).cpu().numpy().astype(np.float64)
        l2 = np.linalg.norm(g - g_ref[k]) / max(np.linalg.norm(g_ref[k]), 1e-30)
        cos = (g * g_ref[k]).sum() / max(np.linalg.norm(g) * np.linalg.norm(g_ref[k]), 1e-30)
        assert (l2 < 1e-2) if f32 else (cos > 0.9), f'grad {k}: relL2 {l2:.3e} cos {cos:.4f}'


def test_retrain_model_from_keras_h5(mt, tmp_path):
    """retrain_model (utils/model_tools.py:1128-1176) on a Keras .h5 model file plus a separate weights file: evaluates, seeds the
    checkpoint's best value, sets the learning rate, freezes all but the head, returns (model, checkpoint)."""
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    rng = np.random.default_rng(9)
    x = rng.random((4, 32, 32, 4)).astype(np.float32)
    y = np.eye(2, dtype=np.float32)[(x[..., 0] > 0.5).astype(int)]
    ck = mt.ModelCheckpoint(str(tmp_path / 'best.npz'), monitor='val_mean_io_u', save_best_only=True, mode='max')
    comp = dict(optimizer=mt.Adam(1e-3), loss=lambda t, p: mt.weighted_bce(t, p, 2.0), metrics=[mt.MeanIoU(2)])
    m, ck2 = mt.retrain_model(os.path.join(gold, 'keras_unet_model.h5'), ck, [(x, y)], 'mean_io_u', weights_file=os.path.join(gold, 'keras_unet_weights.h5'),
                              by_name=True, skip_mismatch=True, custom_objects={'compile': comp}, lr=5e-4, freeze=True)
    assert ck2 is ck and 0.0 <= ck.best <= 1.0 and abs(float(m.optimizer.learning_rate.numpy()) - 5e-4) < 1e-9
    assert [l.trainable for l in m.layers][-1] and not any(l.trainable for l in m.layers[:-1])
    m2, _ = mt.retrain_model(m, ck, [(x, y)], 'loss')                      # a Model object is accepted as well
    assert m2 is m and ck.best > 0
    mt.reset_uids()
    with pytest.raises(RuntimeError, match='no loss'):
        mt.retrain_model(os.path.join(gold, 'keras_unet_model.h5'), ck, [(x, y)], 'loss')


def test_models_save_and_load_as_keras_hdf5(mt, tmp_path):
    """Model.save / save_weights / ModelCheckpoint with .h5 / .hdf5 paths write Keras-layout HDF5 (encoder / centre blocks grouped
    as the reference's custom layers) and load back bit-identically, optimizer state included; other families round-trip through
    the stored builder arguments."""
    from satellite_computervision_amd import hdf5_io as H
    rng = np.random.default_rng(3)
    x = rng.random((4, 32, 32, 4)).astype(np.float32)
    y = np.eye(2, dtype=np.float32)[(x[..., 0] > 0.5).astype(int)]
    mt.reset_uids(); mt.set_seed(6)
    m = mt.get_unet_model(2, 4, filters=[32, 64], factors=[2, 2])
    m.compile(optimizer=mt.Adam(1e-3), loss=lambda t, p: mt.weighted_bce(t, p, 2.0), metrics=[mt.MeanIoU(2)])
    ck = mt.ModelCheckpoint(str(tmp_path / 'best.hdf5'), monitor='loss', save_best_only=True, mode='min')
    m.fit(x, y, batch_size=4, epochs=2, verbose=0, callbacks=[ck])
    assert H.is_hdf5(str(tmp_path / 'best.hdf5'))
    path = str(tmp_path / 'model.h5')
    m.save(path)
    layers = H.read_keras_weights(path)
    assert [l for l, _ in layers][:3] == ['encoder_0', 'encoder_1', 'conv_block'] and [len(w) for _, w in layers][:5] == [6, 6, 6, 2, 4]
    assert layers[0][1][0][0] == 'encoder_0/conv_block/conv_batch_act/conv2d/kernel:0' and layers[0][1][5][0].endswith('batch_normalization/moving_variance:0')
    m2 = mt.load_model(path)
    a, b = m.predict(x), m2.predict(x)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.array_equal(m2.runtime.adam_m.cpu().numpy(), m.runtime.adam_m.cpu().numpy())
    m.save_weights(str(tmp_path / 'w.h5'))
    mt.reset_uids(); mt.set_seed(99)
    m3 = mt.get_unet_model(2, 4, filters=[32, 64], factors=[2, 2])
    m3.load_weights(str(tmp_path / 'w.h5'))
    assert np.array_equal(m3.predict(x)[0], a[0])
    mt.reset_uids(); mt.set_seed(98)
    m4 = mt.get_unet_model(2, 4, filters=[32, 64], factors=[2, 2])
    m4.set_weights(m.get_weights())                                      # Keras list form
    assert np.array_equal(m4.predict(x)[0], a[0]) and len(m.get_weights()) == len(m.param_specs)
    with pytest.raises(ValueError, match='weight list of length'):
        m4.set_weights(m.get_weights()[:-1])
    mt.reset_uids(); mt.set_seed(2)
    ac = mt.get_acnn_model2(3, 4, nfilters=16, depth=2)
    ac.save(str(tmp_path / 'acnn.h5'))
    ac2 = mt.load_model(str(tmp_path / 'acnn.h5'))
    assert [l for l, _ in H.read_keras_weights(str(tmp_path / 'acnn.h5'))][:2] == ['Conv0_1', 'bn0_1'] and np.array_equal(ac.predict(x), ac2.predict(x))


def test_training_is_bit_reproducible(mt):
    """Two runs of the same training steps from the same weights and batches end in BIT-IDENTICAL parameters, Adam slots and moving
    statistics: weight-gradient slabs and head dW / db rows are added in fixed order, BN statistics go through double-precision
    rows, the bias of a convolution under BatchNormalization has an exactly zero gradient."""
    rng = np.random.default_rng(12)
    xs = [rng.random((8, 64, 64, 4)).astype(np.float32) for _ in range(3)]
    ys = [np.eye(2, dtype=np.float32)[(x[..., 0] + x[..., 3] > 1.0).astype(int)] for x in xs]
    runs = []
    for rep in range(2):                      # the atrous family too (plain Conv2D bias gradients, residual joins)
        mt.reset_uids(); mt.set_seed(5)
        ac = mt.get_acnn_model(2, 16, 4, 3)
        ac.compile(optimizer=mt.Adam(2e-3), loss=lambda t, p: mt.weighted_categorical_crossentropy(t, p, [1.0, 3.0]))
        for step in range(4):
            ac.train_on_batch(xs[step % 3], ys[step % 3])
        runs.append(ac.get_weights_dict())
    assert not [k for k in runs[0] if not np.array_equal(runs[0][k], runs[1][k])]
    assert np.abs(runs[0]['Conv2D_1_2/bias']).max() > 0                  # a Conv2D without BatchNormalization: its bias does train
    runs = []
    for rep in range(2):
        mt.reset_uids(); mt.set_seed(5)
        m = mt.get_unet_model(2, 4, filters=[32, 64, 128], factors=[2, 2, 2])
        m.compile(optimizer=mt.Adam(2e-3), loss=lambda t, p: mt.weighted_categorical_crossentropy(t, p, [1.0, 3.0]))
        for step in range(6):
            m.train_on_batch(xs[step % 3], ys[step % 3])
        rt = m.runtime
        torch.cuda.synchronize()
        runs.append((m.get_weights_dict(), rt.adam_m.cpu().numpy().copy(), rt.adam_v.cpu().numpy().copy()))
    (w0, m0, v0), (w1, m1, v1) = runs
    diff = [k for k in w0 if not np.array_equal(w0[k], w1[k])]
    assert not diff, diff
    assert np.array_equal(m0, m1) and np.array_equal(v0, v1)
    assert all(np.all(w0[k] == 0) for k in w0 if k.endswith('/bias') and not k.startswith('probs'))      # exact zero gradient: never moved


def test_resident_float32_batches_are_read_in_place(mt):
    """A float32 batch that already lives on the device is read in place by the ingest and loss kernels (no staging copy); a host array, or a
    tensor of another dtype, goes through the staging tensors.  Either way the step is the same step: bit-identical weights."""
    rng = np.random.default_rng(31)
    xs = [rng.random((4, 64, 64, 4)).astype(np.float32) for _ in range(2)]
    ys = [np.eye(2, dtype=np.float32)[(x[..., 1] + x[..., 2] > 1.0).astype(int)] for x in xs]
    runs = []
    for mode in ('host', 'device', 'device-f64'):
        mt.reset_uids(); mt.set_seed(9)
        m = mt.get_unet_model(2, 4, filters=[32, 64], factors=[2, 2])
        m.compile(optimizer=mt.Adam(2e-3), loss=lambda t, p: mt.weighted_categorical_crossentropy(t, p, [1.0, 3.0]))
        losses = []
        for step in range(4):
            x, y = xs[step % 2], ys[step % 2]
            if mode == 'device':
                x, y = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
            elif mode == 'device-f64':
                x, y = torch.from_numpy(x).cuda().double(), torch.from_numpy(y).cuda().double()
            losses.append(m.train_on_batch(x, y))
            plan = m._head_plan(4, 64, 64, True)
            assert bool(plan.x_src) == (mode == 'device') and (plan.y_src is not None) == (mode == 'device')
        ev = m.evaluate(xs[0], ys[0], batch_size=4, verbose=0)
        runs.append((losses, m.get_weights_dict(), ev))
    for other in runs[1:]:
        assert np.allclose(other[0], runs[0][0], rtol=1e-6, atol=0)           # (the scalar loss is summed with float atomics: last-bit noise)
        assert not [k for k in runs[0][1] if not np.array_equal(runs[0][1][k], other[1][k])]
        assert np.allclose(np.asarray(other[2], dtype=np.float64), np.asarray(runs[0][2], dtype=np.float64), rtol=1e-6, atol=0)


def test_frozen_layers_run_batchnorm_in_inference_mode(mt):
    """tf.keras semantics of `layer.trainable = False` (retrain_model(freeze=True), utils/model_tools.py:1174-1175): a frozen
    BatchNormalization normalises with its MOVING statistics inside fit() and does not update them; only the head trains.  Loss and
    head gradients against autograd of the PyTorch-CPU restatement run with inference-mode BatchNorm."""
    from oracle import torch_unet as TU
    filters, factors = [32, 64], [2, 2]
    mt.reset_uids(); mt.set_seed(21)
    m = mt.get_unet_model(2, 4, filters=filters, factors=factors)
    m.compute_dtype = 'float32'
    rng = np.random.default_rng(6)
    w = m.get_weights_dict()
    for k in w:                                       # moving statistics that differ clearly from any batch statistics
        if k.endswith('moving_mean'):
            w[k] = (0.3 * rng.standard_normal(w[k].shape)).astype(np.float32)
        elif k.endswith('moving_var'):
            w[k] = (0.5 + rng.random(w[k].shape)).astype(np.float32)
    m.set_weights_dict(w)
    names = mt.structural_names(m)
    tp = TU.params_to_torch({rn: w[kn] for rn, kn in names.items()}, torch.float64)
    x = rng.random((4, 32, 32, 4)).astype(np.float32)
    t = np.eye(2, dtype=np.float32)[(x[..., 1] > 0.5).astype(int)]
    m.compile(optimizer=mt.Adam(1e-3), loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, [1.0, 2.0]))
    for layer in m.layers[:-1]:
        layer.trainable = False
    pr, _ = TU.unet_forward(tp, torch.tensor(x, dtype=torch.float64), filters, factors, training=False)
    lt = TU.weighted_cce_mean(torch.tensor(t, dtype=torch.float64), pr, [1.0, 2.0]); lt.backward()
    loss = m.train_on_batch(x, t)
    np.testing.assert_allclose(loss, lt.item(), rtol=3e-5)
    rt = m.runtime
    for rn in ('probs.kernel', 'probs.bias'):
        g = rt.get_grad(names[rn]).cpu().numpy().astype(np.float64)
        r = tp[rn].grad.numpy()
        assert np.linalg.norm(g - r) / np.linalg.norm(r) < 1e-4, rn
    w1 = m.get_weights_dict()
    changed = [k for k in w if not np.array_equal(w[k], w1[k])]
    assert changed and all(k.startswith('probs/') for k in changed), changed          # moving statistics untouched as well
    # un-freezing restores batch-statistics training (a different plan)
    for layer in m.layers:
        layer.trainable = True
    pr2, _ = TU.unet_forward(tp, torch.tensor(x, dtype=torch.float64), filters, factors, training=True)
    w2 = m.get_weights_dict()
    tp2 = TU.params_to_torch({rn: w2[kn] for rn, kn in names.items()}, torch.float64)
    pr2, _ = TU.unet_forward(tp2, torch.tensor(x, dtype=torch.float64), filters, factors, training=True)
    l2 = TU.weighted_cce_mean(torch.tensor(t, dtype=torch.float64), pr2, [1.0, 2.0]).item()
    np.testing.assert_allclose(m.train_on_batch(x, t), l2, rtol=3e-5)
    assert any(not np.array_equal(w2[k], v) for k, v in m.get_weights_dict().items() if k.endswith('moving_mean'))


def test_fit_uploads_host_batches_one_ahead_and_stays_bit_identical(mt):
    """round 4: fit / evaluate upload host ndarray batches one batch ahead on a copy stream (model_tools._prefetch_to_device).  The
    result must not depend on it: same seed, same data, with and without the prefetcher -> identical parameters, moving statistics,
    loss history and evaluation; ragged last batches, steps_per_epoch on an endless generator and a two-input model included."""
    rng = np.random.default_rng(5)
    x = rng.random((22, 32, 32, 4)).astype(np.float32)             # 22 = 2 x 8 + 6: a ragged last batch
    y = np.eye(2, dtype=np.float32)[(x[..., 0] + x[..., 3] > 1.0).astype(np.int64)]

    def endless():
        while True:
            for i in range(0, 16, 8):
                yield x[i:i + 8], y[i:i + 8]

    def run(flag):
        os.environ['SATCV_PREFETCH'] = flag
        try:
            mt.reset_uids(); mt.set_seed(0)
            m = mt.get_unet_model(2, 4, filters=[32, 64], factors=[2, 2])
            m.compile(optimizer=mt.Adam(2e-3), loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, [1.0, 2.0]),
                      metrics=['categorical_accuracy', mt.MeanIoU(2)])
            h1 = m.fit(x, y, batch_size=8, epochs=3, validation_data=(x, y), verbose=0, shuffle=False)
            h2 = m.fit(endless(), steps_per_epoch=5, epochs=2, verbose=0)
            ev = m.evaluate(x, y, batch_size=8)
            return m.get_weights_dict(), h1.history, h2.history, ev
        finally:
            os.environ.pop('SATCV_PREFETCH', None)

    w1, ha1, hb1, ev1 = run('1')
    w0, ha0, hb0, ev0 = run('0')
    # (the reported loss scalar is a float atomic sum: last-bit differences between any two runs)
    for a_, b_ in ((ha1, ha0), (hb1, hb0)):
        assert a_.keys() == b_.keys()
        for k in a_:
            np.testing.assert_allclose(a_[k], b_[k], rtol=1e-5, err_msg=k)
    np.testing.assert_allclose(ev1, ev0, rtol=1e-5)
    for k in w0:
        assert np.array_equal(w0[k], w1[k]), k
    # two inputs (Siamese): lists of arrays go through the same slots
    mt.reset_uids(); mt.set_seed(1)
    sm = mt.make_siamese_unet(4, filters=[32, 64], factors=[2, 2])
    xa, xb_ = x[:8], x[8:16]
    ys = (rng.random((8, 32, 32, 1)) < 0.3).astype(np.float32)
    sm.compile(optimizer=mt.Adam(1e-3), loss=lambda yt, yp: mt.weighted_bce(yt, yp, 2.0))
    h = sm.fit([xa, xb_], ys, batch_size=4, epochs=2, verbose=0, shuffle=False)
    assert np.isfinite(h.history['loss']).all()
