"""FP8 (OCP e4m3fn) inference path, BASELINE config 5.  The reference has no reduced-precision path: parity is against
this build's fp32 path / the float64 oracle.  Operator tests use values that are exactly representable, so the fp8
MFMA operand layout, the folded epilogue and the e4m3 rounding are checked BIT-EXACTLY; the model test states its
tolerance (IoU of the fp8 mask vs the oracle's, both against ground truth)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    from satellite_computervision_amd import ops, model_tools as mt
    from satellite_computervision_amd._lib import lib, check, FP8
    return dict(ops=ops, mt=mt, lib=lib, check=check, FP8=FP8)


def _to_f8(t):
    return t.to(torch.float8_e4m3fn)


@pytest.mark.parametrize('cin,cout,k,dil,scaled', [(32, 32, 3, 1, 0), (16, 64, 3, 1, 0), (64, 96, 3, 1, 0), (128, 128, 3, 1, 0), (64, 32, 1, 1, 0), (32, 32, 3, 2, 0),
                                                  (64, 64, 3, 1, 1), (128, 128, 3, 1, 1), (192, 64, 3, 1, 1), (256, 32, 3, 1, 1), (64, 128, 1, 1, 1), (128, 96, 3, 2, 1)])
def test_fp8_conv_bit_exact(env, cin, cout, k, dil, scaled):
    """scaled=1: SATCV_FP8X -- weights in 16-channel granules, block-scaled K=64 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4)"""
    ops, lib, check = env['ops'], env['lib'], env['check']
    from satellite_computervision_amd._lib import FP8X
    FP8 = FP8X if scaled else env['FP8']
    rng = np.random.default_rng(cin * 7 + cout + k)
    n, h, w = 3, 20, 40
    x = torch.tensor(rng.integers(-3, 4, (n, h, w, cin)), dtype=torch.float32)
    kern = torch.tensor(rng.integers(-2, 3, (k, k, cin, cout)), dtype=torch.float32)
    osc = torch.tensor(2.0 ** rng.integers(-6, -3, cout), dtype=torch.float32)
    bias = torch.tensor(rng.integers(-4, 5, cout), dtype=torch.float32)
    pad = dil * (k - 1) // 2
    acc = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), kern.permute(3, 2, 0, 1).double(), padding=pad, dilation=dil).permute(0, 2, 3, 1)
    ref = _to_f8((acc * osc.double() + bias.double()).clamp_min(0).clamp_max(448).float())
    xd = _to_f8(x).cuda()
    w8, _ = ops.pack_weights(kern.cuda(), cin, FP8, want_dgrad=False)
    y = torch.zeros(n, h, w, cout, dtype=torch.uint8, device='cuda')
    oscd, bd = osc.cuda(), bias.cuda()
    d = ops.make_conv_desc(x0=xd.data_ptr(), c0=cin, w=w8.data_ptr(), bias=bd.data_ptr(), out_scale=oscd.data_ptr(), y=y.data_ptr(), ldy=cout, n=n, h=h, w_=w,
                           cout=cout, cout_pad=ops.rup(cout, 32), kh=k, kw=k, dil=dil, dtype=FP8, out_relu=1)
    check(lib.satcv_conv2d_igemm(C.byref(d), ops.stream_ptr()))
    got = y.cpu()
    assert torch.equal(got, ref.view(torch.uint8)), (got.view(torch.float8_e4m3fn).float() - ref.float()).abs().max()


@pytest.mark.parametrize('cin,cout,c1,affine', [(32, 32, 0, False), (64, 32, 32, True), (32, 64, 0, True), (64, 64, 0, False), (64, 64, 32, True)])
def test_fp8_thin_layer_kernel_bit_exact(env, cin, cout, c1, affine):
    """The persistent weights-stationary kernel in e4m3 storage (conv_igemm_ws.hip, T = fp8: the thin layers of the folded fp8 graph): whole
    8 x 32 tiles, optional second source and input affine + ReLU (the skip half of a decoder concatenation), folded epilogue.  Exactly
    representable data: bit-exact against float64."""
    ops, lib, check, FP8 = env['ops'], env['lib'], env['check'], env['FP8']
    import ctypes
    rng = np.random.default_rng(cin * 11 + cout + c1)
    n, h, w = 3, 24, 64
    c0 = cin - c1
    x = torch.tensor(rng.integers(-3, 4, (n, h, w, cin)), dtype=torch.float32)
    kern = torch.tensor(rng.integers(-2, 3, (3, 3, cin, cout)), dtype=torch.float32)
    osc = torch.tensor(2.0 ** rng.integers(-7, -4, cout), dtype=torch.float32)
    bias = torch.tensor(rng.integers(-4, 5, cout), dtype=torch.float32)
    isc = torch.tensor(2.0 ** rng.integers(-1, 2, cin), dtype=torch.float32)
    ish = torch.tensor(rng.integers(-2, 3, cin), dtype=torch.float32)
    a = (x * isc + ish).clamp_min(0) if affine else x              # (small integers: exact in e4m3)
    assert torch.equal(_to_f8(a).float(), a)
    acc = torch.nn.functional.conv2d(a.permute(0, 3, 1, 2).double(), kern.permute(3, 2, 0, 1).double(), padding=1).permute(0, 2, 3, 1)
    ref = _to_f8((acc * osc.double() + bias.double()).clamp_min(0).clamp_max(448).float())
    x0 = _to_f8(x[..., :c0].contiguous()).cuda()
    x1 = _to_f8(x[..., c0:].contiguous()).cuda() if c1 else None
    w8, _ = ops.pack_weights(kern.cuda(), cin, FP8, want_dgrad=False)
    y = torch.zeros(n, h, w, cout, dtype=torch.uint8, device='cuda')
    oscd, bd, iscd, ishd = osc.cuda(), bias.cuda(), isc.cuda(), ish.cuda()
    before = ctypes.c_int32()
    check(lib.satcv_get_option(b'igemm_thin_launches', ctypes.byref(before)))
    d = ops.make_conv_desc(x0=x0.data_ptr(), c0=c0, x1=x1.data_ptr() if c1 else None, c1=c1, w=w8.data_ptr(), bias=bd.data_ptr(), out_scale=oscd.data_ptr(),
                           y=y.data_ptr(), ldy=cout, n=n, h=h, w_=w, cout=cout, cout_pad=ops.rup(cout, 32), kh=3, kw=3, dil=1, dtype=FP8, out_relu=1,
                           in_scale=iscd.data_ptr() if affine else None, in_shift=ishd.data_ptr() if affine else None, in_relu=1 if affine else 0)
    check(lib.satcv_conv2d_igemm(C.byref(d), ops.stream_ptr()))
    after = ctypes.c_int32()
    check(lib.satcv_get_option(b'igemm_thin_launches', ctypes.byref(after)))
    assert after.value - before.value == 1, 'served by the thin-layer kernel'
    got = y.cpu()
    assert torch.equal(got, ref.view(torch.uint8)), (got.view(torch.float8_e4m3fn).float() - ref.float()).abs().max()


def test_fp8_transposed_conv_into_concat_slice_and_requant(env):
    ops, lib, check, FP8 = env['ops'], env['lib'], env['check'], env['FP8']
    rng = np.random.default_rng(3)
    n, h, w, cin, cout, f, ca = 2, 8, 16, 64, 32, 2, 32
    x = torch.tensor(rng.integers(-3, 4, (n, h, w, cin)), dtype=torch.float32)
    kern = torch.tensor(rng.integers(-2, 3, (f, f, cout, cin)), dtype=torch.float32)          # Keras Conv2DTranspose layout
    osc = torch.tensor(2.0 ** rng.integers(-5, -2, cout), dtype=torch.float32)
    bias = torch.tensor(rng.integers(-4, 5, cout), dtype=torch.float32)
    up = torch.nn.functional.conv_transpose2d(x.permute(0, 3, 1, 2).double(), kern.permute(3, 2, 0, 1).double(), stride=f).permute(0, 2, 3, 1)
    ref_up = _to_f8((up * osc.double() + bias.double()).clamp_min(0).clamp_max(448).float())
    skip = torch.tensor(rng.integers(0, 9, (n, h * f, w * f, ca)), dtype=torch.float32)
    rs = torch.tensor(2.0 ** rng.integers(-2, 2, ca), dtype=torch.float32)
    rsh = torch.tensor(rng.integers(-3, 4, ca), dtype=torch.float32)
    ref_skip = _to_f8((skip * rs + rsh).clamp_min(0))
    cat = torch.zeros(n, h * f, w * f, ca + cout, dtype=torch.uint8, device='cuda')
    xd, sd = _to_f8(x).cuda(), _to_f8(skip).cuda()
    w8, _ = ops.pack_weights(kern.cuda(), cin, FP8, transposed=True, want_dgrad=False)
    oscd, bd, rsd, rshd = osc.cuda(), bias.cuda(), rs.cuda(), rsh.cuda()
    d = ops.make_conv_desc(x0=xd.data_ptr(), c0=cin, w=w8.data_ptr(), bias=bd.data_ptr(), out_scale=oscd.data_ptr(), y=cat.data_ptr() + ca, ldy=ca + cout,
                           n=n, h=h, w_=w, cout=f * f * cout, cout_pad=ops.rup(f * f * cout, 32), kh=1, kw=1, dil=1, mode_out=1, f=f, cstat=cout,
                           dtype=FP8, out_relu=1)
    check(lib.satcv_conv2d_igemm(C.byref(d), ops.stream_ptr()))
    check(lib.satcv_affine_requant(sd.data_ptr(), ca, rsd.data_ptr(), rshd.data_ptr(), 1, cat.data_ptr(), ca + cout, n * h * f * w * f, ca, FP8, FP8, ops.stream_ptr()))
    got = cat.cpu()
    assert torch.equal(got[..., ca:], ref_up.view(torch.uint8))
    assert torch.equal(got[..., :ca], ref_skip.view(torch.uint8))
    # max-pool on the fp8 values
    pooled = torch.zeros(n, h, w, ca + cout, dtype=torch.uint8, device='cuda')
    check(lib.satcv_maxpool(cat.data_ptr(), pooled.data_ptr(), n, h * f, w * f, ca + cout, 2, 2, 0, FP8, ops.stream_ptr()))
    refp = torch.nn.functional.max_pool2d(got.view(torch.float8_e4m3fn).float().permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
    assert torch.equal(pooled.cpu().view(torch.float8_e4m3fn).float(), refp)


def test_fp8_unet_inference_iou(env):
    """Trained small U-Net: fp8 (e4m3 weights + activations, fp32 accumulate) mask vs the float64 oracle's mask, both scored
    against ground truth.  Stated tolerance for the fp8 path (3 mantissa bits, per-tensor activation scales): |IoU_fp8 - IoU_oracle| <= 5e-3,
    >= 98.5 % identical pixels, >= 99.5 % where the oracle's probability is beyond 0.5 +- 0.25."""
    from oracle.unet import UNetOracle
    mt = env['mt']
    mt.reset_uids(); mt.set_seed(1)
    filters, factors = [32, 64], [2, 2]
    m = mt.get_unet_model(2, 4, filters=filters, factors=factors)
    m.compute_dtype = 'float32'
    rng = np.random.default_rng(7)

    def make(n):
        lo = torch.tensor(rng.random((n, 4, 8, 8)), dtype=torch.float32)
        x = torch.nn.functional.interpolate(lo, size=(64, 64), mode='bilinear', align_corners=False)
        x = (x.permute(0, 2, 3, 1).numpy() + 0.05 * rng.standard_normal((n, 64, 64, 4))).astype(np.float32)
        return x, (x[..., 0] + x[..., 3] > 1.0).astype(np.int64)
    x, lab = make(32)
    m.compile(optimizer=mt.Adam(2e-3), loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, [1.0, 1.0]))
    m.fit(x, np.eye(2, dtype=np.float32)[lab], batch_size=8, epochs=100, verbose=0)
    xt, labt = make(8)
    w = m.get_weights_dict()
    names = mt.structural_names(m)
    o = UNetOracle(2, 4, filters, factors, dtype=np.float64)
    for k in o.params:
        o.params[k] = w[names[k]].astype(np.float64)
    p_ref, c_ref = o.forward(xt, training=False)

    def iou(a, b):
        return (np.logical_and(a == 1, b == 1).sum()) / max(np.logical_or(a == 1, b == 1).sum(), 1)
    iou_ref = iou(c_ref, labt)
    assert iou_ref > 0.85
    m.enable_fp8_inference(x[:8])
    p8, c8 = m.predict(xt)
    m.disable_fp8_inference()
    p32, c32 = m.predict(xt)
    assert np.array_equal(c32, c_ref) or (c32 == c_ref).mean() > 0.9999
    agree, d_iou, d_p = (c8 == c_ref).mean(), abs(iou(c8, labt) - iou_ref), np.abs(p8 - p_ref).mean()
    print(f'fp8 vs oracle: pixel agreement {agree:.4f}, IoU {iou(c8, labt):.4f} vs {iou_ref:.4f}, mean |dp| {d_p:.4f}')
    margin = np.abs(p_ref[..., 1] - 0.5) > 0.25                 # confidently classified pixels must not flip
    assert (c8[margin] == c_ref[margin]).mean() > 0.995, (c8[margin] == c_ref[margin]).mean()
    assert agree >= 0.985 and d_iou <= 5e-3 and d_p < 0.015, (agree, d_iou, d_p)


def test_folded_bf16_inference_matches_regular_plan(env):
    """the folded graph with bf16 tensors (no quantisation) against the regular bf16 plan and the float64 oracle"""
    from oracle.unet import UNetOracle
    mt = env['mt']
    mt.reset_uids(); mt.set_seed(6)
    filters, factors = [32, 64, 128], [2, 2, 2]
    m = mt.get_unet_model(2, 4, filters=filters, factors=factors)
    m.compute_dtype = 'bfloat16'
    rng = np.random.default_rng(3)
    w = m.get_weights_dict()
    for k in w:                                   # non-trivial BN statistics and biases
        if k.endswith('moving_mean') or k.endswith('/bias') or k.endswith('/beta'):
            w[k] = (0.2 * rng.standard_normal(w[k].shape)).astype(np.float32)
        elif k.endswith('moving_var') or k.endswith('/gamma'):
            w[k] = (0.5 + rng.random(w[k].shape)).astype(np.float32)
    m.set_weights_dict(w)
    names = mt.structural_names(m)
    o = UNetOracle(2, 4, filters, factors, dtype=np.float64)
    for k in o.params:
        o.params[k] = w[names[k]].astype(np.float64)
    x = rng.random((3, 48, 80, 4)).astype(np.float32)
    p_ref, _ = o.forward(x, training=False)
    p_reg, _ = m.predict(x)
    m.enable_folded_inference()
    p_fold, c_fold = m.predict(x)
    m.disable_folded_inference()
    assert np.abs(p_fold - p_ref).max() < 0.05 and np.abs(p_reg - p_ref).max() < 0.05
    assert np.abs(p_fold - p_ref).mean() < 1.5 * np.abs(p_reg - p_ref).mean() + 1e-4      # same bf16 noise level as the regular plan
    assert c_fold.shape == (3, 48, 80)


def test_folded_inference_tracks_weight_updates(env):
    """the default bf16 inference plan bakes BatchNorm / bias / weight images in at build time: it must follow training steps
    and weight loads (no stale predictions), and SATCV-regular and folded paths must agree after them."""
    mt = env['mt']
    mt.reset_uids(); mt.set_seed(8)
    m = mt.get_unet_model(2, 4, filters=[32, 64], factors=[2, 2])
    m.compute_dtype = 'bfloat16'
    m.compile(optimizer=mt.Adam(5e-3), loss=lambda t, p: mt.weighted_categorical_crossentropy(t, p, [1.0, 1.0]))
    rng = np.random.default_rng(1)
    x = rng.random((4, 32, 32, 4)).astype(np.float32)
    y = np.eye(2, dtype=np.float32)[(x[..., 0] > 0.5).astype(np.int64)]
    p0, _ = m.predict(x)
    for _ in range(20):
        m.train_on_batch(x, y)
    p1, _ = m.predict(x)
    assert np.abs(p1 - p0).max() > 0.05                       # the model moved, and predict saw it
    plan = m._head_plan(4, 32, 32, False)                      # regular (training-style) inference plan on the same weights
    m._stage_x(plan, x); plan.run_forward(env['ops'].stream_ptr())
    p_reg = plan.outputs[m.outputs[0].id].cpu().numpy()
    assert np.abs(p1 - p_reg).max() < 0.06 and np.abs(p1 - p_reg).mean() < 0.01
    w = m.get_weights_dict()
    m.set_weights_dict({k: np.zeros_like(v) if k == 'probs/kernel' else v for k, v in w.items()})
    p2, _ = m.predict(x)
    assert np.allclose(p2[..., 0], p2[0, 0, 0, 0], atol=1e-6)  # zero head kernel -> constant output: the load was seen


@pytest.mark.parametrize('dt', ['bf16', 'fp8'])
@pytest.mark.parametrize('n,h,w,cin,cout', [(2, 16, 64, 32, 32), (3, 24, 40, 16, 64), (5, 8, 8, 64, 128), (2, 4, 12, 32, 32)])
def test_fused_maxpool_epilogue_equals_separate_pool(env, dt, n, h, w, cin, cout):
    """satcv_conv_desc.pool_y: the pooled tensor written by the conv epilogue equals max-pooling the stored output (partial tiles,
    several images per workgroup), bit for bit."""
    ops, lib, check = env['ops'], env['lib'], env['check']
    from satellite_computervision_amd._lib import BF16, FP8
    code, tdt = (BF16, torch.bfloat16) if dt == 'bf16' else (FP8, torch.uint8)
    rng = np.random.default_rng(n * h + w)
    if dt == 'bf16':
        x = torch.tensor(rng.standard_normal((n, h, w, cin)), dtype=torch.float32).cuda().to(torch.bfloat16)
        kern = torch.tensor(rng.standard_normal((3, 3, cin, cout)) * 0.1, dtype=torch.float32).cuda()
    else:
        x = _to_f8(torch.tensor(rng.integers(-3, 4, (n, h, w, cin)), dtype=torch.float32)).cuda().view(torch.uint8)
        kern = torch.tensor(rng.integers(-2, 3, (3, 3, cin, cout)), dtype=torch.float32).cuda()
    wp, _ = ops.pack_weights(kern, cin, code, want_dgrad=False)
    osc = torch.full((cout,), 0.05, device='cuda'); b = torch.tensor(rng.standard_normal(cout), dtype=torch.float32).cuda()
    y = torch.zeros(n, h, w, cout, dtype=tdt, device='cuda')
    p = torch.zeros(n, h // 2, w // 2, cout, dtype=tdt, device='cuda')
    d = ops.make_conv_desc(x0=x.data_ptr(), c0=cin, w=wp.data_ptr(), bias=b.data_ptr(), out_scale=osc.data_ptr(), y=y.data_ptr(), ldy=cout, n=n, h=h, w_=w,
                           cout=cout, cout_pad=ops.rup(cout, 32), dtype=code, out_relu=1, pool_y=p.data_ptr(), pool_ld=cout, pool_f=2)
    assert lib.satcv_conv2d_igemm_pipelined(C.byref(d)) == 1
    check(lib.satcv_conv2d_igemm(C.byref(d), ops.stream_ptr()))
    p2 = torch.zeros_like(p)
    check(lib.satcv_maxpool(y.data_ptr(), p2.data_ptr(), n, h, w, cout, 2, 2, 0, code, ops.stream_ptr()))
    torch.cuda.synchronize()
    assert y.float().abs().sum() > 0 if dt == 'bf16' else y.any()
    assert torch.equal(p, p2)
