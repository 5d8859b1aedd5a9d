"""Op-level parity: every HIP kernel, through the C ABI, against the NumPy oracle (float64)
on the same seeded inputs.  fp32 storage: tight tolerance (exact fp32 MFMA products);
bf16 storage: inputs are pre-rounded to bf16 so only accumulation order and the final
bf16 rounding differ (tolerance 2^-7 relative to the output scale)."""
import os

import numpy as np
import pytest
import torch

from oracle import keras_ops as K
from oracle import losses as OL

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DT = [torch.float32, torch.bfloat16]


@pytest.fixture(scope='module')
def ops():
    from satellite_computervision_amd import ops as _ops
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    return _ops


def dev():
    return torch.device('cuda')


def rnd(rng, shape, td, scale=1.0):
    """float64 array whose values are exactly representable in the storage dtype."""
    x = torch.tensor(rng.standard_normal(shape) * scale, dtype=torch.float32)
    return x.to(td).to(torch.float64).numpy()


def to_dev(x64, td, cpad=None):
    t = torch.tensor(x64, dtype=torch.float32)
    if cpad is not None and cpad > t.shape[-1]:
        t = torch.nn.functional.pad(t, (0, cpad - t.shape[-1]))
    return t.to(td).to(dev()).contiguous()


def f32dev(x64):
    return torch.tensor(x64, dtype=torch.float32, device=dev()).contiguous()


def back(t, c=None):
    a = t.detach().float().cpu().numpy().astype(np.float64)
    return a if c is None else a[..., :c]


def close(got, ref, td, what='', k=1.0):
    scale = max(np.abs(ref).max(), 1e-6)
    tol = (2e-5 if td == torch.float32 else 1.2e-2) * k
    err = np.abs(got - ref).max() / scale
    assert err < tol, f'{what}: rel-to-max err {err:.3e} >= {tol:.1e} (scale {scale:.3e})'


def rup(a, b):
    return (a + b - 1) // b * b


# ------------------------------------------------------------------------- ingest
@pytest.mark.parametrize('td', DT)
def test_ingest(ops, td):
    rng = np.random.default_rng(0)
    x = rng.random((2, 5, 7, 4)).astype(np.float32)
    y = ops.ingest_nhwc(torch.tensor(x, device=dev()), 16, ops.DTYPE_CODE[td])
    ref = torch.tensor(x).to(td).float().numpy()
    assert np.array_equal(back(y, 4), ref.astype(np.float64))
    assert not back(y)[..., 4:].any()
    planes = torch.tensor(rng.integers(0, 10000, (2, 4, 6, 5)).astype(np.int16), device=dev())
    z = ops.ingest_chw(planes, 1.0 / 10000, 16, ops.DTYPE_CODE[td])
    refz = (planes.cpu().numpy().astype(np.float32) * np.float32(1.0 / 10000)).transpose(0, 2, 3, 1)
    np.testing.assert_allclose(back(z, 4), torch.tensor(refz).to(td).float().numpy(), rtol=1e-6, atol=1e-7)


# --------------------------------------------------------------------- conv forward
CONV_CASES = [
    # n, h, w, cin, cout, k, dil
    (2, 32, 32, 4, 32, 3, 1),      # first layer: 4 real channels padded to 16, BN=32 config
    (2, 32, 64, 32, 64, 3, 1),     # BN=64 config, TW=32
    (1, 16, 16, 64, 128, 3, 1),    # BN=128 config, TW=16
    (3, 8, 8, 32, 128, 3, 1),      # TW=8, several images per tile (H < TH), odd image count
    (2, 20, 24, 16, 32, 3, 1),     # ragged: W=24 -> TW=8, H not a multiple of TH
    (1, 12, 12, 32, 32, 3, 1),     # 384-chip bottleneck size
    (1, 40, 40, 32, 32, 3, 3),     # dilated (ASPP rate 3)
    (1, 32, 32, 32, 64, 3, 6),     # dilated (ASPP rate 6)
    (3, 8, 8, 32, 128, 3, 3),      # dilated, several images per tile (tap-loop form of the pipelined kernel)
    (2, 20, 24, 64, 32, 3, 2),     # dilated, ragged tile grid
    (2, 16, 16, 64, 32, 1, 1),     # 1x1
    (1, 32, 32, 48, 96, 3, 1),     # channel counts that are not powers of two
    (1, 256, 256, 16, 32, 3, 1),   # full-resolution tile
    # the deep shapes of get_unet_model(2, 4) at 256x256 -- the instantiations bench.py times (K up to 9 * 1024)
    (2, 16, 16, 1024, 512, 3, 1),  # dec4.conv1 (consumes concat([skip, up]))
    (2, 16, 16, 512, 512, 3, 1),   # dec4.conv2
    (2, 8, 8, 512, 1024, 3, 1),    # centre block
    (2, 32, 32, 512, 256, 3, 1),   # dec3.conv1
    (2, 32, 32, 128, 256, 3, 1),   # enc3
    (1, 64, 64, 256, 128, 3, 1),   # dec2.conv1
    (1, 128, 128, 128, 64, 3, 1),  # dec1.conv1
    (1, 128, 128, 64, 64, 3, 1),   # dec1.conv2
    (1, 256, 256, 64, 32, 3, 1),   # dec0.conv1
    (1, 256, 256, 32, 32, 3, 1),   # dec0.conv2
]


@pytest.mark.parametrize('td', DT)
@pytest.mark.parametrize('case', CONV_CASES)
def test_conv2d_fwd(ops, td, case):
    n, h, w, cin, cout, k, dil = case
    rng = np.random.default_rng(hash(case) % 2**31)
    x = rnd(rng, (n, h, w, cin), td)
    kern = rnd(rng, (k, k, cin, cout), td, 0.2)
    b = rng.standard_normal(cout)
    ref = K.conv2d_same(x, kern, b, dil)
    cpad = rup(cin, 16)
    wf, _ = ops.pack_weights(f32dev(kern), cpad, ops.DTYPE_CODE[td], want_dgrad=False)
    stats = ops.new_stats(rup(cout, 16), dev())
    y = ops.conv2d(to_dev(x, td, cpad), wf, cout, kh=k, kw=k, dil=dil, bias=f32dev(b), stats=stats)
    torch.cuda.synchronize()
    got = back(y, cout)
    close(got, ref, td, f'conv {case}')
    # epilogue statistics are those of the STORED values
    s = stats.sum(0).double().cpu().numpy()
    np.testing.assert_allclose(s[0, :cout], got.sum((0, 1, 2)), rtol=2e-4, atol=2e-3 * np.sqrt(n * h * w))
    np.testing.assert_allclose(s[1, :cout], (got ** 2).sum((0, 1, 2)), rtol=2e-4)


@pytest.mark.parametrize('td', DT)
def test_conv2d_dual_source_affine(ops, td):
    """decoder conv1: concat([skip, up]) -> BN -> ReLU fused into the loader (model_tools.py:307-312)."""
    rng = np.random.default_rng(11)
    n, h, w, c0, c1, cout = 2, 32, 32, 32, 32, 32
    xa, xb = rnd(rng, (n, h, w, c0), td), rnd(rng, (n, h, w, c1), td)
    sc, sh = rng.standard_normal(c0 + c1), rng.standard_normal(c0 + c1)
    kern = rnd(rng, (3, 3, c0 + c1, cout), td, 0.2)
    sc32, sh32 = sc.astype(np.float32).astype(np.float64), sh.astype(np.float32).astype(np.float64)
    a = np.maximum(np.concatenate([xa, xb], -1) * sc32 + sh32, 0)
    if td == torch.bfloat16:
        a = torch.tensor(a, dtype=torch.float32).to(td).double().numpy()     # the staged tile is rounded to bf16
    ref = K.conv2d_same(a, kern, None, 1)
    wf, _ = ops.pack_weights(f32dev(kern), c0 + c1, ops.DTYPE_CODE[td], want_dgrad=False)
    y = ops.conv2d(to_dev(xa, td), wf, cout, x1=to_dev(xb, td), in_scale=f32dev(sc), in_shift=f32dev(sh), in_relu=True)
    close(back(y, cout), ref, td, 'dual-source conv', k=2.0)


# ---------------------------------------------------------------------- conv backward
BWD_CASES = [
    (2, 32, 32, 16, 32),     # wgrad cfg (1,1)
    (2, 32, 32, 64, 32),     # wgrad cfg (2,1)
    (2, 16, 16, 32, 64),     # (1,2), TW=16
    (2, 16, 32, 64, 64),     # (2,2)
    (3, 8, 8, 64, 128),      # (1,4), TW=8 multi-image
    (1, 20, 24, 32, 32),     # ragged
    (1, 64, 64, 32, 32),     # many pixel tiles -> split-K
    (2, 32, 32, 64, 64),     # (2,2) at TW=32
    (2, 24, 24, 64, 64),     # ragged rows, TW=8
    (2, 48, 48, 32, 32),     # 3 column tiles of 16
    # deep shapes of the benchmarked model (data gradient: Cout -> Cin; weight gradient tiles of 128 channels and more)
    (2, 16, 16, 512, 512),   # dec4.conv2
    (2, 16, 16, 1024, 512),  # dec4.conv1
    (2, 8, 8, 512, 1024),    # centre block
    (2, 32, 32, 512, 256),   # dec3.conv1
    (2, 16, 16, 256, 512),   # enc4
    (1, 64, 64, 256, 128),   # dec2.conv1
    (1, 64, 64, 128, 128),   # dec2.conv2
    (1, 128, 128, 128, 64),  # dec1.conv1
    (1, 256, 256, 64, 32),   # dec0.conv1
]


@pytest.mark.parametrize('td', DT)
@pytest.mark.parametrize('case', BWD_CASES)
def test_conv2d_dgrad_wgrad(ops, td, case):
    n, h, w, cin, cout = case
    rng = np.random.default_rng(hash(case) % 2**31)
    x = rnd(rng, (n, h, w, cin), td)
    kern = rnd(rng, (3, 3, cin, cout), td, 0.2)
    dy = rnd(rng, (n, h, w, cout), td)
    dx_ref, dk_ref, _ = K.conv2d_same_bwd(x, kern, dy, 1)
    _, wd = ops.pack_weights(f32dev(kern), cin, ops.DTYPE_CODE[td])
    dx = ops.conv2d_dgrad(to_dev(dy, td), wd, cin)
    close(back(dx, cin), dx_ref, td, f'dgrad {case}')
    dk = ops.conv2d_wgrad(to_dev(x, td), to_dev(dy, td), cin, cout)
    close(back(dk), dk_ref, torch.float32 if td == torch.float32 else td, f'wgrad {case}', k=(5.0 if td == torch.float32 else 0.5))


@pytest.mark.parametrize('td', DT)
def test_wgrad_padded_input_and_affine(ops, td):
    """first layer (4 real of 16 stored channels) and a fused input BN+ReLU, dilation 3."""
    rng = np.random.default_rng(5)
    n, h, w, cin, cout = 2, 32, 32, 4, 32
    x = rnd(rng, (n, h, w, cin), td); dy = rnd(rng, (n, h, w, cout), td)
    _, dk_ref, _ = K.conv2d_same_bwd(x, np.zeros((3, 3, cin, cout)), dy, 1)
    dk = ops.conv2d_wgrad(to_dev(x, td, 16), to_dev(dy, td), cin, cout)
    close(back(dk), dk_ref, td, 'wgrad cin=4', k=(5.0 if td == torch.float32 else 0.5))
    cin = 32
    x = rnd(rng, (n, h, w, cin), td)
    sc, sh = rng.standard_normal(cin).astype(np.float32), rng.standard_normal(cin).astype(np.float32)
    a = np.maximum(x * sc.astype(np.float64) + sh.astype(np.float64), 0)
    if td == torch.bfloat16:
        a = torch.tensor(a, dtype=torch.float32).to(td).double().numpy()
    _, dk_ref, _ = K.conv2d_same_bwd(a, np.zeros((3, 3, cin, cout)), dy, 3)
    dk = ops.conv2d_wgrad(to_dev(x, td), to_dev(dy, td), cin, cout, dil=3, in_scale=f32dev(sc), in_shift=f32dev(sh), in_relu=True)
    close(back(dk), dk_ref, td, 'wgrad affine dil3', k=(5.0 if td == torch.float32 else 0.5))


# ------------------------------------------------- the double-buffered 256 x 128 tile, forced on small shapes
@pytest.fixture
def force_db(ops):
    from satellite_computervision_amd._lib import lib, check
    import ctypes
    old = ctypes.c_int32()
    check(lib.satcv_get_option(b'igemm_db', ctypes.byref(old)))
    check(lib.satcv_set_option(b'igemm_db', 2))
    yield
    check(lib.satcv_set_option(b'igemm_db', old.value))


DB_CASES = [
    # n, h, w, cin, cout: Cout % 128 == 0 and Cin >= 64 (the kernel's shape limits); every tile width (32 / 16 / 8), several images
    # per tile, ragged grids, odd image counts, one and many K-chunks (chunk parity of the two LDS stages)
    (1, 16, 16, 64, 128), (2, 16, 16, 1024, 512), (2, 16, 16, 512, 512), (3, 8, 8, 512, 1024), (2, 32, 32, 512, 256),
    (2, 32, 32, 128, 256), (1, 64, 64, 256, 128), (3, 20, 24, 64, 128), (5, 8, 8, 80, 128), (1, 40, 72, 96, 128), (7, 4, 4, 64, 128),
    (2, 12, 12, 64, 256),
    # round 5, the 16x16x32 tile (csrc/conv_igemm_m16.hip: Cin % 64 == 0, maps at least 16 pixels wide and a tile high): ragged tile rows for
    # both tile widths, an odd image count, two and many 32-channel chunks
    (1, 40, 48, 128, 128), (3, 20, 64, 64, 128), (1, 24, 32, 192, 256),
]


@pytest.mark.parametrize('td', [torch.bfloat16])
@pytest.mark.parametrize('case', DB_CASES)
def test_conv2d_double_buffered_tile(ops, td, case, force_db):
    """forward (with BN statistics) and data gradient of the 3x3 conv through the 8-wave double-buffered configuration."""
    n, h, w, cin, cout = case
    rng = np.random.default_rng(hash(case) % 2**31)
    x = rnd(rng, (n, h, w, cin), td)
    kern = rnd(rng, (3, 3, cin, cout), td, 0.2)
    b = rng.standard_normal(cout)
    ref = K.conv2d_same(x, kern, b, 1)
    wf, _ = ops.pack_weights(f32dev(kern), cin, ops.DTYPE_CODE[td], want_dgrad=False)
    stats = ops.new_stats(cout, dev())
    y = ops.conv2d(to_dev(x, td), wf, cout, bias=f32dev(b), stats=stats)
    got = back(y, cout)
    close(got, ref, td, f'db conv {case}')
    s = stats.sum(0).double().cpu().numpy()
    np.testing.assert_allclose(s[0, :cout], got.sum((0, 1, 2)), rtol=2e-4, atol=2e-3 * np.sqrt(n * h * w))
    np.testing.assert_allclose(s[1, :cout], (got ** 2).sum((0, 1, 2)), rtol=2e-4)
    # data gradient of the transposed problem: dy has `cin` channels here, dx `cout` (a multiple of 128)
    kern2 = rnd(rng, (3, 3, cout, cin), td, 0.2)
    dy = rnd(rng, (n, h, w, cin), td)
    dx_ref, _, _ = K.conv2d_same_bwd(np.zeros((n, h, w, cout)), kern2, dy, 1)
    _, wd = ops.pack_weights(f32dev(kern2), cout, ops.DTYPE_CODE[td])
    dx = ops.conv2d_dgrad(to_dev(dy, td), wd, cout)
    close(back(dx, cout), dx_ref, td, f'db dgrad {case}')


@pytest.mark.parametrize('case', [(4, 16, 16, 512, 512, 0), (2, 32, 32, 256, 128, 0), (2, 16, 16, 128, 256, 128), (1, 64, 64, 128, 128, 64)])
def test_16x16x32_tile_fused_bn_backward_sums(ops, case, force_db):
    """the fused BatchNorm-backward sums (bst_*) in the epilogue of the 16x16x32 tile, whose accumulator layout differs from the 32x32x16
    one (a lane owns two columns): against float64 sums of the stored gradient; the stored gradient itself against the oracle."""
    n, h, w, cin, cout, split = case
    td = torch.bfloat16
    rng = np.random.default_rng(hash(case) % 2**31)
    x = rnd(rng, (n, h, w, cin), td)
    kern = rnd(rng, (3, 3, cin, cout), td, 0.1)
    ref = K.conv2d_same(x, kern, np.zeros(cout), 1)
    v = torch.tensor(rnd(rng, (n, h, w, cout), td) * 1.5 + 0.25, dtype=torch.float32).to(td).double().numpy()
    sc, sh = rng.standard_normal(cout).astype(np.float32), rng.standard_normal(cout).astype(np.float32) * 0.5
    mu, rs = rng.standard_normal(cout).astype(np.float32) * 0.3, (0.5 + rng.random(cout)).astype(np.float32)
    wf, _ = ops.pack_weights(f32dev(kern), cin, ops.DTYPE_CODE[td])
    v0 = to_dev(v[..., :split] if split else v, td)
    v1 = to_dev(v[..., split:], td) if split else None
    for relu in (1, 0):
        stats = ops.new_stats(cout, dev())
        bst = dict(y=v0, ld=v0.shape[-1], scale=f32dev(sc), shift=f32dev(sh), mean=f32dev(mu), rstd=f32dev(rs), relu=relu)
        if split:
            bst.update(y1=v1, ld1=v1.shape[-1], split=split)
        y = ops.conv2d(to_dev(x, td), wf, cout, stats=stats, bst=bst)
        g = back(y, cout)
        close(g, ref, td, f'16x16x32 conv {case}')
        mask = (v * sc.astype(np.float64) + sh.astype(np.float64) > 0) if relu else np.ones_like(v, bool)
        gg = np.where(mask, g, 0.0)
        xh = (v - mu.astype(np.float64)) * rs.astype(np.float64)
        got = stats.sum(0).double().cpu().numpy()
        tol = 2e-4 * np.sqrt(n * h * w) * max(1.0, float(np.abs(g).max()))
        np.testing.assert_allclose(got[0], gg.sum((0, 1, 2)), rtol=1e-4, atol=tol, err_msg=f'sum g {case} relu={relu}')
        np.testing.assert_allclose(got[1], (gg * xh).sum((0, 1, 2)), rtol=1e-4, atol=tol * float(np.abs(xh).max()), err_msg=f'sum g xhat {case} relu={relu}')


def m16p_launches():
    from satellite_computervision_amd._lib import lib, check
    import ctypes
    v = ctypes.c_int32()
    check(lib.satcv_get_option(b'm16p_launches', ctypes.byref(v)))
    return v.value


@pytest.fixture
def force_m16p(ops):
    """every launch conv_igemm_m16p.hip serves runs on it (the default leaves launches with one tile per workgroup to the one-tile kernels),
    inference-style launches (no statistics) included"""
    from satellite_computervision_amd._lib import lib, check
    import ctypes
    old = {}
    # (igemm_thin = 0: the 64 -> 64 shape would otherwise run on the persistent weights-stationary kernel, which is asked first)
    for k, v in ((b'm16p', 2), (b'igemm_m16', 2), (b'igemm_thin', 0)):
        o = ctypes.c_int32()
        check(lib.satcv_get_option(k, ctypes.byref(o)))
        old[k] = o.value
        check(lib.satcv_set_option(k, v))
    yield
    for k, v in old.items():
        check(lib.satcv_set_option(k, v))


# n, h, w, cin, cout, split of the input (two sources) / of the raw outputs of the fused sums
M16P_CASES = [(2, 32, 32, 64, 128, 0), (1, 64, 64, 128, 256, 64), (9, 128, 96, 64, 128, 32), (3, 40, 64, 256, 128, 128), (40, 64, 64, 64, 128, 0), (2, 16, 64, 192, 384, 64),
              # round 6: the 64-channel output block (64 filters: 64 -> 64, 64 + 64 -> 64 and the data gradient 128 -> 64 of the U-Net; 192 = three blocks)
              (2, 32, 32, 64, 64, 0), (5, 64, 96, 128, 64, 64), (3, 24, 64, 128, 192, 32), (33, 64, 64, 64, 64, 32)]


@pytest.mark.parametrize('case', M16P_CASES)
def test_conv2d_persistent_16x16x32_kernel(ops, case, force_m16p):
    """conv_igemm_m16p.hip (persistent, cross-tile pipelined 16x16x32 tile; staging / matrix wave roles, weights by LDS-DMA, swapped MFMA
    operands, swizzled staging image): one, two and many tiles per workgroup, ranges of unequal length, 1 - 3 output-channel blocks, 2 - 8
    chunks per tile.  Forward with bias + statistics of the stored values; two sources + the producing layer's BatchNorm + ReLU in the loader;
    the data gradient with the fused BatchNorm-backward sums over one or two raw-output tensors."""
    n, h, w, cin, cout, split = case
    td = torch.bfloat16
    rng = np.random.default_rng(hash(case) % 2**31 + 11)
    x = rnd(rng, (n, h, w, cin), td)
    kern = rnd(rng, (3, 3, cin, cout), td, 0.1)
    b = rng.standard_normal(cout)
    ref = K.conv2d_same(x, kern, b, 1)
    wf, _ = ops.pack_weights(f32dev(kern), cin, ops.DTYPE_CODE[td], want_dgrad=False)
    stats = ops.new_stats(cout, dev())
    before = m16p_launches()
    y = ops.conv2d(to_dev(x, td), wf, cout, bias=f32dev(b), stats=stats)
    assert m16p_launches() - before == 1, 'path taken'
    got = back(y, cout)
    close(got, ref, td, f'persistent conv {case}')
    s = stats.sum(0).double().cpu().numpy()
    np.testing.assert_allclose(s[0, :cout], got.sum((0, 1, 2)), rtol=2e-4, atol=2e-3 * np.sqrt(n * h * w))
    np.testing.assert_allclose(s[1, :cout], (got ** 2).sum((0, 1, 2)), rtol=2e-4)
    # no statistics (the launch of a training plan whose BatchNorm is frozen / an inference launch under option igemm_m16 = 2): same values
    y_ns = ops.conv2d(to_dev(x, td), wf, cout, bias=f32dev(b))
    assert torch.equal(y_ns, y)
    if split:
        sc, sh = (rng.random(cin) + 0.5).astype(np.float32), (rng.standard_normal(cin) * 0.3).astype(np.float32)
        a_ref = np.maximum(x * sc.astype(np.float64) + sh.astype(np.float64), 0)
        a_ref = torch.tensor(a_ref, dtype=torch.float32).to(td).double().numpy()
        ref2 = K.conv2d_same(a_ref, kern, b, 1)
        before = m16p_launches()
        y2 = ops.conv2d(to_dev(x[..., :split], td), wf, cout, x1=to_dev(x[..., split:], td), bias=f32dev(b), in_scale=f32dev(sc), in_shift=f32dev(sh), in_relu=True)
        assert m16p_launches() - before == 1, 'path taken'
        close(back(y2, cout), ref2, td, f'persistent conv, two sources + affine {case}', k=2.0)
    # fused BatchNorm-backward sums (the launch writes dL/d act of a conv -> BN -> ReLU layer whose raw outputs are v)
    ref0 = ref - b
    v = torch.tensor(rnd(rng, (n, h, w, cout), td) * 1.5 + 0.25, dtype=torch.float32).to(td).double().numpy()
    bsc, bsh = rng.standard_normal(cout).astype(np.float32), rng.standard_normal(cout).astype(np.float32) * 0.5
    mu, rs = rng.standard_normal(cout).astype(np.float32) * 0.3, (0.5 + rng.random(cout)).astype(np.float32)
    vs = cout // 2 if split else 0
    v0 = to_dev(v[..., :vs] if vs else v, td)
    v1 = to_dev(v[..., vs:], td) if vs else None
    for relu in (1, 0):
        st = ops.new_stats(cout, dev())
        bst = dict(y=v0, ld=v0.shape[-1], scale=f32dev(bsc), shift=f32dev(bsh), mean=f32dev(mu), rstd=f32dev(rs), relu=relu)
        if vs:
            bst.update(y1=v1, ld1=v1.shape[-1], split=vs)
        before = m16p_launches()
        yb = ops.conv2d(to_dev(x, td), wf, cout, stats=st, bst=bst)
        assert m16p_launches() - before == 1, 'path taken'
        g = back(yb, cout)
        close(g, ref0, td, f'persistent conv with fused sums {case}')
        mask = (v * bsc.astype(np.float64) + bsh.astype(np.float64) > 0) if relu else np.ones_like(v, bool)
        gg = np.where(mask, g, 0.0)
        xh = (v - mu.astype(np.float64)) * rs.astype(np.float64)
        gs = st.sum(0).double().cpu().numpy()
        tol = 2e-4 * np.sqrt(n * h * w) * max(1.0, float(np.abs(g).max()))
        np.testing.assert_allclose(gs[0], gg.sum((0, 1, 2)), rtol=1e-4, atol=tol, err_msg=f'sum g {case} relu={relu}')
        np.testing.assert_allclose(gs[1], (gg * xh).sum((0, 1, 2)), rtol=1e-4, atol=tol * float(np.abs(xh).max()), err_msg=f'sum g xhat {case} relu={relu}')


def test_reduce_slabs_batched(ops):
    """satcv_reduce_slabs_batched: several layers' fp32 partial slabs summed in ONE launch (few / many slabs -> 1 ... 16 lanes per output, plain HWIO
    and transposed-conv layouts, accumulate, padded slab rows) against float64 sums; two runs are bit-identical (fixed summation order)."""
    import ctypes
    from satellite_computervision_amd._lib import lib, check, ReduceJob
    rng = np.random.default_rng(3)
    specs = [  # nslab, taps, cin, nvalid, kpad, npad, transposed, accumulate
        (2, 9, 64, 128, 64, 128, 0, 0), (8, 9, 32, 32, 32, 32, 0, 0), (128, 9, 16, 32, 32, 32, 0, 1), (37, 1, 24, 64, 32, 64, 0, 0),
        # 9 x 13 x 8 = 936 items of one lane (8 mod 16) in front of a 16-lane job: item counts are padded to 16 so that its groups stay inside a wave
        (2, 9, 13, 32, 16, 32, 0, 0), (64, 9, 16, 64, 16, 64, 0, 0),
        (16, 1, 128, 4 * 64, 128, 256, 1, 0), (5, 1, 32, 4 * 32, 32, 128, 1, 1)]
    jobs, keep, prefix, tot = [], [], [], 0
    for nslab, taps, cin, nvalid, kpad, npad, tr, acc in specs:
        ws = torch.tensor(rng.standard_normal((nslab, taps, kpad, npad)).astype(np.float32), device=dev())
        dw0 = rng.standard_normal((taps, cin, nvalid) if not tr else (nvalid, cin)).astype(np.float32)
        dw = torch.tensor(dw0, device=dev())
        j = ReduceJob()
        j.ws, j.dw, j.nslab, j.taps, j.kpad, j.npad, j.cin, j.nvalid, j.transposed, j.accumulate = ws.data_ptr(), dw.data_ptr(), nslab, taps, kpad, npad, cin, nvalid, tr, acc
        lanes = 1
        while lanes < 16 and 2 * lanes <= nslab // 2:
            lanes *= 2
        j.lanes = lanes
        prefix.append(tot)
        tot += int(lib.satcv_reduce_job_items(ctypes.byref(j)))
        jobs.append(j); keep.append((ws, dw, dw0))
    arr = (ReduceJob * len(jobs))(*jobs)
    jd = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev())
    pd = torch.tensor(prefix, dtype=torch.int64, device=dev())
    outs = []
    for rep in range(2):
        for (ws, dw, dw0) in keep:
            dw.copy_(torch.tensor(dw0))
        check(lib.satcv_reduce_slabs_batched(jd.data_ptr(), pd.data_ptr(), len(jobs), tot, ops.stream_ptr()))
        torch.cuda.synchronize()
        outs.append([dw.clone() for (_, dw, _) in keep])
    for (nslab, taps, cin, nvalid, kpad, npad, tr, acc), (ws, dw, dw0), o0, o1 in zip(specs, keep, outs[0], outs[1]):
        assert torch.equal(o0, o1)
        ref = ws.double().sum(0)[:, :cin, :nvalid].cpu().numpy()
        if tr:
            ref = ref[0].T                                       # (f, f, cout, cin) flattened = [nvalid][cin]
        if acc:
            ref = ref + dw0
        np.testing.assert_allclose(o0.cpu().numpy(), ref, rtol=2e-6, atol=2e-5 * np.sqrt(nslab))


def test_conv2d_double_buffered_dual_source_affine(ops, force_db):
    """decoder conv1 through the double-buffered tile: concat([skip, up]) -> BN -> ReLU in the loader, 512 + 512 -> 128 channels."""
    td = torch.bfloat16
    rng = np.random.default_rng(12)
    n, h, w, c0, c1, cout = 2, 16, 16, 256, 256, 128
    xa, xb = rnd(rng, (n, h, w, c0), td), rnd(rng, (n, h, w, c1), td)
    sc, sh = rng.standard_normal(c0 + c1), rng.standard_normal(c0 + c1)
    kern = rnd(rng, (3, 3, c0 + c1, cout), td, 0.2)
    sc32, sh32 = sc.astype(np.float32).astype(np.float64), sh.astype(np.float32).astype(np.float64)
    a = np.maximum(np.concatenate([xa, xb], -1) * sc32 + sh32, 0)
    a = torch.tensor(a, dtype=torch.float32).to(td).double().numpy()
    ref = K.conv2d_same(a, kern, None, 1)
    wf, _ = ops.pack_weights(f32dev(kern), c0 + c1, ops.DTYPE_CODE[td], want_dgrad=False)
    y = ops.conv2d(to_dev(xa, td), wf, cout, x1=to_dev(xb, td), in_scale=f32dev(sc), in_shift=f32dev(sh), in_relu=True)
    close(back(y, cout), ref, td, 'db dual-source conv', k=2.0)


# ------------------------------------------------- the persistent weights-stationary kernel of the thin layers, forced on small shapes
@pytest.fixture
def force_thin(ops):
    from satellite_computervision_amd._lib import lib, check
    import ctypes
    old = ctypes.c_int32()
    check(lib.satcv_get_option(b'igemm_thin', ctypes.byref(old)))
    check(lib.satcv_set_option(b'igemm_thin', 2))
    yield
    check(lib.satcv_set_option(b'igemm_thin', old.value))


WS_CASES = [
    # n, h, w, cin (real), cout: every (stored Cin, Cout) pair the kernel serves on maps of whole 8 x 32 tiles; one and many tiles per
    # workgroup, more tiles than workgroups (9 x 136 x 224 = 1071 tiles); the two ragged maps fall back to the general kernel
    (2, 32, 32, 4, 32), (1, 64, 64, 32, 32), (2, 40, 96, 64, 32), (1, 256, 256, 64, 32), (3, 24, 32, 32, 64), (2, 64, 96, 16, 64),
    (1, 8, 32, 32, 32), (2, 16, 64, 64, 32), (9, 136, 224, 16, 32), (2, 9, 33, 64, 32), (3, 20, 24, 32, 64),
    (2, 24, 64, 64, 64), (1, 128, 128, 64, 64),      # the 8-wave form
]


def ws_launches():
    from satellite_computervision_amd._lib import lib, check
    import ctypes
    v = ctypes.c_int32()
    check(lib.satcv_get_option(b'igemm_thin_launches', ctypes.byref(v)))
    return v.value


@pytest.mark.parametrize('case', WS_CASES)
def test_conv2d_weights_stationary_thin_kernel(ops, case, force_thin):
    """forward (bias + BN statistics) and data gradient through conv_igemm_ws.hip"""
    td = torch.bfloat16
    n, h, w, cin, cout = case
    rng = np.random.default_rng(hash(case) % 2**31)
    x = rnd(rng, (n, h, w, cin), td)
    kern = rnd(rng, (3, 3, cin, cout), td, 0.2)
    b = rng.standard_normal(cout)
    ref = K.conv2d_same(x, kern, b, 1)
    cpad = rup(cin, 16)
    wf, _ = ops.pack_weights(f32dev(kern), cpad, ops.DTYPE_CODE[td], want_dgrad=False)
    stats = ops.new_stats(cout, dev())
    before = ws_launches()
    y = ops.conv2d(to_dev(x, td, cpad), wf, cout, bias=f32dev(b), stats=stats)
    assert ws_launches() - before == (1 if (h % 8 == 0 and w % 32 == 0) else 0), 'path taken'
    got = back(y, cout)
    close(got, ref, td, f'ws conv {case}')
    s = stats.sum(0).double().cpu().numpy()
    np.testing.assert_allclose(s[0, :cout], got.sum((0, 1, 2)), rtol=2e-4, atol=2e-3 * np.sqrt(n * h * w))
    np.testing.assert_allclose(s[1, :cout], (got ** 2).sum((0, 1, 2)), rtol=2e-4)
    if cin >= 16 and cout in (16, 32, 64):
        # data gradient of the transposed problem (dy: cout channels -> dx: cin channels) when that pair is served too
        cin2, cout2 = cout, cin if cin in (32, 64) else 32
        kern2 = rnd(rng, (3, 3, cout2, cin2), td, 0.2)
        dy = rnd(rng, (n, h, w, cin2), td)
        dx_ref, _, _ = K.conv2d_same_bwd(np.zeros((n, h, w, cout2)), kern2, dy, 1)
        _, wd = ops.pack_weights(f32dev(kern2), cout2, ops.DTYPE_CODE[td])
        dx = ops.conv2d_dgrad(to_dev(dy, td), wd, cout2)
        close(back(dx, cout2), dx_ref, td, f'ws dgrad {case}')


def roles_launches():
    from satellite_computervision_amd._lib import lib, check
    import ctypes
    v = ctypes.c_int32()
    check(lib.satcv_get_option(b'thin_roles_launches', ctypes.byref(v)))
    return v.value


@pytest.fixture
def force_roles(ops):
    """every shape conv_thin_roles.hip serves runs on it (the default routes only the shapes it measured faster on)"""
    from satellite_computervision_amd._lib import lib, check
    import ctypes
    old = ctypes.c_int32()
    check(lib.satcv_get_option(b'thin_roles', ctypes.byref(old)))
    check(lib.satcv_set_option(b'thin_roles', 2))
    yield
    check(lib.satcv_set_option(b'thin_roles', old.value))


@pytest.mark.parametrize('case', [(2, 32, 32, 4, 32), (1, 64, 64, 32, 32), (2, 40, 96, 64, 32), (3, 24, 32, 32, 64), (2, 64, 96, 16, 64), (1, 8, 32, 32, 32),
                                  (9, 136, 224, 16, 32), (5, 12, 64, 64, 32), (1, 256, 256, 32, 32)])
def test_conv2d_thin_layers_with_wave_roles(ops, case, force_thin, force_roles):
    """conv_thin_roles.hip (staging / matrix wave roles), all five instantiations: forward with bias + BatchNorm statistics of the stored values,
    the input BatchNorm + ReLU in the loader with a two-source input, one and many tiles per workgroup, 4-row tiles (64 -> 32: also maps
    whose height is a multiple of 4 only), the folded inference epilogue (multiplier, ReLU, fused 2 x 2 max-pool)."""
    td = torch.bfloat16
    n, h, w, cin, cout = case
    rng = np.random.default_rng(hash(case) % 2**31 + 5)
    cpad = rup(cin, 16)
    x = rnd(rng, (n, h, w, cin), td)
    kern = rnd(rng, (3, 3, cin, cout), td, 0.2)
    b = rng.standard_normal(cout)
    ref = K.conv2d_same(x, kern, b, 1)
    wf, _ = ops.pack_weights(f32dev(kern), cpad, ops.DTYPE_CODE[td], want_dgrad=False)
    stats = ops.new_stats(cout, dev())
    before = roles_launches()
    y = ops.conv2d(to_dev(x, td, cpad), wf, cout, bias=f32dev(b), stats=stats)
    assert roles_launches() - before == 1, 'path taken'
    got = back(y, cout)
    close(got, ref, td, f'roles conv {case}')
    s = stats.sum(0).double().cpu().numpy()
    np.testing.assert_allclose(s[0, :cout], got.sum((0, 1, 2)), rtol=2e-4, atol=2e-3 * np.sqrt(n * h * w))
    np.testing.assert_allclose(s[1, :cout], (got ** 2).sum((0, 1, 2)), rtol=2e-4)
    if cin >= 32:
        # two sources (concat([skip, up])) + the producing layer's BatchNorm + ReLU in the loader
        c0 = cin // 2
        sc, sh = (rng.random(cin) + 0.5).astype(np.float32), (rng.standard_normal(cin) * 0.3).astype(np.float32)
        a_ref = np.maximum(x * sc.astype(np.float64) + sh.astype(np.float64), 0)
        a_ref = torch.tensor(a_ref, dtype=torch.float32).to(td).double().numpy()
        ref2 = K.conv2d_same(a_ref, kern, b, 1)
        y2 = ops.conv2d(to_dev(x[..., :c0], td), wf, cout, x1=to_dev(x[..., c0:], td), bias=f32dev(b), in_scale=f32dev(sc), in_shift=f32dev(sh), in_relu=True)
        close(back(y2, cout), ref2, td, f'roles conv, two sources + affine {case}', k=2.0)
    if h % 8 == 0:
        # folded inference epilogue: per-channel multiplier, ReLU, 2 x 2 max-pool of the stored tile
        osc = (rng.random(cout) + 0.5).astype(np.float32)
        ref3 = np.maximum(K.conv2d_same(x, kern, np.zeros(cout), 1) * osc.astype(np.float64) + b, 0)
        pooled = torch.zeros(n, h // 2, w // 2, cout, dtype=td, device=dev())
        y3 = ops.conv2d(to_dev(x, td, cpad), wf, cout, bias=f32dev(b), out_scale=f32dev(osc), out_relu=True, pool_y=pooled, pool_f=2)
        g3 = back(y3, cout)
        close(g3, ref3, td, f'roles conv, folded epilogue {case}')
        assert np.array_equal(back(pooled, cout), g3.reshape(n, h // 2, 2, w // 2, 2, cout).max((2, 4)))


@pytest.mark.parametrize('case', [(2, 32, 64, 16, 16), (1, 64, 64, 32, 32), (3, 16, 32, 4, 16), (2, 24, 96, 32, 16), (1, 8, 32, 16, 32)])
def test_conv2d_weights_stationary_dilation_3(ops, case, force_thin):
    """the thin persistent kernel with a 3-pixel halo: the dilated convolutions of the atrous CNNs (utils/model_tools.py:935, 968), 16 / 32 channels,
    incl. 16 OUTPUT channels on a 32-column tile (upper column groups not stored) -- forward with bias + statistics, the fused input BatchNorm +
    ReLU, and the data gradient (the transposed problem)."""
    td = torch.bfloat16
    n, h, w, cin, cout = case
    rng = np.random.default_rng(hash(case) % 2**31 + 3)
    cpad = rup(cin, 16)
    x = rnd(rng, (n, h, w, cin), td)
    kern = rnd(rng, (3, 3, cin, cout), td, 0.2)
    b = rng.standard_normal(cout)
    ref = K.conv2d_same(x, kern, b, 3)
    wf, wd = ops.pack_weights(f32dev(kern), cpad, ops.DTYPE_CODE[td])
    stats = ops.new_stats(rup(cout, 16), dev())
    before = ws_launches()
    y = ops.conv2d(to_dev(x, td, cpad), wf, cout, dil=3, bias=f32dev(b), stats=stats)
    assert ws_launches() - before == 1, 'path taken'
    got = back(y, cout)
    close(got, ref, td, f'ws dilated conv {case}')
    if cout == 16:
        # the same 32-column tile for 16 output channels without dilation (the 16-filter atrous CNN's plain convolutions)
        before = ws_launches()
        y1 = ops.conv2d(to_dev(x, td, cpad), wf, cout, bias=f32dev(b))
        assert ws_launches() - before == 1, 'path taken (dilation 1, 16 output channels)'
        close(back(y1, cout), K.conv2d_same(x, kern, b, 1), td, f'ws conv, 16 output channels {case}')
    s = stats.sum(0).double().cpu().numpy()
    np.testing.assert_allclose(s[0, :cout], got.sum((0, 1, 2)), rtol=2e-4, atol=2e-3 * np.sqrt(n * h * w))
    np.testing.assert_allclose(s[1, :cout], (got ** 2).sum((0, 1, 2)), rtol=2e-4)
    if cin >= 16:
        sc, sh = (rng.random(cin) + 0.5).astype(np.float32), (rng.standard_normal(cin) * 0.3).astype(np.float32)
        a_ref = np.maximum(x * sc.astype(np.float64) + sh.astype(np.float64), 0)
        a_ref = torch.tensor(a_ref, dtype=torch.float32).to(td).double().numpy()
        ref2 = K.conv2d_same(a_ref, kern, b, 3)
        y2 = ops.conv2d(to_dev(x, td), wf, cout, dil=3, bias=f32dev(b), in_scale=f32dev(sc), in_shift=f32dev(sh), in_relu=True)
        close(back(y2, cout), ref2, td, f'ws dilated conv + affine {case}', k=2.0)
        # data gradient: dy has `cout` channels, dx `cin`
        dy = rnd(rng, (n, h, w, cout), td)
        dx_ref, _, _ = K.conv2d_same_bwd(np.zeros((n, h, w, cin)), kern, dy, 3)
        before = ws_launches()
        dx = ops.conv2d_dgrad(to_dev(dy, td), wd, cin, dil=3)
        assert ws_launches() - before == 1, 'path taken (data gradient)'
        close(back(dx, cin), dx_ref, td, f'ws dilated dgrad {case}')
        # accumulate: dx += (the residual sum's second consumer)
        g0 = rnd(rng, (n, h, w, cin), td)
        acc = to_dev(g0, td)
        before = ws_launches()
        ops.conv2d_dgrad(to_dev(dy, td), wd, cin, dil=3, out=acc, accumulate=True)
        assert ws_launches() - before == 1, 'path taken (accumulating data gradient)'
        want = torch.tensor(torch.tensor(g0, dtype=torch.float32).to(td).double().numpy() + back(dx, cin), dtype=torch.float32).to(td).double().numpy()
        np.testing.assert_allclose(back(acc, cin), want, rtol=0, atol=float(np.abs(want).max()) * 2 ** -7)


def test_conv2d_weights_stationary_dual_source_affine(ops, force_thin):
    """dec0.conv1 through the thin kernel: concat([skip 32, up 32]) -> BN -> ReLU in the loader, -> 32 channels"""
    td = torch.bfloat16
    rng = np.random.default_rng(13)
    n, h, w, c0, c1, cout = 2, 40, 96, 32, 32, 32
    xa, xb = rnd(rng, (n, h, w, c0), td), rnd(rng, (n, h, w, c1), td)
    sc, sh = rng.standard_normal(c0 + c1), rng.standard_normal(c0 + c1)
    kern = rnd(rng, (3, 3, c0 + c1, cout), td, 0.2)
    sc32, sh32 = sc.astype(np.float32).astype(np.float64), sh.astype(np.float32).astype(np.float64)
    a = np.maximum(np.concatenate([xa, xb], -1) * sc32 + sh32, 0)
    a = torch.tensor(a, dtype=torch.float32).to(td).double().numpy()
    ref = K.conv2d_same(a, kern, None, 1)
    wf, _ = ops.pack_weights(f32dev(kern), c0 + c1, ops.DTYPE_CODE[td], want_dgrad=False)
    y = ops.conv2d(to_dev(xa, td), wf, cout, x1=to_dev(xb, td), in_scale=f32dev(sc), in_shift=f32dev(sh), in_relu=True)
    close(back(y, cout), ref, td, 'ws dual-source conv', k=2.0)
    # a single 16-channel source with an affine and no ReLU
    x = rnd(rng, (n, h, w, 16), td)
    sc, sh = rng.standard_normal(16), rng.standard_normal(16)
    kern = rnd(rng, (3, 3, 16, 64), td, 0.2)
    a = x * sc.astype(np.float32).astype(np.float64) + sh.astype(np.float32).astype(np.float64)
    a = torch.tensor(a, dtype=torch.float32).to(td).double().numpy()
    ref = K.conv2d_same(a, kern, None, 1)
    wf, _ = ops.pack_weights(f32dev(kern), 16, ops.DTYPE_CODE[td], want_dgrad=False)
    y = ops.conv2d(to_dev(x, td), wf, 64, in_scale=f32dev(sc), in_shift=f32dev(sh), in_relu=False)
    close(back(y, 64), ref, td, 'ws affine conv', k=2.0)


@pytest.mark.parametrize('td', DT)
@pytest.mark.parametrize('case', [(2, 32, 32, 32, 32, 32), (2, 16, 16, 64, 64, 128), (1, 64, 64, 32, 0, 64), (3, 8, 8, 128, 128, 128), (1, 40, 72, 64, 32, 32),
                                  (2, 16, 16, 512, 512, 256)])
def test_wgrad_dual_source_with_affine(ops, td, case):
    """decoder conv1 weight gradient: X = ReLU(BN(concat([skip, up]))) formed in the loader from two sources (utils/model_tools.py:307-312),
    dilation 1 -- the double-buffered kernel's path (per-thread source / scale / shift, LDS scale table)."""
    n, h, w, c0, c1, cout = case
    rng = np.random.default_rng(hash(case) % 2**31)
    xa = rnd(rng, (n, h, w, c0), td)
    xb = rnd(rng, (n, h, w, c1), td) if c1 else None
    cin = c0 + c1
    dy = rnd(rng, (n, h, w, cout), td)
    sc, sh = rng.standard_normal(cin).astype(np.float32), rng.standard_normal(cin).astype(np.float32)
    x = np.concatenate([xa, xb], -1) if c1 else xa
    a = np.maximum(x * sc.astype(np.float64) + sh.astype(np.float64), 0)
    if td == torch.bfloat16:
        a = torch.tensor(a, dtype=torch.float32).to(td).double().numpy()
    _, dk_ref, _ = K.conv2d_same_bwd(a, np.zeros((3, 3, cin, cout)), dy, 1)
    dk = ops.conv2d_wgrad(to_dev(xa, td), to_dev(dy, td), cin, cout, x1=to_dev(xb, td) if c1 else None, in_scale=f32dev(sc), in_shift=f32dev(sh), in_relu=True)
    close(back(dk), dk_ref, td, f'wgrad dual+affine {case}', k=(5.0 if td == torch.float32 else 0.5))


# ------------------------------------------------------------------ transposed conv
@pytest.mark.parametrize('td', DT)
@pytest.mark.parametrize('case', [(2, 8, 8, 64, 32, 2), (1, 16, 16, 128, 64, 2), (2, 4, 4, 256, 128, 2), (1, 6, 6, 32, 32, 3),
                                  (2, 32, 32, 64, 32, 2), (2, 24, 24, 64, 32, 2), (1, 64, 64, 64, 32, 2), (2, 32, 32, 128, 64, 2),
                                  (2, 8, 8, 1024, 512, 2), (2, 16, 16, 512, 256, 2),
                                  (3, 5, 7, 128, 64, 2), (2, 24, 40, 256, 64, 2), (1, 12, 12, 128, 128, 2)])      # ragged maps on the 128 x 256 wgrad block
def test_conv2d_transpose(ops, td, case):
    n, h, w, cin, cout, f = case
    rng = np.random.default_rng(hash(case) % 2**31)
    x = rnd(rng, (n, h, w, cin), td)
    kt = rnd(rng, (f, f, cout, cin), td, 0.2)
    b = rng.standard_normal(cout)
    dy = rnd(rng, (n, h * f, w * f, cout), td)
    ref = K.conv2d_transpose_ks(x, kt, b)
    dx_ref, dk_ref, _ = K.conv2d_transpose_ks_bwd(x, kt, dy)
    wf, wd = ops.pack_weights(f32dev(kt), cin, ops.DTYPE_CODE[td], transposed=True)
    stats = ops.new_stats(cout, dev())
    y = ops.conv2d_transpose(to_dev(x, td), wf, cout, f, bias=f32dev(b), stats=stats)
    got = back(y, cout)
    close(got, ref, td, f'convT fwd {case}')
    s = stats.sum(0).double().cpu().numpy()
    np.testing.assert_allclose(s[0], got.sum((0, 1, 2)), rtol=2e-4, atol=2e-3 * np.sqrt(got.size / cout))
    dx = ops.conv2d_transpose_dgrad(to_dev(dy, td), wd, cin, cout, f)
    close(back(dx, cin), dx_ref, td, f'convT dgrad {case}')
    dk = ops.conv2d_wgrad(to_dev(x, td), to_dev(dy, td), cin, cout, transposed_f=f)
    close(back(dk), dk_ref, td, f'convT wgrad {case}', k=(5.0 if td == torch.float32 else 0.5))


@pytest.mark.parametrize('case', [(2, 32, 32, 64, 32), (1, 16, 64, 128, 64), (3, 8, 96, 64, 32), (1, 64, 64, 64, 32), (2, 32, 32, 256, 128), (1, 8, 32, 256, 128), (5, 16, 64, 256, 128)])
def test_conv2d_transpose_streaming_kernel_with_input_batchnorm(ops, case):
    """The thin transposed convolutions (64 -> 32, 128 -> 64 channels, maps a multiple of 32 wide) run on the streaming kernel
    (conv_transpose_thin.hip): fused input BatchNorm + ReLU, bias, depth-to-space stores, statistics of the stored values, and the
    same result as the tiled kernel (SATCV_CONVT_THIN=0 is read once per process, so the tiled result comes from a narrower twin map)."""
    td = torch.bfloat16
    n, h, w, cin, cout = case
    rng = np.random.default_rng(sum(case))
    x = rnd(rng, (n, h, w, cin), td)
    kt = rnd(rng, (2, 2, cout, cin), td, 0.2)
    b = rng.standard_normal(cout)
    sc, sh = (rng.standard_normal(cin) * 0.5 + 1).astype(np.float32), rng.standard_normal(cin).astype(np.float32)
    a = np.maximum(x * sc.astype(np.float64) + sh.astype(np.float64), 0)
    a = torch.tensor(a, dtype=torch.float32).to(td).double().numpy()
    ref = K.conv2d_transpose_ks(a, kt, b)
    wf, _ = ops.pack_weights(f32dev(kt), cin, ops.DTYPE_CODE[td], transposed=True)
    for relu in (True, False):
        stats = ops.new_stats(cout, dev())
        y = ops.conv2d_transpose(to_dev(x, td), wf, cout, 2, bias=f32dev(b), in_scale=f32dev(sc), in_shift=f32dev(sh), in_relu=relu, stats=stats)
        got = back(y, cout)
        if relu:
            close(got, ref, td, f'convT streaming fwd {case}')
        else:
            al = torch.tensor(x * sc.astype(np.float64) + sh.astype(np.float64), dtype=torch.float32).to(td).double().numpy()
            close(got, K.conv2d_transpose_ks(al, kt, b), td, f'convT streaming fwd, linear input {case}')
        s = stats.sum(0).double().cpu().numpy()
        np.testing.assert_allclose(s[0], got.sum((0, 1, 2)), rtol=2e-4, atol=2e-3 * np.sqrt(got.size / cout))
        np.testing.assert_allclose(s[1], (got ** 2).sum((0, 1, 2)), rtol=2e-4, atol=1e-2)
    # without statistics / bias / input transform
    y2 = ops.conv2d_transpose(to_dev(x, td), wf, cout, 2)
    close(back(y2, cout), K.conv2d_transpose_ks(x, kt, np.zeros(cout)), td, f'convT streaming fwd plain {case}')
    # the folded inference form: per-channel multiplier on the accumulator, bias, ReLU in the epilogue
    osc = (rng.standard_normal(cout) * 0.5).astype(np.float32)
    y3 = ops.conv2d_transpose(to_dev(x, td), wf, cout, 2, bias=f32dev(b), out_scale=f32dev(osc), out_relu=True)
    ref3 = np.maximum(K.conv2d_transpose_ks(x, kt, np.zeros(cout)) * osc.astype(np.float64) + b, 0)
    close(back(y3, cout), ref3, td, f'convT streaming fwd, folded epilogue {case}')


@pytest.mark.parametrize('case', [(2, 32, 32, 64, 32), (1, 16, 64, 128, 64), (3, 8, 96, 64, 32), (1, 64, 64, 64, 32), (2, 32, 64, 128, 64)])
def test_conv2d_transpose_streaming_data_gradient_with_fused_bn_sums(ops, case):
    """Data gradient of the thin transposed convolutions on the streaming kernel (maps a multiple of 32 wide), with and without the fused
    BatchNorm-backward sums of the layer below: the stored gradient is the same either way, the sums are those of the STORED gradient."""
    td = torch.bfloat16
    n, h, w, cin, cout = case
    rng = np.random.default_rng(sum(case) + 1)
    x = rnd(rng, (n, h, w, cin), td)
    kt = rnd(rng, (2, 2, cout, cin), td, 0.2)
    dy = rnd(rng, (n, 2 * h, 2 * w, cout), td)
    dx_ref, _, _ = K.conv2d_transpose_ks_bwd(x, kt, dy)
    _, wd = ops.pack_weights(f32dev(kt), cin, ops.DTYPE_CODE[td], transposed=True)
    plain = ops.conv2d_transpose_dgrad(to_dev(dy, td), wd, cin, cout, 2)
    close(back(plain, cin), dx_ref, td, f'convT streaming dgrad {case}')
    v = rnd(rng, (n, h, w, cin), td) * 1.5 + 0.25
    v = torch.tensor(v, dtype=torch.float32).to(td).double().numpy()
    sc, sh = rng.standard_normal(cin).astype(np.float32), rng.standard_normal(cin).astype(np.float32) * 0.5
    mu, rs = rng.standard_normal(cin).astype(np.float32) * 0.3, (0.5 + rng.random(cin)).astype(np.float32)
    for relu in (1, 0):
        stats = ops.new_stats(cin, dev())
        got = ops.conv2d_transpose_dgrad(to_dev(dy, td), wd, cin, cout, 2, stats=stats,
                                         bst=dict(y=to_dev(v, td), ld=cin, scale=f32dev(sc), shift=f32dev(sh), mean=f32dev(mu), rstd=f32dev(rs), relu=relu))
        assert torch.equal(got, plain), 'the fused sums must not change the stored gradient'
        g = back(got, cin)
        mask = (v * sc.astype(np.float64) + sh.astype(np.float64) > 0) if relu else np.ones_like(v, bool)
        gg = np.where(mask, g, 0.0)
        xh = (v - mu.astype(np.float64)) * rs.astype(np.float64)
        s = stats.sum(0).double().cpu().numpy()
        tol = 2e-4 * np.sqrt(n * h * w) * max(1.0, float(np.abs(g).max()))
        np.testing.assert_allclose(s[0], gg.sum((0, 1, 2)), rtol=1e-4, atol=tol, err_msg=f'sum g {case} relu={relu}')
        np.testing.assert_allclose(s[1], (gg * xh).sum((0, 1, 2)), rtol=1e-4, atol=tol * float(np.abs(xh).max()), err_msg=f'sum g xhat {case} relu={relu}')


@pytest.mark.parametrize('case', [
    # n, h, w, cin, cout, input BatchNorm + ReLU, fused sums of the layer below, linear BatchNorm
    (2, 8, 64, 64, 32, True, True, False),        # dec0's shape class: one channel block, 64-pixel tiles
    (1, 4, 128, 64, 32, False, False, False),     # activated input (no affine)
    (3, 6, 64, 128, 64, True, True, False),       # dec1: one 128-channel block, 32-pixel tiles; odd image count
    (2, 16, 32, 192, 64, True, True, False),      # Cin not a multiple of 128: three 64-channel blocks share the g / y tiles
    (1, 5, 64, 128, 64, True, False, True),       # BatchNormalization without ReLU; more workgroups than tiles per slab
])
def test_convt_bwd_fused(ops, case):
    """satcv_convt_bwd_fused: the `up` half of decoder_block's backward (utils/model_tools.py:306-309 differentiated) in one launch -- BatchNorm-backward
    apply of the concatenation's `up` channels, space-to-depth data gradient and weight gradient of the transposed convolution, sums of the
    BatchNorm backward of the layer below -- against the float64 oracle on the bf16-rounded dup (the value both products consume on the device)."""
    n, h, w, cin, cout, aff, sums, linear = case
    td = torch.bfloat16
    rng = np.random.default_rng(sum(case[:5]) + 7)
    cs = cout                                      # skip channels in front of the `up` channels
    xr = rnd(rng, (n, h, w, cin), td) * 1.2 + 0.1
    kt = rnd(rng, (2, 2, cout, cin), td, 0.15)
    gcat = rnd(rng, (n, 2 * h, 2 * w, cs + cout), td)
    yup = torch.tensor(rnd(rng, (n, 2 * h, 2 * w, cout), td) * 1.3 + 0.2, dtype=torch.float32).to(td).double().numpy()
    ctot = cs + cout
    sc, sh = rng.standard_normal(ctot).astype(np.float32), rng.standard_normal(ctot).astype(np.float32) * 0.5
    mu, rs = rng.standard_normal(ctot).astype(np.float32) * 0.3, (0.5 + rng.random(ctot)).astype(np.float32)
    c1, c2 = rng.standard_normal(ctot).astype(np.float32) * 0.05, rng.standard_normal(ctot).astype(np.float32) * 0.05
    xsc, xsh = (rng.random(cin) + 0.5).astype(np.float32), (rng.standard_normal(cin) * 0.3).astype(np.float32)
    xmu, xrs = rng.standard_normal(cin).astype(np.float32) * 0.3, (0.5 + rng.random(cin)).astype(np.float32)
    if aff:
        # keep the input's pre-activations away from 0: the device forms them in fp32, the oracle in float64, and a ReLU mask that flips on a
        # pre-activation of ~1e-7 moves a fused sum by a whole gradient value
        for _ in range(3):
            pre = xr * xsc.astype(np.float64) + xsh.astype(np.float64)
            xr = np.where(np.abs(pre) < 0.03, xr + 0.25 / xsc.astype(np.float64), xr)
            xr = torch.tensor(xr, dtype=torch.float32).to(td).double().numpy()
    # oracle
    u = slice(cs, ctot)
    g_up = gcat[..., u]
    act = yup * sc[u].astype(np.float64) + sh[u].astype(np.float64)
    gm = g_up if linear else np.where(act > 0, g_up, 0.0)
    xhat = (yup - mu[u].astype(np.float64)) * rs[u].astype(np.float64)
    dup = sc[u].astype(np.float64) * (gm - c1[u].astype(np.float64) - xhat * c2[u].astype(np.float64))
    dup = torch.tensor(dup, dtype=torch.float32).to(td).double().numpy()          # the device rounds dup to bf16 before both products
    if aff:
        xa = np.maximum(xr * xsc.astype(np.float64) + xsh.astype(np.float64), 0)
        xa = torch.tensor(xa, dtype=torch.float32).to(td).double().numpy()
    else:
        xa = xr
    dx_ref, dk_ref, _ = K.conv2d_transpose_ks_bwd(xa, kt, dup)
    # device
    _, wd = ops.pack_weights(f32dev(kt), cin, ops.DTYPE_CODE[td], transposed=True)
    coef = torch.tensor(np.concatenate([c1, c2]), device=dev())
    stats = ops.new_stats(cin, dev()) if sums else None
    bst = dict(sums=stats, mean=f32dev(xmu), rstd=f32dev(xrs)) if sums else None
    dx, dk = ops.convt_bwd_fused(to_dev(gcat, td), cs, to_dev(yup, td), f32dev(sc), f32dev(sh), f32dev(mu), f32dev(rs), coef, to_dev(xr, td), wd, cin, cout,
                                 in_scale=f32dev(xsc) if aff else None, in_shift=f32dev(xsh) if aff else None, in_relu=aff, linear=linear, bst=bst)
    torch.cuda.synchronize()
    close(back(dx, cin), dx_ref, td, f'convT fused dx {case}')
    close(back(dk), dk_ref, td, f'convT fused dK {case}', k=0.5)
    if sums:
        g = back(dx, cin)
        v_mask = xa > 0
        gg = np.where(v_mask, g, 0.0)
        # xhat of the layer below from its raw output: (x - mean) * rstd
        xh = (xr - xmu.astype(np.float64)) * xrs.astype(np.float64)
        sgot = stats.sum(0).double().cpu().numpy()
        tol = 3e-3 * np.sqrt(n * h * w) * max(1.0, float(np.abs(g).max()))
        np.testing.assert_allclose(sgot[0], gg.sum((0, 1, 2)), rtol=2e-3, atol=tol, err_msg=f'sum g {case}')
        # the device rebuilds xhat from the bf16 activation: one more storage rounding per element
        np.testing.assert_allclose(sgot[1], (gg * xh).sum((0, 1, 2)), rtol=2e-2, atol=8 * tol * float(np.abs(xh).max()), err_msg=f'sum g xhat {case}')
    # unsupported shapes are refused by the workspace query (the engine then keeps the three launches)
    from satellite_computervision_amd._lib import lib
    import ctypes
    bad = ops.make_ctbf_desc(g=1 << 20, ldg=64, yup=1 << 20, ldy=48, bn_scale=None, bn_shift=None, bn_mean=None, bn_rstd=None, bn_c1=None, bn_c2=None, x=1 << 20, ldx=64,
                             w_dgrad=1 << 20, w_npad=64, dx=1 << 20, lddx=64, dw=1 << 20, cin=64, cout=48, n=1, h=8, w_=64, dtype=1)
    assert lib.satcv_convt_bwd_fused_workspace(ctypes.byref(bad)) < 0


# ------------------------------------------------------------------------ batch norm
@pytest.mark.parametrize('td', DT)
@pytest.mark.parametrize('f', [2, 3])
def test_bn_relu_pool_and_backward(ops, td, f):
    rng = np.random.default_rng(21 + f)
    n, h, w, c = 2, 12, 18, 32
    y = rnd(rng, (n, h, w, c), td) * 1.5 + 0.3
    if td == torch.bfloat16:
        y = torch.tensor(y, dtype=torch.float32).to(td).double().numpy()
    g, b = rng.standard_normal(c).astype(np.float32).astype(np.float64), rng.standard_normal(c).astype(np.float32).astype(np.float64)
    # training statistics from the (conv-epilogue style) sum / sumsq rows
    stats = ops.new_stats(c, dev())
    stats[3, 0] = f32dev(y.sum((0, 1, 2))); stats[7, 1] = f32dev((y ** 2).sum((0, 1, 2)))
    mm, mv = torch.zeros(c, device=dev()), torch.ones(c, device=dev())
    scale, shift, mean, rstd = ops.bn_finalize_train(stats, n * h * w, f32dev(g), f32dev(b), mm, mv, updates=2)
    z, m_ref, v_ref = K.batchnorm_train(y, g, b)
    np.testing.assert_allclose(back(mean), m_ref, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(back(rstd), 1 / np.sqrt(v_ref + 1e-3), rtol=1e-4)
    np.testing.assert_allclose(back(mm), m_ref * (1 - 0.99 ** 2), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(back(mv), 0.99 ** 2 + v_ref * (1 - 0.99 ** 2), rtol=1e-4)
    assert not stats.any()                                   # consumed and zeroed
    act_ref = K.relu(z)
    if td == torch.bfloat16:
        # the stored activation (and hence the pooling arg-max / ties) is the bf16-rounded value
        act_ref = torch.tensor(act_ref, dtype=torch.float32).to(td).double().numpy()
    st2 = ops.new_stats(c, dev())
    act, pooled = ops.bn_relu_pool(to_dev(y, td), scale, shift, f, stats=st2)
    close(back(act), act_ref, td, 'act')
    close(back(pooled), K.maxpool(act_ref, f), td, 'pooled')
    np.testing.assert_allclose(st2.sum(0)[0].double().cpu().numpy(), back(act).sum((0, 1, 2)), rtol=1e-4, atol=1e-2)
    # backward with gradient arriving both densely (skip) and through the pool
    da = rnd(rng, (n, h, w, c), td)
    dp = rnd(rng, (n, h // f, w // f, c), td)
    dz = K.relu_bwd(act_ref, da + K.maxpool_bwd(act_ref, f, dp))
    dy_ref, dg_ref, db_ref = K.batchnorm_train_bwd(y, g, m_ref, v_ref, dz)
    dy, dgamma, dbeta, dbias = ops.bn_relu_bwd(to_dev(y, td), scale, shift, mean, rstd, da=to_dev(da, td), dpool=to_dev(dp, td), f=f,
                                               want_dbias=True)
    close(back(dy), dy_ref, td, 'bn bwd dy', k=4.0)
    close(back(dgamma), dg_ref, td, 'dgamma', k=4.0)
    close(back(dbeta), db_ref, td, 'dbeta', k=4.0)
    np.testing.assert_allclose(back(dbias), back(dy).sum((0, 1, 2)), rtol=1e-3, atol=2e-2)
    # dense-only path
    dz = K.relu_bwd(act_ref, da)
    dy_ref, _, _ = K.batchnorm_train_bwd(y, g, m_ref, v_ref, dz)
    dy, _, _, _ = ops.bn_relu_bwd(to_dev(y, td), scale, shift, mean, rstd, da=to_dev(da, td))
    close(back(dy), dy_ref, td, 'bn bwd dense', k=4.0)
    # inference affine
    s2, h2 = ops.bn_affine_infer(f32dev(g), f32dev(b), f32dev(m_ref), f32dev(v_ref))
    np.testing.assert_allclose(back(s2), g / np.sqrt(v_ref + 1e-3), rtol=1e-5)
    np.testing.assert_allclose(back(h2), b - m_ref * g / np.sqrt(v_ref + 1e-3), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('td', DT)
@pytest.mark.parametrize('case', [(2, 12, 18, 32, 32), (1, 16, 16, 64, 32), (3, 5, 7, 16, 48)])
def test_bn_backward_of_a_concatenation_in_one_pass(ops, td, case):
    """BatchNormalization + ReLU of concat([skip, up]) (utils/model_tools.py:307-309): the dual-source dense backward against the oracle's
    backward of the materialised concatenation"""
    n, h, w, c0, c1 = case
    c = c0 + c1
    rng = np.random.default_rng(31 + c0)
    y = rnd(rng, (n, h, w, c), td) * 1.5 + 0.3
    if td == torch.bfloat16:
        y = torch.tensor(y, dtype=torch.float32).to(td).double().numpy()
    g, b = rng.standard_normal(c).astype(np.float32).astype(np.float64), rng.standard_normal(c).astype(np.float32).astype(np.float64)
    stats = ops.new_stats(c, dev())
    stats[0, 0] = f32dev(y.sum((0, 1, 2))); stats[1, 1] = f32dev((y ** 2).sum((0, 1, 2)))
    mm, mv = torch.zeros(c, device=dev()), torch.ones(c, device=dev())
    scale, shift, mean, rstd = ops.bn_finalize_train(stats, n * h * w, f32dev(g), f32dev(b), mm, mv)
    z, m_ref, v_ref = K.batchnorm_train(y, g, b)
    da = rnd(rng, (n, h, w, c), td)
    dy_ref, dg_ref, db_ref = K.batchnorm_train_bwd(y, g, m_ref, v_ref, K.relu_bwd(K.relu(z), da))
    dy0, dy1, dgamma, dbeta = ops.bn_relu_bwd_concat(to_dev(y[..., :c0], td), to_dev(y[..., c0:], td), scale, shift, mean, rstd, to_dev(da, td))
    close(back(dy0), dy_ref[..., :c0], td, 'concat bn bwd dy0', k=4.0)
    close(back(dy1), dy_ref[..., c0:], td, 'concat bn bwd dy1', k=4.0)
    close(back(dgamma), dg_ref, td, 'concat dgamma', k=4.0)
    close(back(dbeta), db_ref, td, 'concat dbeta', k=4.0)


@pytest.fixture
def general_kernel_only(ops):
    """thin shapes on the general kernel too: the test below compares a fused launch bit for bit with the plain one, and the fused form
    is not served by the persistent thin-layer kernel (a different summation order)"""
    from satellite_computervision_amd._lib import lib, check
    import ctypes
    old = ctypes.c_int32()
    check(lib.satcv_get_option(b'igemm_thin', ctypes.byref(old)))
    check(lib.satcv_set_option(b'igemm_thin', 0))
    yield
    check(lib.satcv_set_option(b'igemm_thin', old.value))


@pytest.mark.parametrize('case', [
    (2, 32, 32, 32, 32, 0, 3),       # thin 256 x 32 tile
    (2, 32, 64, 64, 64, 0, 3),       # 128 x 64
    (1, 32, 32, 128, 128, 0, 3),     # 128 x 128 tile (row vectors parked in LDS at once)
    (2, 16, 16, 256, 128, 0, 3),     # TW = 16
    (4, 16, 16, 512, 512, 0, 3),     # double-buffered 256 x 128 tile
    (2, 32, 32, 32, 64, 32, 3),      # gradient of a decoder concatenation: raw outputs in two tensors
    (2, 16, 16, 128, 128, 64, 3),
    (2, 32, 32, 64, 64, 0, 1),       # 1 x 1
])
def test_data_gradient_with_fused_bn_backward_sums(ops, case, general_kernel_only):
    """The data gradient of a conv whose input is ReLU(BN(v)) writes dL/d act; its epilogue also forms sum g and sum g * xhat of that
    BatchNorm's backward (include/satcv.h: bst_*; utils/model_tools.py:178-180 differentiated).  Checked against float64 sums of the
    STORED gradient, and against the separate reduce launch it replaces."""
    n, h, w, cin, cout, split, k = case
    td = torch.bfloat16
    rng = np.random.default_rng(hash(case) % 2**31)
    x = rnd(rng, (n, h, w, cin), td)
    kern = rnd(rng, (k, k, cin, cout), td, 0.2)
    v = rnd(rng, (n, h, w, cout), td) * 1.5 + 0.25
    v = torch.tensor(v, dtype=torch.float32).to(td).double().numpy()
    sc, sh = rng.standard_normal(cout).astype(np.float32), rng.standard_normal(cout).astype(np.float32) * 0.5
    mu, rs = rng.standard_normal(cout).astype(np.float32) * 0.3, (0.5 + rng.random(cout)).astype(np.float32)
    wf, _ = ops.pack_weights(f32dev(kern), cin, ops.DTYPE_CODE[td])
    v0 = to_dev(v[..., :split] if split else v, td)
    v1 = to_dev(v[..., split:], td) if split else None
    for relu in (1, 0):
        stats = ops.new_stats(cout, dev())
        bst = dict(y=v0, ld=v0.shape[-1], scale=f32dev(sc), shift=f32dev(sh), mean=f32dev(mu), rstd=f32dev(rs), relu=relu)
        if split:
            bst.update(y1=v1, ld1=v1.shape[-1], split=split)
        y = ops.conv2d(to_dev(x, td), wf, cout, kh=k, kw=k, stats=stats, bst=bst)
        plain = ops.conv2d(to_dev(x, td), wf, cout, kh=k, kw=k)
        assert torch.equal(y, plain), 'the fused sums must not change the stored gradient'
        g = back(y, cout)
        mask = (v * sc.astype(np.float64) + sh.astype(np.float64) > 0) if relu else np.ones_like(v, bool)
        gg = np.where(mask, g, 0.0)
        xh = (v - mu.astype(np.float64)) * rs.astype(np.float64)
        s1_ref, s2_ref = gg.sum((0, 1, 2)), (gg * xh).sum((0, 1, 2))
        got = stats.sum(0).double().cpu().numpy()
        tol = 2e-4 * np.sqrt(n * h * w) * max(1.0, float(np.abs(g).max()))
        np.testing.assert_allclose(got[0], s1_ref, rtol=1e-4, atol=tol, err_msg=f'sum g {case} relu={relu}')
        np.testing.assert_allclose(got[1], s2_ref, rtol=1e-4, atol=tol * float(np.abs(xh).max()), err_msg=f'sum g xhat {case} relu={relu}')


@pytest.mark.parametrize('case', [
    # n, h, w, c0, c1, cout, input affine + ReLU, linear BN
    (2, 16, 64, 32, 0, 32, True, False),
    (1, 8, 32, 32, 0, 32, False, False),       # one tile, plain input
    (3, 24, 96, 32, 32, 32, True, False),      # concat([skip, up]) input (decoder_block's first conv)
    (2, 16, 32, 32, 0, 32, True, True),        # BatchNormalization without ReLU
    (5, 40, 64, 64, 0, 32, True, False),       # more tiles than resident workgroups' first round (ragged tile ranges)
    (2, 16, 64, 64, 0, 64, True, False),       # 64 -> 64 (one wave per SIMD, 36 weight-gradient accumulator tiles)
    (1, 24, 32, 32, 32, 64, False, False),
])
def test_fused_thin_layer_backward(ops, case):
    """satcv_conv2d_bwd_fused: dy = scale * (g * mask - c1 - xhat * c2) formed in registers, data gradient and weight gradient from
    the one dy tile (Conv2D :178 + BatchNormalization :179 + Activation :180 of utils/model_tools.py differentiated).  Against the
    float64 oracle fed with the bf16-rounded dy, and against the three launches it replaces."""
    n, h, w, c0, c1, cout, affine, linear = case
    cin = c0 + c1
    td = torch.bfloat16
    rng = np.random.default_rng(abs(hash(case)) % 2**31)
    xr = rnd(rng, (n, h, w, cin), td)
    g = rnd(rng, (n, h, w, cout), td)
    v = rnd(rng, (n, h, w, cout), td) * 1.5 + 0.25
    v = torch.tensor(v, dtype=torch.float32).to(td).double().numpy()
    kern = rnd(rng, (3, 3, cin, cout), td, 0.2)
    sc, sh = (0.5 + rng.random(cout)).astype(np.float32) * rng.choice([-1, 1], cout), rng.standard_normal(cout).astype(np.float32) * 0.5
    mu, rs = rng.standard_normal(cout).astype(np.float32) * 0.3, (0.5 + rng.random(cout)).astype(np.float32)
    coef = (rng.standard_normal((2, cout)) * 0.1).astype(np.float32)
    isc, ish = (0.5 + rng.random(cin)).astype(np.float32), (rng.standard_normal(cin) * 0.3).astype(np.float32)
    # reference: the activated input, dy rounded to bf16 (what the three-launch path stores), then the float64 conv backward
    a = np.maximum(xr * isc.astype(np.float64) + ish.astype(np.float64), 0) if affine else xr
    a = torch.tensor(a, dtype=torch.float32).to(td).double().numpy()          # the loader rounds the activated input to bf16
    f64 = lambda t: t.astype(np.float64)
    mask = np.ones_like(v, bool) if linear else (v * f64(sc) + f64(sh) > 0)
    xh = (v - f64(mu)) * f64(rs)
    dy = f64(sc) * (np.where(mask, g, 0.0) - f64(coef[0]) - xh * f64(coef[1]))
    dy = torch.tensor(dy, dtype=torch.float32).to(td).double().numpy()
    dx_ref, dk_ref, _ = K.conv2d_same_bwd(a, kern, dy, 1)
    _, wd = ops.pack_weights(f32dev(kern), cin, ops.DTYPE_CODE[td])
    x0 = to_dev(xr[..., :c0], td)
    x1 = to_dev(xr[..., c0:], td) if c1 else None
    out = ops.conv_bwd_fused(to_dev(g, td), to_dev(v, td), f32dev(sc), f32dev(sh), f32dev(mu), f32dev(rs), f32dev(coef), x0, wd, cin, cout, x1=x1,
                             in_scale=f32dev(isc) if affine else None, in_shift=f32dev(ish) if affine else None, in_relu=affine, linear=linear)
    assert out is not None, 'shape must be served by the fused kernel'
    dx, dw = out
    if affine and not (cin == 64 and cout == 64):
        # bst_*: the sums of the BatchNorm backward of the layer below (whose scale / shift are the input affine), from the stored dx
        bmu, brs = (rng.standard_normal(cin) * 0.3).astype(np.float32), (0.5 + rng.random(cin)).astype(np.float32)
        stats = ops.new_stats(cin, dev())
        out2 = ops.conv_bwd_fused(to_dev(g, td), to_dev(v, td), f32dev(sc), f32dev(sh), f32dev(mu), f32dev(rs), f32dev(coef), x0, wd, cin, cout, x1=x1,
                                  in_scale=f32dev(isc), in_shift=f32dev(ish), in_relu=True, linear=linear,
                                  bst=dict(sums=stats, mean=f32dev(bmu), rstd=f32dev(brs)))
        assert torch.equal(out2[0], dx), 'the fused sums must not change the stored gradient'
        gst = back(dx)
        gm = np.where(xr * isc.astype(np.float64) + ish.astype(np.float64) > 0, gst, 0.0)
        xh = (xr - bmu.astype(np.float64)) * brs.astype(np.float64)
        s1_ref, s2_ref = gm.sum((0, 1, 2)), (gm * xh).sum((0, 1, 2))
        got = stats.sum(0).double().cpu().numpy()
        tol = 2e-4 * np.sqrt(n * h * w) * max(1.0, float(np.abs(gst).max()))
        np.testing.assert_allclose(got[0], s1_ref, rtol=1e-4, atol=tol, err_msg=f'fused sum g {case}')
        # (xhat is rebuilt from the staged bf16 activation: one storage rounding per element more than the separate pass, unbiased)
        np.testing.assert_allclose(got[1], s2_ref, rtol=2e-3, atol=8 * tol * float(np.abs(xh).max()), err_msg=f'fused sum g xhat {case}')
    close(back(dx), dx_ref, td, f'fused dx {case}', k=1.5)          # (dy differs from the reference's by at most one bf16 rounding per element)
    close(back(dw), dk_ref, td, f'fused dw {case}', k=1.5)
    # accumulate flag (shared weights): dw += result
    dw2 = dw.clone()
    ops.conv_bwd_fused(to_dev(g, td), to_dev(v, td), f32dev(sc), f32dev(sh), f32dev(mu), f32dev(rs), f32dev(coef), x0, wd, cin, cout, x1=x1,
                       in_scale=f32dev(isc) if affine else None, in_shift=f32dev(ish) if affine else None, in_relu=affine, linear=linear,
                       accumulate_into=dw2)
    np.testing.assert_allclose(back(dw2), 2 * back(dw), rtol=1e-5, atol=1e-6)
    # shapes outside the limits are refused, not mis-served
    assert ops.conv_bwd_fused(to_dev(g[:, :6], td), to_dev(v[:, :6], td), f32dev(sc), f32dev(sh), f32dev(mu), f32dev(rs), f32dev(coef),
                              to_dev(xr[:, :6, :, :c0], td), wd, cin, cout, x1=to_dev(xr[:, :6, :, c0:], td) if c1 else None) is None


@pytest.mark.parametrize('case', [(2, 16, 64, True), (3, 24, 32, False)])
def test_fused_backward_forms_the_head_gradient_in_its_loader(ops, case):
    """The block under the 1 x 1 head (utils/model_tools.py:405): with hg_dlogits / hg_w the fused launch forms g = bf16(dlogits w^T) itself
    instead of reading the tensor satcv_head_bwd would have stored; same gradients as the launch fed with that tensor."""
    n, h, w, affine = case
    td = torch.bfloat16
    cin = cout = 32
    rng = np.random.default_rng(n * 100 + h)
    x = to_dev(rnd(rng, (n, h, w, cin), td), td)
    v = to_dev(rnd(rng, (n, h, w, cout), td) * 1.5 + 0.25, td)
    kern = rnd(rng, (3, 3, cin, cout), td, 0.2)
    _, wd = ops.pack_weights(f32dev(kern), cin, ops.DTYPE_CODE[td])
    dl = f32dev(rng.standard_normal((n * h * w, 2)) * 0.1)
    wh = f32dev(rng.standard_normal((cout, 2)))
    g = (dl.double() @ wh.double().t()).float().to(td).reshape(n, h, w, cout).contiguous()      # what satcv_head_bwd stores (up to the last fp32 bit before rounding)
    one = lambda c, s_=1.0: f32dev((rng.random(c) + 0.5) * s_)
    sc, sh, mu, rs = one(cout), one(cout, 0.1), one(cout, 0.1), one(cout)
    coef = f32dev(rng.standard_normal((2, cout)) * 0.1)
    isc, ish = (one(cin), one(cin, 0.1)) if affine else (None, None)
    ref = ops.conv_bwd_fused(g, v, sc, sh, mu, rs, coef, x, wd, cin, cout, in_scale=isc, in_shift=ish, in_relu=affine)
    got = ops.conv_bwd_fused(None, v, sc, sh, mu, rs, coef, x, wd, cin, cout, in_scale=isc, in_shift=ish, in_relu=affine, head=(dl, wh))
    assert ref is not None and got is not None
    # (a product sum that lands within an fp32 ulp of a bf16 rounding boundary may round the other way: a handful of elements of g differ by
    #  one bf16 ulp, nothing else does)
    assert float((got[0].float() - ref[0].float()).abs().max()) <= 2e-2 * float(ref[0].float().abs().max())
    assert float((got[0].float() - ref[0].float()).abs().mean()) <= 1e-4 * float(ref[0].float().abs().mean())
    assert float((got[1] - ref[1]).abs().max()) <= 2e-3 * float(ref[1].abs().max())


@pytest.mark.parametrize('case', [
    # n, h, w, stored cin, real cin, cout, data gradient
    (2, 16, 64, 32, 32, 64, True),          # encoder_block 2 of get_unet_model (32 -> 64 channels)
    (3, 24, 32, 16, 4, 32, False),          # the first block: 4 bands stored as 16 channels -> 32, fed by the model input (no dx)
])
def test_fused_pooled_encoder_backward(ops, case):
    """The pooled form of satcv_conv2d_bwd_fused (encoder_block, utils/model_tools.py:262-286: the block's output is a skip AND is
    max-pooled): g = da + MaxPooling2D's gradient routed by the arg-max bytes of satcv_bn_relu_pool_amax.  Against the float64 oracle
    (first maximum in row-major window order) and the arg-max bytes themselves against NumPy."""
    n, h, w, cs, cin, cout, want_dx = case
    td = torch.bfloat16
    rng = np.random.default_rng(abs(hash(case)) % 2**31)
    xr = rnd(rng, (n, h, w, cs), td)
    xr[..., cin:] = 0
    da = rnd(rng, (n, h, w, cout), td)
    dp = rnd(rng, (n, h // 2, w // 2, cout), td)
    v = rnd(rng, (n, h, w, cout), td) * 1.5 + 0.25
    v = torch.tensor(v, dtype=torch.float32).to(td).double().numpy()
    kern = rnd(rng, (3, 3, cin, cout), td, 0.2)
    sc, sh = (0.5 + rng.random(cout)).astype(np.float32) * rng.choice([-1, 1], cout), rng.standard_normal(cout).astype(np.float32) * 0.5
    mu, rs = rng.standard_normal(cout).astype(np.float32) * 0.3, (0.5 + rng.random(cout)).astype(np.float32)
    coef = (rng.standard_normal((2, cout)) * 0.1).astype(np.float32)
    f64 = lambda t: t.astype(np.float64)
    # forward: activation (rounded to bf16 as stored), pooled maximum and its first position
    act_d, pooled_d, amax_d = ops.bn_relu_pool_amax(to_dev(v, td), f32dev(sc), f32dev(sh), 2)
    act = torch.tensor(np.maximum(v * f64(sc) + f64(sh), 0), dtype=torch.float32)
    act = (torch.tensor(v, dtype=torch.float32) * torch.tensor(sc) + torch.tensor(sh)).clamp_min(0).to(td).double().numpy()     # fp32 affine, as the kernel
    win = act.reshape(n, h // 2, 2, w // 2, 2, cout).transpose(0, 1, 3, 5, 2, 4).reshape(n, h // 2, w // 2, cout, 4)
    am_ref = win.argmax(-1)                                                  # first maximum
    np.testing.assert_array_equal(back(act_d), act)
    np.testing.assert_array_equal(amax_d.cpu().numpy(), am_ref)
    np.testing.assert_array_equal(back(pooled_d), win.max(-1))
    # backward reference
    route = np.zeros((n, h // 2, w // 2, cout, 4))
    np.put_along_axis(route, am_ref[..., None], dp[..., None], -1)
    g = da + route.reshape(n, h // 2, w // 2, cout, 2, 2).transpose(0, 1, 4, 2, 5, 3).reshape(n, h, w, cout)
    mask = v * f64(sc) + f64(sh) > 0
    dy = f64(sc) * (np.where(mask, g, 0.0) - f64(coef[0]) - (v - f64(mu)) * f64(rs) * f64(coef[1]))
    dy = torch.tensor(dy, dtype=torch.float32).to(td).double().numpy()
    dx_ref, dk_ref, _ = K.conv2d_same_bwd(xr[..., :cin], kern, dy, 1)
    _, wd = ops.pack_weights(f32dev(kern), cs, ops.DTYPE_CODE[td])
    out = ops.conv_bwd_fused(to_dev(da, td), to_dev(v, td), f32dev(sc), f32dev(sh), f32dev(mu), f32dev(rs), f32dev(coef), to_dev(xr, td), wd, cin, cout,
                             dpool=to_dev(dp, td), amax=amax_d, want_dx=want_dx)
    assert out is not None, 'shape must be served by the pooled fused kernel'
    dx, dw = out
    close(back(dw), dk_ref, td, f'pooled fused dw {case}', k=1.5)
    if want_dx:
        close(back(dx)[..., :cin], dx_ref, td, f'pooled fused dx {case}', k=1.5)
        # the pooled part of the sums of the block below, in the activated form (its pooled output is this block's input x): sum dx [x > 0], sum dx x
        stats = ops.new_stats(cs, dev())
        xa = np.abs(xr)                                                       # an activation: non-negative, with exact zeros
        xa[xa < 0.3] = 0
        out2 = ops.conv_bwd_fused(to_dev(da, td), to_dev(v, td), f32dev(sc), f32dev(sh), f32dev(mu), f32dev(rs), f32dev(coef), to_dev(xa, td), wd, cin, cout,
                                  dpool=to_dev(dp, td), amax=amax_d, bst=dict(sums=stats, act_form=1))
        assert out2 is not None
        gst = back(out2[0])
        got = stats.sum(0).double().cpu().numpy()
        tol = 2e-4 * np.sqrt(n * h * w) * max(1.0, float(np.abs(gst).max()))
        np.testing.assert_allclose(got[0], np.where(xa > 0, gst, 0.0).sum((0, 1, 2)), rtol=1e-4, atol=tol)
        np.testing.assert_allclose(got[1], (gst * xa).sum((0, 1, 2)), rtol=1e-4, atol=tol * float(xa.max()))
    else:
        assert dx is None


def test_fused_bn_backward_sums_are_refused_on_partial_tiles(ops):
    """a map that is not a whole number of tiles keeps the separate reduce launch: the query says so and conv2d raises"""
    td = torch.bfloat16
    rng = np.random.default_rng(3)
    n, h, w, c = 1, 12, 20, 32
    wf, _ = ops.pack_weights(f32dev(rnd(rng, (3, 3, c, c), td, 0.2)), c, ops.DTYPE_CODE[td])
    v = to_dev(rnd(rng, (n, h, w, c), td), td)
    one = torch.ones(c, device=dev())
    with pytest.raises(ValueError):
        ops.conv2d(to_dev(rnd(rng, (n, h, w, c), td), td), wf, c, stats=ops.new_stats(c, dev()),
                   bst=dict(y=v, ld=c, scale=one, shift=one, mean=one, rstd=one, relu=1))


# ------------------------------------------------------------------- head and losses
@pytest.mark.parametrize('td', DT)
@pytest.mark.parametrize('ncls,activation', [(2, 'softmax'), (5, 'softmax'), (1, 'sigmoid')])
def test_head_and_losses(ops, td, ncls, activation):
    rng = np.random.default_rng(31 + ncls)
    n, h, w, c = 2, 16, 24, 32
    yraw = rnd(rng, (n, h, w, c), td)
    sc, sh = rng.standard_normal(c).astype(np.float32), rng.standard_normal(c).astype(np.float32)
    wh = rng.standard_normal((c, ncls)).astype(np.float32) * 0.3
    bh = rng.standard_normal(ncls).astype(np.float32)
    a = np.maximum(yraw * sc.astype(np.float64) + sh.astype(np.float64), 0)
    logits = a @ wh.astype(np.float64) + bh
    p_ref = K.softmax(logits) if activation == 'softmax' else K.sigmoid(logits)
    probs, classes = ops.head_fwd(to_dev(yraw, td), f32dev(wh), f32dev(bh), activation, f32dev(sc), f32dev(sh))
    np.testing.assert_allclose(back(probs), p_ref, rtol=2e-4, atol=2e-6)
    if activation == 'softmax':
        margin = np.sort(p_ref, -1)[..., -1] - np.sort(p_ref, -1)[..., -2]
        ok = margin > 1e-4
        assert np.array_equal(classes.cpu().numpy()[ok], K.argmax_classes(p_ref)[ok])
        assert classes.dtype == torch.int32
        lab = rng.integers(0, ncls, (n, h, w)); t = np.eye(ncls)[lab]
        wts = (1.0 + np.arange(ncls) * 3.0)
        pg = back(probs)
        l_ref, g_ref, _ = OL.weighted_categorical_crossentropy(t, pg, wts)
        dl_ref = K.softmax_bwd(pg, g_ref)
        loss, dl = ops.loss_fwd_bwd('weighted_categorical_crossentropy', probs, f32dev(t), f32dev(wts))
        np.testing.assert_allclose(loss.item(), l_ref, rtol=1e-4)
        np.testing.assert_allclose(back(dl), dl_ref, rtol=2e-3, atol=1e-9)
        l_ref, g_ref = OL.weighted_bce(t, pg, 5.0)
        loss, dl2 = ops.loss_fwd_bwd('weighted_bce', probs, f32dev(t), f32dev(np.array([5.0])))
        np.testing.assert_allclose(loss.item(), l_ref, rtol=1e-4)
        np.testing.assert_allclose(back(dl2), K.softmax_bwd(pg, g_ref), rtol=2e-3, atol=1e-9)
        conf = ops.confusion(classes, f32dev(t), ncls).cpu().numpy()
        cref = np.zeros((ncls, ncls), np.int64)
        np.add.at(cref, (lab.ravel(), classes.cpu().numpy().ravel()), 1)
        assert np.array_equal(conf, cref)
    else:
        assert np.array_equal(classes.cpu().numpy(), (back(probs) > 0.5).astype(np.int32))
        t = (rng.random((n, h, w, 1)) < 0.3).astype(np.float64)
        pg = back(probs)
        l_ref, g_ref = OL.weighted_bce(t, pg, 5.0)
        loss, dl = ops.loss_fwd_bwd('weighted_bce', probs, f32dev(t), f32dev(np.array([5.0])), activation='sigmoid')
        np.testing.assert_allclose(loss.item(), l_ref, rtol=1e-4)
        np.testing.assert_allclose(back(dl), g_ref * pg * (1 - pg), rtol=2e-3, atol=1e-9)
    # head backward
    dlg = rng.standard_normal((n, h, w, ncls)).astype(np.float32)
    dx, dw, db = ops.head_bwd(to_dev(yraw, td), f32dev(wh), f32dev(dlg), f32dev(sc), f32dev(sh))
    dl64 = dlg.astype(np.float64)
    close(back(dx), dl64 @ wh.astype(np.float64).T, td, 'head dx')
    np.testing.assert_allclose(back(dw), a.reshape(-1, c).T @ dl64.reshape(-1, ncls), rtol=2e-3, atol=2e-2)
    np.testing.assert_allclose(back(db), dl64.reshape(-1, ncls).sum(0), rtol=1e-3, atol=1e-3)


def test_adam_keras_formulation(ops):
    rng = np.random.default_rng(41)
    nel = 1003
    p, g = rng.standard_normal(nel), rng.standard_normal(nel)
    pd, m, v = f32dev(p), torch.zeros(nel, device=dev()), torch.zeros(nel, device=dev())
    state = torch.tensor([9e-4, 0.0, 1.0, 0.0], device=dev())
    pr, mr, vr = p.copy(), np.zeros(nel), np.zeros(nel)
    for t in range(1, 4):
        gd = f32dev(g * t)
        ops.adam_step(pd, gd, m, v, state)
        gr = g.astype(np.float32).astype(np.float64) * t
        mr = 0.9 * mr + 0.1 * gr; vr = 0.999 * vr + 0.001 * gr * gr
        alpha = 9e-4 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        pr = pr - alpha * mr / (np.sqrt(vr) + 1e-7)
    np.testing.assert_allclose(back(pd), pr, rtol=1e-5, atol=1e-6)
    assert state[1].item() == 3.0


@pytest.mark.parametrize('ncls', [2, 4])
def test_ratio_losses_dice_iou_mse(ops, ncls):
    """gen_dice (global and per-image weights), iou_loss, mse_4d: loss and dL/dlogits vs the oracle."""
    rng = np.random.default_rng(51 + ncls)
    n, h, w = 3, 12, 10
    logits = rng.standard_normal((n, h, w, ncls))
    p = K.softmax(logits).astype(np.float32).astype(np.float64)
    lab = rng.integers(0, ncls, (n, h, w)); lab[1][lab[1] == ncls - 1] = 0      # a class absent from one image
    t = np.eye(ncls)[lab]
    pd, td = f32dev(p), f32dev(t)
    for gw in ([0.3, 0.5] + [0.1] * (ncls - 2), None):
        l_ref, g_ref = OL.gen_dice(t, p, global_weights=gw)
        loss, dl = ops.loss_fwd_bwd('gen_dice', pd, td, f32dev(np.array(gw)) if gw else None)
        np.testing.assert_allclose(loss.item(), l_ref, rtol=2e-4)
        ref = K.softmax_bwd(p, g_ref)
        np.testing.assert_allclose(back(dl), ref, rtol=5e-3, atol=1e-6 * np.abs(ref).max())
    l_ref, g_ref = OL.iou_loss(t, p)
    loss, dl = ops.loss_fwd_bwd('iou_loss', pd, td, None)
    np.testing.assert_allclose(loss.item(), l_ref, rtol=2e-4)
    ref = K.softmax_bwd(p, g_ref)
    np.testing.assert_allclose(back(dl), ref, rtol=5e-3, atol=1e-6 * np.abs(ref).max())
    tn = rng.random((n, h, w, ncls)); tn[0, 0, 0, 0] = np.nan
    l_ref, g_ref = OL.mse_4d(tn, p)
    loss, dl = ops.loss_fwd_bwd('mse_4d', pd, f32dev(tn), None)
    np.testing.assert_allclose(loss.item(), l_ref, rtol=2e-4)
    ref = K.softmax_bwd(p, g_ref)
    np.testing.assert_allclose(back(dl), ref, rtol=5e-3, atol=1e-6 * np.abs(ref).max())


@pytest.mark.parametrize('td', DT)
@pytest.mark.parametrize('case', [(2, 24, 24, 32, 64, 3, 3), (2, 24, 24, 32, 64, 3, 6), (2, 24, 24, 32, 64, 3, 12), (2, 24, 24, 256, 64, 1, 1),
                                  (1, 32, 32, 128, 256, 3, 12), (2, 24, 24, 32, 64, 1, 1)])
def test_aspp_shapes_backward_and_accumulate(ops, td, case):
    """the ASPP convolutions (1x1 and 3x3 with rates 3/6/12) on a non-power-of-two map: dgrad, wgrad (incl. the
    per-tap fallback for halo tiles that exceed the LDS) and the accumulate-into-output form of dgrad."""
    n, h, w, cin, cout, k, dil = case
    rng = np.random.default_rng(hash(case) % 2**31)
    x = rnd(rng, (n, h, w, cin), td)
    kern = rnd(rng, (k, k, cin, cout), td, 0.2)
    dy = rnd(rng, (n, h, w, cout), td)
    y_ref = K.conv2d_same(x, kern, None, dil)
    dx_ref, dk_ref, _ = K.conv2d_same_bwd(x, kern, dy, dil)
    wf, wd = ops.pack_weights(f32dev(kern), cin, ops.DTYPE_CODE[td])
    y = ops.conv2d(to_dev(x, td), wf, cout, kh=k, kw=k, dil=dil)
    close(back(y, cout), y_ref, td, f'fwd {case}')
    dx = ops.conv2d_dgrad(to_dev(dy, td), wd, cin, kh=k, kw=k, dil=dil)
    close(back(dx, cin), dx_ref, td, f'dgrad {case}')
    base = rnd(rng, (n, h, w, cin), td)
    acc = to_dev(base, td)
    ops.conv2d_dgrad(to_dev(dy, td), wd, cin, kh=k, kw=k, dil=dil, out=acc, accumulate=True)
    close(back(acc, cin), dx_ref + base, td, f'dgrad accumulate {case}', k=2.0)
    dk = ops.conv2d_wgrad(to_dev(x, td), to_dev(dy, td), cin, cout, kh=k, kw=k, dil=dil)
    close(back(dk), dk_ref, td, f'wgrad {case}', k=(5.0 if td == torch.float32 else 0.5))


def test_dilated_convs_run_on_the_pipelined_kernel(ops):
    """ASPP rates on a 32x32 map: the pipelined kernel takes them (tap-loop form) instead of the generic fallback."""
    import ctypes as C
    from satellite_computervision_amd._lib import lib, BF16
    x = torch.zeros(2, 32, 32, 128, dtype=torch.bfloat16, device='cuda'); y = torch.zeros(2, 32, 32, 256, dtype=torch.bfloat16, device='cuda')
    w = torch.zeros(9 * 128 * 256, dtype=torch.bfloat16, device='cuda')
    for dil in (2, 3, 6, 12):
        d = ops.make_conv_desc(x0=x.data_ptr(), c0=128, w=w.data_ptr(), y=y.data_ptr(), ldy=256, n=2, h=32, w_=32, cout=256, cout_pad=256, kh=3, kw=3,
                               dil=dil, dtype=BF16)
        assert lib.satcv_conv2d_igemm_pipelined(C.byref(d)) == 1, dil


@pytest.mark.parametrize('td', DT)
@pytest.mark.parametrize('case', [(2, 64, 64, 4, 64, 7, 2, 1), (2, 33, 47, 64, 64, 3, 2, 1), (3, 32, 32, 64, 128, 1, 2, 1), (1, 31, 31, 32, 32, 3, 2, 2),
                                  (2, 24, 40, 16, 32, 5, 1, 1)])
def test_strided_and_large_kernel_convs(ops, td, case):
    """ResNet-style convolutions of the build-defined DeepLab backbone: 7x7 / stride-2 stem, 3x3 and 1x1 stride-2 transitions, odd
    input sizes (symmetric padding dil*(k-1)/2, output (h-1)//stride+1) -- the tap-loop form of the pipelined kernel -- against
    torch.nn.functional.conv2d in float64, with the fused input affine + ReLU and the epilogue statistics."""
    import ctypes as C
    from satellite_computervision_amd._lib import lib
    n, h, w, cin, cout, k, stride, dil = case
    rng = np.random.default_rng(hash(case) % 2**31)
    x = rnd(rng, (n, h, w, cin), td)
    kern = rnd(rng, (k, k, cin, cout), td, 0.2)
    b = rng.standard_normal(cout)
    sc, sh = 1 + 0.2 * rng.standard_normal(cin), 0.2 * rng.standard_normal(cin)
    act = np.maximum(x * sc + sh, 0)
    pad = dil * (k - 1) // 2
    ref = torch.nn.functional.conv2d(torch.tensor(act).permute(0, 3, 1, 2), torch.tensor(kern).permute(3, 2, 0, 1), torch.tensor(b), stride=stride,
                                     padding=pad, dilation=dil).permute(0, 2, 3, 1).numpy()
    cpad = rup(cin, 16)
    wf, _ = ops.pack_weights(f32dev(kern), cpad, ops.DTYPE_CODE[td], want_dgrad=False)
    scp, shp = np.zeros(cpad), np.zeros(cpad)
    scp[:cin], shp[:cin] = sc, sh
    stats = ops.new_stats(rup(cout, 16), dev())
    xd = to_dev(x, td, cpad)
    y = ops.conv2d(xd, wf, cout, kh=k, kw=k, dil=dil, bias=f32dev(b), stats=stats, stride=stride, in_scale=f32dev(scp), in_shift=f32dev(shp), in_relu=True)
    torch.cuda.synchronize()
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    assert tuple(y.shape[:3]) == (n, ho, wo) and ref.shape[1:3] == (ho, wo)
    got = back(y, cout)
    close(got, ref, td, f'strided conv {case}', k=2.0)
    s = stats.sum(0).double().cpu().numpy()
    np.testing.assert_allclose(s[0, :cout], got.sum((0, 1, 2)), rtol=2e-4, atol=2e-3 * np.sqrt(n * ho * wo))
    d = ops.make_conv_desc(x0=xd.data_ptr(), c0=cpad, w=wf.data_ptr(), y=y.data_ptr(), ldy=y.shape[-1], n=n, h=ho, w_=wo, cout=cout, cout_pad=rup(cout, 32),
                           kh=k, kw=k, dil=dil, dtype=ops.DTYPE_CODE[td], stride=stride, hin=h if stride > 1 else 0, win=w if stride > 1 else 0)
    assert lib.satcv_conv2d_igemm_pipelined(C.byref(d)) == 1


# ------------------------------------------------- split-K of under-filled launches (opt-in: SATCV_SPLITK=1 read at first use)
@pytest.mark.parametrize('case', [(2, 8, 8, 512, 1024, 3), (4, 8, 8, 1024, 512, 3), (1, 32, 32, 512, 512, 3), (1, 32, 32, 1024, 256, 1), (3, 8, 8, 256, 128, 3)])
def test_conv2d_split_k(case):
    """igemm_fast_kernel<..., SK = true>: 2 - 4 workgroups share an output tile, each sums a contiguous range of the K chunks into an fp32
    slab, the finish kernel adds the slabs in order, applies bias and forms the BatchNorm statistics of the stored values.  Run in a
    subprocess (the switch is read once per process); compared against the float64 oracle and against the single-pass launch."""
    import subprocess, sys, os, json
    code = r"""
import sys, json, numpy as np, torch
sys.path.insert(0, %r)
from satellite_computervision_amd import ops
from oracle import keras_ops as K
n, h, w, cin, cout, k = %r
rng = np.random.default_rng(1)
def rnd(shape, s=1.0): return torch.tensor(rng.standard_normal(shape) * s, dtype=torch.float32).bfloat16()
x, kern, b = rnd((n, h, w, cin)), rnd((k, k, cin, cout), 0.2), torch.tensor(rng.standard_normal(cout), dtype=torch.float32).cuda()
wf, _ = ops.pack_weights(kern.float().cuda(), cin, 1)
st = ops.new_stats(cout, torch.device('cuda'))
y = ops.conv2d(x.cuda(), wf, cout, kh=k, kw=k, bias=b, stats=st)
ref = K.conv2d_same(x.double().numpy(), kern.double().numpy(), b.double().cpu().numpy(), 1)
got = y.double().cpu().numpy()
s = st.sum(0).double().cpu().numpy()
print(json.dumps(dict(err=float(np.abs(got - ref).max() / np.abs(ref).max()), s1=float(np.abs(s[0] - got.reshape(-1, cout).sum(0)).max() / max(np.abs(s[0]).max(), 1.0)),
                      s2=float(np.abs(s[1] - (got.reshape(-1, cout) ** 2).sum(0)).max() / np.abs(s[1]).max()), y=got.ravel()[:4096:7].tolist())))
""" % (ROOT, case)
    outs = {}
    for sk in ('1', '0'):
        r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, SATCV_SPLITK=sk), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[sk] = json.loads(r.stdout.strip().splitlines()[-1])
    assert outs['1']['err'] < 1.2e-2 and outs['1']['s1'] < 1e-3 and outs['1']['s2'] < 1e-3, outs['1']
    # the split sum differs from the single-pass one by fp32 association only: at most one bf16 ulp on a few elements
    a, b_ = np.array(outs['1']['y']), np.array(outs['0']['y'])
    assert np.abs(a - b_).max() <= 2 ** -7 * max(np.abs(b_).max(), 1.0)


@pytest.mark.parametrize('td', DT)
@pytest.mark.parametrize('case', [(2, 32, 32, 128, 256, 1), (1, 17, 23, 64, 64, 1), (2, 32, 32, 64, 128, 3)])
def test_conv2d_residual_join_in_place(ops, td, case):
    """satcv_conv_desc.accumulate = 2 (round 4): y = ReLU(y + out_scale * conv(x) + bias) written in place over the shortcut, the
    conv result rounded to the storage type before the addition -- the residual join of a ResNet bottleneck whose last convolution
    applies its own inference BatchNormalization (bias = scale * b + shift from satcv_bn_affine_infer_batched, checked here too)."""
    import ctypes as C
    from satellite_computervision_amd._lib import lib, check
    n, h, w, cin, cout, k = case
    rng = np.random.default_rng(hash(case) % 2**31)
    x = rnd(rng, (n, h, w, cin), td)
    kern = rnd(rng, (k, k, cin, cout), td, 0.1)
    short = rnd(rng, (n, h, w, cout), td)
    gamma, beta = rng.uniform(0.5, 1.5, cout), rng.normal(0, 0.3, cout)
    mm, mv, b = rng.normal(0, 0.2, cout), rng.uniform(0.5, 2.0, cout), rng.normal(0, 0.2, cout)
    g_, b_, mm_, mv_, cb_ = (f32dev(v) for v in (gamma, beta, mm, mv, b))
    scale, shift, beff = (torch.zeros(cout, dtype=torch.float32, device=dev()) for _ in range(3))
    tab = torch.tensor([[g_.data_ptr(), b_.data_ptr(), mm_.data_ptr(), mv_.data_ptr(), scale.data_ptr(), shift.data_ptr(), cout,
                         cb_.data_ptr(), beff.data_ptr()]], dtype=torch.int64, device=dev())
    check(lib.satcv_bn_affine_infer_batched(tab.data_ptr(), 1, 1e-3, ops.stream_ptr()))
    sc_ref = gamma.astype(np.float32).astype(np.float64) / np.sqrt(mv.astype(np.float32).astype(np.float64) + 1e-3)
    sh_ref = beta.astype(np.float32) - mm.astype(np.float32) * sc_ref
    np.testing.assert_allclose(back(scale), sc_ref, rtol=2e-6)
    np.testing.assert_allclose(back(shift), sh_ref, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(back(beff), sh_ref + b.astype(np.float32) * sc_ref, rtol=2e-5, atol=1e-6)
    wf, _ = ops.pack_weights(f32dev(kern), cin, ops.DTYPE_CODE[td])
    y = to_dev(short, td)
    d = ops.make_conv_desc(x0=to_dev(x, td).data_ptr(), c0=cin, w=wf.data_ptr(), bias=beff.data_ptr(), out_scale=scale.data_ptr(), y=y.data_ptr(), ldy=cout,
                           n=n, h=h, w_=w, cout=cout, cout_pad=rup(cout, 32), kh=k, kw=k, dil=1, dtype=ops.DTYPE_CODE[td], accumulate=2)
    xd = to_dev(x, td); d.x0 = xd.data_ptr()
    check(lib.satcv_conv2d_igemm(C.byref(d), ops.stream_ptr()))
    conv = K.conv2d_same(x, kern, None, 1) * back(scale) + back(beff)
    conv = torch.tensor(conv, dtype=torch.float32).to(td).to(torch.float64).numpy()          # rounded to the storage type first
    ref = np.maximum(short + conv, 0.0)
    close(back(y, cout), ref, td, f'residual join {case}')
    assert (back(y, cout) >= 0).all()


@pytest.mark.parametrize('case', [(8, 64, 64, 128, 128, 6, 0), (4, 64, 64, 192, 256, 2, 1), (16, 32, 32, 512, 256, 12, 0)])
def test_dilated_conv_double_buffered_taploop(ops, case):
    """the double-buffered tap-loop form (round 4: 256-pixel x 128-channel tile, 64-channel chunks, gather table re-derived at tap
    boundaries, validity of an item captured when it is loaded): dilated 3x3 convolutions large enough to select it (>= 96 tiles),
    with the fused input BatchNorm + ReLU and a dual-source input, against the float64 oracle."""
    n, h, w, cin, cout, dil, dual = case
    td = torch.bfloat16
    rng = np.random.default_rng(hash(case) % 2**31)
    x = rnd(rng, (n, h, w, cin), td)
    kern = rnd(rng, (3, 3, cin, cout), td, 0.05)
    sc = rng.uniform(0.5, 1.5, cin).astype(np.float32).astype(np.float64)
    sh = rng.normal(0, 0.5, cin).astype(np.float32).astype(np.float64)
    wf, _ = ops.pack_weights(f32dev(kern), cin, ops.DTYPE_CODE[td])
    act = torch.tensor(np.maximum(x * sc + sh, 0.0), dtype=torch.float32).to(td).to(torch.float64).numpy()     # the loader rounds the activated item
    y_ref = K.conv2d_same(act, kern, None, dil)
    if dual:
        c0 = 128
        y = ops.conv2d(to_dev(x[..., :c0], td), wf, cout, kh=3, kw=3, dil=dil, x1=to_dev(x[..., c0:], td), in_scale=f32dev(sc), in_shift=f32dev(sh), in_relu=True)
    else:
        y = ops.conv2d(to_dev(x, td), wf, cout, kh=3, kw=3, dil=dil, in_scale=f32dev(sc), in_shift=f32dev(sh), in_relu=True)
    close(back(y, cout), y_ref, td, f'double-buffered tap loop {case}')


def test_round6_small_entry_points(ops):
    """round 6 additions to the C ABI: satcv_zero2 (gradient buffer + loss scalar cleared by one launch), satcv_adam_step_part (the update on
    sub-ranges, the step counter bumped once: two parts == one whole step, bit for bit), satcv_pack_weights_batched with the item counts of
    satcv_pack_job_items rounded to the block size (every layer's images equal the per-layer satcv_pack_weights, ragged channel counts included)."""
    import ctypes
    from satellite_computervision_amd._lib import lib, check, PackJob
    st = ops.stream_ptr()
    # ---- zero2
    a = torch.randn(4 * 1000 + 12, device=dev())
    b = torch.randn(5, device=dev())
    check(lib.satcv_zero2(a.data_ptr(), a.numel() * 4, b.data_ptr() + 4, 8, st))
    torch.cuda.synchronize()
    assert a.abs().max().item() == 0 and b[1:3].abs().max().item() == 0 and b[0].item() != 0 and b[3].item() != 0
    assert lib.satcv_zero2(a.data_ptr() + 4, 16, None, 0, st) != 0           # misaligned
    # ---- Adam in two parts
    rng = np.random.default_rng(5)
    n = 4096 + 640
    p0, g = f32dev(rng.standard_normal(n)), f32dev(rng.standard_normal(n))
    outs = []
    for parts in (((0, n, 1),), ((1024, n, 0), (0, 1024, 1))):
        p, m, v = p0.clone(), torch.zeros(n, device=dev()), torch.zeros(n, device=dev())
        state = torch.tensor([1e-2, 0.0, 1.0, 0.0], dtype=torch.float32, device=dev())
        for _ in range(3):
            for lo, hi, bump in parts:
                check(lib.satcv_adam_step_part(p.data_ptr() + 4 * lo, g.data_ptr() + 4 * lo, m.data_ptr() + 4 * lo, v.data_ptr() + 4 * lo, hi - lo, 0.9, 0.999, 1e-7,
                                               state.data_ptr(), None, bump, st))
        torch.cuda.synchronize()
        assert state[1].item() == 3.0
        outs.append((p, m, v))
    for x, y in zip(outs[0], outs[1]):
        assert torch.equal(x, y)
    # ---- batched pack against the per-layer pack
    shapes = [((3, 3, 4, 32), False), ((3, 3, 13, 32), False), ((3, 3, 64, 64), False), ((1, 1, 32, 2), False), ((2, 2, 32, 64), True), ((3, 3, 24, 16), False)]
    jobs, keep, refs = [], [], []
    for shp, tr in shapes:
        k = f32dev(rng.standard_normal(shp))
        cin, cout = (shp[3], shp[2]) if tr else (shp[2], shp[3])
        cp, taps = ops.rup(cin, 16), shp[0] * shp[1]
        rf, rd = ops.pack_weights(k, cp, ops.BF16, transposed=tr)
        fw, dg = torch.zeros_like(rf), torch.zeros_like(rd)
        if not tr:
            jobs += [PackJob(k.data_ptr(), fw.data_ptr(), 0, taps, cin, cout, cp, ops.rup(cout, 32)), PackJob(k.data_ptr(), dg.data_ptr(), 1, taps, cin, cout, ops.rup(cout, 16), ops.rup(cin, 32))]
        else:
            jobs += [PackJob(k.data_ptr(), fw.data_ptr(), 2, taps, cin, cout, cp, ops.rup(taps * cout, 32)), PackJob(k.data_ptr(), dg.data_ptr(), 3, taps, cin, cout, ops.rup(taps * cout, 16), ops.rup(cin, 32))]
        keep.append(k); refs.append((rf, rd, fw, dg))
    prefix, tot = [], 0
    for j in jobs:
        prefix.append(tot)
        items = int(lib.satcv_pack_job_items(ctypes.byref(j)))
        assert items % 256 == 0
        tot += items
    arr = (PackJob * len(jobs))(*jobs)
    jd = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev())
    pd = torch.tensor(prefix, dtype=torch.int64, device=dev())
    check(lib.satcv_pack_weights_batched(jd.data_ptr(), pd.data_ptr(), len(jobs), tot, ops.BF16, st))
    torch.cuda.synchronize()
    for (rf, rd, fw, dg), (shp, tr) in zip(refs, shapes):
        assert torch.equal(rf, fw), ('forward image', shp, tr)
        assert torch.equal(rd, dg), ('data-gradient image', shp, tr)
    assert lib.satcv_pack_weights_batched(jd.data_ptr(), pd.data_ptr(), len(jobs), tot - 8, ops.BF16, st) != 0      # a prefix table not built from satcv_pack_job_items


@pytest.mark.parametrize('case', [(2, 32, 32, 64, 128, 0), (3, 16, 48, 128, 256, 64), (8, 8, 8, 192, 128, 0), (2, 64, 32, 64, 128, 32)])
def test_wgrad_dma_kernel_16x16x32_variant(ops, case):
    """round 6 (review item 2a): wgrad_dma_kernel<TW, M16 = true> -- the deep 3x3 weight gradient on v_mfma_f32_16x16x32_bf16 (option wgrad_m16; 2 x 2 blocks per
    (ci, co) tile, k order 4 g ... / 16 + 4 g ... per lane group, X rows of 160 bytes, dY granules swizzled by row & 7).  Correct; measured 33 % slower per launch
    than the 32x32x16 form and therefore off by default (profiles/r06_ab_wgrad_m16.txt).  Tile widths 32 / 16 / 8, one and two sources, the fused input transform;
    both forms against the float64 oracle and against each other."""
    import ctypes
    from satellite_computervision_amd._lib import lib, check
    n, h, w, cin, cout, split = case
    td = torch.bfloat16
    rng = np.random.default_rng(hash(case) % 2**31 + 5)
    x = rnd(rng, (n, h, w, cin), td)
    dy = rnd(rng, (n, h, w, cout), td)
    sc, sh = (rng.random(cin) + 0.5).astype(np.float32), (rng.standard_normal(cin) * 0.3).astype(np.float32)
    a_ref = np.maximum(x * sc.astype(np.float64) + sh.astype(np.float64), 0)
    a_ref = torch.tensor(a_ref, dtype=torch.float32).to(td).double().numpy()
    _, dk_ref, _ = K.conv2d_same_bwd(a_ref, np.zeros((3, 3, cin, cout)), dy, 1)
    out = {}
    old = ctypes.c_int32()
    check(lib.satcv_get_option(b'wgrad_m16', ctypes.byref(old)))
    try:
        for m16 in (0, 1):
            check(lib.satcv_set_option(b'wgrad_m16', m16))
            if split:
                dk = ops.conv2d_wgrad(to_dev(x[..., :split], td), to_dev(dy, td), cin, cout, x1=to_dev(x[..., split:], td), in_scale=f32dev(sc), in_shift=f32dev(sh), in_relu=True)
            else:
                dk = ops.conv2d_wgrad(to_dev(x, td), to_dev(dy, td), cin, cout, in_scale=f32dev(sc), in_shift=f32dev(sh), in_relu=True)
            out[m16] = back(dk)
            close(out[m16], dk_ref, td, f'wgrad (m16 = {m16}) {case}', k=0.5)
    finally:
        check(lib.satcv_set_option(b'wgrad_m16', old.value))
    # the same bf16 products in fp32, another summation order
    assert np.abs(out[0] - out[1]).max() <= 2e-4 * max(np.abs(out[0]).max(), 1e-30)
