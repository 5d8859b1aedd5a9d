"""Device tile input pipeline against the NumPy restatement oracle/input_pipeline.py, whose building
blocks are pinned by outputs of the reference's real utils/array_tools.py (tests/test_oracle_cpu.py)."""
import os
import random
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_morph_kernel_matches_reference_fixtures():
    """all 16 flip/flip/rot90 combinations of the reference's aug_array_morph outputs, through satcv_tile_ingest"""
    from satellite_computervision_amd import processing as P
    z = np.load(os.path.join(GOLD, 'array_tools_reference.npz'))
    x = z['morph_in']                                                # (2, 8, 8, 3) NHWC
    chw = np.ascontiguousarray(np.moveaxis(x, 3, 1))
    for v in (0, 1):
        for h in (0, 1):
            for r in range(4):
                dst = torch.zeros(2, 8, 8, 3, device='cuda')
                P.device_source(chw, (8, 8), dst, 0, morph=(v, h, r), to_fit=False)
                torch.cuda.synchronize()
                assert np.array_equal(dst.cpu().numpy(), z[f'morph_{v}{h}{r}']), (v, h, r)


@pytest.mark.parametrize('kind', ['uint16', 'uint8', 'float32'])
def test_source_rescale_trim_color_morph(kind):
    from oracle import input_pipeline as ip
    from satellite_computervision_amd import processing as P
    rng = np.random.default_rng(3)
    n, c, hin, win, h, w = 3, 4, 40, 36, 32, 32
    if kind == 'uint16':
        planes, rescale = rng.integers(0, 10000, (n, c, hin, win)).astype(np.uint16), 10000.0
    elif kind == 'uint8':
        planes, rescale = rng.integers(0, 256, (n, c, hin, win)).astype(np.uint8), 255.0
    else:
        planes, rescale = (rng.random((n, c, hin, win)) * 3).astype(np.float32), 0.0
    for morph, color in (((0, 0, 0), None), ((1, 0, 3), (0.97, 1.04)), ((0, 1, 2), (1.05, 0.95))):
        ref = ip.get_unet_data(list(planes), (h, w), rescale if rescale else False)
        if color:
            ref = ip.aug_array_color(ref, *color)
        ref = ip.aug_array_morph(ref, *morph).astype(np.float32)
        dst = torch.zeros(n, h, w, c + 2, device='cuda')
        P.device_source(planes, (h, w), dst, 1, rescale_val=rescale, color=color, morph=morph, to_fit=True)
        torch.cuda.synchronize()
        got = dst.cpu().numpy()
        assert np.all(got[..., 0] == 0) and np.all(got[..., -1] == 0)                # neighbours of the channel slice untouched
        if color is None:
            assert np.array_equal(got[..., 1:1 + c], ref)                             # bit-exact (float64 arithmetic, one rounding)
        else:
            # the channel mean is summed in a different order than np.nanmean (float64 for integer planes: <= 1 ulp of the fp32
            # result; float32 planes: NumPy's mean itself is a float32 pairwise sum)
            np.testing.assert_allclose(got[..., 1:1 + c], ref, rtol=2.4e-7 if kind != 'float32' else 2e-6, atol=1e-9 if kind != 'float32' else 1e-6)


def test_nan_mask_channel_and_replacement():
    from oracle import input_pipeline as ip
    from satellite_computervision_amd import processing as P
    rng = np.random.default_rng(5)
    n, c, hin, win, h, w = 2, 3, 20, 20, 16, 16
    planes = (rng.random((n, c, hin, win)) * 50).astype(np.float32)
    planes[0, 1, 5, 6] = np.nan                                                      # flags channels 1 and 2 (mask accumulates)
    planes[1, 0, 9, 9] = -999999.0                                                     # flags every channel of image 1
    planes[1, 2, 3, 3] = np.nan                                                      # only the last channel
    ref = ip.get_unet_data([p.copy() for p in planes], (h, w), 100, add_nan_mask=True, to_fit=True, fill=lambda k: np.full(k, 777.0))
    dst = torch.zeros(n, h, w, c + 1, device='cuda')
    P.device_source(planes, (h, w), dst, 0, rescale_val=100, add_nan_mask=True, to_fit=True, seed=11)
    torch.cuda.synchronize()
    got = dst.cpu().numpy()
    assert np.array_equal(got[..., c], ref[..., c].astype(np.float32))               # mask channel
    replaced = ref[..., :c] == 777.0
    assert replaced.sum() == 2 + 3 + 1
    assert np.array_equal(got[..., :c][~replaced], ref[..., :c][~replaced].astype(np.float32))
    assert np.isfinite(got).all() and np.abs(got[..., :c][replaced]).max() < 8        # N(0,1) draws
    # prediction mode (to_fit False): the reference appends an all-zero mask and replaces nothing
    dst2 = torch.zeros(n, h, w, c + 1, device='cuda')
    P.device_source(planes, (h, w), dst2, 0, rescale_val=100, add_nan_mask=True, to_fit=False)
    torch.cuda.synchronize()
    g2 = dst2.cpu().numpy()
    assert np.all(g2[..., c] == 0) and np.isnan(g2[0, 5 - 2, 6 - 2, 1])


def test_labels_merge_onehot_morph():
    from oracle import input_pipeline as ip
    from satellite_computervision_amd import processing as P
    rng = np.random.default_rng(9)
    n, hin, win, h, w, ncls = 3, 24, 24, 16, 16, 8
    lc = rng.integers(0, 14, (n, 1, hin, win)).astype(np.uint8)
    lc[0, 0, 10, 10] = 255
    lu = rng.choice([0, 82, 84, 5], size=(n, 1, hin, win)).astype(np.float32)
    lc_t, lu_t = [(12, 3), (11, 3), (10, 3), (9, 8), (255, 0)], [(82, 9), (84, 10)]
    for morph in ((0, 0, 0), (1, 1, 1)):
        ref = ip.aug_array_morph(ip.process_y(list(lc), (h, w), ncls, lc_t, list(lu), lu_t), *morph)
        y = torch.zeros(n, h, w, ncls, device='cuda')
        P.device_labels(lc, (h, w), ncls, y, 0, P.merge_lut(lc_t, y.device), lu, P.merge_lut(lu_t, y.device), morph)
        torch.cuda.synchronize()
        assert np.array_equal(y.cpu().numpy(), ref)
        assert (ref.sum(-1) == 0).any()                                               # classes 9/10 fall outside depth 8 -> all-zero rows


def test_generator_feeds_fit():
    """UNETDataGenerator mirror: same draws as the reference under random.seed, device tensors accepted by Model.fit."""
    from oracle import input_pipeline as ip
    from satellite_computervision_amd import processing as P, model_tools as mt
    rng = np.random.default_rng(2)
    N, h = 8, 32
    s2 = [rng.integers(0, 10000, (4, h + 4, h + 4)).astype(np.uint16) for _ in range(N)]
    lab = [rng.integers(0, 2, (1, h + 4, h + 4)).astype(np.uint8) for _ in range(N)]
    gen = P.UNETDataGenerator(labelfiles=lab, s2files=s2, batch_size=4, unet_dim=(h, h), n_channels=4, n_classes=2, shuffle=False, lc_transitions=None)
    assert len(gen) == 2
    random.seed(123)
    x, y = gen[1]
    random.seed(123)
    color = (random.uniform(0.95, 1.05), random.uniform(0.95, 1.05))
    morph = (random.uniform(0, 1) < 0.5, random.uniform(0, 1) < 0.5, random.randint(0, 3))
    fx, fy = ip.getitem([dict(arrays=s2[4:], rescale_val=10000.0, color=color)], lab[4:], (h, h), 2, None, morph)
    np.testing.assert_allclose(x.cpu().numpy(), fx, rtol=2.4e-7, atol=1e-9)
    assert np.array_equal(y.cpu().numpy(), fy)
    mt.reset_uids(); mt.set_seed(0)
    m = mt.get_unet_model(2, 4, filters=[32, 64], factors=[2, 2])
    m.compile(optimizer=mt.Adam(1e-3), loss=lambda t, p: mt.weighted_categorical_crossentropy(t, p, [1.0, 1.0]))
    hist = m.fit(gen, epochs=2, verbose=0)
    assert np.isfinite(hist.history['loss']).all() and len(hist.history['loss']) == 2


@pytest.mark.parametrize('kind', ['uint16', 'float32'])
def test_siamese_generator_against_oracle(kind):
    """SiameseDataGenerator (utils/processing.py:757-893): two dates, validity mask over both, binary labels x mask, per-date colour
    augmentation, one flip / rotation for the stack -- against the NumPy restatement with the same `random` draws.  Without NaNs
    the batches agree to float32 rounding; with NaNs the mask / labels are exact and the replaced values are U[0,1) draws."""
    from oracle import input_pipeline as ip
    from satellite_computervision_amd import processing as P
    rng = np.random.default_rng(8)
    n, c, hin, win, h, w = 4, 4, 40, 36, 32, 32
    def planes():
        p = rng.integers(0, 10000, (n, c, hin, win))
        return p.astype(np.uint16) if kind == 'uint16' else p.astype(np.float32)
    bef, aft = planes(), planes()
    labs = rng.integers(0, 4, (n, hin, win)).astype(np.uint8)
    kw = dict(labelfiles=list(labs), batch_size=n, unet_dim=(h, w), n_channels=c, shuffle=False)
    for seed in (0, 1, 2):
        g = P.SiameseDataGenerator(list(bef), list(aft), False, **kw)
        assert len(g) == 1
        random.seed(seed)
        (xb, xa), y = g[0]
        random.seed(seed)
        cb = (random.uniform(0.95, 1.05), random.uniform(0.95, 1.05)); ca = (random.uniform(0.95, 1.05), random.uniform(0.95, 1.05))
        morph = (random.uniform(0, 1) < 0.5, random.uniform(0, 1) < 0.5, random.randint(0, 3))
        (rb, ra), ry = ip.siamese_getitem(list(bef), list(aft), list(labs), (h, w), c, False, True, cb, ca, morph)
        assert tuple(xb.shape) == rb.shape and tuple(y.shape) == ry.shape == (n, h, w, 1)
        np.testing.assert_allclose(xb.cpu().numpy(), rb, rtol=2e-6, atol=2e-7)
        np.testing.assert_allclose(xa.cpu().numpy(), ra, rtol=2e-6, atol=2e-7)
        assert np.array_equal(y.cpu().numpy(), ry) and set(np.unique(ry)) <= {0.0, 1.0}
    # invalid pixels: NaN in one date (float planes only), a band below -1 after the rescale in the other
    if kind == 'float32':
        bef2, aft2 = bef.copy(), aft.copy()
        bef2[0, 1, 10:14, 9:12] = np.nan
        aft2[1, 2, 20:22, 5:30] = -20000.0
        aft2[0, 0, 11, 10] = np.nan
        g = P.SiameseDataGenerator(list(bef2), list(aft2), True, to_fit=False, **kw)
        xb, xa = g[0]
        rb, ra = ip.siamese_getitem(list(bef2), list(aft2), list(labs), (h, w), c, True, to_fit=False, fill=lambda k: np.full(k, 0.5))
        nb, na = np.isnan(bef2[:, :, 4:36, 2:34]).transpose(0, 2, 3, 1), np.isnan(aft2[:, :, 4:36, 2:34]).transpose(0, 2, 3, 1)
        gb, ga = xb.cpu().numpy(), xa.cpu().numpy()
        assert np.array_equal(gb[~nb], rb[~nb]) and np.array_equal(ga[~na], ra[~na])
        assert nb.sum() == 12 and (gb[nb] >= 0).all() and (gb[nb] < 1).all() and len(np.unique(gb[nb])) > 6
        g = P.SiameseDataGenerator(list(bef2), list(aft2), True, **kw)
        random.seed(5)
        (xb, xa), y = g[0]
        random.seed(5)
        cb = (random.uniform(0.95, 1.05), random.uniform(0.95, 1.05)); ca = (random.uniform(0.95, 1.05), random.uniform(0.95, 1.05))
        morph = (random.uniform(0, 1) < 0.5, random.uniform(0, 1) < 0.5, random.randint(0, 3))
        _, ry = ip.siamese_getitem(list(bef2), list(aft2), list(labs), (h, w), c, True, True, cb, ca, morph, fill=lambda k: np.full(k, 0.5))
        assert np.array_equal(y.cpu().numpy(), ry) and (ry == 0).sum() > (np.where(labs > 1, 1, labs)[:, 4:36, 2:34] == 0).sum()
        # the colour augmentation of the date with NaNs used a mean that includes the replaced values: finite everywhere
        assert torch.isfinite(xb).all() and torch.isfinite(xa).all()


def test_siamese_generator_feeds_the_siamese_model():
    """SiameseDataGenerator -> make_siamese_unet.fit / predict: device batches of two dates, no host round trip."""
    from satellite_computervision_amd import processing as P
    from satellite_computervision_amd import model_tools as mt
    rng = np.random.default_rng(2)
    n, c, hw = 8, 4, 32
    bef = rng.integers(0, 10000, (n, c, hw, hw)).astype(np.uint16)
    aft = bef.copy()
    aft[:, :, 8:24, 8:24] = rng.integers(0, 10000, (n, c, 16, 16)).astype(np.uint16)          # the change to detect
    labs = np.zeros((n, hw, hw), np.uint8); labs[:, 8:24, 8:24] = 1
    g = P.SiameseDataGenerator(list(bef), list(aft), False, labelfiles=list(labs), batch_size=4, unet_dim=(hw, hw), n_channels=c, shuffle=True)
    mt.reset_uids(); mt.set_seed(0)
    m = mt.make_siamese_unet(c, [32, 64], [2, 2])
    m.compile(optimizer=mt.Adam(2e-3), loss=lambda t, p: mt.weighted_bce(t, p, 2.0))
    random.seed(0)
    h = m.fit(g, epochs=4, verbose=0)
    assert len(h.history['loss']) == 4 and np.isfinite(h.history['loss']).all() and h.history['loss'][-1] < h.history['loss'][0]
    gp = P.SiameseDataGenerator(list(bef), list(aft), False, to_fit=False, batch_size=4, unet_dim=(hw, hw), n_channels=c, shuffle=False)
    probs, classes = m.predict(gp)
    assert probs.shape == (n, hw, hw, 1) and classes.shape == (n, hw, hw, 1)
