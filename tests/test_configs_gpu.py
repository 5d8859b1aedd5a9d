"""The BASELINE.json configurations at the depth / size bench.py times them, against the float64 oracle:

  * configs[1]: the five-level get_unet_model(2, 4) TRAINED in bf16 (the benchmarked mode) lands where fp32 training lands;
  * configs[4]: the folded fp8 (e4m3) five-level graph on the 384x384 chips of a 1024x1024 scene, through predict_chips;
  * configs[2]: DeepLab-v3 / ResNet-50 on a 512x512x4 tile at batch 1 is covered by test_model_gpu.py::test_deeplabv3_resnet50_inference.

Reference call sites: utils/model_tools.py:394-415 (get_unet_model), utils/prediction_tools.py:87-156 (chip loop).
"""
import numpy as np
import pytest
import torch

from oracle import tiling as OT
from oracle.unet import UNetOracle

pytestmark = pytest.mark.gpu


def iou(a, b, cls=1):
    inter = np.logical_and(a == cls, b == cls).sum()
    union = np.logical_or(a == cls, b == cls).sum()
    return inter / union if union else 1.0


def make_tiles(rng, n, size=256):
    """bright rectangles on a smooth background + pixel noise; label = rectangle (the task of
    test_model_gpu.py::test_trained_five_level_model_bf16_iou_within_1e3, at any tile size)"""
    lo = torch.tensor(rng.random((n, 4, size // 32, size // 32)), dtype=torch.float32)
    x = torch.nn.functional.interpolate(lo, size=(size, size), mode='bilinear', align_corners=False).permute(0, 2, 3, 1).numpy() * 0.5
    lab = np.zeros((n, size, size), np.int64)
    for i in range(n):
        for _ in range(4 * (size // 256) ** 2):
            hh, ww = rng.integers(24, 96, 2)
            y0, x0 = rng.integers(0, size - hh), rng.integers(0, size - ww)
            lab[i, y0:y0 + hh, x0:x0 + ww] = 1
    x = x + 0.35 * lab[..., None] * np.array([1.0, 0.6, 0.8, 1.2], np.float32) + 0.05 * rng.standard_normal((n, size, size, 4))
    return x.astype(np.float32), lab


@pytest.fixture(scope='module')
def trained():
    """get_unet_model(2, 4), default five levels, trained for 240 Adam steps of batch 8 on the same seeded batches, once with fp32
    storage and once with bf16 storage (the mode bench.py times: bf16 activations / weight images / gradients, fp32 master weights)."""
    from satellite_computervision_amd import model_tools as mt
    assert torch.cuda.is_available()
    data_rng = np.random.default_rng(9)
    x, lab = make_tiles(data_rng, 32)
    xt, labt = make_tiles(data_rng, 4)
    y = np.eye(2, dtype=np.float32)[lab]
    out = dict(mt=mt, xt=xt, labt=labt, calib=x[:8])
    for dtype in ('float32', 'bfloat16'):
        mt.reset_uids(); mt.set_seed(2)                    # identical initial weights and shuffle order for both runs
        m = mt.get_unet_model(2, 4)
        m.compute_dtype = dtype
        m.compile(optimizer=mt.Adam(1e-3), loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, [1.0, 2.0]))
        hist = m.fit(x, y, batch_size=8, epochs=60, verbose=0)            # 240 steps
        out[dtype] = dict(model=m, weights=m.get_weights_dict(), names=mt.structural_names(m), loss=hist.history['loss'])
    return out


def oracle_for(tr, dtype):
    o = UNetOracle(2, 4, dtype=np.float64)
    w, names = tr[dtype]['weights'], tr[dtype]['names']
    for k in o.params:
        o.params[k] = w[names[k]].astype(np.float64)
    return o


# Two training runs are two TRAJECTORIES: rounding differences (bf16 vs fp32 storage, or just another summation order) grow through
# 240 Adam steps, and the nets they end in differ by more than any single-step tolerance.  Measured on MI355X over three seeds on 16
# held-out tiles (tools/diag_train_iou.py): fp32-trained 0.9915 / 0.9959 / 0.9969, bf16-trained 0.9930 / 0.9943 / 0.9939 (three
# launches per thin layer) and 0.9837 / 0.9965 / 0.9946 (fused thin-layer backward) -- a spread of 3e-3 ... 8e-3 in either direction
# with final losses equal to three digits.  So: the PARITY bar (1e-3) is asserted where it is defined, device mask vs float64 oracle on
# the SAME weights; between the two trained nets the test asserts the trajectory-noise bound.
TRAJECTORY_IOU_TOL = 1.5e-2


def test_bf16_training_lands_where_fp32_training_lands(trained):
    """configs[1] is bf16 TRAINING: 240 steps in bf16 and in fp32 from the same seed and batches.  Held-out 256x256 tiles:
    (1) each trained net's device mask scores within 1e-3 IoU of the float64 oracle forward on that net's own weights;
    (2) the bf16-trained net learns the task as well as the fp32-trained one: IoU within the trajectory-noise bound above;
    (3) the loss curves track each other (bf16 storage rounding does not derail optimisation)."""
    xt, labt = trained['xt'], trained['labt']
    res = {}
    for dtype in ('float32', 'bfloat16'):
        o = oracle_for(trained, dtype)
        _, c_ref = o.forward(xt, training=False)
        _, c_dev = trained[dtype]['model'].predict(xt, batch_size=4)
        res[dtype] = dict(ref=iou(c_ref, labt), dev=iou(c_dev, labt), diff=int((c_dev != c_ref).sum()))
    lf, lb = np.asarray(trained['float32']['loss']), np.asarray(trained['bfloat16']['loss'])
    print(f"five-level net, 240 steps: fp32-trained IoU oracle {res['float32']['ref']:.5f} device {res['float32']['dev']:.5f} "
          f"({res['float32']['diff']} px differ); bf16-trained IoU oracle {res['bfloat16']['ref']:.5f} device(bf16) {res['bfloat16']['dev']:.5f} "
          f"({res['bfloat16']['diff']} px differ); final epoch loss fp32 {lf[-1]:.5f} bf16 {lb[-1]:.5f}")
    assert res['float32']['ref'] > 0.9 and res['bfloat16']['ref'] > 0.9          # both have learned the task
    for dtype in ('float32', 'bfloat16'):
        assert abs(res[dtype]['dev'] - res[dtype]['ref']) <= 1e-3, res
    assert abs(res['bfloat16']['dev'] - res['float32']['dev']) <= TRAJECTORY_IOU_TOL, res
    # loss curves: same shape (first epochs identical to a few %, last epochs both converged to the same level)
    assert abs(lb[0] - lf[0]) < 0.05 * lf[0] and lb[-1] < 0.25 * lb[0] and lf[-1] < 0.25 * lf[0]
    assert abs(np.mean(lb[-10:]) - np.mean(lf[-10:])) < 0.1 * np.mean(lf[-10:]) + 1e-3


# Tolerance the fp8 path holds at full depth (measured on MI355X): on the trained five-level net the fp8 mask's
# IoU is 3.8e-4 from the oracle's and 37 of 589,824 stitched pixels differ (all inside the +-0.25 probability margin), so the
# north-star's 1e-3 IoU bar is asserted for fp8 too, with 99.9 % pixel agreement.
FP8_IOU_TOL = 1e-3
FP8_AGREE = 0.999


def test_fp8_five_level_chips_of_a_1024_scene_vs_oracle(trained):
    """configs[4] at the depth bench.py times it: the five-level fp32-trained net, folded fp8 plan, on the nine 384x384 chips
    (buff 128, kernel 256) of a 1024x1024 scene through prediction_tools.predict_chips; scored against the float64 oracle run
    chip by chip through the restated reference loop (oracle/tiling.py) on the same weights."""
    from satellite_computervision_amd import prediction_tools as pt
    m = trained['float32']['model']
    o = oracle_for(trained, 'float32')
    rng = np.random.default_rng(21)
    scene, lab = make_tiles(rng, 1, 1024)
    scene, lab = scene[0], lab[0]
    idx = pt.generate_chip_indices(scene, 128, 256)
    assert idx == OT.generate_chip_indices(scene, 128, 256) and len(idx) == 9
    ref = OT.predict_chips(scene, idx, np.zeros(scene.shape[:2]), lambda chip: o.forward(chip, training=False)[0][..., ::-1], kernel=256, buff=128)
    # (channel 0 of the reversed probabilities = P(class 1): the reference loop accumulates channel 0 of what predict returns)
    got32 = pt.predict_chips(scene, idx, np.zeros(scene.shape[:2]), m, kernel=256, buff=128, batch_size=9, channel=1)
    m.enable_fp8_inference(trained['calib'])
    try:
        got8 = pt.predict_chips(scene, idx, np.zeros(scene.shape[:2]), m, kernel=256, buff=128, batch_size=9, channel=1)
    finally:
        m.disable_fp8_inference()
    core = np.zeros(scene.shape[:2], bool)
    for y, x in idx:
        core[y:y + 256, x:x + 256] = True                  # the stitched centres (the buff // 2 border is never predicted)
    c_ref, c32, c8 = (ref > 0.5)[core], (got32 > 0.5)[core], (got8 > 0.5)[core]
    labc = lab[core]
    i_ref, i32, i8 = iou(c_ref, labc), iou(c32, labc), iou(c8, labc)
    agree = (c8 == c_ref).mean()
    sure = np.abs(ref[core] - 0.5) > 0.25
    print(f'fp8 five-level, 9 chips of 384x384: IoU oracle {i_ref:.5f} fp32 {i32:.5f} fp8 {i8:.5f}; fp8 pixel agreement {agree:.5f} '
          f'({(c8 != c_ref).sum()} of {c8.size} differ), mean |dp| {np.abs(got8 - ref)[core].mean():.4f}, confident pixels flipped '
          f'{(c8[sure] != c_ref[sure]).sum()}')
    assert i_ref > 0.8
    np.testing.assert_allclose(got32[core], ref[core], atol=2e-3)          # fp32 plan: the stitched probabilities themselves
    assert abs(i32 - i_ref) <= 1e-3
    assert abs(i8 - i_ref) <= FP8_IOU_TOL, (i8, i_ref)
    assert agree >= FP8_AGREE, agree
    assert (c8[sure] == c_ref[sure]).mean() > 0.999                          # confidently classified pixels do not flip
    assert got8[~core].max() == 0
