"""Training-step / inference time of the model families beside the headline U-Net: Siamese U-Net + ASPP
(utils/model_tools.py:576-663), atrous CNNs get_acnn_model / get_acnn_model2 (:922-1014), at a production-like shape (batch 16, 256 x 256 x 4)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from satellite_computervision_amd import model_tools as mt

mt.set_compute_dtype(os.environ.get('DTYPE', 'bfloat16'))
B, S, C = int(os.environ.get('B', '16')), int(os.environ.get('S', '256')), 4
rng = np.random.default_rng(0)


def timed(fn, n=8, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


x = rng.random((B, S, S, C), dtype=np.float32)
x2 = rng.random((B, S, S, C), dtype=np.float32)
only = os.environ.get('ONLY', '')
if only in ('', 'siamese'):
    mt.reset_uids(); mt.set_seed(0)
    sm = mt.make_siamese_unet(C, filters=[32, 64, 128, 256], factors=[2, 2, 2, 2])
    sm.compile(optimizer=mt.Adam(1e-3), loss=lambda yt, yp: mt.weighted_bce(yt, yp, 2.0))
    ys = (rng.random((B, S, S, 1)) < 0.3).astype(np.float32)
    print(f'make_siamese_unet b{B} {S}x{S}x{C}: train step {timed(lambda: sm.train_on_batch([x, x2], ys)):8.2f} ms   predict {timed(lambda: sm.predict([x, x2], batch_size=B)):8.2f} ms', flush=True)
if only in ('', 'acnn'):
    mt.reset_uids(); mt.set_seed(0)
    ac = mt.get_acnn_model(3, 32, C, 6)
    ac.compile(optimizer=mt.Adam(1e-3), loss=lambda a, b: mt.weighted_categorical_crossentropy(a, b, [1.0, 1.0, 2.0]))
    y3 = np.eye(3, dtype=np.float32)[rng.integers(0, 3, (B, S, S))]
    print(f'get_acnn_model(3, 32, {C}, 6) b{B} {S}x{S}: train step {timed(lambda: ac.train_on_batch(x, y3)):8.2f} ms   predict {timed(lambda: ac.predict(x, batch_size=B)):8.2f} ms', flush=True)
if only in ('', 'acnn2'):
    mt.reset_uids(); mt.set_seed(0)
    a2 = mt.get_acnn_model2(3, C, nfilters=16, depth=16)
    a2.compile(optimizer=mt.Adam(1e-3), loss=lambda a, b: mt.weighted_categorical_crossentropy(a, b, [1.0, 1.0, 2.0]))
    y3 = np.eye(3, dtype=np.float32)[rng.integers(0, 3, (B, S, S))]
    print(f'get_acnn_model2(3, {C}, 16, 16) b{B} {S}x{S}: train step {timed(lambda: a2.train_on_batch(x, y3)):8.2f} ms   predict {timed(lambda: a2.predict(x, batch_size=B)):8.2f} ms', flush=True)
