"""Diagnostic (GPU): per-parameter gradient error of the device U-Net vs the float64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_model_gpu import build_pair
from satellite_computervision_amd import model_tools as mt
from oracle import losses as OL

for dtype in ('float32', 'bfloat16'):
    filters, factors = [32, 64], [2, 2]
    o, m, names = build_pair(mt, dtype, 2, 4, filters, factors)
    rng = np.random.default_rng(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    x = rng.random((n, S, S, 4)).astype(np.float32)
    lab = (rng.random((n, S, S)) < 0.3).astype(np.int64)
    t = np.eye(2)[lab].astype(np.float32)
    m.compile(optimizer=mt.Adam(9e-4), loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, [1.0, 20.0]))
    pr, _ = o.forward(x, training=True)
    loss_ref, dprobs, _ = OL.weighted_categorical_crossentropy(t.astype(np.float64), pr, [1.0, 20.0])
    g_ref = o.backward(dprobs)
    loss = m.train_on_batch(x, t)
    rt = m.runtime
    print(dtype, 'loss', loss, loss_ref)
    for k in o.trainable:
        g = rt.get_grad(names[k]).cpu().numpy().astype(np.float64)
        r = g_ref[k]
        l2 = np.linalg.norm(g - r) / max(np.linalg.norm(r), 1e-30)
        mx = np.abs(g - r).max() / max(np.abs(r).max(), 1e-30)
        cos = (g * r).sum() / max(np.linalg.norm(g) * np.linalg.norm(r), 1e-30)
        print(f'  {k:22s} relL2 {l2:9.3e}  relmax {mx:9.3e}  cos {cos:8.5f}  |ref|max {np.abs(r).max():9.3e}')
