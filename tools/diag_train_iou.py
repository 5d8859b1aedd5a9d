"""Diagnostic: held-out IoU of the five-level net after 240 steps, fp32 vs bf16 storage with the fused thin-layer backward on / off,
over a few seeds (how much of an IoU difference is trajectory noise?)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from test_configs_gpu import make_tiles, iou
from satellite_computervision_amd import model_tools as mt
data_rng = np.random.default_rng(9)
x, lab = make_tiles(data_rng, 32)
xt, labt = make_tiles(data_rng, 16)
y = np.eye(2, dtype=np.float32)[lab]
for seed in (2, 5, 7):
    for dtype, fuse in (('float32', True), ('bfloat16', False), ('bfloat16', True)):
        mt.reset_uids(); mt.set_seed(seed)
        m = mt.get_unet_model(2, 4)
        m.compute_dtype = dtype
        m.fuse_thin_bwd = fuse
        m.compile(optimizer=mt.Adam(1e-3), loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, [1.0, 2.0]))
        h = m.fit(x, y, batch_size=8, epochs=60, verbose=0)
        _, c = m.predict(xt, batch_size=8)
        print(f'seed {seed} {dtype:9s} fused_bwd={fuse!s:5s} IoU(16 held-out tiles) {iou(c, labt):.5f}  final loss {h.history["loss"][-1]:.5f}', flush=True)
