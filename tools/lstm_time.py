"""ConvLSTM2D family: time of one training step / one inference pass at the generators' default shapes (6 time steps of 4-band
images, utils/processing.py:901; LSTM tiles 32 x 32 against a 3x finer 96 x 96 U-Net tile for the hybrid, utils/model_tools.py:874)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from satellite_computervision_amd import model_tools as mt, lstm_tools as lt

dtype = os.environ.get('DTYPE', 'bfloat16')
mt.set_compute_dtype(dtype)
B, T, H, W, C, NCLS = int(os.environ.get('B', '16')), 6, 32, 32, 4, 3
rng = np.random.default_rng(0)


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


ONLY = os.environ.get('ONLY', '')
mt.reset_uids(); mt.set_seed(0)
m = lt.get_lstm_model(C, NCLS, T)
m.compile(optimizer=mt.Adam(1e-3), loss=mt.mse_4d)
x = rng.random((B, T, H, W, C), dtype=np.float32)
y = np.eye(NCLS, dtype=np.float32)[rng.integers(0, NCLS, (B, H, W))]
if ONLY != 'hybrid':
    print(f'get_lstm_model  b{B} T{T} {H}x{W}x{C}: train step {timed(lambda: m.train_on_batch(x, y)):8.2f} ms   predict {timed(lambda: m.predict(x)):8.2f} ms', flush=True)

mt.reset_uids(); mt.set_seed(0)
hy = lt.get_hybrid_model([96, 96, C], [T, H, W, C], NCLS)
hy.compile(optimizer=mt.Adam(1e-3), loss=mt.mse_4d)
xu = rng.random((B, 96, 96, C), dtype=np.float32)
yu = np.eye(NCLS, dtype=np.float32)[rng.integers(0, NCLS, (B, 96, 96))]
print(f'get_hybrid_model b{B} unet 96x96 + lstm T{T} {H}x{W}: train step {timed(lambda: hy.train_on_batch([xu, x], yu)):8.2f} ms   predict {timed(lambda: hy.predict([xu, x])):8.2f} ms', flush=True)
