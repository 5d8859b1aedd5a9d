"""launches per get_lstm_model training step: run under `rocprofv3 --kernel-trace --stats` and divide the total calls by STEPS (the first three steps run eagerly,
the rest replay the captured graph: same launches)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from satellite_computervision_amd import model_tools as mt, lstm_tools as lt
mt.set_compute_dtype('bfloat16'); mt.reset_uids(); mt.set_seed(0)
B, T, H, W, C, NCLS, STEPS = 16, 6, 32, 32, 4, 3, 40
m = lt.get_lstm_model(C, NCLS, T)
m.compile(optimizer=mt.Adam(1e-3), loss=mt.mse_4d)
rng = np.random.default_rng(0)
x = rng.random((B, T, H, W, C), dtype=np.float32)
y = np.eye(NCLS, dtype=np.float32)[rng.integers(0, NCLS, (B, H, W))]
for _ in range(STEPS):
    m.train_on_batch(x, y)
torch.cuda.synchronize()
print('steps', STEPS)
