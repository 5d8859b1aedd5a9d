"""Training step time vs batch size (where does the launch path / per-kernel latency start to dominate?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from satellite_computervision_amd import model_tools as mt
mt.reset_uids(); mt.set_seed(0); mt.set_compute_dtype('bfloat16')
m = mt.get_unet_model(2, 4)
m.compile(optimizer=mt.Adam(9e-4), loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, [1.0, 20.0]))
rng = np.random.default_rng(0)
for B in (1, 4, 8, 16, 32, 64):
    x, y = bench.synth_batch(rng, B)
    xb, yb = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    for _ in range(5): m.train_step_device(xb, yb)
    torch.cuda.synchronize()
    K = 20
    t0 = time.perf_counter()
    for _ in range(K): m.train_step_device(xb, yb)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'B={B:3d}: enqueue {1e3*(t1-t0)/K:.2f} ms/step, total {1e3*(t2-t0)/K:.2f} ms/step, {B*K/(t2-t0):.0f} tiles/s')
