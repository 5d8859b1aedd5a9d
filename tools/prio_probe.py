"""GPU probe: does stream priority change the training step (main stream high priority, weight-gradient stream normal)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from satellite_computervision_amd import model_tools as mt
mt.reset_uids(); mt.set_seed(0); mt.set_compute_dtype('bfloat16')
m = mt.get_unet_model(2, 4)
m.compile(optimizer=mt.Adam(9e-4), loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, [1.0, 20.0]))
rng = np.random.default_rng(0)
x, y = bench.synth_batch(rng, 64)
xb, yb = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()


def run(K=20):
    for _ in range(5): m.train_step_device(xb, yb)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): m.train_step_device(xb, yb)
    torch.cuda.synchronize(); return 64 * K / (time.perf_counter() - t0)


print('default priority      :', round(run(), 1), 'tiles/s')
hp = torch.cuda.Stream(priority=-1)
with torch.cuda.stream(hp):
    print('main stream high prio :', round(run(), 1), 'tiles/s')
print('default priority again:', round(run(), 1), 'tiles/s')
