"""GPU probe: PCIe-inclusive rates of the Keras-style surface (host ndarray in, host ndarray out) next to the HBM-resident rates."""
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
xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()


def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps


a = t(lambda: m.predict_on_device(xd)); b = t(lambda: m.predict(x, batch_size=64))
print(f'inference  : HBM-resident {64/a:8.1f} tiles/s   host ndarray in / ndarray out {64/b:8.1f} tiles/s')
a = t(lambda: m.train_step_device(xd, yd)); b = t(lambda: m.train_on_batch(x, y))
print(f'training   : HBM-resident {64/a:8.1f} tiles/s   host ndarray batches (incl. loss read-back) {64/b:8.1f} tiles/s')
xp, yp = torch.from_numpy(x).pin_memory(), torch.from_numpy(y).pin_memory()
b = t(lambda: m.train_step_device(xp, yp))
print(f'training   : pinned host tensors {64/b:8.1f} tiles/s')
xs, ys = np.concatenate([x] * 6), np.concatenate([y] * 6)
for flag in ('1', '0'):
    os.environ['SATCV_PREFETCH'] = flag
    m.fit(xs, ys, batch_size=64, epochs=1, verbose=0, shuffle=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.fit(xs, ys, batch_size=64, epochs=2, verbose=0, shuffle=False)
    torch.cuda.synchronize()
    print(f'fit(host ndarrays, 12 steps of 64) SATCV_PREFETCH={flag}: {12 * 64 / (time.perf_counter() - t0):8.1f} tiles/s')
