"""GPU probe: one training step captured into a hipGraph (torch.cuda.CUDAGraph) and replayed, against eager launches."""
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


def eager(K=20):
    for _ in range(5): m.train_step_device(xb, yb)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): m.train_step_device(xb, yb)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e3


print('eager          :', round(eager(), 3), 'ms/step')
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): m.train_step_device(xb, yb)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        m.train_step_device(xb, yb)
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    print('graph replay   :', round((time.perf_counter() - t0) / 20 * 1e3, 3), 'ms/step')
except Exception as e:
    print('capture failed :', type(e).__name__, str(e)[:300])
print('eager again    :', round(eager(), 3), 'ms/step')
