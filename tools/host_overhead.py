"""GPU probe: host-side enqueue time per training step vs GPU time per step (is the launch path the bottleneck?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from satellite_computervision_amd import model_tools as mt
if os.environ.get('WITH_PG'):
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
    kw = dict(device_id=torch.device('cuda', 0)) if os.environ['WITH_PG'] == 'eager' else {}
    dist.init_process_group('nccl', rank=0, world_size=1, **kw)
    if os.environ['WITH_PG'] != 'lazy0':
        t = torch.ones(4, device='cuda'); dist.all_reduce(t); torch.cuda.synchronize()
mt.reset_uids(); mt.set_seed(0); mt.set_compute_dtype('bfloat16')
m = mt.get_unet_model(2, 4)
m.compile(optimizer=mt.Adam(9e-4), loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, [1.0, 20.0]))
rng = np.random.default_rng(0)
x, y = bench.synth_batch(rng, 64)
xb, yb = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
for _ in range(5): m.train_step_device(xb, yb)
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K): m.train_step_device(xb, yb)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'enqueue {1e3*(t1-t0)/K:.2f} ms/step, total {1e3*(t2-t0)/K:.2f} ms/step, OMP={os.environ.get("OMP_NUM_THREADS")}, threads={torch.get_num_threads()}')
