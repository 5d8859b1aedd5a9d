"""GPU probe: inference throughput of the full U-Net, bf16 plan vs folded fp8 plan, 256^2 tiles / 384^2 chips / 1024^2 scenes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from satellite_computervision_amd import model_tools as mt
mt.reset_uids(); mt.set_seed(0); mt.set_compute_dtype('bfloat16')
m = mt.get_unet_model(2, 4)
rng = np.random.default_rng(0)


def timeit(x, reps=10):
    for _ in range(3): m.predict_on_device(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): m.predict_on_device(x)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps


for (n, s) in ((8, 256), (16, 256), (32, 256), (64, 256), (128, 256), (36, 384), (4, 1024)):
    x = torch.from_numpy(rng.beta(2, 5, (n, s, s, 4)).astype(np.float32)).cuda()
    m.disable_fp8_inference()
    tb = timeit(x)
    m.enable_folded_inference()
    tf = timeit(x)
    m.enable_fp8_inference(x[:min(n, 8)])
    t8 = timeit(x)
    k = (s // 256) ** 2 if s != 384 else 1
    print(f'n{n} {s}x{s}: bf16 {tb*1e3:7.2f} ms ({n/tb:8.1f} img/s)  folded-bf16 {tf*1e3:7.2f} ms ({n/tf:8.1f} img/s)   fp8 {t8*1e3:7.2f} ms ({n/t8:8.1f} img/s)   speed-up {tb/t8:.2f}', flush=True)
