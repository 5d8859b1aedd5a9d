"""GPU probe: standalone time of the weight-gradient kernel (incl. slab reduce) for every U-Net 3x3 shape."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from satellite_computervision_amd import ops
dev = torch.device('cuda')


def run(n, h, w, cin, cout, reps=20):
    x = torch.randn(n, h, w, cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(n, h, w, cout, device=dev).to(torch.bfloat16)
    sc = torch.rand(cin, device=dev) + 0.5; sh = torch.randn(cin, device=dev)
    dw = torch.zeros(3, 3, cin, cout, device=dev)
    f = lambda: ops.conv2d_wgrad(x, dy, cin, cout, in_scale=sc, in_shift=sh, in_relu=True, dw=dw)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    fl = 2.0 * n * h * w * cin * cout * 9
    by = n * h * w * (cin + cout) * 2
    print(f'  n{n} {h}x{w} {cin}->{cout}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s  {by/t/1e12:5.2f} TB/s(alg)', flush=True)


# every 3x3 weight gradient of get_unet_model(2, 4) at batch 64: encoder / centre Cin->Cout, decoder conv1 (2f->f) and conv2 (f->f)
SH = [(64, 256, 256, 16, 32), (64, 256, 256, 64, 32), (64, 256, 256, 32, 32), (64, 128, 128, 32, 64), (64, 128, 128, 128, 64), (64, 128, 128, 64, 64),
      (64, 64, 64, 64, 128), (64, 64, 64, 256, 128), (64, 64, 64, 128, 128), (64, 32, 32, 128, 256), (64, 32, 32, 512, 256), (64, 32, 32, 256, 256),
      (64, 16, 16, 256, 512), (64, 16, 16, 1024, 512), (64, 16, 16, 512, 512), (64, 8, 8, 512, 1024)]
if os.environ.get('PROBE_DEEP'):
    SH = [(64, 64, 64, 128, 128), (64, 32, 32, 256, 256), (64, 32, 32, 512, 256), (64, 16, 16, 512, 512)]
for shp in SH:
    run(*shp)
