"""GPU probe: 3x3 conv throughput per storage dtype (bf16 / fp8 K=16 MFMA / fp8 block-scaled K=64 MFMA)."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from satellite_computervision_amd import ops
from satellite_computervision_amd._lib import lib, check, BF16, FP8, FP8X
dev = torch.device('cuda')


def run(n, h, w, cin, cout, dt, reps=30):
    td = {BF16: torch.bfloat16}.get(dt, torch.uint8)
    x = (torch.randn(n, h, w, cin, device=dev) * 0.5).to(torch.bfloat16) if dt == BF16 else torch.randint(0, 120, (n, h, w, cin), device=dev, dtype=torch.uint8)
    kern = torch.randn(3, 3, cin, cout, device=dev) * 0.1
    wf, _ = ops.pack_weights(kern, cin, dt, want_dgrad=False)
    y = torch.empty(n, h, w, cout, device=dev, dtype=td)
    b = torch.zeros(cout, device=dev); osc = torch.full((cout,), 1e-3, device=dev)
    d = ops.make_conv_desc(x0=x.data_ptr(), c0=cin, w=wf.data_ptr(), y=y.data_ptr(), ldy=cout, n=n, h=h, w_=w, cout=cout, cout_pad=ops.rup(cout, 32),
                           dtype=dt, bias=b.data_ptr(), out_scale=osc.data_ptr(), out_relu=1)
    s = ops.stream_ptr()
    f = lambda: check(lib.satcv_conv2d_igemm(C.byref(d), s))
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    return t, 2.0 * n * h * w * cin * cout * 9 / t / 1e12


for shp in [(64, 256, 256, 64, 64), (64, 128, 128, 64, 64), (64, 128, 128, 192, 64), (64, 64, 64, 128, 128), (64, 64, 64, 384, 128), (64, 32, 32, 256, 256), (64, 32, 32, 768, 256),
            (64, 16, 16, 512, 512), (64, 8, 8, 512, 1024)]:
    r = [run(*shp, dt) for dt in (BF16, FP8, FP8X)]
    print(f'n{shp[0]} {shp[1]}x{shp[2]} {shp[3]}->{shp[4]}: ' + '   '.join(f'{nm} {t*1e6:7.1f} us {tf:7.1f} TF/s' for nm, (t, tf) in zip(('bf16', 'fp8', 'fp8x'), r)), flush=True)
