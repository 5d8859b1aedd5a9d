"""GPU probe: time one conv shape through the C ABI with HIP events (ablation via env SATCV_DBG)."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from satellite_computervision_amd import ops
from satellite_computervision_amd._lib import lib, check
dev = torch.device('cuda')
def run(n, h, w, cin, cout, stats=True, reps=40, k=3):
    x = torch.randn(n, h, w, cin, device=dev).to(torch.bfloat16)
    kern = torch.randn(k, k, cin, cout, device=dev) * 0.1
    wf, _ = ops.pack_weights(kern, cin, 1, want_dgrad=False)
    st = ops.new_stats(cout, dev) if stats else None
    y = torch.empty(n, h, w, cout, device=dev, dtype=torch.bfloat16)
    b = torch.zeros(cout, device=dev)
    d = ops.make_conv_desc(x0=x.data_ptr(), c0=cin, w=wf.data_ptr(), y=y.data_ptr(), ldy=cout, n=n, h=h, w_=w, cout=cout, cout_pad=ops.rup(cout, 32),
                           dtype=1, bias=b.data_ptr(), stats=st.data_ptr() if stats else None, stats_ld=cout, kh=k, kw=k)
    s = ops.stream_ptr()
    f = lambda: check(lib.satcv_conv2d_igemm(C.byref(d), s))
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.3: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(reps): f()
    e1.record(); th = (time.perf_counter() - t0) / reps
    torch.cuda.synchronize(); t = e0.elapsed_time(e1) / reps * 1e-3
    fl = 2.0 * n * h * w * cin * cout * k * k
    by = n * h * w * (cin + cout) * 2
    print(f'  n{n} {h}x{w} {cin}->{cout} k{k} stats={int(stats)}: gpu {t*1e6:8.1f} us (host {th*1e6:6.1f} us/call)  {fl/t/1e12:7.1f} TF/s  {by/t/1e12:5.2f} TB/s(alg)', flush=True)
print('lib', os.environ.get('SATCV_LIB'))
# the 3x3 shapes of get_unet_model(2, 4) at batch 64 (forward and data gradient), thin to deep
SHAPES = [(64, 256, 256, 16, 32), (64, 256, 256, 32, 32), (64, 256, 256, 64, 32), (64, 256, 256, 32, 64), (64, 128, 128, 32, 64), (64, 128, 128, 64, 64),
            (64, 128, 128, 128, 64), (64, 128, 128, 64, 128), (64, 128, 128, 64, 32), (64, 64, 64, 128, 128), (64, 32, 32, 256, 256), (64, 16, 16, 512, 512)]
if os.environ.get('PROBE_1X1'):
    SHAPES = [(64, 256, 256, 32, 32, True, 40, 1), (64, 256, 256, 32, 32, False, 40, 1), (64, 256, 256, 64, 64, True, 40, 1), (64, 256, 256, 32, 32, True, 40, 3), (64, 256, 256, 32, 32, False, 40, 3)]
if os.environ.get('PROBE_DEEP'):
    SHAPES = [(64, 64, 64, 128, 128), (64, 32, 32, 256, 256), (64, 32, 32, 768, 256), (64, 16, 16, 512, 512)]
for shp in SHAPES:
    run(*shp)
