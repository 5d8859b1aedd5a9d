"""GPU probe: time conv shapes through the C ABI with HIP events.  Variants: --opt key=value (satcv_set_option), SATCV_LIB=<ablation build>
(python -m satellite_computervision_amd.build -DSATCV_ABLATE=bits).

    python tools/conv_probe.py [--opt igemm_db=2] [--shapes deep|thin|all|n,h,w,cin,cout[,k] ...] [--affine] [--reps 30]
"""
import argparse, sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from satellite_computervision_amd import ops
from satellite_computervision_amd._lib import lib, check
ap = argparse.ArgumentParser()
ap.add_argument('--opt', action='append', default=[])
ap.add_argument('--shapes', nargs='*', default=['deep'])
ap.add_argument('--affine', action='store_true', help='fused input BN affine + ReLU (decoder layers)')
ap.add_argument('--nostats', action='store_true')
ap.add_argument('--reps', type=int, default=30)
args = ap.parse_args()
for kv in args.opt:
    k, v = kv.split('=')
    check(lib.satcv_set_option(k.encode(), int(v)))
dev = torch.device('cuda')
DEEP = [(64, 64, 64, 128, 128), (64, 32, 32, 256, 256), (64, 32, 32, 512, 256), (64, 16, 16, 512, 512), (64, 16, 16, 1024, 512), (64, 8, 8, 512, 1024)]
THIN = [(64, 256, 256, 16, 32), (64, 256, 256, 32, 32), (64, 256, 256, 64, 32), (64, 256, 256, 32, 64), (64, 128, 128, 32, 64), (64, 128, 128, 64, 64),
        (64, 128, 128, 128, 64), (64, 128, 128, 64, 128), (64, 128, 128, 64, 32)]
shapes = []
for s in args.shapes:
    shapes += DEEP if s == 'deep' else THIN if s == 'thin' else DEEP + THIN if s == 'all' else [tuple(int(v) for v in s.split(','))]


def run(n, h, w, cin, cout, k=3):
    x = torch.randn(n, h, w, cin, device=dev).to(torch.bfloat16)
    kern = torch.randn(k, k, cin, cout, device=dev) * 0.1
    wf, _ = ops.pack_weights(kern, cin, 1, want_dgrad=False)
    st = None if args.nostats else ops.new_stats(cout, dev)
    y = torch.empty(n, h, w, cout, device=dev, dtype=torch.bfloat16)
    b = torch.zeros(cout, device=dev)
    sc = torch.rand(cin, device=dev) + 0.5 if args.affine else None
    sh = torch.randn(cin, device=dev) * 0.1 if args.affine else None
    d = ops.make_conv_desc(x0=x.data_ptr(), c0=cin, w=wf.data_ptr(), y=y.data_ptr(), ldy=cout, n=n, h=h, w_=w, cout=cout, cout_pad=ops.rup(cout, 32),
                           dtype=1, bias=b.data_ptr(), stats=st.data_ptr() if st is not None else None, stats_ld=cout, kh=k, kw=k,
                           in_scale=sc.data_ptr() if args.affine else None, in_shift=sh.data_ptr() if args.affine else None, in_relu=1 if args.affine else 0)
    s = ops.stream_ptr()
    f = lambda: check(lib.satcv_conv2d_igemm(C.byref(d), s))
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps): f()
    e1.record()
    torch.cuda.synchronize(); t = e0.elapsed_time(e1) / args.reps * 1e-3
    fl = 2.0 * n * h * w * cin * cout * k * k
    by = n * h * w * (cin + cout) * 2
    print(f'  n{n} {h}x{w} {cin}->{cout} k{k}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s  {by/t/1e12:5.2f} TB/s(alg)', flush=True)


print('lib', os.environ.get('SATCV_LIB'), 'opts', args.opt, 'affine', args.affine)
for shp in shapes:
    run(*shp)
_v = C.c_int32()
if lib.satcv_get_option(b'm16p_launches', C.byref(_v)) == 0:
    print('launches on the persistent 16x16x32 kernel:', _v.value)
