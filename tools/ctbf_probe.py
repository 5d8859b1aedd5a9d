"""GPU probe: satcv_convt_bwd_fused at the U-Net's decoder shapes (batch 64): time, algorithmic HBM rate (g, y, x in; dx out), TFLOP/s.
    python tools/ctbf_probe.py [--reps 20]"""
import argparse, sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from satellite_computervision_amd import ops
from satellite_computervision_amd._lib import lib, check
ap = argparse.ArgumentParser()
ap.add_argument('--reps', type=int, default=20)
ap.add_argument('--shapes', nargs='*', default=['64,128,128,64,32', '64,64,64,128,64', '64,32,32,256,128'])
args = ap.parse_args()
dev = torch.device('cuda')
for sp in args.shapes:
    n, h, w, cin, cout = (int(v) for v in sp.split(','))
    cs = cout
    bf = torch.bfloat16
    g = torch.randn(n, 2 * h, 2 * w, cs + cout, device=dev).to(bf)
    y = torch.randn(n, 2 * h, 2 * w, cout, device=dev).to(bf)
    x = torch.randn(n, h, w, cin, device=dev).to(bf)
    kt = torch.randn(2, 2, cout, cin, device=dev) * 0.1
    _, wd = ops.pack_weights(kt, cin, 1, transposed=True)
    vec = lambda c: (torch.rand(c, device=dev) + 0.5)
    sc, sh, mu, rs = vec(cs + cout), vec(cs + cout) - 1, vec(cs + cout) - 1, vec(cs + cout)
    coef = torch.randn(2 * (cs + cout), device=dev) * 0.01
    xsc, xsh, xmu, xrs = vec(cin), vec(cin) - 1, vec(cin) - 1, vec(cin)
    stats = ops.new_stats(cin, dev)
    dx = torch.empty(n, h, w, cin, dtype=bf, device=dev)
    dw = torch.empty(2, 2, cout, cin, device=dev)
    off = lambda v: v.data_ptr() + 4 * cs
    d = ops.make_ctbf_desc(g=g.data_ptr() + 2 * cs, ldg=cs + cout, yup=y.data_ptr(), ldy=cout, bn_scale=off(sc), bn_shift=off(sh), bn_mean=off(mu), bn_rstd=off(rs),
                           bn_c1=coef.data_ptr() + 4 * cs, bn_c2=coef.data_ptr() + 4 * (cs + cout + cs), x=x.data_ptr(), ldx=cin, w_dgrad=wd.data_ptr(),
                           w_npad=ops.rup(cin, 32), dx=dx.data_ptr(), lddx=cin, dw=dw.data_ptr(), cin=cin, cout=cout, n=n, h=h, w_=w, dtype=1,
                           in_scale=xsc.data_ptr(), in_shift=xsh.data_ptr(), in_relu=1, bst_sums=stats.data_ptr(), bst_sums_ld=cin, bst_mean=xmu.data_ptr(), bst_rstd=xrs.data_ptr())
    nb = lib.satcv_convt_bwd_fused_workspace(C.byref(d))
    ws = torch.empty(max(nb // 4, 1), device=dev)
    d.workspace, d.workspace_bytes = ws.data_ptr(), nb
    s = ops.stream_ptr()
    f = lambda: check(lib.satcv_convt_bwd_fused(C.byref(d), s))
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / args.reps * 1e-3
    P = n * h * w
    by = (2 * 4 * P * cout + 2 * P * cin) * 2
    fl = 4.0 * P * 4 * cout * cin
    print(f'  convT bwd fused n{n} {h}x{w} {cin}<-4x{cout}: {t*1e6:8.1f} us (incl. the slab sum)  {by/t/1e12:5.2f} TB/s(alg)  {fl/t/1e12:6.1f} TF/s  HBM roofline {by/8e12*1e6:6.1f} us', flush=True)
