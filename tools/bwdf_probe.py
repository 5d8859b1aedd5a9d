"""GPU probe: the fused thin-layer backward (satcv_conv2d_bwd_fused) on the U-Net's level-0 / level-1 shapes at batch 64, against the
three launches it replaces (satcv_bn_bwd_apply + data gradient + weight gradient), each timed stand-alone with HIP events.

    python tools/bwdf_probe.py [--batch 64] [--reps 20]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--reps', type=int, default=20)
args = ap.parse_args()
from satellite_computervision_amd import ops
from satellite_computervision_amd._lib import lib, check

dev = torch.device('cuda')
td = torch.bfloat16


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (hw, c0, c1, cout) in ((256, 32, 0, 32), (256, 32, 32, 32), (128, 64, 0, 64)):
    n, cin = args.batch, c0 + c1
    g = torch.randn(n, hw, hw, cout, device=dev).to(td)
    v = torch.randn(n, hw, hw, cout, device=dev).to(td)
    x0 = torch.randn(n, hw, hw, c0, device=dev).to(td)
    x1 = torch.randn(n, hw, hw, c1, device=dev).to(td) if c1 else None
    kern = torch.randn(3, 3, cin, cout, device=dev) * 0.1
    _, wd = ops.pack_weights(kern, cin, ops.DTYPE_CODE[td])
    one = lambda c, s=1.0: (torch.rand(c, device=dev) + 0.5) * s
    sc, sh, mu, rs, isc, ish = one(cout), one(cout, 0.1), one(cout, 0.1), one(cout), one(cin), one(cin, 0.1)
    coef = torch.randn(2, cout, device=dev) * 0.1
    dx = torch.empty(n, hw, hw, cin, dtype=td, device=dev)
    dw = torch.empty(3, 3, cin, cout, device=dev)
    d = ops.make_bwdf_desc(g=g.data_ptr(), yraw=v.data_ptr(), ldg=cout, bn_scale=sc.data_ptr(), bn_shift=sh.data_ptr(), bn_mean=mu.data_ptr(),
                           bn_rstd=rs.data_ptr(), bn_coef=coef.data_ptr(), x0=x0.data_ptr(), c0=c0, x1=x1.data_ptr() if c1 else None, c1=c1,
                           in_scale=isc.data_ptr(), in_shift=ish.data_ptr(), in_relu=1, w_dgrad=wd.data_ptr(), dx=dx.data_ptr(), lddx=cin,
                           dw=dw.data_ptr(), cin=cin, cout=cout, n=n, h=hw, w_=hw, dtype=ops.DTYPE_CODE[td])
    nb = lib.satcv_conv2d_bwd_fused_workspace(C.byref(d))
    label = f'n{n} {hw}x{hw} {c0}+{c1}->{cout}'
    px = n * hw * hw
    if nb < 0:
        print(f'{label}: not served by the fused kernel')
        continue
    ws = torch.empty(nb // 4, device=dev)
    d.workspace, d.workspace_bytes = ws.data_ptr(), nb
    st = ops.stream_ptr()
    us = timed(lambda: check(lib.satcv_conv2d_bwd_fused(C.byref(d), st)), args.reps)
    # the three launches
    dy = torch.empty_like(v)
    bd = ops.make_bnbwd_desc(yraw=v.data_ptr(), ldy=cout, scale=sc.data_ptr(), shift=sh.data_ptr(), mean=mu.data_ptr(), rstd=rs.data_ptr(), n=n, h=hw, w_=hw,
                             c=cout, dtype=ops.DTYPE_CODE[td], da=g.data_ptr(), ldda=cout, sums=ops.new_stats(cout, dev).data_ptr(), sums_ld=cout,
                             coef=coef.data_ptr(), dy=dy.data_ptr(), lddy_out=cout)
    t_app = timed(lambda: check(lib.satcv_bn_bwd_apply(C.byref(bd), st)), args.reps)
    t_dg = timed(lambda: ops.conv2d_dgrad(dy, wd, cin, out=dx), args.reps)
    t_wg = timed(lambda: ops.conv2d_wgrad(x0, dy, cin, cout, x1=x1, in_scale=isc, in_shift=ish, in_relu=True, dw=dw), args.reps)
    by = px * (2 * cout + 2 * cin) * 2
    print(f'{label}: fused {us:7.1f} us ({by / us / 1e6:.2f} TB/s over g, y, x, dx; {4.0 * px * cin * cout * 9 / us / 1e6:.0f} TF/s)   '
          f'three launches {t_app + t_dg + t_wg:7.1f} us (apply {t_app:.1f} + dgrad {t_dg:.1f} + wgrad {t_wg:.1f})', flush=True)
