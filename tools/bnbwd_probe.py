"""Isolated timing of the BN-backward reduce / apply passes (HBM-bound; tools for DESIGN.md section 6)."""
import ctypes as C
import torch
from satellite_computervision_amd import ops
from satellite_computervision_amd._lib import lib, check, BF16

def run(n, h, w, c, pool=False):
    dev = 'cuda'
    y = torch.randn(n, h, w, c, device=dev).to(torch.bfloat16)
    da = torch.randn(n, h, w, c, device=dev).to(torch.bfloat16)
    dy = torch.empty_like(y)
    sc = torch.ones(c, device=dev); sh = torch.zeros(c, device=dev); mu = torch.zeros(c, device=dev); rs = torch.ones(c, device=dev)
    sums = ops.new_stats(c, dev); coef = torch.zeros(2, c, device=dev); dbias = torch.zeros(c, device=dev)
    kw = dict(yraw=y.data_ptr(), ldy=c, scale=sc.data_ptr(), shift=sh.data_ptr(), mean=mu.data_ptr(), rstd=rs.data_ptr(), n=n, h=h, w_=w, c=c, dtype=BF16,
              da=da.data_ptr(), ldda=c)
    if pool:
        dp = torch.randn(n, h // 2, w // 2, c, device=dev).to(torch.bfloat16)
        kw.update(dpool=dp.data_ptr(), lddp=c, f=2)
    dr = ops.make_bnbwd_desc(sums=sums.data_ptr(), sums_ld=c, **kw)
    da_ = ops.make_bnbwd_desc(coef=coef.data_ptr(), dy=dy.data_ptr(), lddy_out=c, dbias=dbias.data_ptr(), **kw)
    st = ops.stream_ptr()
    S = y.numel() * 2 / 1e9
    for name, fn, d, nb in (('reduce', lib.satcv_bn_bwd_reduce, dr, 2 + (0.25 if pool else 0)), ('apply', lib.satcv_bn_bwd_apply, da_, 3 + (0.25 if pool else 0))):
        for _ in range(3):
            check(fn(C.byref(d), st))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            check(fn(C.byref(d), st))
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print(f'n{n} {h}x{w}x{c} pool={int(pool)} {name:6s} {us:7.1f} us  {nb * S / us * 1e6 / 1e3:6.2f} TB/s')

def pair(n, h, w, c):
    dev = 'cuda'
    y = torch.randn(n, h, w, c, device=dev).to(torch.bfloat16)
    da = torch.randn(n, h, w, c, device=dev).to(torch.bfloat16)
    dy = torch.empty_like(y)
    sc = torch.ones(c, device=dev); sh = torch.zeros(c, device=dev); mu = torch.zeros(c, device=dev); rs = torch.ones(c, device=dev)
    sums = ops.new_stats(c, dev); coef = torch.zeros(2, c, device=dev)
    kw = dict(yraw=y.data_ptr(), ldy=c, scale=sc.data_ptr(), shift=sh.data_ptr(), mean=mu.data_ptr(), rstd=rs.data_ptr(), n=n, h=h, w_=w, c=c, dtype=BF16,
              da=da.data_ptr(), ldda=c)
    dr = ops.make_bnbwd_desc(sums=sums.data_ptr(), sums_ld=c, **kw)
    da_ = ops.make_bnbwd_desc(coef=coef.data_ptr(), dy=dy.data_ptr(), lddy_out=c, **kw)
    st = ops.stream_ptr()
    def go():
        check(lib.satcv_bn_bwd_reduce(C.byref(dr), st)); check(lib.satcv_bn_bwd_apply(C.byref(da_), st))
    for _ in range(3):
        go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        go()
    e1.record(); torch.cuda.synchronize()
    print(f'pair n{n} {h}x{w}x{c}: {e0.elapsed_time(e1) * 100:7.1f} us')

for shp in [(64, 256, 256, 32), (64, 256, 256, 64), (64, 128, 128, 64), (64, 128, 128, 128), (64, 64, 64, 128)]:
    pair(*shp)
for shp in [(64, 256, 256, 32), (64, 256, 256, 64), (64, 128, 128, 64), (64, 128, 128, 128), (64, 64, 64, 128), (64, 32, 32, 512)]:
    run(*shp)
run(64, 256, 256, 32, True)
run(64, 128, 128, 64, True)
