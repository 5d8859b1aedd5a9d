"""GPU probe: achievable HBM bandwidth of torch copy vs the BN-backward kernels on a dec0-sized tensor."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from satellite_computervision_amd import ops
dev = torch.device('cuda')
n, h, w, c = 64, 256, 256, 32
x = torch.randn(n, h, w, c, device=dev).to(torch.bfloat16)
g = torch.randn(n, h, w, c, device=dev).to(torch.bfloat16)
y = torch.empty_like(x)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
nbytes = x.numel() * 2
t = timeit(lambda: y.copy_(x)); print(f'torch copy      : {t*1e6:8.1f} us  {2*nbytes/t/1e12:.2f} TB/s (r+w)')
t = timeit(lambda: torch.add(x, g, out=y)); print(f'torch add (2r+1w): {t*1e6:8.1f} us  {3*nbytes/t/1e12:.2f} TB/s')
t = timeit(lambda: x.float().sum()); print(f'torch sum        : {t*1e6:8.1f} us')
sc = torch.rand(c, device=dev) + 0.5; sh = torch.randn(c, device=dev); mean = torch.randn(c, device=dev); rstd = torch.rand(c, device=dev) + 0.5
import ctypes as C
from satellite_computervision_amd._lib import lib, check
sums = ops.new_stats(c, dev); coef = torch.zeros(2, c, device=dev); dy = torch.empty_like(x)
d = ops.make_bnbwd_desc(yraw=x.data_ptr(), ldy=c, scale=sc.data_ptr(), shift=sh.data_ptr(), mean=mean.data_ptr(), rstd=rstd.data_ptr(), n=n, h=h, w_=w, c=c,
                        dtype=1, da=g.data_ptr(), ldda=c, sums=sums.data_ptr(), sums_ld=c, coef=coef.data_ptr(), dy=dy.data_ptr(), lddy_out=c)
st = ops.stream_ptr()
t = timeit(lambda: check(lib.satcv_bn_bwd_reduce(C.byref(d), st))); print(f'bn_bwd reduce (2r): {t*1e6:8.1f} us  {2*nbytes/t/1e12:.2f} TB/s')
t = timeit(lambda: check(lib.satcv_bn_bwd_apply(C.byref(d), st))); print(f'bn_bwd apply (2r+1w): {t*1e6:8.1f} us  {3*nbytes/t/1e12:.2f} TB/s')
act, pooled = None, None
t = timeit(lambda: ops.bn_relu_pool(x, sc, sh, 2)); print(f'bn_relu_pool (1r+1.25w, incl alloc): {t*1e6:8.1f} us  {2.25*nbytes/t/1e12:.2f} TB/s')
