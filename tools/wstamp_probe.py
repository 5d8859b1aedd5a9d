"""diagnostic: per-phase cycle shares of the weight-gradient kernel's pixel-tile loop (needs SATCV_LIB = a -DSATCV_STAMP build)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from satellite_computervision_amd import ops
from satellite_computervision_amd._lib import lib, check
dev = torch.device('cuda')
aff = os.environ.get('AFF', '1') == '1'
for (n, h, w, cin, cout) in [tuple(int(v) for v in s.split(',')) for s in sys.argv[1:]]:
    x = torch.randn(n, h, w, cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(n, h, w, cout, device=dev).to(torch.bfloat16)
    sc = torch.rand(cin, device=dev) + 0.5; sh = torch.randn(cin, device=dev) * 0.1
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(6):
        if i == 1: e0.record()
        dk = ops.conv2d_wgrad(x, dy, cin, cout, in_scale=sc if aff else None, in_shift=sh if aff else None, in_relu=aff)
    e1.record(); torch.cuda.synchronize()
    buf = (C.c_ulonglong * (8 * 4 * 8))()
    lib.satcv_debug_read_wstamps.argtypes = [C.c_void_p]
    assert lib.satcv_debug_read_wstamps(buf) == 0
    print(f'{n}x{h}x{w} {cin}->{cout} affine={aff}: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us/call (incl. reduce);  per tile: load-issue | mfma | barrier | store | barrier ; setup ; tiles ; tail')
    for b in (0, 5):
        for wv in range(4):
            v = [buf[(b * 4 + wv) * 8 + i] for i in range(8)]
            print(f'  b{b} w{wv}: ' + ' '.join(f'{x_:7d}' for x_ in v[:5]) + f' ; setup {v[5]:6d} ; tiles {v[6]:4d} ; tail {v[7]:7d}')
