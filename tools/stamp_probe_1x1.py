"""diagnostic: phases of the 1x1 / transposed-conv forward tile (needs SATCV_LIB = a -DSATCV_STAMP build)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from satellite_computervision_amd import ops
from satellite_computervision_amd._lib import lib, check
dev = torch.device('cuda')
for (n, h, w, cin, cout, f) in [tuple(int(v) for v in s.split(',')) for s in sys.argv[1:]]:
    x = torch.randn(n, h, w, cin, device=dev).to(torch.bfloat16)
    kt = torch.randn(f, f, cout, cin, device=dev) * 0.1
    wf, _ = ops.pack_weights(kt, cin, 1, transposed=True)
    y = ops.conv2d_transpose(x, wf, cout, f)
    for _ in range(20):
        y = ops.conv2d_transpose(x, wf, cout, f)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (8 * 8 * 8))()
    lib.satcv_debug_read_stamps.argtypes = [C.c_void_p]
    assert lib.satcv_debug_read_stamps(buf) == 0
    nch = max(cin // 32, 1)
    print(f'{n}x{h}x{w} {cin}->{f}x{f}x{cout}: per chunk (issue | compute | barrier1 | store | barrier2); setup | first load+store | epilogue')
    for b in (0, 5):
        for wv in range(4):
            v = [buf[(b * 8 + wv) * 8 + i] for i in range(8)]
            if sum(v) == 0:
                continue
            per = [v[0] / nch, v[1] / nch, v[2] / nch, v[3] / nch, v[7] / nch]
            print(f'  b{b} w{wv}: ' + ' '.join(f'{x_:7.0f}' for x_ in per) + f' | loop {sum(per) * nch:8.0f} | setup {v[4]:6d} first {v[5]:6d} epilogue {v[6]:6d}')
