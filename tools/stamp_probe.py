"""diagnostic: per-phase cycle shares of the double-buffered conv loop (needs SATCV_LIB = a -DSATCV_STAMP build)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from satellite_computervision_amd import ops
from satellite_computervision_amd._lib import lib, check
check(lib.satcv_set_option(b'igemm_db', int(os.environ.get('DB', '2'))))
check(lib.satcv_set_option(b'igemm_thin', int(os.environ.get('THIN', '1'))))
dev = torch.device('cuda')
for (n, h, w, cin, cout) in [tuple(int(v) for v in s.split(',')) for s in sys.argv[1:]]:
    x = torch.randn(n, h, w, cin, device=dev).to(torch.bfloat16)
    kern = torch.randn(3, 3, cin, cout, device=dev) * 0.1
    wf, _ = ops.pack_weights(kern, cin, 1, want_dgrad=False)
    y = torch.empty(n, h, w, cout, device=dev, dtype=torch.bfloat16)
    d = ops.make_conv_desc(x0=x.data_ptr(), c0=cin, w=wf.data_ptr(), y=y.data_ptr(), ldy=cout, n=n, h=h, w_=w, cout=cout, cout_pad=ops.rup(cout, 32), dtype=1, kh=3, kw=3)
    for _ in range(20):
        check(lib.satcv_conv2d_igemm(C.byref(d), ops.stream_ptr()))
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (8 * 8 * 8))()
    lib.satcv_debug_read_stamps.argtypes = [C.c_void_p]
    assert lib.satcv_debug_read_stamps(buf) == 0
    nch = max(cin // 16, 1)
    print(f'{n}x{h}x{w} {cin}->{cout}: cycles, DB: per chunk (wait | store+issue | compute | barrier); non-DB: per chunk (issue | compute | barrier1 | store | barrier2); then setup | first load+store | epilogue (per tile)')
    for b in (0, 5):
        for wv in range(8):
            v = [buf[(b * 8 + wv) * 8 + i] for i in range(8)]
            if sum(v) == 0:
                continue
            per = [v[0] / nch, v[1] / nch, v[2] / nch, v[3] / nch, v[7] / nch]
            if os.environ.get('THIN', '1') != '0' and cin <= 64 and cout <= 64:
                print(f'  b{b} w{wv} per tile: wait {v[0]:6d} store {v[1]:6d} bar {v[2]:6d} issue {v[3]:6d} mfma {v[4]:6d} bar {v[5]:6d} epilogue {v[6]:6d} bar {v[7]:6d}  total {sum(v):7d}')
                continue
            print(f'  b{b} w{wv}: ' + ' '.join(f'{x_:7.0f}' for x_ in per) + f' | loop {sum(per) * nch:8.0f} | setup {v[4]:6d} first {v[5]:6d} epilogue {v[6]:6d}')
