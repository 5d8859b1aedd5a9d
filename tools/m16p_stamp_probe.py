"""Interval anatomy of the persistent 16x16x32 kernel (s_memtime stamps, diagnostic build): build with
    bash tools/scripts/build_m16p_variants.sh "-DSATCV_STAMP_M16P"      ->  libsatcv_m16pDSATCV_STAMP_M16P.so
and run  SATCV_LIB=<that .so> python tools/m16p_stamp_probe.py n,h,w,cin,cout [--affine].  Prints, per wave of workgroup 0, the mean cycles (s_memtime
counts at 100 MHz: x (clock / 100 MHz) shader cycles) per interval spent in its work, in the wait for the weight pieces and at the barrier."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from satellite_computervision_amd import ops
from satellite_computervision_amd._lib import lib, check
shape = tuple(int(v) for v in sys.argv[1].split(','))
aff = '--affine' in sys.argv
check(lib.satcv_set_option(b'igemm_m16', 2))
n, h, w, cin, cout = shape
dev = torch.device('cuda')
x = torch.randn(n, h, w, cin, device=dev).to(torch.bfloat16)
kern = torch.randn(3, 3, cin, cout, device=dev) * 0.1
wf, _ = ops.pack_weights(kern, cin, 1, want_dgrad=False)
st = ops.new_stats(cout, dev)
sc = torch.rand(cin, device=dev) + 0.5 if aff else None
sh = torch.randn(cin, device=dev) * 0.1 if aff else None
for _ in range(3):
    y = ops.conv2d(x, wf, cout, stats=st, in_scale=sc, in_shift=sh, in_relu=aff)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (8 * 12 * 4))()
lib.satcv_debug_read_stamps_m16p.argtypes = [C.c_void_p]
assert lib.satcv_debug_read_stamps_m16p(buf) == 0
for wg in (0, 1):
    print(f'workgroup {wg}:')
    for wv in range(12):
        v = [buf[(wg * 12 + wv) * 4 + i] for i in range(4)]
        k = max(v[3], 1)
        role = 'matrix ' if wv < 8 else 'staging'
        print(f'  wave {wv:2d} {role}: intervals {v[3]:4d}   work {v[0] / k:8.1f}   weight wait {v[1] / k:8.1f}   barrier {v[2] / k:8.1f}   (memtime ticks per interval; total {sum(v[:3]) / k:8.1f})')
