"""GPU probe: every convolution launch of the DeepLab-v3 / ResNet-50 inference plan (BASELINE configs[2], 512 x 512 x 4) timed IN
ISOLATION with its algorithmic FLOPs / bytes and roofline max(flops / 2.5 PF, bytes / 8 TB/s); per-class totals at the end.

    B=1 python tools/deeplab_probe.py      (B=16 for the batch the bench line also reports)

(Launches that write a residual join in place keep adding into their output while they are replayed: timing only.)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from satellite_computervision_amd import model_tools as mt, ops
from satellite_computervision_amd._lib import lib
B = int(os.environ.get('B', '1'))
mt.reset_uids(); mt.set_seed(0); mt.set_compute_dtype('bfloat16')
m = mt.get_deeplabv3_model(2, 4)
x = torch.rand(B, 512, 512, 4, device='cuda')
for _ in range(3):
    m.predict_on_device(x)
torch.cuda.synchronize()
plan = m._infer_plan(B, 512, 512)
if getattr(m, '_infer_splitk', False) and B <= 2:
    lib.satcv_set_option(b'splitk', 2)
st = ops.stream_ptr()
rows = []
for fn in plan.fwd:
    label, w = getattr(fn, 'label', None), getattr(fn, 'work', None)
    if label is None or w is None or w.get('kind') != 'conv':
        continue
    for _ in range(3):
        fn(st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn(st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    fl = 2.0 * w['px'] * w['cin'] * w['cout'] * w['taps']
    by = (w['px'] * (w['cin'] + w['cout']) + w['taps'] * w['cin'] * w['cout']) * w['esize']
    roof = max(fl / 2.5e15, by / 8e12) * 1e6
    cls = '1x1' if w['taps'] == 1 else ('3x3 dilated / strided / 7x7 (tap loop)' if (' d1 ' not in label or w['taps'] != 9 or 'stride' in label) else '3x3')
    rows.append((cls, label, us, fl, by, roof))
    print(f"{label:62s} {us:8.1f} us {fl / us / 1e6:7.1f} TF/s {by / us / 1e6:5.2f} TB/s(alg)  roof {roof:6.1f} us  frac {roof / us:.2f}", flush=True)
print(f'== DeepLab-v3 / ResNet-50, batch {B} x 512 x 512 x 4, bf16: convolution launches in isolation')
tot_us = sum(r[2] for r in rows)
for cls in sorted(set(r[0] for r in rows)):
    sel = [r for r in rows if r[0] == cls]
    us, fl, roof = sum(r[2] for r in sel), sum(r[3] for r in sel), sum(r[5] for r in sel)
    print(f'TOTAL {cls:42s}: {len(sel):3d} launches {us:9.1f} us ({100 * us / tot_us:4.1f} %)  {fl / 1e9:8.1f} GFLOP  {fl / us / 1e6:7.1f} TF/s  sum-of-rooflines {roof:7.1f} us  frac {roof / us:.3f}')
fl = sum(r[3] for r in rows)
print(f'TOTAL all convolutions: {len(rows)} launches {tot_us:9.1f} us  {fl / 1e9:.1f} GFLOP per pass = {fl / 1e9 / B:.1f} per tile  {fl / tot_us / 1e6:.1f} TF/s = {fl / tot_us / 1e6 / 2500:.3f} of the dense bf16 peak')
