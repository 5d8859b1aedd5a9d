"""GPU probe: every launch of one U-Net training step timed IN ISOLATION (HIP events, back-to-back repeats of the same
launch), with its algorithmic FLOPs / bytes and the per-layer roofline max(flops / 2.5 PF, bytes / 8 TB/s).

    python tools/step_probe.py [--batch 64] [--dtype bfloat16] [--reps 20] [--json out.json]

The table is what DESIGN.md section 3/6 quotes; commit its output under profiles/.

Replaying single launches leaves the BatchNorm statistics / backward sums buffers with several launches' worth of sums (the
finalize kernels that clear them are not replayed): one whole training step runs at the end so that the model is usable again,
but do not draw numerical conclusions from a process that ran this probe.
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('SATCV_WGRAD_STREAM', '0')           # every launch on the current stream
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--tile', type=int, default=256)
ap.add_argument('--channels', type=int, default=4)
ap.add_argument('--dtype', default='bfloat16')
ap.add_argument('--reps', type=int, default=20)
ap.add_argument('--json', default=None)
ap.add_argument('--only', default=None, help='substring filter on the label')
ap.add_argument('--opt', action='append', default=[], help='kernel-selection knob key=value (satcv_set_option), repeatable')
args = ap.parse_args()

from satellite_computervision_amd import model_tools as mt, ops
from satellite_computervision_amd._lib import lib, check
# the launches are replayed one by one outside Plan._run, which raises option igemm_m16 to 2 around a training plan's steps (every eligible
# launch on the 16x16x32 tiles, not only those that write statistics): do the same here, so that the probe times the kernels the step runs
check(lib.satcv_set_option(b'igemm_m16', 2))
for kv in args.opt:
    k, v = kv.split('=')
    check(lib.satcv_set_option(k.encode(), int(v)))
mt.reset_uids(); mt.set_seed(0); mt.set_compute_dtype(args.dtype)
model = mt.get_unet_model(2, args.channels)
model.compile(optimizer=mt.Adam(9e-4), loss=lambda a, b: mt.weighted_categorical_crossentropy(a, b, [1.0, 20.0]))
rng = np.random.default_rng(0)
B, T = args.batch, args.tile
x = torch.from_numpy(rng.beta(2, 5, (B, T, T, args.channels)).astype(np.float32)).cuda()
y = torch.from_numpy(np.eye(2, dtype=np.float32)[(rng.random((B, T, T)) < 0.05).astype(np.int64)]).cuda()
for _ in range(2):
    plan = model.train_step_device(x, y)
torch.cuda.synchronize()
PEAK_F = 2.5e15 if args.dtype == 'bfloat16' else 157.3e12
PEAK_B = 8.0e12
rows = []
st = ops.stream_ptr()
for phase, steps in (('fwd', plan.fwd), ('bwd', plan.bwd)):
    for fn in steps:
        label = getattr(fn, 'label', None)
        if label is None or (args.only and args.only not in label):
            continue
        for _ in range(3):
            fn(st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn(st)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / args.reps * 1e3
        w = fn.work
        es = w['esize']
        if w['kind'] == 'conv':
            fl = 2.0 * w['px'] * w['cin'] * w['cout'] * w['taps'] if 'd2s' not in label else 2.0 * w['px'] * w['cin'] * w['cout']
            by = w['px'] * (w['cin'] + w['cout']) * es + w['taps'] * w['cin'] * w['cout'] * es
            if 'd2s' in label or 's2d' in label:      # transposed conv: cout counts all f*f sub-pixel positions already
                by = w['px'] * (w['cin'] + w['cout']) * es
        elif w['kind'] == 'wgrad':
            fl = 2.0 * w['px'] * w['cin'] * w['cout'] * w['taps']
            by = w['px'] * (w['cin'] + w['cout'] * (w['taps'] if 'convT' in label else 1)) * es
        elif w['kind'] == 'bwd_fused':          # BatchNorm apply + data gradient + weight gradient in one launch: g, y, x in, dx out
            fl = 4.0 * w['px'] * w['cin'] * w['cout'] * w['taps']
            by = w['px'] * (2 * w['cout'] + 2 * w['cin']) * es
        elif w['kind'] == 'convt_bwd_fused':    # concat-BN apply of the `up` half + transposed-conv data + weight gradient: g, y_up (4 px x cout each), x in, dx out
            fl = 4.0 * w['px'] * w['cin'] * 4 * w['cout']
            by = (2 * 4 * w['px'] * w['cout'] + 2 * w['px'] * w['cin']) * es
        elif w['kind'] == 'bn_bwd_reduce':
            fl, by = 0.0, 2 * w['px'] * w['c'] * es
        else:
            fl, by = 0.0, 3 * w['px'] * w['c'] * es
        roof_us = max(fl / PEAK_F, by / PEAK_B) * 1e6
        rows.append(dict(phase=phase, label=label, us=round(us, 1), gflop=round(fl / 1e9, 2), mbytes=round(by / 1e6, 1),
                         tflops=round(fl / us / 1e6, 1), tbps=round(by / us / 1e6, 2), roof_us=round(roof_us, 1), frac=round(roof_us / us, 3)))
        r = rows[-1]
        print(f"{phase} {label:58s} {r['us']:8.1f} us {r['tflops']:7.1f} TF/s {r['tbps']:5.2f} TB/s(alg)  roof {r['roof_us']:6.1f} us  frac {r['frac']:.2f}", flush=True)

def tot(pred):
    sel = [r for r in rows if pred(r)]
    return sum(r['us'] for r in sel), sum(r['gflop'] for r in sel), sum(r['roof_us'] for r in sel), len(sel)
for name, pred in (('3x3 fwd+dgrad', lambda r: r['label'].startswith('conv_') and ' k3 ' in r['label']),
                   ('1x1/convT', lambda r: r['label'].startswith('conv_') and ' k1 ' in r['label']),
                   ('fused bwd', lambda r: r['label'].startswith('bwd_fused')),
                   ('convT fused bwd', lambda r: r['label'].startswith('convt_bwd_fused')),
                   ('wgrad', lambda r: r['label'].startswith('wgrad')),
                   ('bn_bwd', lambda r: r['label'].startswith('bn_bwd'))):
    us, gf, roof, cnt = tot(pred)
    if cnt:
        print(f"TOTAL {name:15s}: {cnt:3d} launches {us:9.1f} us  {gf / max(us, 1e-9) / 1e3:7.1f} TF/s  sum-of-rooflines {roof:8.1f} us  frac {roof / us:.3f}")
for t in plan.keep:            # sums buffers are the float64 tensors of the plan: clear them, then one clean step
    if isinstance(t, torch.Tensor) and t.dtype == torch.float64:
        t.zero_()
model.train_step_device(x, y)
torch.cuda.synchronize()
if args.json:
    json.dump(dict(args=vars(args), rows=rows), open(args.json, 'w'), indent=1)
