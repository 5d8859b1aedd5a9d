"""GPU probe: the folded inference plan (bf16 or fp8 storage) of the five-level U-Net at batch 64 of 256 x 256 x 4 -- run a few passes under
`rocprofv3 --kernel-trace --stats` to see where an inference pass spends its time.

    rocprofv3 --kernel-trace --stats --output-format csv -d out -o inf -- python3 tools/infer_profile.py [--fp8] [--reps 5]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
ap = argparse.ArgumentParser()
ap.add_argument('--fp8', action='store_true')
ap.add_argument('--reps', type=int, default=5)
ap.add_argument('--batch', type=int, default=64)
args = ap.parse_args()
from satellite_computervision_amd import model_tools as mt, fp8_infer
mt.set_compute_dtype('bfloat16')
model = mt.get_unet_model(2, 4)
rng = np.random.default_rng(0)
x = rng.random((args.batch, 256, 256, 4)).astype(np.float32)
q = fp8_infer.calibrate(model, x[:8]) if args.fp8 else None
plan = fp8_infer.Fp8Plan(model, args.batch, 256, 256, q, store=fp8_infer.FP8 if args.fp8 else fp8_infer.BF16)
xt = torch.from_numpy(x).cuda()
for t, xin in plan.x_by_tid.items():
    xin.copy_(xt)
from satellite_computervision_amd import ops
st = ops.stream_ptr()
for _ in range(2):
    [f(st) for f in plan.fwd]
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(args.reps):
    [f(st) for f in plan.fwd]
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / args.reps
print(f'{"fp8" if args.fp8 else "bf16"} folded inference: {ms:.3f} ms per batch of {args.batch} = {args.batch / ms * 1e3:.0f} tiles/s')
