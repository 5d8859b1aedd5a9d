"""GPU soak: the bench workload (five-level U-Net, 256 x 256 x 4, batch 64, bf16) trained for a few hundred steps on a learnable synthetic task
(class = a threshold on two bands); prints the loss every 50 steps and the final mask IoU on held-out tiles.  A guard against kernels that
are fast and subtly wrong: the loss must fall and stay finite, the IoU must end high.

    python tools/soak_train.py [--steps 300]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=300)
ap.add_argument('--batch', type=int, default=64)
args = ap.parse_args()
from satellite_computervision_amd import model_tools as mt
mt.set_compute_dtype('bfloat16')
mt.reset_uids(); mt.set_seed(3)
model = mt.get_unet_model(2, 4)
model.compile(optimizer=mt.Adam(9e-4), loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, [1.0, 2.0]))
rng = np.random.default_rng(0)


def batch(b):
    x = rng.random((b, 256, 256, 4)).astype(np.float32)
    # smooth the bands so that the task needs spatial context, label = threshold on a mix of two smoothed bands
    t = torch.from_numpy(x).permute(0, 3, 1, 2)
    t = torch.nn.functional.avg_pool2d(t, 9, 1, 4)
    xs = t.permute(0, 2, 3, 1).contiguous().numpy()
    lab = ((xs[..., 0] + xs[..., 3]) > 1.0).astype(np.int64)
    return torch.from_numpy(x).cuda(), torch.from_numpy(np.eye(2, dtype=np.float32)[lab]).cuda(), lab


pool = [batch(args.batch) for _ in range(4)]
losses = []
for step in range(args.steps):
    x, y, _ = pool[step % len(pool)]
    losses.append(model.train_on_batch(x, y))
    if step % 50 == 0 or step == args.steps - 1:
        print(f'step {step:4d} loss {losses[-1]:.5f}', flush=True)
assert all(np.isfinite(losses)), 'non-finite loss'
xh, _, lab = batch(16)
pred = model.predict(xh.cpu().numpy(), batch_size=16)
cls = np.asarray(pred[1] if isinstance(pred, (list, tuple)) else pred.argmax(-1)).reshape(lab.shape)
inter = np.logical_and(cls == 1, lab == 1).sum(); union = np.logical_or(cls == 1, lab == 1).sum()
print(f'first loss {losses[0]:.4f} last loss {losses[-1]:.4f} held-out IoU {inter / max(union, 1):.4f}')
assert losses[-1] < 0.5 * losses[0], 'the loss did not fall'
