"""Diagnostic (GPU): compare every intermediate gradient tensor of the device plan with the oracle's."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_model_gpu import build_pair
from satellite_computervision_amd import model_tools as mt
from oracle import losses as OL
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
o, m, names = build_pair(mt, 'float32', 2, 4, [32, 64], [2, 2])
rng = np.random.default_rng(0)
x = rng.random((n, S, S, 4)).astype(np.float32)
t = np.eye(2)[(rng.random((n, S, S)) < 0.3).astype(np.int64)].astype(np.float32)
m.compile(optimizer=mt.Adam(9e-4), loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, [1.0, 20.0]))
pr, _ = o.forward(x, training=True)
_, dprobs, _ = OL.weighted_categorical_crossentropy(t.astype(np.float64), pr, [1.0, 20.0])
o.backward(dprobs)
m.train_on_batch(x, t)
plan = m.runtime.plan(n, S, S, True)
inv = {}
for k, v in names.items():
    if k.endswith('.kernel'):
        inv[v[:-len('/kernel')]] = k[:-len('.kernel')]
    if k.endswith('.gamma'):
        inv[v[:-len('/gamma')]] = k[:-len('.gamma')]
for key, ten in plan.dbg.items():
    kind, lname = key.split(':')
    oname = f'{kind}:{inv.get(lname, lname)}'
    if oname not in o.dbg:
        print('no oracle tensor for', key, oname); continue
    ref = o.dbg[oname]
    got = ten.float().cpu().numpy()[..., :ref.shape[-1]]
    err = np.abs(got - ref)
    idx = np.unravel_index(err.argmax(), err.shape)
    print(f'{oname:26s} shape {str(ref.shape):22s} relmax {err.max()/max(np.abs(ref).max(),1e-30):9.2e} at {idx}  frac>1e-4 {(err > 1e-4*np.abs(ref).max()).mean():.4f}')

# ---- dissect the first bad BN backward (dec1.conv2)
lname = [v[:-len('/kernel')] for k, v in names.items() if k == 'dec1.conv2.kernel'][0]
cx = plan.dbg['_ctx:' + lname]
ref_dy = o.dbg['dy:dec1.conv2']
got = plan.dbg['dy:' + lname].float().cpu().numpy()
err = np.abs(got - ref_dy)
bad = np.argwhere(err > 1e-3 * np.abs(ref_dy).max())
print('bad elements', len(bad), 'first', bad[:12].tolist())
import collections
print('by channel', collections.Counter(bad[:, 3].tolist()).most_common(8))
print('by row', collections.Counter(bad[:, 1].tolist()).most_common(8))
print('by col', collections.Counter(bad[:, 2].tolist()).most_common(8))
print('by img', collections.Counter(bad[:, 0].tolist()).most_common(8))
da = cx['da'][0].float().cpu().numpy(); y = cx['y'].float().cpu().numpy()
aff = {k: v.cpu().numpy() for k, v in cx['aff'].items()}
x_, yv, st, a_, _ = o.cache['dec1.conv2']
print('yraw err', np.abs(y - yv).max(), 'mean err', np.abs(aff['mean'] - st[0]).max(), 'rstd err', np.abs(aff['rstd'] - 1/np.sqrt(st[1] + 1e-3)).max())
i = tuple(bad[0])
sc, sh = aff['scale'][i[3]], aff['shift'][i[3]]
print('elem', i, 'da', da[i], 'y', y[i], 'a_dev', sc * y[i] + sh, 'a_ref', a_[i], 'dy_dev', got[i], 'dy_ref', ref_dy[i])
