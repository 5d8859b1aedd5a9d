import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from satellite_computervision_amd import model_tools as mt
def run(fuse, n=2, hw=256):
    mt.reset_uids(); mt.set_seed(3)
    m = mt.get_unet_model(2, 4)
    m.compute_dtype = 'bfloat16'
    m.fuse_thin_bwd = fuse
    m.compile(optimizer=mt.Adam(1e-3), loss=lambda a, b: mt.weighted_categorical_crossentropy(a, b, [1.0, 3.0]))
    rng = np.random.default_rng(0)
    x = rng.random((n, hw, hw, 4)).astype(np.float32)
    y = np.eye(2, dtype=np.float32)[(rng.random((n, hw, hw)) < 0.3).astype(np.int64)]
    plan = m.train_step_device(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
    torch.cuda.synchronize()
    rt = m.runtime
    g = {p.name: rt.get_grad(p.name).clone() for p in m.param_specs if p.name in rt.offsets}
    dbg = {k: v.clone() for k, v in plan.dbg.items() if isinstance(v, torch.Tensor)}
    labels = [getattr(f, 'label', None) for f in plan.bwd]
    inv = {v: k for k, v in mt.structural_names(m).items()}
    g = {inv.get(k, k): v for k, v in g.items()}
    return g, dbg, [l for l in labels if l]
for n, hw in ((3, 64),):
    g0, d0, _ = run(False, n, hw)
    g1, d1, lab = run(True, n, hw)
    print('case', n, hw); print('\n'.join(l for l in lab if 'bn_bwd' in l or 'fused' in l or 'bnred' in l))
    for k in g0:
        a, b = g0[k].double().flatten(), g1[k].double().flatten()
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
        if cos < 0.995 and float(a.norm()) > 0:
            print('  grad', k, 'cos', round(cos, 4), 'norms', float(a.norm()), float(b.norm()))
    for k in d0:
        if k in d1 and k.startswith('dx:'):
            a, b = d0[k].double().flatten(), d1[k].double().flatten()
            cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
            print('  ', k, 'cos', round(cos, 5), 'maxabs diff', float((a - b).abs().max()), 'scale', float(a.abs().max()))
