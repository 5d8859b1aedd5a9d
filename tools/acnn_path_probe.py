"""GPU probe: every step of an atrous-CNN training plan timed alone (wall clock, 10 repeats) with a mark where the thin persistent kernel served it."""
import sys, os, ctypes
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from satellite_computervision_amd import model_tools as mt, ops
from satellite_computervision_amd._lib import lib, check
mt.set_compute_dtype('bfloat16'); mt.reset_uids(); mt.set_seed(0)
a2 = mt.get_acnn_model2(3, 4, nfilters=16, depth=3)
a2.compile(optimizer=mt.Adam(1e-3), loss=lambda a, b: mt.weighted_categorical_crossentropy(a, b, [1.0, 1.0, 2.0]))
rng = np.random.default_rng(0)
B, S = 16, 256
x = rng.random((B, S, S, 4), dtype=np.float32); y3 = np.eye(3, dtype=np.float32)[rng.integers(0, 3, (B, S, S))]
a2.train_on_batch(x, y3)
plan = a2._head_plan(B, S, S, True)
def wsl():
    v = ctypes.c_int32(); check(lib.satcv_get_option(b'igemm_thin_launches', ctypes.byref(v))); return v.value
st = ops.stream_ptr()
lib.satcv_set_option(b'igemm_m16', 2)
import time
for name, steps in (('fwd', plan.fwd), ('bwd', plan.bwd)):
    for s in steps:
        lab = getattr(s, 'label', '') or getattr(s, '__name__', '?')
        b = wsl()
        for _ in range(3): s(st)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): s(st)
        torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 10 * 1e6
        print(f'{name} {us:8.1f} us  {"ws" if wsl() > b else "  "}  {lab}')
