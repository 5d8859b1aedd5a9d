"""Data-parallel training path on the GPU: two replicas x batch B/2 with SyncBN and the bucketed
gradient all-reduce must reproduce one device x batch B.  Only one GPU is available to the tests, so both ranks
share cuda:0 and talk through gloo (RCCL refuses two ranks on one device); the exchange code is the same
`parallel.GradSync` / `parallel.allreduce_mean_` that the 8-GPU bench drives over RCCL."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ["REPO"])
import torch.distributed as dist
from satellite_computervision_amd import model_tools as mt, parallel
rank, world = parallel.init_from_env("gloo")
torch.cuda.set_device(0)
sync_bn = os.environ["SYNC_BN"] == "1"
rng = np.random.default_rng(5)
B = 8
x = rng.random((B, 32, 32, 4)).astype(np.float32)
lab = (rng.random((B, 32, 32)) < 0.3).astype(np.int64)
y = np.eye(2, dtype=np.float32)[lab]


def make():
    mt.reset_uids(); mt.set_seed(11)
    m = mt.get_unet_model(2, 4, filters=[32, 64], factors=[2, 2])
    m.compute_dtype = "float32"
    m.compile(optimizer=mt.Adam(1e-2), loss=lambda t, p: mt.weighted_categorical_crossentropy(t, p, [1.0, 3.0]))
    return m


dp = make(); dp.sync_bn = sync_bn
dp.runtime
parallel.make_grad_sync(dp, bucket_bytes=4096)
losses = [dp.train_on_batch(x[rank::world], y[rank::world])]
g_dp = dp.runtime.gflat.clone() / world                 # all-reduced sum -> mean (Adam applies the 1/world scale itself)
losses += [dp.train_on_batch(x[rank::world], y[rank::world]) for _ in range(2)]
lt = torch.tensor(losses, dtype=torch.float64, device="cuda")
parallel.allreduce_mean_(lt)
w_dp = dp.runtime.pflat.clone(); s_dp = dp.runtime.sflat.clone()
gathered = [torch.empty_like(w_dp) for _ in range(world)]
dist.all_gather(gathered, w_dp)
assert torch.equal(gathered[0], gathered[1]), "replicas diverged"
if rank == 0:
    one = make()
    order = np.concatenate([np.arange(B)[r::world] for r in range(world)])
    l1 = [one.train_on_batch(x[order], y[order])]
    g1 = one.runtime.gflat.clone()
    l1 += [one.train_on_batch(x[order], y[order]) for _ in range(2)]
    w1, s1 = one.runtime.pflat, one.runtime.sflat
    dw = ((w_dp - w1).norm() / w1.norm()).item(); ds = ((s_dp - s1).norm() / s1.norm()).item()
    dl = float(np.abs(lt.cpu().numpy() - np.array(l1)).max())
    dg = ((g_dp - g1).norm() / g1.norm()).item()
    print("RESULT", dw, ds, dl, dg)
dist.barrier(); dist.destroy_process_group()
'''


def _run(tmp_path, sync_bn):
    script = tmp_path / 'dp_worker.py'
    script.write_text(WORKER)
    env = dict(os.environ, REPO=ROOT, MASTER_ADDR='127.0.0.1', MASTER_PORT='29641', WORLD_SIZE='2', SYNC_BN='1' if sync_bn else '0')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK='0'),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    line = [l for l in outs[0].splitlines() if l.startswith('RESULT')][0]
    return [float(v) for v in line.split()[1:]]


@pytest.mark.gpu
def test_two_replicas_with_syncbn_equal_one_device(tmp_path):
    dw, ds, dl, dg = _run(tmp_path, True)
    # fp32: only the summation order differs (per-replica partial sums, bucketed all-reduce).  First-step gradients, three
    # losses and the BN moving statistics agree to rounding; the weights after three Adam steps are looser because Adam
    # turns the ~0 gradients of the conv biases that precede a BatchNorm into +-lr steps of rounding-noise sign.
    assert dg < 1e-4 and ds < 1e-4 and dl < 1e-4 and dw < 5e-3, (dw, ds, dl, dg)


@pytest.mark.gpu
def test_two_replicas_per_replica_bn_stay_in_lockstep(tmp_path):
    """default mode (tf.distribute semantics): statistics per replica -> weights identical across ranks (asserted in the
    worker) but not equal to the single-device run."""
    dw, ds, dl, dg = _run(tmp_path, False)
    assert dg > 1e-3 and dw < 0.5, (dw, dg)


@pytest.mark.gpu
def test_bench_two_ranks_end_to_end(tmp_path):
    """bench.py exactly as the driver launches it for N=2 (torch.distributed.run, one JSON line from rank 0), with the two
    ranks sharing the only GPU over gloo: barriers, max-over-ranks timing, overlapped gradient exchange, whole-job value."""
    import json
    env = dict(os.environ, SATCV_BENCH_BACKEND='gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', '29655',
           os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '2', '--batch', '8']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 3 and out['config']['global_batch'] == 16 and out['scaling'] == 'weak'
    assert out['value'] > 0 and np.isfinite(out['extra']['loss_last']) and 'cpu_baseline' not in out
    assert abs(out['value'] - 16 * 3 / (out['ms_per_step'] * 3 / 1000)) / out['value'] < 1e-3


CHIP_WORKER = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ["REPO"])
import torch.distributed as dist
from satellite_computervision_amd import model_tools as mt, parallel, prediction_tools as pt
rank, world = parallel.init_from_env("gloo")
torch.cuda.set_device(0)
mt.reset_uids(); mt.set_seed(3)
m = mt.get_unet_model(2, 4, filters=[32, 64], factors=[2, 2])
rng = np.random.default_rng(0)
scene = rng.random((224, 288, 4)).astype(np.float32)
idx = pt.generate_chip_indices(scene, buff=32, kernel=64)
assert len(idx) >= 4
out = pt.predict_chips_sharded(scene, idx, np.zeros(scene.shape[:2]), m, kernel=64, buff=32, batch_size=3, channel=1)
if rank == 0:
    ref = pt.predict_chips(scene, idx, np.zeros(scene.shape[:2]), m, kernel=64, buff=32, batch_size=5, channel=1)
    print("RESULT", float(np.abs(out - ref).max()), float(np.abs(ref).max()), len(idx))
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.gpu
def test_sharded_chip_inference_equals_single_process(tmp_path):
    """inference partitioning: chips split round-robin over two ranks, templates summed once"""
    script = tmp_path / 'chip_worker.py'
    script.write_text(CHIP_WORKER)
    env = dict(os.environ, REPO=ROOT, MASTER_ADDR='127.0.0.1', MASTER_PORT='29671', WORLD_SIZE='2')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK='0'),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    err, mag, nchips = [float(v) for v in [l for l in outs[0].splitlines() if l.startswith('RESULT')][0].split()[1:]]
    assert err == 0.0 and mag > 0 and nchips >= 4          # inference is bit-identical across batch splits


BUCKET_WORKER = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ["REPO"])
import torch.distributed as dist
from satellite_computervision_amd import model_tools as mt, parallel
rank, world = parallel.init_from_env("gloo")
torch.cuda.set_device(0)
rng = np.random.default_rng(5)
B = 8
x = rng.random((B, 32, 32, 4)).astype(np.float32)
y = np.eye(2, dtype=np.float32)[(rng.random((B, 32, 32)) < 0.3).astype(np.int64)]


def run(overlap, per=None):
    mt.reset_uids(); mt.set_seed(11)
    m = mt.get_unet_model(2, 4, filters=[32, 64], factors=[2, 2])
    m.compute_dtype = "float32"
    m.compile(optimizer=mt.Adam(1e-2), loss=lambda t, p: mt.weighted_categorical_crossentropy(t, p, [1.0, 3.0]))
    rt = m.runtime
    if per is None:
        # bucket boundaries sit at numel - k * per: put every one of the first few INSIDE the gamma / beta range of a
        # BatchNormalization that belongs to a conv_batch_act node (the ranges the overlap bookkeeping used to miss)
        hits = []
        for l in m.layers:
            for ps in l.specs:
                if ps.kind in ("gamma", "beta") and ps.name in rt.offsets:
                    hits.append((rt.offsets[ps.name], ps.size, ps.name))
        hits.sort(reverse=True)
        numel = rt.gflat.numel()
        off, size, name = hits[2]
        per = numel - (off + size // 2)
        assert 0 < per < numel
    sync = parallel.make_grad_sync(m, bucket_bytes=4 * per, overlap=overlap)
    inside = [a for a, b in sync.bounds if any(o < a < o + s for o, s, _ in [(rt.offsets[ps.name], ps.size, 0) for l in m.layers for ps in l.specs if ps.kind in ("gamma", "beta") and ps.name in rt.offsets])]
    for _ in range(3):
        m.train_on_batch(x[rank::world], y[rank::world])
    torch.cuda.synchronize()
    return rt.pflat.clone(), per, len(inside)


w_ov, per, n_inside = run(True)
w_no, _, _ = run(False, per)
assert n_inside >= 1, "no bucket boundary inside a BatchNorm gamma/beta range"
same = bool(torch.equal(w_ov, w_no))
g = [torch.empty_like(w_ov) for _ in range(world)]
dist.all_gather(g, w_ov)
if rank == 0:
    print("RESULT", int(same), int(torch.equal(g[0], g[1])), n_inside)
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.gpu
def test_overlapped_buckets_cut_inside_batchnorm_ranges(tmp_path):
    """a bucket boundary inside the gamma / beta range of a conv_batch_act's BatchNormalization: the overlapped exchange may
    only start a bucket once that node has written dgamma / dbeta -- the parameters after three steps must be bit-identical
    to the run with a single exchange after the backward pass (a + b is order-free for two ranks), and equal on both ranks."""
    script = tmp_path / 'bucket_worker.py'
    script.write_text(BUCKET_WORKER)
    env = dict(os.environ, REPO=ROOT, MASTER_ADDR='127.0.0.1', MASTER_PORT='29683', WORLD_SIZE='2')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK='0'),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    same, lockstep, n_inside = [int(v) for v in [l for l in outs[0].splitlines() if l.startswith('RESULT')][0].split()[1:]]
    assert same == 1 and lockstep == 1 and n_inside >= 1


RCCL_WORKER = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ["REPO"])
import torch.distributed as dist
from satellite_computervision_amd import model_tools as mt, parallel
assert parallel.FORCE
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
rng = np.random.default_rng(5)
B = 8
x = rng.random((B, 64, 64, 4)).astype(np.float32)
y = np.eye(2, dtype=np.float32)[(rng.random((B, 64, 64)) < 0.3).astype(np.int64)]


def run(sync_on, sync_bn=False, overlap=True, payload=None):
    mt.reset_uids(); mt.set_seed(11)
    m = mt.get_unet_model(2, 4, filters=[32, 64, 128], factors=[2, 2, 2])
    m.compute_dtype = "bfloat16"
    m.sync_bn = sync_bn
    m.compile(optimizer=mt.Adam(1e-3), loss=lambda t, p: mt.weighted_categorical_crossentropy(t, p, [1.0, 3.0]))
    sync = parallel.make_grad_sync(m, bucket_bytes=256 << 10, overlap=overlap) if sync_on else None
    if sync_on:
        assert sync._active() and len(sync.bounds) > 3
        if payload:
            sync.payload = payload
    for _ in range(3):
        m.train_step_device(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), sync)
    torch.cuda.synchronize()
    dist.barrier(device_ids=[0])
    return m.runtime.pflat.clone(), m.runtime.sflat.clone()


w0, s0 = run(False)
w1, s1 = run(True)                      # overlapped buckets on the weight-gradient stream, ReduceOp.SUM over one rank
w2, s2 = run(True, sync_bn=True)        # + ReduceOp.AVG of the BatchNorm statistics buffers
w3, s3 = run(True, overlap=False)
t = torch.arange(8, dtype=torch.float64, device="cuda")
parallel.allreduce_mean_(t)
use_cabi = os.environ.get("SATCV_CABI_COMM", "0") == "1"
import ctypes as C
from satellite_computervision_amd._lib import lib, check
if use_cabi:
    # the exchange went through libsatcv's own communicator (include/satcv.h: satcv_comm_init / satcv_allreduce_grads / satcv_allreduce)
    via_cabi = parallel.cabi_comm() is not None and parallel._comm["calls"] > 10
    # bf16 wire format: a one-rank sum returns every element rounded to bf16 once (ranges cut into buckets from the end, ragged tail)
    g = torch.randn(100003, device="cuda")
    ref = g.clone()
    ref[7:100001] = ref[7:100001].bfloat16().float()
    scratch = torch.empty(100003, dtype=torch.bfloat16, device="cuda")
    check(lib.satcv_allreduce_grads(parallel.cabi_comm(), g.data_ptr(), 7, 100001, 4096, 1, scratch.data_ptr(), torch.cuda.current_stream().cuda_stream))
    g2 = torch.randn(100003, device="cuda"); ref2 = g2.clone()
    check(lib.satcv_allreduce_grads(parallel.cabi_comm(), g2.data_ptr(), 0, 100003, 4096, 0, None, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    rank, world = C.c_int32(-1), C.c_int32(-1)
    check(lib.satcv_comm_info(parallel.cabi_comm(), C.byref(rank), C.byref(world)))
    okrw = rank.value == 0 and world.value == 1
    w4, s4 = run(True, payload="bf16")      # training with the 37 MB payload: close to, not identical with, the fp32 exchange
    close = bool(torch.isfinite(w4).all()) and float((w4 - w0).abs().max()) < 2e-2 and not torch.equal(w4, w0)
else:
    # default: torch.distributed's ProcessGroupNCCL carries the exchange (async_op all-reduces on the weight-gradient stream); the
    # library's communicator is never created, and a bf16 payload request falls back to the fp32 exchange with a warning
    via_cabi = parallel.cabi_comm() is None and parallel._comm["calls"] == 0
    g = ref = g2 = ref2 = torch.zeros(1)
    okrw = True
    import warnings
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter("always")
        w4, s4 = run(True, payload="bf16")
    close = torch.equal(w4, w0) and any("bf16" in str(w_.message) for w_ in wl)
print("RESULT", int(torch.equal(w0, w1)), int(torch.equal(s0, s1)), int(torch.equal(w0, w2)), int(torch.equal(s0, s2)), int(torch.equal(w0, w3)),
      int(torch.equal(t.cpu(), torch.arange(8, dtype=torch.float64))), int(via_cabi), int(torch.equal(g, ref)), int(torch.equal(g2, ref2)),
      int(okrw), int(close))
parallel.destroy_cabi_comm()
dist.destroy_process_group()
'''


@pytest.mark.gpu
@pytest.mark.parametrize('cabi', ['1', '0'])
def test_rccl_single_rank_path(tmp_path, cabi):
    """the RCCL code path on the one GPU a test box has: backend "nccl" with a one-rank communicator and
    SATCV_FORCE_COLLECTIVES=1, so the bucketed async all-reduce on the weight-gradient stream, the stream waits, ReduceOp.AVG
    (SyncBN buffers) and barrier(device_ids=...) all execute.  A one-rank sum / mean is the identity: parameters and moving
    statistics after three steps must be BIT-IDENTICAL to the run without any exchange.  cabi = '1' (opt-in, SATCV_CABI_COMM=1): the
    exchange runs through the C ABI's own communicator (satcv_comm_init over an id broadcast on the process group,
    satcv_allreduce_grads, satcv_allreduce); its bf16 payload returns each element rounded once.  cabi = '0' (the default): the
    torch.distributed path -- async_op all-reduces on the weight-gradient stream -- carries everything."""
    script = tmp_path / 'rccl_worker.py'
    script.write_text(RCCL_WORKER)
    env = dict(os.environ, REPO=ROOT, MASTER_ADDR='127.0.0.1', MASTER_PORT='29691' if cabi == '1' else '29693', WORLD_SIZE='1', RANK='0',
               LOCAL_RANK='0', SATCV_FORCE_COLLECTIVES='1', HSA_ENABLE_IPC_MODE_LEGACY='0', SATCV_CABI_COMM=cabi)
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    flags = [int(v) for v in [l for l in r.stdout.splitlines() if l.startswith('RESULT')][0].split()[1:]]
    assert flags == [1] * 11, flags


@pytest.mark.gpu
def test_bench_single_rank_under_torchrun_over_rccl(tmp_path):
    """bench.py launched the way the driver launches it (torch.distributed.run, RANK / WORLD_SIZE from the environment) with
    one rank on the one GPU, backend nccl, collectives forced: init, barriers with device_ids, the overlapped exchange,
    max-over-ranks timing and teardown all run over RCCL.  A fresh subprocess: nothing that has touched the GPU is exec'd."""
    import json
    env = dict(os.environ, SATCV_FORCE_COLLECTIVES='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('SATCV_BENCH_BACKEND', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1', '--master-port', '29657',
           os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '2', '--batch', '8', '--no-cpu-baseline', '--no-infer']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert out['n_gpus'] == 1 and out['value'] > 0 and np.isfinite(out['extra']['loss_last'])
