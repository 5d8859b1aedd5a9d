"""Headline benchmark: U-Net 256x256x4 training throughput (tiles/s) on N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one optimisation step of get_unet_model(2, 4) (as coded in the reference,
utils/model_tools.py:394) on a batch of 64 synthetic Sentinel-2-like 256x256x4 tiles per GPU:
forward (train-mode BN) + weighted categorical cross-entropy + backward + [RCCL gradient
all-reduce] + Keras-Adam + weight repack, bf16 storage / fp32 accumulate.  Inputs are resident
in HBM when the timed region starts.  Rank 0 prints ONE JSON line.

roofline: the dominant kernel class is the 3x3 implicit-GEMM convolution (forward + data-gradient
launches); `achieved` = algorithmic FLOPs of those launches / their summed HIP-event durations
inside the timed region; peak = 2.5 PFLOP/s dense bf16 MFMA (MI355X_MICROARCH.md).
cpu_baseline: the same training step on the host cores with the PyTorch-CPU (oneDNN) restatement
of the graph (oracle/torch_unet.py; TensorFlow is absent here) on a bounded sample.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TILE, CH, NCLS, BATCH = 256, 4, 2, 64
FWD_GFLOP_PER_TILE = 22.641
TRAIN_GFLOP_PER_TILE = 67.77
PEAK_BF16_TFLOPS = 2500.0
# algorithmic HBM bytes of the 3x3 conv launches of one training step at batch 64 (each launch reads its input once and
# writes its output once, bf16, weights once): forward 11 layers + data gradient 10 layers, see DESIGN.md section 3
def _alg_3x3_bytes(batch, esize=2, tile=TILE, filters=(32, 64, 128, 256, 512), cin0=16):
    L, cin = [], cin0
    for i, c in enumerate(filters):
        L.append((batch * (tile >> i) ** 2, cin, c)); cin = c
    L.append((batch * (tile >> len(filters)) ** 2, filters[-1], 2 * filters[-1]))
    for j in range(len(filters) - 1, -1, -1):
        px = batch * (tile >> j) ** 2
        L += [(px, 2 * filters[j], filters[j]), (px, filters[j], filters[j])]       # conv1 reads concat([skip f, up f])
    one = lambda px, ci, co: px * (ci + co) * esize + 9 * ci * co * esize
    return sum(one(*l) for l in L) + sum(one(*l) for l in L[1:])        # forward + data gradient (none for the first layer)


ALG_3X3_BYTES_PER_STEP = _alg_3x3_bytes(BATCH)                          # 6.45 GB at batch 64: 16 forward + 15 dgrad launches


def synth_batch(rng, n):
    """Sentinel-2-like reflectance /10000 (gamma-ish) and ~5 % positive rectangular masks."""
    x = rng.beta(2, 5, (n, TILE, TILE, CH)).astype(np.float32)
    lab = np.zeros((n, TILE, TILE), np.int64)
    for i in range(n):
        for _ in range(3):
            h, w = rng.integers(16, 64, 2)
            y0, x0 = rng.integers(0, TILE - h), rng.integers(0, TILE - w)
            lab[i, y0:y0 + h, x0:x0 + w] = 1
    y = np.eye(NCLS, dtype=np.float32)[lab]
    return x, y


def pmc_traffic_per_launch():
    """HBM bytes per 3x3 implicit-GEMM launch from the committed rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes of this same
    command (profiles/r01_pmc_traffic.json, produced by tools/pmc_summary.py with the gfx950 FETCH_SIZE x2 correction).
    PMC counters cannot be collected from inside the timed run; None when the summary is absent."""
    try:
        d = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')))['classes']
        c = d.get('igemm_3x3') or d.get('igemm_3x3_1x1_convT')
        return round(c['hbm_bytes_per_launch'] / 1e6, 1)
    except Exception:
        return None


def cpu_baseline(seconds_budget=20.0):
    """torch-CPU port of the identical training step (fp32), batch 2, all host cores."""
    from oracle.unet import UNetOracle
    from oracle import torch_unet as TU
    cores = min(os.cpu_count() or 1, 64)       # more oneDNN threads than this only slows the 256^2 convs
    torch.set_num_threads(cores)
    o = UNetOracle(NCLS, CH, dtype=np.float32, seed=0)
    p = TU.params_to_torch(o.params, torch.float32)
    train = [k for k, v in p.items() if v.requires_grad]
    m = {k: torch.zeros_like(p[k]) for k in train}
    v = {k: torch.zeros_like(p[k]) for k in train}
    rng = np.random.default_rng(0)
    bs = 1
    x, y = synth_batch(rng, bs)
    xt, yt = torch.from_numpy(x), torch.from_numpy(y)
    filters, factors = [32, 64, 128, 256, 512], [2, 2, 2, 2, 2]

    def step(t):
        for k in train:
            p[k].grad = None
        pr, _ = TU.unet_forward(p, xt, filters, factors, training=True)
        loss = TU.weighted_cce_mean(yt, pr, [1.0, 20.0])
        loss.backward()
        TU.keras_adam_(p, {k: p[k].grad for k in train}, m, v, t)
    step(1)                                  # warm-up (oneDNN primitive creation)
    t0 = time.perf_counter()
    nsteps = 0
    while (time.perf_counter() - t0 < seconds_budget and nsteps < 50) or nsteps < 1:
        step(nsteps + 2)
        nsteps += 1
    dt = time.perf_counter() - t0
    return dict(value=round(bs * nsteps / dt, 3), unit='tiles/s', cores=cores, kind='port',
                sample=f'{nsteps} training steps of batch {bs} (256x256x4, fp32, torch-CPU/oneDNN stand-in: TensorFlow absent)')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=BATCH)
    ap.add_argument('--dtype', default='bfloat16')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--infer', action='store_true', help='also time inference (reported under "extra")')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('SATCV_BENCH_BACKEND') == 'gloo':   # test hook: several ranks on ONE GPU (RCCL refuses that), see tests/test_dp_gpu.py
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or 'RANK' in os.environ:              # launched by torch.distributed.run: one rank per GPU over RCCL
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        # (no device_id=: the eager communicator init it triggers was measured to slow EVERY kernel launch of the process,
        #  14.7 vs 13.05 ms/step; the lazy init on the first collective does not)
        dist.init_process_group(os.environ.get('SATCV_BENCH_BACKEND', 'nccl'), rank=rank, world_size=world)

    BARRIER_KW = dict(device_ids=[local_rank]) if (dist is not None and dist.get_backend() == 'nccl') else {}

    from satellite_computervision_amd import model_tools as mt
    from satellite_computervision_amd import parallel
    from satellite_computervision_amd._lib import lib, check

    mt.reset_uids()
    mt.set_seed(0)                                     # identical initial weights on every rank
    mt.set_compute_dtype(args.dtype)
    model = mt.get_unet_model(NCLS, CH)
    model.compile(optimizer=mt.Adam(9e-4), loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, [1.0, 20.0]))
    sync = parallel.make_grad_sync(model) if dist is not None else None

    rng = np.random.default_rng(1000 + rank)           # per-rank data shard
    B = args.batch
    pool = []
    for _ in range(2):
        x, y = synth_batch(rng, B)
        pool.append((torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()))

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier(**BARRIER_KW)
        torch.cuda.synchronize()

    model._head_plan(B, TILE, TILE, True)              # buffers and launch descriptors exist before any (warm-up or timed) step
    for i in range(args.warmup):
        xb, yb = pool[i % len(pool)]
        model.train_step_device(xb, yb, sync)
    barrier()
    check(lib.satcv_prof_enable(0b111))
    t0 = time.perf_counter()
    for i in range(args.steps):
        xb, yb = pool[i % len(pool)]
        plan = model.train_step_device(xb, yb, sync)
    barrier()
    dt = time.perf_counter() - t0
    check(lib.satcv_prof_enable(0))
    loss = float(plan.loss_buf.item())
    if dist is not None:
        tmax = torch.tensor([dt], device='cuda')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    prof = {}
    for kind, name in ((0, 'conv3x3_igemm_fwd_dgrad'), (1, 'conv1x1_convT_gemm'), (2, 'conv_wgrad')):
        ms, cnt, fl = C.c_double(), C.c_int64(), C.c_double()
        check(lib.satcv_prof_collect(kind, C.byref(ms), C.byref(cnt), C.byref(fl)))
        prof[name] = dict(ms=ms.value, launches=cnt.value, flops=fl.value)

    extra = {'loss_last': loss, 'kernel_ms_per_step': {k: round(v['ms'] / args.steps, 3) for k, v in prof.items()},
             'kernel_tflops': {k: round(v['flops'] / max(v['ms'], 1e-9) / 1e9, 1) for k, v in prof.items()}}
    if args.infer:
        xb, _ = pool[0]
        for _ in range(3):
            model.predict_on_device(xb)
        barrier()
        t1 = time.perf_counter()
        for _ in range(10):
            model.predict_on_device(xb)
        barrier()
        extra['infer_tiles_per_s'] = round(world * B * 10 / (time.perf_counter() - t1), 1)
        # BASELINE configs[4]: sliding-window inference on 1024^2 scenes = 9 chips of 384^2 per scene (buff 128, kernel 256,
        # utils/prediction_tools.py:87-156), bf16 plan vs folded fp8 (e4m3) plan; output "kernel tiles" = 256^2 centres kept
        chips = torch.from_numpy(np.random.default_rng(5).beta(2, 5, (36, 384, 384, CH)).astype(np.float32)).cuda()      # 4 scenes
        for tag in ('bf16', 'fp8'):
            if tag == 'fp8':
                model.enable_fp8_inference(chips[:8])
            for _ in range(3):
                model.predict_on_device(chips)
            barrier()
            t1 = time.perf_counter()
            for _ in range(10):
                model.predict_on_device(chips)
            barrier()
            extra[f'config5_{tag}_kernel_tiles_per_s'] = round(world * 36 * 10 / (time.perf_counter() - t1), 1)
        for _ in range(3):
            model.predict_on_device(xb)
        barrier()
        t1 = time.perf_counter()
        for _ in range(10):
            model.predict_on_device(xb)
        barrier()
        extra['infer_fp8_tiles_per_s'] = round(world * B * 10 / (time.perf_counter() - t1), 1)
        model.disable_fp8_inference()
        # BASELINE configs[2]: DeepLab-v3 (ResNet-50 OS16 + the reference's ASPP block), NAIP-like 512x512x4 tiles, inference
        dl = mt.get_deeplabv3_model(2, 4)
        for bs in (1, 16):
            xd = torch.from_numpy((np.random.default_rng(6).integers(0, 256, (bs, 512, 512, 4)) / 255.0).astype(np.float32)).cuda()
            for _ in range(3):
                dl.predict_on_device(xd)
            barrier()
            t1 = time.perf_counter()
            for _ in range(10):
                dl.predict_on_device(xd)
            barrier()
            extra[f'config3_deeplab_b{bs}_tiles_per_s'] = round(world * bs * 10 / (time.perf_counter() - t1), 1)

    if rank == 0:
        tiles = world * B * args.steps
        value = tiles / dt
        d = prof['conv3x3_igemm_fwd_dgrad']
        ach = d['flops'] / max(d['ms'], 1e-9) / 1e9
        out = {
            'metric': 'tiles/sec (train) 256x256x4 U-Net', 'value': round(value, 2), 'unit': 'tiles/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1000 * dt / args.steps, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16' if args.dtype == 'bfloat16' else 'f32', 'data': 'synthetic',
            'config': {'workload': f'U-Net 256x256x4 {args.dtype} training, batch {B} per GPU (BASELINE configs[1])',
                       'global_batch': world * B, 'parallelism': f'dp{world}', 'loss': 'weighted_categorical_crossentropy',
                       'optimizer': 'adam(9e-4)'},
            'roofline': {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': PEAK_BF16_TFLOPS if args.dtype == 'bfloat16' else 157.3,
                         'unit': 'TFLOP/s', 'frac': round(ach / (PEAK_BF16_TFLOPS if args.dtype == 'bfloat16' else 157.3), 4),
                         'traffic': pmc_traffic_per_launch(), 'traffic_unit': 'MB per launch (rocprofv3 PMC, profiles/r01_pmc_traffic.json)',
                         'algorithmic_bytes_per_launch_MB': round(ALG_3X3_BYTES_PER_STEP / max(d['launches'] / args.steps, 1) / 1e6, 1),
                         'kernel': 'igemm_fast_kernel (3x3 conv fwd + dgrad)',
                         'avg_launch_us': round(1000 * d['ms'] / max(d['launches'], 1), 2)},
            'model_tflops': round(value * TRAIN_GFLOP_PER_TILE / 1000, 2),
            'extra': extra,
        }
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier(**BARRIER_KW)
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
