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
PEAK_HBM_TBPS = 8.0


def unet_layers(batch, tile=TILE, cin0=CH, filters=(32, 64, 128, 256, 512), ncls=NCLS):
    """(kind, pixels of the GEMM grid, Cin, Cout, taps) of every convolution of get_unet_model(2, 4) as coded
    (utils/model_tools.py:321-415): 6 encoder / centre convs, per decoder level one transposed conv (k = s = 2: per INPUT
    pixel Cin x 4*Cout) and two 3x3 convs (the first reads concat([skip, up])), the 1x1 head.  Cin is the ALGORITHMIC one
    (4 for the first layer, although 16 channels are stored)."""
    L, cin = [], cin0
    for i, c in enumerate(filters):
        L.append(('conv3', batch * (tile >> i) ** 2, cin, c, 9)); cin = c
    L.append(('conv3', batch * (tile >> len(filters)) ** 2, filters[-1], 2 * filters[-1], 9))
    cin = 2 * filters[-1]
    for j in range(len(filters) - 1, -1, -1):
        L.append(('convT', batch * (tile >> (j + 1)) ** 2, cin, 4 * filters[j], 1))
        px = batch * (tile >> j) ** 2
        L += [('conv3', px, 2 * filters[j], filters[j], 9), ('conv3', px, filters[j], filters[j], 9)]
        cin = filters[j]
    L.append(('head', batch * tile * tile, filters[0], ncls, 1))
    return L


def alg_work(batch, esize=2, cin0=CH):
    """algorithmic FLOPs / unfused HBM bytes (input once, output once, weights once) of the launches of one training step,
    per class, and the per-layer roofline time sum(max(flops / MFMA peak, bytes / HBM peak))."""
    out = {'conv3_fwd_dgrad': [0.0, 0.0, 0.0, 0], 'all': [0.0, 0.0, 0.0, 0]}

    def add(key, fl, by):
        t = max(fl / (PEAK_BF16_TFLOPS * 1e12), by / (PEAK_HBM_TBPS * 1e12))
        for k in (key, 'all'):
            if k in out:
                out[k][0] += fl; out[k][1] += by; out[k][2] += t; out[k][3] += 1
    first = True
    for kind, px, ci, co, taps in unet_layers(batch, cin0=cin0):
        fl = 2.0 * px * ci * co * taps
        by = px * (ci + co) * esize + taps * ci * co * esize
        key = 'conv3_fwd_dgrad' if kind == 'conv3' else 'other'
        add(key, fl, by)                                  # forward
        if not first:
            add(key, fl, by)                              # data gradient (none for the first layer)
        add('wgrad', fl, px * (ci + co) * esize + taps * ci * co * 4)          # weight gradient: X and dY once, fp32 dW
        first = False
    return {k: dict(flops=v[0], bytes=v[1], roof_s=v[2], launches=v[3]) for k, v in out.items()}


def synth_batch(rng, n, ch=CH):
    """Sentinel-2-like reflectance /10000 (gamma-ish; ch = 4 NAIP-like bands or all 13 Sentinel-2 bands, utils/ee_tools.py:100)
    and ~5 % positive rectangular masks."""
    x = rng.beta(2, 5, (n, TILE, TILE, ch)).astype(np.float32)
    lab = np.zeros((n, TILE, TILE), np.int64)
    for i in range(n):
        for _ in range(3):
            h, w = rng.integers(16, 64, 2)
            y0, x0 = rng.integers(0, TILE - h), rng.integers(0, TILE - w)
            lab[i, y0:y0 + h, x0:x0 + w] = 1
    y = np.eye(NCLS, dtype=np.float32)[lab]
    return x, y


def pmc_traffic_per_launch():
    """HBM bytes per 3x3 implicit-GEMM launch from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same
    command (profiles/rNN_pmc_traffic.json, produced by tools/pmc_summary.py with the gfx950 FETCH_SIZE x2 correction).  PMC
    counters cannot be collected from inside the timed run, so the line carries the newest committed summary and names the
    file and the commit it was measured at; (None, None) when there is no summary."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')), reverse=True):
        try:
            d = json.load(open(path))
            c = d['classes'].get('igemm_3x3') or d['classes'].get('igemm_3x3_1x1_convT')
            tot, nl = c['hbm_bytes_per_launch'] * c['launches'], c['launches']
            src = f"{os.path.basename(path)} @ {d.get('commit', 'round-1 tree 138affc')}"
            return round(tot / nl / 1e6, 1), src, str(d.get('csrc_digest', ''))
        except Exception:
            continue
    return None, None, ''


def csrc_digest():
    """sha256[:16] over the kernel sources (csrc/*.hip, *.hpp, include/satcv.h): what a PMC summary must have been measured at to describe the
    library being run (tools/pmc_summary.py stores the same digest; documentation-only commits do not make a summary stale, kernel edits do)."""
    import glob, hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, 'satellite_computervision_amd', 'csrc', '*.hip')) + glob.glob(os.path.join(ROOT, 'satellite_computervision_amd', 'csrc', '*.hpp')))
    for f in files + [os.path.join(ROOT, 'include', 'satcv.h')]:
        h.update(os.path.basename(f).encode()); h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def box_probe():
    """How fast is THIS box?  Boxes of the pool differ by +-4 %: two fixed launches through the C ABI, timed with HIP
    events before the timed regions, so that the headline can be read against the box it ran on -- a streaming launch (BatchNorm-backward
    apply over 64 x 256 x 256 x 32 bf16: two reads + one write, 805 MB) and a matrix-bound one (3x3 convolution 1024 -> 512 at 16 x 16,
    batch 64: 154.6 GFLOP).  ~50 ms in total."""
    from satellite_computervision_amd import ops
    from satellite_computervision_amd._lib import lib, check
    dev = torch.device('cuda')
    out = {}

    def timeit(f, reps):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    n, h, w, c = 64, 256, 256, 32
    x = torch.randn(n, h, w, c, device=dev, generator=g).to(torch.bfloat16)
    gr = torch.randn(n, h, w, c, device=dev, generator=g).to(torch.bfloat16)
    dy = torch.empty_like(x)
    one, zero = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    sums, coef = ops.new_stats(c, dev), torch.zeros(2, c, device=dev)
    d = ops.make_bnbwd_desc(yraw=x.data_ptr(), ldy=c, scale=one.data_ptr(), shift=zero.data_ptr(), mean=zero.data_ptr(), rstd=one.data_ptr(), n=n, h=h, w_=w,
                            c=c, dtype=1, da=gr.data_ptr(), ldda=c, sums=sums.data_ptr(), sums_ld=c, coef=coef.data_ptr(), dy=dy.data_ptr(), lddy_out=c)
    st = ops.stream_ptr()
    t = timeit(lambda: check(lib.satcv_bn_bwd_apply(C.byref(d), st)), 40)
    out['stream_2r1w_TBps'] = round(3 * x.numel() * 2 / t / 1e12, 3)
    n, h, w, cin, cout = 64, 16, 16, 1024, 512
    x = torch.randn(n, h, w, cin, device=dev, generator=g).to(torch.bfloat16)
    kern = torch.randn(3, 3, cin, cout, device=dev, generator=g) * 0.05
    wf, _ = ops.pack_weights(kern, cin, 1, want_dgrad=False)
    y = torch.empty(n, h, w, cout, device=dev, dtype=torch.bfloat16)
    b = torch.zeros(cout, device=dev)
    stt = ops.new_stats(cout, dev)
    dc = ops.make_conv_desc(x0=x.data_ptr(), c0=cin, w=wf.data_ptr(), y=y.data_ptr(), ldy=cout, n=n, h=h, w_=w, cout=cout, cout_pad=ops.rup(cout, 32), dtype=1,
                            bias=b.data_ptr(), stats=stt.data_ptr(), stats_ld=cout, kh=3, kw=3)
    t = timeit(lambda: check(lib.satcv_conv2d_igemm(C.byref(dc), st)), 100)
    out['conv3x3_1024_512_16x16_us'] = round(t * 1e6, 1)
    out['conv3x3_1024_512_16x16_TFLOPs'] = round(2.0 * n * h * w * cin * cout * 9 / t / 1e12, 1)
    return out


def head_commit():
    """commit of the tree being measured: git when the checkout has its history, else the stamp build() left beside the
    library (satellite_computervision_amd/_build_commit.txt travels to the GPU box, .git does not)."""
    import subprocess
    try:
        r = subprocess.run(['git', '-C', ROOT, 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True, timeout=10)
        if r.returncode == 0 and r.stdout.strip():
            return r.stdout.strip()
    except Exception:
        pass
    try:
        return open(os.path.join(ROOT, 'satellite_computervision_amd', '_build_commit.txt')).read().strip()
    except Exception:
        return 'unknown'


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown'


def _tf_cpu_baseline(seconds_budget, CH=CH):
    """when TensorFlow is importable, time the Keras graph of the reference's builder semantics on the
    host cores (GPUs hidden).  The graph is rebuilt here from tf.keras layers following utils/model_tools.py:174-415 as
    coded (one conv per block) -- none of the reference's files travel to the GPU box."""
    os.environ.setdefault('CUDA_VISIBLE_DEVICES', '')
    import tensorflow as tf                                   # noqa: F401  (absent in this image: the caller falls back)
    L = tf.keras.layers

    def cba(x, f):
        return L.Activation('relu')(L.BatchNormalization()(L.Conv2D(f, 3, padding='same')(x)))
    inp = L.Input([TILE, TILE, CH])
    x, skips = inp, []
    for f in (32, 64, 128, 256, 512):
        e = cba(x, f); skips.append(e); x = L.MaxPooling2D(2)(e)
    x = cba(x, 1024)
    for f, sk in zip((512, 256, 128, 64, 32), skips[::-1]):
        x = L.Conv2DTranspose(f, 2, strides=2, padding='same')(x)
        x = L.Activation('relu')(L.BatchNormalization()(L.concatenate([sk, x])))
        x = cba(cba(x, f), f)
    probs = L.Conv2D(NCLS, 1, activation='softmax')(x)
    m = tf.keras.Model(inp, probs)
    m.compile(optimizer=tf.keras.optimizers.Adam(9e-4), loss='categorical_crossentropy')
    rng = np.random.default_rng(0)
    res = {}
    for bs in (1, 16):
        xb, _ = synth_batch(rng, bs, CH)
        m.predict(xb, verbose=0)
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < seconds_budget / 4 or n < 1:
            m.predict(xb, batch_size=bs, verbose=0); n += 1
        res[f'predict_b{bs}_tiles_per_s'] = round(bs * n / (time.perf_counter() - t0), 3)
    xb, yb = synth_batch(rng, 1, CH)
    m.train_on_batch(xb, yb)
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds_budget / 2 or n < 1:
        m.train_on_batch(xb, yb); n += 1
    return dict(value=round(n / (time.perf_counter() - t0), 3), unit='tiles/s', cores=os.cpu_count(), kind='reference',
                sample=f'{n} tf.keras train_on_batch steps of batch 1 (256x256x{CH}, fp32, TensorFlow {tf.__version__} CPU)', **res)


def cpu_baseline(seconds_budget=20.0, CH=CH):
    """The reference's CPU path timed on this host (rank 0, N = 1 only, bounded sample): TensorFlow's Keras graph when
    TensorFlow can be imported (kind "reference"), else the PyTorch-CPU (oneDNN) restatement of the identical graph
    (oracle/torch_unet.py, kind "port"): one training step of batch 1 plus inference at batch 1 and 16
    (utils/prediction_tools.py:152 predicts chip by chip at batch 1)."""
    try:
        out = _tf_cpu_baseline(seconds_budget, CH)
        out['cpu_model'] = _cpu_model()
        return out
    except ImportError:
        pass
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
    x, y = synth_batch(rng, bs, CH)
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
    while (time.perf_counter() - t0 < seconds_budget * 0.6 and nsteps < 50) or nsteps < 1:
        step(nsteps + 2)
        nsteps += 1
    dt = time.perf_counter() - t0
    res = {}
    with torch.no_grad():
        for b in (1, 16):
            xb = torch.from_numpy(synth_batch(rng, b, CH)[0])
            TU.unet_forward(p, xb, filters, factors, training=False)
            t1, n = time.perf_counter(), 0
            while (time.perf_counter() - t1 < seconds_budget * 0.2 and n < 50) or n < 1:
                TU.unet_forward(p, xb, filters, factors, training=False)
                n += 1
            res[f'predict_b{b}_tiles_per_s'] = round(b * n / (time.perf_counter() - t1), 3)
    return dict(value=round(bs * nsteps / dt, 3), unit='tiles/s', cores=cores, kind='port', cpu_model=_cpu_model(),
                threads=f'torch.set_num_threads({cores}); os.cpu_count()={os.cpu_count()}',
                sample=f'{nsteps} training steps of batch {bs} (256x256x{CH}, fp32, torch-CPU/oneDNN stand-in: TensorFlow absent), '
                       f'predict at batch 1 and 16 beside it', **res)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=BATCH)
    ap.add_argument('--dtype', default='bfloat16')
    ap.add_argument('--channels', type=int, default=CH, choices=(4, 13),
                    help='input bands: 4 (BASELINE configs[1], the default line) or 13 = all Sentinel-2 bands (BASELINE configs[3])')
    ap.add_argument('--repeats', type=int, default=5,
                    help='the timed region of exactly --steps steps is run this many times (same work each); value = the MEDIAN region')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--prof-in-timed', action='store_true',
                    help='(rounds 1-5 behaviour) record the per-class HIP events inside the timed regions; default: the timed regions run with '
                         'satcv_prof_enable(0) and one EXTRA, untimed-for-the-headline region with the events on supplies the class times')
    ap.add_argument('--infer', action='store_true', help='(default) also time bf16 / fp8 inference and the config-5 chip rate, reported under "extra"')
    ap.add_argument('--no-infer', action='store_true', help='skip the inference timings')
    ap.add_argument('--full-infer', action='store_true', help='(kept for old command lines: the DeepLab-v3 config-3 timings are part of the default run)')
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
    CHN = args.channels
    model = mt.get_unet_model(NCLS, CHN)
    model.compile(optimizer=mt.Adam(9e-4), loss=lambda yt, yp: mt.weighted_categorical_crossentropy(yt, yp, [1.0, 20.0]))
    sync = parallel.make_grad_sync(model) if dist is not None else None

    rng = np.random.default_rng(1000 + rank)           # per-rank data shard
    B = args.batch
    pool = []
    for _ in range(2):
        x, y = synth_batch(rng, B, CHN)
        pool.append((torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()))

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier(**BARRIER_KW)
        torch.cuda.synchronize()

    model._head_plan(B, TILE, TILE, True)              # buffers and launch descriptors exist before any (warm-up or timed) step
    probe = box_probe() if rank == 0 else None
    for i in range(args.warmup):
        xb, yb = pool[i % len(pool)]
        model.train_step_device(xb, yb, sync)
    barrier()
    # the timed region: EXACTLY --steps steps between barrier + synchronize pairs, max over ranks.  Boxes of the pool differ by several
    # percent and a 0.17-s region is short, so the region is repeated (identical work) and the MEDIAN region is the reported one; the
    # spread goes to extra.region_ms_per_step
    # Round 6: the HIP events that time the kernel classes (two per conv-class launch, ~55 launches per step on two streams) are OFF in the
    # timed regions; one more region of the same --steps steps with them ON supplies the class times (roofline.achieved, extra.kernel_*) and
    # its own ms_per_step goes to extra.ms_per_step_with_prof_events, so that the cost of the events is a number.
    def region():
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            xb, yb = pool[i % len(pool)]
            pl = model.train_step_device(xb, yb, sync)
        barrier()
        dtr = time.perf_counter() - t0
        if dist is not None:
            tmax = torch.tensor([dtr], device='cuda')
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dtr = float(tmax.item())
        return dtr, pl
    if args.prof_in_timed:
        check(lib.satcv_prof_enable(0b1111))
    regions = []
    for rep in range(max(args.repeats, 1)):
        dtr, plan = region()
        regions.append(dtr)
    if args.prof_in_timed:
        nprof = args.steps * len(regions)           # steps the HIP-event sums cover
        dt_prof = None
    else:
        check(lib.satcv_prof_enable(0b1111))
        dt_prof, plan = region()
        nprof = args.steps
    check(lib.satcv_prof_enable(0))
    loss = float(plan.loss_buf.item())
    dt = float(np.median(regions))

    prof = {}
    for kind, name in ((0, 'conv3x3_igemm_fwd_dgrad'), (1, 'conv1x1_convT_gemm'), (2, 'conv_wgrad'), (3, 'conv3x3_fused_dgrad_wgrad')):
        ms, cnt, fl = C.c_double(), C.c_int64(), C.c_double()
        check(lib.satcv_prof_collect(kind, C.byref(ms), C.byref(cnt), C.byref(fl)))
        prof[name] = dict(ms=ms.value, launches=cnt.value, flops=fl.value)

    extra = {'loss_last': loss, 'kernel_ms_per_step': {k: round(v['ms'] / nprof, 3) for k, v in prof.items()},
             'region_ms_per_step': {'median': round(1000 * dt / args.steps, 3), 'min': round(1000 * min(regions) / args.steps, 3),
                                    'max': round(1000 * max(regions) / args.steps, 3), 'repeats': len(regions)},
             'kernel_tflops': {k: round(v['flops'] / max(v['ms'], 1e-9) / 1e9, 1) for k, v in prof.items()},
             'prof_events_in_timed_regions': bool(args.prof_in_timed),
             'ms_per_step_with_prof_events': round(1000 * dt_prof / args.steps, 3) if dt_prof is not None else round(1000 * dt / args.steps, 3),
             'box_probe': probe}
    if dist is not None:
        extra['grad_exchange'] = {'via': 'satcv_allreduce_grads (C ABI, RCCL)' if parallel.cabi_comm() is not None else f'torch.distributed {dist.get_backend()}',
                                  'payload': sync.payload, 'bucket_MiB': sync.per * 4 / 2 ** 20, 'calls': parallel._comm['calls']}
    if not args.no_infer:
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
        chips = torch.from_numpy(np.random.default_rng(5).beta(2, 5, (36, 384, 384, CHN)).astype(np.float32)).cuda()      # 4 scenes
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
        # (batch 1 = the configuration BASELINE names; batch 16 shows what batching the tiles buys.  < 1 s together)
        mt.reset_uids()
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
        d, d3 = prof['conv3x3_igemm_fwd_dgrad'], prof['conv3x3_fused_dgrad_wgrad']
        peak = PEAK_BF16_TFLOPS if args.dtype == 'bfloat16' else 157.3
        aw = alg_work(B, 2 if args.dtype == 'bfloat16' else 4, CHN)
        a3 = aw['conv3_fwd_dgrad']
        # achieved = ALGORITHMIC FLOPs of the 3x3 implicit-GEMM launches of the timed steps (forward + data gradient; true Cin: 4 for the
        # first layer although 16 channels are stored) / their summed HIP-event durations on the launch stream.  The data gradients of the
        # thin layers whose whole backward is ONE fused launch (BatchNorm apply + data gradient + weight gradient, csrc/conv_bwd_fused.hip)
        # are not executed by this kernel class any more: their FLOPs leave the numerator as their time left the denominator, and the fused
        # launches (HBM-bound) are reported with their own roofline under extra.fused_backward.
        fused = [f.work for f in plan.bwd if getattr(f, 'work', {}).get('kind') == 'bwd_fused']
        fused_dgrad_flops = sum(2.0 * w['px'] * w['cin'] * w['cout'] * 9 for w in fused if not w.get('nodx'))
        fused_bytes = sum(w['px'] * (2 * w['cout'] + (1 if w.get('nodx') else 2) * w['cin']) * w['esize']
                          + (w['px'] // 4 * w['cout'] * (w['esize'] + 1) if w.get('pooled') else 0) for w in fused)
        fused_dgrad_bytes = sum(w['px'] * (w['cin'] + w['cout']) * w['esize'] + 9 * w['cin'] * w['cout'] * w['esize'] for w in fused if not w.get('nodx'))
        n_fused_dgrad = sum(1 for w in fused if not w.get('nodx'))
        ach = (a3['flops'] - fused_dgrad_flops) * nprof / max(d['ms'], 1e-9) / 1e9
        # like-for-like with rounds 1-2 (every 3x3 forward + data-gradient FLOP of the step over the time of every launch that executes
        # one of them, the fused thin-layer launches with their WHOLE duration: a lower bound)
        ach_all = a3['flops'] * nprof / max(d['ms'] + d3['ms'], 1e-9) / 1e9
        extra['fused_backward'] = {'launches_per_step': len(fused), 'ms_per_step': round(d3['ms'] / nprof, 3),
                                   'algorithmic_MB_per_step': round(fused_bytes / 1e6, 1), 'bound': 'hbm',
                                   'achieved_TBps': round(fused_bytes * nprof / max(d3['ms'], 1e-9) / 1e9, 3), 'peak_TBps': PEAK_HBM_TBPS,
                                   'frac': round(fused_bytes * nprof / max(d3['ms'], 1e-9) / 1e9 / PEAK_HBM_TBPS, 4),
                                   'tflops': round(d3['flops'] / max(d3['ms'], 1e-9) / 1e9, 1),
                                   'data_gradient_gflop_per_step_inside': round(fused_dgrad_flops / 1e9, 1)}
        launches_per_step = max(d['launches'] / nprof, 1)
        traffic, traffic_src, traffic_digest = pmc_traffic_per_launch()
        hc = head_commit()
        # the PMC summary is a committed file of an earlier run of this command (counters cannot be collected inside the timed run): it is
        # STALE unless it was measured on these very kernel sources
        built = ''
        try:
            built = open(os.path.join(ROOT, 'satellite_computervision_amd', '_build_commit.txt')).read().strip()
        except Exception:
            pass
        traffic_stale = not (traffic_digest and traffic_digest == csrc_digest())
        out = {
            'metric': f'tiles/sec (train) 256x256x{CHN} U-Net', 'value': round(value, 2), 'unit': 'tiles/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1000 * dt / args.steps, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16' if args.dtype == 'bfloat16' else 'f32', 'data': 'synthetic',
            'config': {'workload': f'U-Net 256x256x{CHN} {args.dtype} training, batch {B} per GPU (BASELINE configs[{1 if CHN == 4 else 3}])',
                       'global_batch': world * B, 'parallelism': f'dp{world}', 'loss': 'weighted_categorical_crossentropy',
                       'optimizer': 'adam(9e-4)'},
            'roofline': {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
                         'traffic': traffic, 'traffic_unit': 'MB per launch (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE)', 'traffic_source': traffic_src, 'traffic_stale': traffic_stale, 'csrc_digest': csrc_digest(), 'head_commit': hc, 'build_commit': built,
                         'algorithmic_bytes_per_launch_MB': round((a3['bytes'] - fused_dgrad_bytes) / max(a3['launches'] - n_fused_dgrad, 1) / 1e6, 1),
                         'algorithmic_gflop_per_step': round(a3['flops'] / 1e9, 1),
                         'kernel': '3x3 implicit-GEMM conv (forward + data gradient launches; the thin layers\' data gradients run inside the fused backward launches: extra.fused_backward)',
                         'algorithmic_gflop_per_step_in_these_launches': round((a3['flops'] - fused_dgrad_flops) / 1e9, 1), 'launches_per_step': launches_per_step,
                         'avg_launch_us': round(1000 * d['ms'] / max(d['launches'], 1), 2),
                         # whole step against the per-layer rooflines: sum over every conv-like layer and pass (forward, data
                         # gradient, weight gradient) of max(flops / MFMA peak, unfused bytes / 8 TB/s), / measured step time
                         'step_roofline_ms': round(1000 * aw['all']['roof_s'], 3),
                         'step_frac': round(1000 * aw['all']['roof_s'] / (1000 * dt / args.steps), 4),
                         'stack_roofline_ms': round(1000 * a3['roof_s'], 3),
                         'stack_roofline_frac': round(1000 * a3['roof_s'] / max((d['ms'] + d3['ms']) / nprof, 1e-9), 4),
                         'frac_like_for_like_r02': round(ach_all / peak, 4)},
            'model_tflops': round(value * (TRAIN_GFLOP_PER_TILE + (2 * 2 * 9 * 32 * (CHN - CH) * TILE * TILE / 1e9 if CHN != CH else 0.0)) / 1000, 2),
            'extra': extra,
        }
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(CH=CHN)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier(**BARRIER_KW)
        parallel.destroy_cabi_comm()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
