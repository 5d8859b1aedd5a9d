"""Graph + executor behind the Keras-style surface of model_tools.py.

A model is a list of block-level nodes (conv+BN+ReLU, max-pool, transposed conv,
concat+BN+ReLU, 1x1 head) built by the layer classes in model_tools.py.  `Plan`
lowers that list, for one (batch, H, W), into a static sequence of C-ABI kernel
launches on preallocated device buffers (torch tensors are only the allocator).

Fusion contract (what the HIP kernels expect):
  * a conv output is stored RAW (pre-BatchNorm); its BN affine + ReLU is applied by
    whichever kernel consumes it (conv loader, pool kernel, head);
  * concat([skip, up]) is never materialised: the consumer conv reads two sources;
  * train-mode BN statistics come from the producing kernel's epilogue.
There is no CPU path: everything here requires the HIP library and a ROCm device.
"""
import ctypes as C
import os
import itertools
from collections import defaultdict
from fractions import Fraction

import numpy as np
import torch

from . import ops, parallel
from ._lib import lib, check, F32, BF16, STAT_ROWS

_M16_DEFAULT = int(os.environ.get('SATCV_M16', '1'))      # library default of option igemm_m16 (csrc/api.hip)

BN_EPS = 1e-3
BN_MOMENTUM = 0.99


# ------------------------------------------------------------------ symbolic graph
class KTensor:
    """Symbolic NHWC tensor (the analogue of a KerasTensor)."""
    _ids = itertools.count()

    def __init__(self, node, channels, down=Fraction(1), name=None):
        self.id = next(KTensor._ids)
        self.node, self.channels, self.down, self.name = node, channels, down, name

    @property
    def shape(self):
        return (None, None, None, self.channels)


class Node:
    def __init__(self, op, inputs, layer=None, **attrs):
        self.op, self.inputs, self.layer, self.attrs = op, list(inputs), layer, attrs
        self.outputs = []

    def out(self, channels, down, name=None):
        t = KTensor(self, channels, down, name)
        self.outputs.append(t)
        return t


class ParamSpec:
    def __init__(self, name, shape, kind, init):
        self.name, self.shape, self.kind, self.init = name, tuple(shape), kind, init

    @property
    def size(self):
        return int(np.prod(self.shape))


# The bias of a convolution that feeds a training-mode BatchNormalization has an identically zero gradient: with
# dy = scale*rstd*(g - mean(g) - xhat*mean(g*xhat)) the sum over the batch vanishes because sum(xhat) = 0.  TensorFlow computes
# reduce_sum(dy) anyway and gets rounding noise (~1e-9), which Adam's normalisation turns into +-lr steps of a parameter that has
# no effect on any output.  Here the gradient is the exact 0 (the bias stays at its value) unless SATCV_BN_BIAS_NOISE=1 asks for
# the summed-noise form; the atomics it needs are also the one remaining source of run-to-run differences in a training step.
BIAS_NOISE = os.environ.get('SATCV_BN_BIAS_NOISE', '0') == '1'
# weight-gradient launches (second stream) enqueued AFTER the data gradient of their layer instead of before it: they then start beside the
# HBM-bound BatchNorm-backward kernels of the next layer rather than beside their own layer's MFMA-bound data gradient
# a layer's weight gradient is enqueued BEHIND its data gradient (both only need dy): with the weight gradients on 160 workgroups the
# step time is the same either way (9.54 ms, three A/B pairs) and the 3x3 data gradients run without their own layer's weight
# gradient beside them (roofline.frac 0.251 -> 0.262); SATCV_WGRAD_LATE=0 restores the earlier order
# round 6, weight gradients on 128 workgroups, profiling events off: enqueued BEFORE the data gradient the step is 7.89-7.90 ms against 7.99-8.00 behind it
# (profiles/r06_ab_env_switches.txt, A/B/A/B on one box) -- the weight gradient then starts beside its own layer's data gradient and the side stream
# finishes earlier; SATCV_WGRAD_LATE=1 restores the round-3 order
WGRAD_LATE = os.environ.get('SATCV_WGRAD_LATE', '0') == '1'
# round 5: decoder_block's up-sampling path backward as one launch (csrc/convt_bwd_fused.hip) for these Conv2DTranspose filter counts (SATCV_CTBF=0: off;
# SATCV_CTBF_COUTS restricts the set)
CTBF = os.environ.get('SATCV_CTBF', '1') != '0'
CTBF_COUTS = tuple(int(v) for v in os.environ.get('SATCV_CTBF_COUTS', '32,64').split(',') if v)
FUSE_RESIDUAL = os.environ.get('SATCV_FUSE_RESIDUAL', '1') == '1'      # inference: residual joins written by the block's last convolution
# round 6 experiment, OFF by default: Adam + operand repack of the parameters whose gradients are final early in the backward pass (99.5 % of them
# once the fourth encoder block is done) on the weight-gradient stream beside the rest of it.  Measured 7.87-7.89 ms against 7.82-7.87 without
# (profiles/r06_ab_early_opt_and_reduce_stream.txt): the trace of a step shows why -- the weight-gradient stream is busy to the last
# microsecond of the backward pass, so work moved onto it only lengthens it.  SATCV_EARLY_OPT=1 turns it on (tests run both).
EARLY_OPT = os.environ.get('SATCV_EARLY_OPT', '0') == '1'
EARLY_OPT_FRAC = 0.05
FUSE_DGRAD_ALL = os.environ.get('SATCV_FUSE_DGRAD_BN_BWD', '1') == '2'      # 2: every eligible data gradient carries the BN-backward sums


def rup(a, b):
    return (a + b - 1) // b * b


def topo_nodes(outputs):
    seen, order = set(), []

    def visit(node):
        if id(node) in seen:
            return
        seen.add(id(node))
        for t in node.inputs:
            visit(t.node)
        order.append(node)
    for t in outputs:
        visit(t.node)
    return order


# -------------------------------------------------------------------- runtime refs
class TRef:
    """Runtime tensor: 1-2 device sources + an optional pending BN affine (+ReLU)."""

    def __init__(self, srcs, n, h, w, affine=None, relu=False):
        self.srcs, self.n, self.h, self.w = srcs, n, h, w      # srcs: [(torch tensor, channels)]
        self.affine, self.relu = affine, relu                   # affine: dict(scale, shift, mean, rstd)

    @property
    def c(self):
        return sum(c for _, c in self.srcs)


def _fp(t, off=0):
    """device address of element `off` of a float32 (parameters, affines) or float64 (statistics rows) tensor, or None."""
    return None if t is None else t.data_ptr() + t.element_size() * off


class Runtime:
    """Device state of one model: flat fp32 parameters / gradients / Adam slots, packed MFMA
    weight images, BN moving statistics."""

    def __init__(self, model, dtype):
        if not torch.cuda.is_available():
            raise RuntimeError('satellite_computervision_amd needs a ROCm GPU (no CPU fallback)')
        self.model, self.dtype = model, dtype
        self.tdtype = ops.TORCH_DTYPE[dtype]
        self.esize = 2 if dtype == BF16 else 4
        self.dev = torch.device('cuda', torch.cuda.current_device())
        # flat layout: trainables (Adam sees one buffer), then non-trainable BN moving statistics
        self.offsets = {}
        off = 0
        for p in model.param_specs:
            if p.kind in ('moving_mean', 'moving_var'):
                continue
            self.offsets[p.name] = off
            off += rup(p.size, 4)
        self.n_train = off
        soff = 0
        self.soffsets = {}
        for p in model.param_specs:
            if p.kind in ('moving_mean', 'moving_var'):
                self.soffsets[p.name] = soff
                soff += rup(p.size, 4)
        self.pflat = torch.zeros(max(off, 4), dtype=torch.float32, device=self.dev)
        self.gflat = torch.zeros_like(self.pflat)
        self.sflat = torch.zeros(max(soff, 4), dtype=torch.float32, device=self.dev)
        self.adam_m = self.adam_v = None
        self.adam_state = torch.tensor([1e-3, 0.0, 1.0, 0.0], dtype=torch.float32, device=self.dev)
        self.lr_mul = None
        self.dropout_seed = 0x5A7C0FFEE
        host = np.zeros(self.pflat.numel(), np.float32)
        hs = np.zeros(self.sflat.numel(), np.float32)
        for p in model.param_specs:
            v = np.asarray(p.init(), np.float32).reshape(-1)
            if p.name in self.offsets:
                host[self.offsets[p.name]:self.offsets[p.name] + p.size] = v
            else:
                hs[self.soffsets[p.name]:self.soffsets[p.name] + p.size] = v
        self.pflat.copy_(torch.from_numpy(host))
        self.sflat.copy_(torch.from_numpy(hs))
        self.specs = {p.name: p for p in model.param_specs}
        # packed weights per conv-like layer
        self.packed = {}
        for node in model.nodes:
            if node.op in ('cba', 'convT'):
                lay = node.layer
                if lay.name in self.packed:
                    continue
                ks = self.specs[lay.kernel_name].shape
                tr = node.op == 'convT'
                cin, cout = (ks[3], ks[2]) if tr else (ks[2], ks[3])
                cin_pad = rup(cin, 16)
                ef, ed, _, _ = ops.packed_sizes(ks[0], ks[1], cin, cout, cin_pad, tr)
                self.packed[lay.name] = dict(
                    fwd=torch.zeros(ef, dtype=self.tdtype, device=self.dev),
                    dgrad=torch.zeros(ed, dtype=self.tdtype, device=self.dev),
                    k=(ks[0], ks[1]), cin=cin, cout=cout, cin_pad=cin_pad, transposed=tr)
        self._ptabs = {}
        self.repack()
        self.plans = {}

    # ---- parameter access
    def pptr(self, name):
        return _fp(self.pflat, self.offsets[name])

    def gptr(self, name):
        return _fp(self.gflat, self.offsets[name])

    def sptr(self, name):
        return _fp(self.sflat, self.soffsets[name])

    def get_param(self, name):
        p = self.specs[name]
        if name in self.offsets:
            return self.pflat[self.offsets[name]:self.offsets[name] + p.size].view(p.shape)
        return self.sflat[self.soffsets[name]:self.soffsets[name] + p.size].view(p.shape)

    def get_grad(self, name):
        p = self.specs[name]
        return self.gflat[self.offsets[name]:self.offsets[name] + p.size].view(p.shape)

    def set_param(self, name, value):
        self.get_param(name).copy_(torch.as_tensor(np.asarray(value, np.float32)).to(self.dev).view(self.specs[name].shape))

    def _pack_table(self, lo=0, hi=None):
        """device table of the pack jobs (forward + data-gradient image) of every layer whose kernel starts in [lo, hi) of the flat parameter
        buffer, for satcv_pack_weights_batched; None when no layer does"""
        from ._lib import PackJob
        jobs = []
        for lname, pk in self.packed.items():
            if not (lo <= self.offsets[lname + '/kernel'] < (self.n_train if hi is None else hi)):
                continue
            taps, cin, cout, cp = pk['k'][0] * pk['k'][1], pk['cin'], pk['cout'], pk['cin_pad']
            src = self.pptr(lname + '/kernel')
            if not pk['transposed']:
                jobs.append(PackJob(src, pk['fwd'].data_ptr(), 0, taps, cin, cout, cp, rup(cout, 32)))
                jobs.append(PackJob(src, pk['dgrad'].data_ptr(), 1, taps, cin, cout, rup(cout, 16), rup(cin, 32)))
            else:
                jobs.append(PackJob(src, pk['fwd'].data_ptr(), 2, taps, cin, cout, cp, rup(taps * cout, 32)))
                jobs.append(PackJob(src, pk['dgrad'].data_ptr(), 3, taps, cin, cout, rup(taps * cout, 16), rup(cin, 32)))
        if not jobs:
            return None
        prefix, tot = [], 0
        for j in jobs:
            prefix.append(tot)
            tot += int(lib.satcv_pack_job_items(C.byref(j)))
        arr = (PackJob * len(jobs))(*jobs)
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.dev)
        pre = torch.tensor(prefix, dtype=torch.int64, device=self.dev)
        return dict(jobs=raw, prefix=pre, n=len(jobs), total=tot)

    def repack(self, lo=0, hi=None, stream=None):
        """fp32 Keras-layout kernels -> MFMA operand images (after every weight update), one launch for all layers (or for the layers whose
        kernels start in [lo, hi) of the flat buffer: the early part of a split optimizer step, engine.Plan)."""
        if not self.packed:
            return
        key = (lo, hi)
        if key not in self._ptabs:
            self._ptabs[key] = self._pack_table(lo, hi)
        t = self._ptabs[key]
        if t is None:
            return
        check(lib.satcv_pack_weights_batched(t['jobs'].data_ptr(), t['prefix'].data_ptr(), t['n'], t['total'], self.dtype,
                                             ops.stream_ptr() if stream is None else stream))

    def ensure_adam(self):
        if self.adam_m is None:
            self.adam_m = torch.zeros_like(self.pflat)
            self.adam_v = torch.zeros_like(self.pflat)

    def plan(self, n, h, w, training):
        # tf.keras runs a BatchNormalization whose `trainable` is False in INFERENCE mode even inside fit() (moving statistics, no
        # update): the set of frozen layers is part of a training plan's identity (retrain_model(freeze=True), utils/model_tools.py:1174)
        frozen = tuple(sorted(l.name for l in self.model.layers if not l.trainable)) if training else ()
        key = (n, h, w, bool(training), frozen)
        if key not in self.plans:
            self.plans[key] = Plan(self, n, h, w, training, frozen)
        return self.plans[key]


class Plan:
    """Static launch sequence for one input shape."""

    def __init__(self, rt, n, h, w, training, frozen=()):
        self.rt, self.n, self.h, self.w, self.training = rt, n, h, w, training
        self.frozen = set(frozen)
        self.tile_policy = 2 if (training and _M16_DEFAULT == 1) else 0      # satcv_conv_desc.tile_policy of this plan's convolution launches
        self.eo_lo, self.early_opt, self.eo_done = None, None, False          # split optimizer step (see _build_backward: `early`)
        self.fwd, self.bwd = [], []
        self.keep = []                        # ctypes descriptors / tensors kept alive
        self.dropouts = []                    # dropout masks (regenerated every training step)
        self.x_inputs = []                    # fp32 staging tensor of every model input
        self.x_by_tid = {}
        self.x_src = {}                       # tensor id -> device pointer of a caller's batch read in place (Model._stage_x), else the staging tensor
        # (SATCV_SIDE_PRIORITY: HIP stream priority of the weight-gradient stream -- positive = lower than the main stream; measured)
        _sp = os.environ.get('SATCV_SIDE_PRIORITY')
        self.side = (torch.cuda.Stream(priority=int(_sp)) if _sp is not None else torch.cuda.Stream()) if (training and rt.model.wgrad_side_stream) else None
        self.step_count = 0
        self.outputs = {}
        self.sync_bn = bool(training and getattr(rt.model, 'sync_bn', False) and parallel.active())
        self.amax_of = {}                     # encoder output tensor id -> arg-max bytes of its 2 x 2 pooling windows (training)
        self._bn_jobs = []
        self._build()
        if self._bn_jobs:
            tab = torch.tensor([[int(v or 0) for v in j] for j in self._bn_jobs], dtype=torch.int64).to(rt.dev)      # satcv_bn_affine_job rows: 6 pointers, c, 2 optional pointers
            self.keep.append(tab)
            nj = len(self._bn_jobs)
            self.fwd.insert(0, lambda st: check(lib.satcv_bn_affine_infer_batched(tab.data_ptr(), nj, BN_EPS, st)))

    # -- helpers
    def _z(self, *shape, dtype=None):
        t = torch.zeros(*shape, dtype=dtype or self.rt.tdtype, device=self.rt.dev)
        self.keep.append(t)
        return t

    def _dims(self, t):
        hh, ww = self.h * t.down, self.w * t.down
        if hh.denominator != 1 or ww.denominator != 1:
            raise ValueError(f'input {self.h}x{self.w} is not divisible by the model downsampling ({1 / t.down})')
        return int(hh), int(ww)

    def _conv_step(self, role='fwd', **kw):
        kw.setdefault('tile_policy', self.tile_policy)
        d = ops.make_conv_desc(**kw)
        self.keep.append(d)
        fn = lambda st, d=d: check(lib.satcv_conv2d_igemm(C.byref(d), st))
        # label + algorithmic work of the launch (tools/step_probe.py, bench.py's per-layer roofline)
        k, cin, cout = kw.get('kh', 1) * kw.get('kw', 1), kw.get('c0', 0) + (kw.get('c1', 0) or 0), kw['cout']
        px = kw['n'] * kw['h'] * kw['w_']
        fn.label = (f"conv_{role} k{kw.get('kh', 1)} d{kw.get('dil', 1)} n{kw['n']} {kw['h']}x{kw['w_']} {kw.get('c0', 0)}+{kw.get('c1', 0) or 0}->{cout}"
                    f"{' d2s' if kw.get('mode_out') else ''}{' s2d' if kw.get('mode_in') else ''}")
        fn.work = dict(kind='conv', role=role, taps=k, px=px, cin=cin, cout=cout, esize=self.rt.esize)
        return fn

    def _src_args(self, r):
        (x0, c0) = r.srcs[0]
        x1, c1 = (r.srcs[1] if len(r.srcs) > 1 else (None, 0))
        a = r.affine
        return dict(x0=x0.data_ptr(), c0=c0, x1=x1.data_ptr() if x1 is not None else None, c1=c1,
                    in_scale=_fp(a['scale']) if a else None, in_shift=_fp(a['shift']) if a else None,
                    in_relu=1 if (a and r.relu) else 0)

    def _bn_forward(self, bnname, stats, ld, off, c, count, updates, aff=None, aoff=0, conv_bias=None, bias_eff=None):
        """emit finalize (train) / affine (infer); returns affine dict of per-channel tensors
        (freshly allocated [c], or the caller's shared arrays written at channel offset aoff)."""
        rt = self.rt
        a = aff or dict(scale=self._z(c, dtype=torch.float32), shift=self._z(c, dtype=torch.float32),
                        mean=self._z(c, dtype=torch.float32), rstd=self._z(c, dtype=torch.float32))
        g, b = rt.pptr(bnname + '/gamma'), rt.pptr(bnname + '/beta')
        mm, mv = rt.sptr(bnname + '/moving_mean'), rt.sptr(bnname + '/moving_var')
        if self.training:
            bessel = 1 if rt.model.bn_bessel else 0
            if self.sync_bn:            # SyncBN: [Σx, Σx²] averaged over replicas before the finalize
                self.fwd.append(lambda st: parallel.allreduce_mean_(stats))
            if bnname in self.frozen:
                # frozen BatchNormalization: the batch statistics are consumed (rows cleared, batch mean / rstd kept for the backward
                # kernels' addressing) but the layer normalises with its MOVING statistics and does not update them
                ts, tt = self._z(c, dtype=torch.float32), self._z(c, dtype=torch.float32)
                self.fwd.append(lambda st: check(lib.satcv_bn_finalize_train(
                    _fp(stats, off), ld, c, float(count), g, b, BN_EPS, BN_MOMENTUM, updates, bessel, None, None,
                    _fp(ts), _fp(tt), _fp(a['mean'], aoff), _fp(a['rstd'], aoff), st)))
                self.fwd.append(lambda st: check(lib.satcv_bn_affine_infer(g, b, mm, mv, BN_EPS, c, _fp(a['scale'], aoff), _fp(a['shift'], aoff), st)))
                return a
            self.fwd.append(lambda st: check(lib.satcv_bn_finalize_train(
                _fp(stats, off), ld, c, float(count), g, b, BN_EPS, BN_MOMENTUM, updates, bessel, mm, mv,
                _fp(a['scale'], aoff), _fp(a['shift'], aoff), _fp(a['mean'], aoff), _fp(a['rstd'], aoff), st)))
        else:
            # inference: scale / shift depend on the parameters only -- every layer's pair is computed by ONE launch at the head of the
            # list (satcv_bn_affine_infer_batched, table built at the end of _build)
            self._bn_jobs.append((g, b, mm, mv, _fp(a['scale'], aoff), _fp(a['shift'], aoff), c, conv_bias, bias_eff))
        return a

    def _materialize(self, t, r, f=1, pooled=None, sink=None, actslot=None):
        """relu(bn(raw)) -> dense activated tensor (and optional pooled / stats).  actslot = (tensor, channel offset,
        channel stride): write into a slice of a shared concatenation buffer instead of a fresh tensor."""
        if not r.relu:
            raise NotImplementedError('materialising a linear (no-ReLU) BatchNorm output')
        (y, c) = r.srcs[0]
        hh, ww = r.h, r.w
        if actslot is None:
            act = self._z(self.n, hh, ww, c)
            act_ptr, act_ld = act.data_ptr(), c
        else:
            act = actslot[0]
            act_ptr, act_ld = actslot[0].data_ptr() + actslot[1] * self.rt.esize, actslot[2]
        st_ptr, ld = (None, 0)
        if sink is not None and self.training:
            st_ptr, ld = _fp(sink[0], sink[1]), sink[2]
        a = r.affine
        dt = self.rt.dtype
        if (self.training and pooled is not None and f == 2 and dt == BF16 and getattr(self.rt.model, 'fuse_pool_bwd', True)
                and hh % 8 == 0 and ww % 32 == 0 and c in (32, 64)):
            # training: also the arg-max byte of every window -- the fused pooled backward (satcv_conv2d_bwd_fused with dpool) routes
            # MaxPooling2D's gradient with it
            amax = self._z(self.n, hh // 2, ww // 2, c, dtype=torch.uint8)
            self.amax_of[t.id] = amax
            self.fwd.append(lambda st: check(lib.satcv_bn_relu_pool_amax(
                y.data_ptr(), _fp(a['scale']), _fp(a['shift']), act_ptr, act_ld, pooled.data_ptr(), amax.data_ptr(),
                st_ptr, ld, self.n, hh, ww, c, f, dt, st)))
            return TRef([(act, c)], self.n, hh, ww) if actslot is None else None
        self.fwd.append(lambda st: check(lib.satcv_bn_relu_pool(
            y.data_ptr(), _fp(a['scale']), _fp(a['shift']), act_ptr, act_ld, pooled.data_ptr() if pooled is not None else None,
            st_ptr, ld, self.n, hh, ww, c, f, dt, st)))
        return TRef([(act, c)], self.n, hh, ww) if actslot is None else None

    # -- lowering
    def _build(self):
        rt, m, n = self.rt, self.rt.model, self.n
        T, dt, es = rt.tdtype, rt.dtype, rt.esize
        training = self.training
        consumers = defaultdict(list)
        for node in m.nodes:
            for t in node.inputs:
                consumers[t.id].append(node)
        vals, acts, ctx = {}, {}, {}
        sinks, cat_stats = {}, {}
        order = {id(nd): i for i, nd in enumerate(m.nodes)}
        fused_join = {}                            # add_relu node id -> the shortcut tensor its sum was written into
        folded_plain = {}                          # tensor id -> a conv -> BatchNorm branch stored normalised, waiting for the join's other branch
        for node in m.nodes:
            if node.op == 'concat_bn_relu':
                a, b = node.inputs
                ctot = a.channels + b.channels
                st = self._z(STAT_ROWS, 2, ctot, dtype=torch.float64)
                cat_stats[id(node)] = st
                sinks[a.id] = (st, 0, ctot)
                sinks[b.id] = (st, a.channels, ctot)

        slots, actslots = {}, {}
        for node in m.nodes:
            if node.op == 'concat':
                if all(t.node.op == 'cba' and len(consumers[t.id]) > 1 for t in node.inputs):
                    # concatenation of ACTIVATED encoder outputs that are also pooled (Siamese skips): the pool kernels write
                    # their activation straight into channel slices of one tensor
                    ctot = sum(t.channels for t in node.inputs)
                    hh, ww = self._dims(node.inputs[0])
                    shared = self._z(n, hh, ww, ctot)
                    off = 0
                    for t in node.inputs:
                        actslots[t.id] = (shared, off, ctot)
                        if node.outputs[0].id in sinks:                      # statistics for the decoder's concat BatchNorm
                            st_, so_, sl_ = sinks[node.outputs[0].id]
                            sinks[t.id] = (st_, so_ + off, sl_)
                        off += t.channels
                    ctx[id(node)] = dict(act=shared, ctot=ctot)
                    continue
                if not all(t.node.op == 'cba' and len(consumers[t.id]) == 1 for t in node.inputs):
                    raise NotImplementedError('concatenate is supported for conv_batch_act outputs consumed only by the concat (ASPP)')
                ctot = sum(t.channels for t in node.inputs)
                hh, ww = self._dims(node.inputs[0])
                yshared = self._z(n, hh, ww, ctot)
                stshared = self._z(STAT_ROWS, 2, ctot, dtype=torch.float64) if training else None
                aff = dict(scale=self._z(ctot, dtype=torch.float32), shift=self._z(ctot, dtype=torch.float32),
                           mean=self._z(ctot, dtype=torch.float32), rstd=self._z(ctot, dtype=torch.float32))
                off = 0
                for t in node.inputs:
                    slots[t.id] = dict(y=yshared, off=off, ctot=ctot, stats=stshared, aff=aff)
                    off += t.channels
                ctx[id(node)] = dict(y=yshared, ctot=ctot, aff=aff)

        for node in m.nodes:
            op = node.op
            if op == 'input':
                t = node.outputs[0]
                cp = rup(t.channels, 16)
                x = self._z(n, self.h, self.w, cp)
                xin = self._z(n, self.h, self.w, t.channels, dtype=torch.float32)
                self.x_f32 = xin
                self.x_inputs.append(xin)
                self.x_by_tid[t.id] = xin
                npix = n * self.h * self.w
                cc = t.channels
                self.fwd.append(lambda st, xin=xin, x=x, npix=npix, cc=cc, cp=cp, tid=t.id: check(lib.satcv_ingest_nhwc(
                    self.x_src.get(tid) or xin.data_ptr(), x.data_ptr(), npix, cc, cp, dt, st)))
                vals[t.id] = TRef([(x, cp)], n, self.h, self.w)
            elif op == 'cba':
                tin, tout = node.inputs[0], node.outputs[0]
                r = vals[tin.id]
                lay = node.layer
                pk = rt.packed[lay.name]
                cout = tout.channels
                if cout % 16:
                    raise NotImplementedError(f'{lay.name}: filters must be a multiple of 16 (got {cout})')
                if r.c != pk['cin_pad']:
                    raise ValueError(f'{lay.name}: input has {r.c} stored channels, kernel expects {pk["cin_pad"]}')
                k, dil = node.attrs['k'], node.attrs['dil']
                stride, relu_out = node.attrs.get('stride', 1), node.attrs.get('relu', True)
                hin, win = r.h, r.w
                if stride > 1:
                    if training:
                        raise NotImplementedError('strided convolutions are inference-only (DeepLab backbone)')
                    r = TRef(r.srcs, r.n, (hin - 1) // stride + 1, (win - 1) // stride + 1, r.affine, r.relu)   # output grid
                sl = slots.get(tout.id)
                if not node.attrs.get('bn', True):
                    # plain Conv2D + bias, no normalisation (atrous CNN family): consumers read the stored values as they are
                    if sl is not None or stride > 1:
                        raise NotImplementedError('a convolution without BatchNormalization as a concatenation branch / with a stride')
                    y = self._z(n, r.h, r.w, cout)
                    self.fwd.append(self._conv_step(w=pk['fwd'].data_ptr(), bias=rt.pptr(lay.name + '/bias'), y=y.data_ptr(), ldy=cout, n=n, h=r.h, w_=r.w,
                                                    cout=cout, cout_pad=rup(cout, 32), kh=k, kw=k, dil=dil, dtype=dt, **self._src_args(r)))
                    vals[tout.id] = TRef([(y, cout)], n, r.h, r.w)
                    ctx[id(node)] = dict(r=r, y=y, yoff=0, ldy=cout, aff=None, aoff=0, cout=cout, k=k, dil=dil)
                    continue
                # inference, residual join (ResNet bottleneck: ReLU(BN(conv) + shortcut), utils-free build-defined DeepLab backbone): when this
                # block's output only feeds the join and nothing reads the shortcut afterwards, the convolution applies its own
                # BatchNormalization in the epilogue (out_scale / bias_eff from the batched affine launch) and writes
                # ReLU(shortcut + result) IN PLACE over the shortcut (accumulate = 2): no add_relu launch, no third tensor
                # (20 launches of ~10 us on a single 512 x 512 tile, 11 % of the kernel time at batch 16; SATCV_FUSE_RESIDUAL=0: separate)
                join = consumers[tout.id][0] if (not training and sl is None and not relu_out and FUSE_RESIDUAL
                                                 and len(consumers[tout.id]) == 1 and consumers[tout.id][0].op == 'add_relu'
                                                 and tout in consumers[tout.id][0].inputs) else None
                if join is not None:
                    tsc = join.inputs[1] if join.inputs[0] is tout else join.inputs[0]      # the join's other addend
                    rs_ = vals.get(tsc.id)
                    fold_args = None
                    if rs_ is None:
                        # (B) the other addend is a conv -> BatchNorm branch that runs LATER in the list (the projection shortcut of a stage's
                        # first block): store this branch normalised (its BatchNorm in the epilogue, nothing pending), the later branch
                        # then adds itself in place (A)
                        if (stride == 1 and tsc.node is not None and tsc.node.op == 'cba' and len(consumers[tsc.id]) == 1
                                and tsc.node.attrs.get('bn', True) and not tsc.node.attrs.get('relu', True) and tsc.channels == cout):
                            y = self._z(n, r.h, r.w, cout)
                            fold_args = dict(y=y.data_ptr(), accumulate=0)
                            folded_plain[tout.id] = y
                            sb = y
                    else:
                        # (A) the other addend is final: a plain activated tensor written by an earlier join, or a branch stored normalised (B)
                        src_ok = (tsc.id in folded_plain) or (tsc.node is not None and tsc.node.op == 'add_relu' and stride == 1)
                        if (src_ok and not rs_.affine and len(rs_.srcs) == 1 and rs_.srcs[0][1] == cout and rs_.srcs[0][0].shape[-1] == cout
                                and (rs_.h, rs_.w) == (r.h, r.w)
                                and all(cn is join or order[id(cn)] < order[id(node)] for cn in consumers[tsc.id])):
                            sb = rs_.srcs[0][0]
                            fold_args = dict(y=sb.data_ptr(), accumulate=2)
                            fused_join[id(join)] = sb
                    if fold_args is not None:
                        be = self._z(cout, dtype=torch.float32)
                        aff = self._bn_forward(lay.bn_name, None, cout, 0, cout, n * r.h * r.w, node.attrs.get('bn_updates', 1),
                                               conv_bias=rt.pptr(lay.name + '/bias'), bias_eff=_fp(be))
                        self.fwd.append(self._conv_step(w=pk['fwd'].data_ptr(), bias=_fp(be), out_scale=_fp(aff['scale']), ldy=cout,
                                                        n=n, h=r.h, w_=r.w, cout=cout, cout_pad=rup(cout, 32), kh=k, kw=k, dil=dil, dtype=dt,
                                                        stride=stride, hin=hin if stride > 1 else 0, win=win if stride > 1 else 0,
                                                        **fold_args, **self._src_args(r)))
                        vals[tout.id] = TRef([(sb, cout)], n, r.h, r.w)
                        ctx[id(node)] = dict(r=r, y=sb, yoff=0, ldy=cout, aff=None, aoff=0, cout=cout, k=k, dil=dil)
                        continue
                if sl is None:
                    y = self._z(n, r.h, r.w, cout)
                    stats = self._z(STAT_ROWS, 2, cout, dtype=torch.float64) if training else None
                    yoff, ldy, aoff, aff_in = 0, cout, 0, None
                else:           # branch of a concatenation: write into the channel slice of the shared tensor
                    y, stats, yoff, ldy, aoff, aff_in = sl['y'], sl['stats'], sl['off'], sl['ctot'], sl['off'], sl['aff']
                self.fwd.append(self._conv_step(w=pk['fwd'].data_ptr(), bias=rt.pptr(lay.name + '/bias'), y=y.data_ptr() + yoff * es, ldy=ldy,
                                                stats=_fp(stats, aoff), stats_ld=ldy, n=n, h=r.h, w_=r.w, cout=cout, cout_pad=rup(cout, 32),
                                                kh=k, kw=k, dil=dil, dtype=dt, stride=stride, hin=hin if stride > 1 else 0,
                                                win=win if stride > 1 else 0, **self._src_args(r)))
                aff = self._bn_forward(lay.bn_name, stats, ldy, aoff, cout, n * r.h * r.w, node.attrs.get('bn_updates', 1), aff_in, aoff)
                if sl is None:
                    vals[tout.id] = TRef([(y, cout)], n, r.h, r.w, affine=aff, relu=relu_out)
                ctx[id(node)] = dict(r=r, y=y, yoff=yoff, ldy=ldy, aff=aff, aoff=aoff, cout=cout, k=k, dil=dil)
            elif op == 'concat':
                cx = ctx[id(node)]
                hh, ww = self._dims(node.inputs[0])
                if 'act' in cx:
                    vals[node.outputs[0].id] = TRef([(cx['act'], cx['ctot'])], n, hh, ww)
                else:
                    vals[node.outputs[0].id] = TRef([(cx['y'], cx['ctot'])], n, hh, ww, affine=cx['aff'], relu=True)
            elif op == 'pool':
                tin, tout = node.inputs[0], node.outputs[0]
                r = vals[tin.id]
                f = node.attrs['f']
                if not (r.affine and len(r.srcs) == 1):
                    raise NotImplementedError('max-pool expects the output of a conv_batch_act block')
                c = r.c
                if r.h % f or r.w % f:
                    raise ValueError(f'input {self.h}x{self.w} is not divisible by the model downsampling (a {r.h}x{r.w} map meets a {f}x{f} pool)')
                pooled = self._z(n, r.h // f, r.w // f, c)
                others = [cn for cn in consumers[tin.id] if cn is not node]
                if others:
                    acts[tin.id] = self._materialize(tin, r, f, pooled, sinks.get(tin.id), actslots.get(tin.id))
                else:
                    (y, _) = r.srcs[0]
                    a = r.affine
                    hh, ww = r.h, r.w
                    self.fwd.append(lambda st, y=y, a=a, pooled=pooled, hh=hh, ww=ww, c=c, f=f: check(lib.satcv_bn_relu_pool(
                        y.data_ptr(), _fp(a['scale']), _fp(a['shift']), None, 0, pooled.data_ptr(), None, 0, n, hh, ww, c, f, dt, st)))
                vals[tout.id] = TRef([(pooled, c)], n, r.h // f, r.w // f)
                ctx[id(node)] = dict(f=f)
            elif op == 'convT':
                tin, tout = node.inputs[0], node.outputs[0]
                r = vals[tin.id]
                if len(r.srcs) != 1:
                    raise NotImplementedError('transposed conv on a concatenated input')
                lay = node.layer
                pk = rt.packed[lay.name]
                cout, f = tout.channels, node.attrs['f']
                if cout % 16:
                    raise NotImplementedError(f'{lay.name}: transposed-conv filters must be a multiple of 16')
                u = self._z(n, r.h * f, r.w * f, cout)
                sink = sinks.get(tout.id) if training else None
                self.fwd.append(self._conv_step(w=pk['fwd'].data_ptr(), bias=rt.pptr(lay.name + '/bias'), y=u.data_ptr(), ldy=cout,
                                                stats=_fp(sink[0], sink[1]) if sink else None, stats_ld=sink[2] if sink else 0,
                                                n=n, h=r.h, w_=r.w, cout=f * f * cout, cout_pad=rup(f * f * cout, 32), kh=1, kw=1, dil=1,
                                                mode_out=1, f=f, cstat=cout, dtype=dt, **self._src_args(r)))
                vals[tout.id] = TRef([(u, cout)], n, r.h * f, r.w * f)
                ctx[id(node)] = dict(r=r, u=u, cout=cout, f=f)
            elif op == 'concat_bn_relu':
                ta, tb = node.inputs
                tout = node.outputs[0]
                if ta.id not in acts:
                    ra0 = vals[ta.id]
                    acts[ta.id] = self._materialize(ta, ra0, 1, None, sinks.get(ta.id)) if ra0.affine else ra0
                ra, rb = acts[ta.id], vals[tb.id]
                if (ra.h, ra.w) != (rb.h, rb.w):
                    raise ValueError(f'concatenation of a {ra.h}x{ra.w} skip with a {rb.h}x{rb.w} up-sampled map: the input size must be divisible by the model downsampling')
                if rb.affine or len(rb.srcs) != 1 or len(ra.srcs) != 1:
                    raise NotImplementedError('concat+BN expects (activated skip, transposed-conv output)')
                ca, cb = ra.c, rb.c
                st = cat_stats[id(node)]
                aff = self._bn_forward(node.layer.name, st, ca + cb, 0, ca + cb, n * ra.h * ra.w, 1)
                vals[tout.id] = TRef([ra.srcs[0], rb.srcs[0]], n, ra.h, ra.w, affine=aff, relu=True)
                ctx[id(node)] = dict(ra=ra, rb=rb, aff=aff, ca=ca, cb=cb)
            elif op == 'maxpool':
                tin, tout = node.inputs[0], node.outputs[0]
                r = vals[tin.id]
                if r.affine:
                    r = self._materialize(tin, r)
                (src, c) = r.srcs[0]
                kk, ss, pp = node.attrs['k'], node.attrs['s'], node.attrs['pad']
                ho, wo = (r.h + 2 * pp - kk) // ss + 1, (r.w + 2 * pp - kk) // ss + 1
                out = self._z(n, ho, wo, c)
                self.fwd.append(lambda st, src=src, out=out, hh=r.h, ww=r.w, c=c, kk=kk, ss=ss, pp=pp: check(lib.satcv_maxpool(
                    src.data_ptr(), out.data_ptr(), n, hh, ww, c, kk, ss, pp, dt, st)))
                vals[tout.id] = TRef([(out, c)], n, ho, wo)
            elif op == 'raw':
                # the convolution output BEFORE its BatchNormalization (build_acnn_layers feeds it to the next Conv2D, utils/model_tools.py:930)
                r = vals[node.inputs[0].id]
                if len(r.srcs) != 1 or node.inputs[0].node.op != 'cba':
                    raise NotImplementedError('raw view of a tensor that is not a single convolution output')
                vals[node.outputs[0].id] = TRef(r.srcs, n, r.h, r.w)
            elif op == 'add_relu':
                ty, tsc = node.inputs
                tout = node.outputs[0]
                if id(node) in fused_join:            # written in place by the block's last convolution (above)
                    sb = fused_join[id(node)]
                    c = sb.shape[-1]
                    vals[tout.id] = TRef([(sb, c)], n, vals[ty.id].h, vals[ty.id].w)
                    ctx[id(node)] = dict(out=sb, c=c, h=vals[ty.id].h, w=vals[ty.id].w)
                    continue
                ry, rs = vals[ty.id], vals[tsc.id]
                if rs.affine and rs.relu and len(rs.srcs) == 1:       # shortcut = ReLU(BN(conv)) kept in raw form: activate it once
                    if tsc.id not in acts:
                        acts[tsc.id] = self._materialize(tsc, rs)
                    rs = acts[tsc.id]
                if len(ry.srcs) != 1 or len(rs.srcs) != 1 or ry.relu or (rs.affine and rs.relu):
                    raise NotImplementedError('residual join expects linear (BN without ReLU) branches')
                (yb, c), (sb, c2) = ry.srcs[0], rs.srcs[0]
                assert c == c2 and (ry.h, ry.w) == (rs.h, rs.w)
                out = self._z(n, ry.h, ry.w, c)
                ya, sa = ry.affine, rs.affine
                npx = n * ry.h * ry.w
                self.fwd.append(lambda st, yb=yb, sb=sb, out=out, ya=ya, sa=sa, npx=npx, c=c: check(lib.satcv_add_act(
                    yb.data_ptr(), _fp(ya['scale']) if ya else None, _fp(ya['shift']) if ya else None, sb.data_ptr(),
                    _fp(sa['scale']) if sa else None, _fp(sa['shift']) if sa else None, 1, out.data_ptr(), npx, c, dt, st)))
                vals[tout.id] = TRef([(out, c)], n, ry.h, ry.w)
                ctx[id(node)] = dict(out=out, c=c, h=ry.h, w=ry.w)
            elif op == 'upsample_head':
                tin, tout = node.inputs[0], node.outputs[0]
                lg = self.outputs[tin.id]                         # fp32 logits written by the linear head
                hctx = ctx[id(tin.node)]
                f = node.attrs['factor']
                act = 0 if node.attrs['activation'] == 'softmax' else 1
                hh, ww, ncls = hctx['r'].h, hctx['r'].w, hctx['ncls']
                probs = self._z(n, hh * f, ww * f, ncls, dtype=torch.float32)
                classes = self._z(*((n, hh * f, ww * f) if act == 0 else (n, hh * f, ww * f, ncls)), dtype=torch.int32)
                th = float(node.attrs.get('thresh', 0.5))
                self.fwd.append(lambda st, lg=lg, probs=probs, classes=classes, hh=hh, ww=ww, ncls=ncls, f=f, act=act, th=th: check(
                    lib.satcv_upsample_head(lg.data_ptr(), n, hh, ww, ncls, f, act, th, probs.data_ptr(), classes.data_ptr(), st)))
                self.outputs[tout.id] = probs
                ctx[id(node)] = dict(classes=classes)
                if training:
                    raise NotImplementedError('the up-sampling head is inference-only')
            elif op == 'dropout':
                tin, tout = node.inputs[0], node.outputs[0]
                r = vals[tin.id]
                if not training:                     # identity at inference (Keras semantics)
                    vals[tout.id] = r
                    if tin.id in acts:
                        acts[tout.id] = acts[tin.id]
                    continue
                ctot = r.c
                spatial = node.attrs['spatial']
                hw = r.h * r.w
                mask = self._z(*((n, ctot) if spatial else (n * hw, ctot)), dtype=torch.float32)
                out = self._z(n, r.h, r.w, ctot)
                rate = float(node.attrs['rate'])
                slot = len(self.dropouts)
                self.dropouts.append(dict(mask=mask, node=node))
                cnt = mask.numel()

                def gen(st, mask=mask, rate=rate, cnt=cnt, slot=slot):
                    # a fresh mask every step: counter = (step, dropout slot)
                    check(lib.satcv_dropout_mask(self.rt.dropout_seed, (self.step_count << 8) + slot, rate, cnt, mask.data_ptr(), st))
                self.fwd.append(gen)
                off = 0
                mode = 0 if spatial else 1
                for (src, cs) in r.srcs:             # dual-source (concat) inputs: one launch per source / channel slice
                    a = r.affine
                    self.fwd.append(lambda st, src=src, cs=cs, off=off, a=a, rl=1 if r.relu else 0, mask=mask, ctot=ctot, mode=mode, out=out,
                                    hw=hw: check(lib.satcv_dropout_apply(
                        src.data_ptr(), cs, _fp(a['scale'], off) if a else None, _fp(a['shift'], off) if a else None, rl,
                        _fp(mask, off), ctot, mode, out.data_ptr() + off * es, ctot, n, hw, cs, dt, st)))
                    off += cs
                vals[tout.id] = TRef([(out, ctot)], n, r.h, r.w)
                ctx[id(node)] = dict(mask=mask, spatial=spatial, ctot=ctot, hw=hw, r=r)
            elif op == 'head':
                tin, tout = node.inputs[0], node.outputs[0]
                r = vals[tin.id]
                if len(r.srcs) != 1:
                    raise NotImplementedError('head on concatenated input')
                lay = node.layer
                ncls = tout.channels
                act = {'softmax': 0, 'sigmoid': 1, 'linear': 2}[node.attrs['activation']]
                (y, c) = r.srcs[0]
                npix = n * r.h * r.w
                probs = self._z(n, r.h, r.w, ncls, dtype=torch.float32)
                classes = self._z(*((n, r.h, r.w) if act != 1 else (n, r.h, r.w, ncls)), dtype=torch.int32)
                hd = ops.make_head_desc(x=y.data_ptr(), ldx=c, cin=c, w=rt.pptr(lay.name + '/kernel'), b=rt.pptr(lay.name + '/bias'),
                                        ncls=ncls, activation=act, npix=npix, dtype=dt,
                                        in_scale=_fp(r.affine['scale']) if r.affine else None,
                                        in_shift=_fp(r.affine['shift']) if r.affine else None,
                                        thresh=node.attrs.get('thresh', 0.5), probs=probs.data_ptr(), classes=classes.data_ptr())
                self.keep.append(hd)
                self.fwd.append(lambda st, hd=hd: check(lib.satcv_head_fwd(C.byref(hd), st)))
                self.outputs[tout.id] = probs
                ctx[id(node)] = dict(r=r, probs=probs, classes=classes, act=act, ncls=ncls)
                self.head = ctx[id(node)]
                self.head_node = node
            elif op == 'classes':
                hctx = ctx[id(node.inputs[0].node)]
                if 'thresh' in node.attrs:
                    node.inputs[0].node.attrs['thresh'] = node.attrs['thresh']
                self.outputs[node.outputs[0].id] = hctx['classes']
            else:
                raise NotImplementedError(f'op {op}')
        self.vals = vals
        self.node_ctx = ctx
        if training:
            self._build_backward(consumers, vals, acts, ctx, cat_stats)

    def _build_backward(self, consumers, vals, acts, ctx, cat_stats):
        rt, m, n = self.rt, self.rt.model, self.n
        dt, es = rt.dtype, rt.esize
        gact, gpool, gpool_f, graw = {}, {}, {}, {}
        self.dbg = {}                       # layer name -> intermediate gradient tensors (diagnostics only)
        _cnt = defaultdict(int)
        for nd in m.nodes:
            if nd.layer is not None:
                _cnt[nd.layer.name] += 1
        shared_layers = {k for k, v in _cnt.items() if v > 1}      # layers applied to several inputs (Siamese encoders)
        ws_need = 0
        wdescs = []
        # deferred, batched slab sums (satcv_reduce_slabs_batched; SATCV_DEFER_REDUCE=0: one sum launch behind every weight-gradient launch):
        # a deferring launch keeps a workspace of its own; the sums run in groups, one launch per ~16 MiB of finished gradients (the bucket size
        # of the gradient exchange, whose checkpoints move to the group boundaries)
        # MEASURED (profiles/r05_ab_defer_reduce.txt, A/B/A/B on one box): 8.59 ms per step with one sum launch per layer, 8.625 with 4 batched
        # launches -- the per-layer sum reads slabs that were written microseconds earlier (37 MB: L2 / Infinity-Cache resident), the deferred one
        # finds 150-220 MB of slabs of several layers pushed out to HBM.  Opt-in (SATCV_DEFER_REDUCE=1).
        DEFER = int(os.environ.get('SATCV_DEFER_REDUCE', '0')) != 0
        # round 6: the slab sum of a side-stream weight gradient on a THIRD stream, behind an event of its launch.  The trace of a step
        # (profiles/r06_step_timeline_before.txt) shows the weight-gradient stream running to the very end of the step, and a quarter of its
        # time in the 21 slab sums (0.8 ms in the step for 0.17 ms of work: small HBM-bound launches between 128-workgroup MFMA launches).
        # With a workspace of its own per layer the next weight gradient does not have to wait for the sum of the previous one.
        # MEASURED, same box: 7.80-7.85 ms against 7.78-7.81 with the sums on the weight-gradient stream -- the step is bound by the chip's total work,
        # not by the order of these launches (profiles/r06_ab_early_opt_and_reduce_stream.txt).  Off by default (SATCV_REDUCE_STREAM=1).
        RED3 = (not DEFER) and self.side is not None and os.environ.get('SATCV_REDUCE_STREAM', '0') == '1'
        self.rstream = torch.cuda.Stream() if RED3 else None
        self._last_red_ev = None
        rpending = []                       # (descriptor, 'w' | 'f', gradient bytes) of launches whose sum has not been scheduled yet
        rgroups = []                        # {'items': [(desc, kind)], 'tab': device job table, filled once the workspaces are assigned}
        fused_ws_need = 0                   # the fused thin-layer backward launches run on the MAIN stream: a workspace of their own
        fdescs = []
        # loss -> dlogits is written by Model.train step into this buffer
        h = self.head
        self.dlogits = self._z(n * h['r'].h * h['r'].w, h['ncls'], dtype=torch.float32)
        self.loss_buf = self._z(1, dtype=torch.float32)

        def bn_bwd_steps(da, ldda, dp, lddp, f, yraw, ldy, aff, aoff, sums, sums_off, sums_ld, c, hh, ww, dy, lddy, dbias,
                         dgamma, dbeta, accum=0, linear=0, frozen=False, second=None, act=None):
            """act: dict(buf=[ROWS][2][c] rows in the ACTIVATED form, parts={'skip', 'pool'}) -- the parts of this layer's gradient whose
            sums a producer already formed (the decoder's concat-BN apply pass for the skip gradient `da`, the data gradient of the next
            encoder block for the pooled gradient `dp`); the separate reduce pass then only covers what is left."""
            coef = self._z(2, c, dtype=torch.float32)
            common = dict(yraw=yraw, ldy=ldy, scale=_fp(aff['scale'], aoff), shift=_fp(aff['shift'], aoff),
                          mean=_fp(aff['mean'], aoff), rstd=_fp(aff['rstd'], aoff), n=n, h=hh, w_=ww, c=c, dtype=dt,
                          f=f, sums=_fp(sums, sums_off), sums_ld=sums_ld, coef=_fp(coef), dy=dy, lddy_out=lddy, dbias=dbias, linear=linear)
            d = ops.make_bnbwd_desc(da=da, ldda=ldda, dpool=dp, lddp=lddp, **common, **(second or {}))
            self.keep.append(d)
            parts = act['parts'] if (act and not frozen) else set()
            need_da, need_dp = da is not None and 'skip' not in parts, dp is not None and 'pool' not in parts
            dr = d
            if parts and (need_da or need_dp):          # reduce pass over the part no producer covered
                dr = ops.make_bnbwd_desc(da=da if need_da else None, ldda=ldda if need_da else 0, dpool=dp if need_dp else None,
                                         lddp=lddp if need_dp else 0, **common)
                self.keep.append(dr)
            cnt = float(n * hh * ww)
            red = lambda st: check(lib.satcv_bn_bwd_reduce(C.byref(dr), st))
            red.label = f"bn_bwd_reduce n{n} {hh}x{ww} c{c} f{f}{' (part)' if dr is not d else ''}"
            red.work = dict(kind='bn_bwd_reduce', px=n * hh * ww, c=c, esize=es)
            if parts and not (need_da or need_dp):
                red = None
            if parts:
                abuf = act['buf']
                fin0 = lambda st: check(lib.satcv_bn_bwd_finalize2(_fp(sums, sums_off) if red is not None else None, sums_ld, _fp(abuf), c, c, cnt,
                                                                   _fp(aff['scale'], aoff), _fp(aff['shift'], aoff), _fp(aff['mean'], aoff),
                                                                   _fp(aff['rstd'], aoff), dgamma, dbeta, _fp(coef), accum, st))
            else:
                fin0 = lambda st: check(lib.satcv_bn_bwd_finalize(_fp(sums, sums_off), sums_ld, c, cnt, dgamma, dbeta, _fp(coef), accum, st))
            if self.sync_bn:
                def fin(st):            # SyncBN: Σdy, Σdy·x̂ averaged over replicas (linear + idempotent, so the whole buffer is reduced)
                    parallel.allreduce_mean_(sums)
                    if parts:
                        parallel.allreduce_mean_(act['buf'])
                    fin0(st)
            else:
                fin = fin0
            app = lambda st: check(lib.satcv_bn_bwd_apply(C.byref(d), st))
            app.label = f"bn_bwd_apply n{n} {hh}x{ww} c{c} f{f}"
            app.work = dict(kind='bn_bwd_apply', px=n * hh * ww, c=c, esize=es)
            app.coef = coef
            if frozen:
                # inference-mode BatchNormalization: dy = scale * g * mask, no batch-statistics terms (coef stays 0), no dgamma / dbeta
                def fin_frozen(st):
                    sums.zero_()
                    coef.zero_()
                    if act:
                        act['buf'].zero_()
                return None, fin_frozen, app
            return red, fin, app

        # ---- sums of an ENCODER block's BatchNorm backward formed by the producers of its two gradients (activated form, see
        # satcv_bn_bwd_finalize2): encoder conv output tensor id -> dict(buf, parts)
        act_sums = {}
        POOL_SUMS = getattr(rt.model, 'fuse_pool_bn_sums', True) and dt == ops.BF16
        ident = {}

        def ident_vecs(c):
            if c not in ident:
                ident[c] = (torch.ones(c, dtype=torch.float32, device=rt.dev), torch.zeros(c, dtype=torch.float32, device=rt.dev))
                self.keep.append(ident[c])
            return ident[c]

        def act_sums_for(t):
            """the registry entry of encoder output t (a conv -> BN -> ReLU node that is pooled AND used as a skip), or None"""
            if not POOL_SUMS or t.node.op != 'cba' or not t.node.attrs.get('bn', True) or not t.node.attrs.get('relu', True):
                return None
            if t.node.layer.bn_name in self.frozen or t.node.layer.name in shared_layers:
                return None
            if t.id not in act_sums:
                act_sums[t.id] = dict(buf=self._z(STAT_ROWS, 2, t.channels, dtype=torch.float64), parts=set())
            return act_sums[t.id]

        seen_layers = set()     # layers applied more than once (shared weights): later visits accumulate their gradients
        fused = {}      # tensor id -> sums buffer whose BN-backward reduce pass was done by the producer of the gradient
        HEAD_FAST = {(1, 16), (2, 16), (1, 32), (2, 32), (3, 32), (4, 32), (1, 64), (2, 64)}

        def fuse_target(t):
            """bnr_* fields (and the sums buffer) if the head's backward kernel can also do the reduce pass of the BN+ReLU
            that produced its input tensor `t` (the raw values are in its registers anyway); None otherwise."""
            if not rt.model.fuse_head_bn_bwd or t.id in gact or len(consumers[t.id]) != 1:
                return None
            P, pc = t.node, ctx.get(id(t.node))
            if P.op != 'cba' or pc['yoff'] != 0 or pc['ldy'] != pc['cout']:
                return None
            c, aff = pc['cout'], pc['aff']
            sums = self._z(STAT_ROWS, 2, c, dtype=torch.float64)
            return dict(mean=_fp(aff['mean']), rstd=_fp(aff['rstd']), sums=_fp(sums), sums_ld=c), sums

        def bst_target(t):
            """(bst fields, sums buffer, channels) if the data-gradient launch that writes dL/d t -- t the activated output of a
            conv -> BN -> ReLU node or of the decoder's BN over [skip, up] -- may also form the sums of that BatchNorm's backward
            (satcv.h: bst_*): the gradient has this one contributor, so the tile in the accumulators IS the whole gradient."""
            if not getattr(rt.model, 'fuse_dgrad_bn_bwd', True) or dt != ops.BF16 or t.id in gact or t.id in gpool or len(consumers[t.id]) != 1:
                return None
            P, pc = t.node, ctx.get(id(t.node))
            if P.op == 'cba' and P.attrs.get('bn', True) and P.layer.bn_name not in self.frozen:
                c, aff, aoff = pc['cout'], pc['aff'], pc['aoff']
                b = dict(y=pc['y'].data_ptr() + pc['yoff'] * es, ld=pc['ldy'], relu=1 if P.attrs.get('relu', True) else 0)
            elif P.op == 'concat_bn_relu' and P.layer.name not in self.frozen:
                ra, rb, aff, ca, cb = pc['ra'], pc['rb'], pc['aff'], pc['ca'], pc['cb']
                if (P.inputs[1].node.op == 'convT' and BIAS_NOISE) or ca % 8:
                    return None            # (the two-launch form of that BN backward keeps its own reduce passes)
                c, aoff = ca + cb, 0
                b = dict(y=ra.srcs[0][0].data_ptr(), ld=ca, y1=rb.srcs[0][0].data_ptr(), ld1=cb, split=ca, relu=1)
            else:
                return None
            b.update(scale=_fp(aff['scale'], aoff), shift=_fp(aff['shift'], aoff), mean=_fp(aff['mean'], aoff), rstd=_fp(aff['rstd'], aoff))
            return b, c

        def dgrad_step(t, **kw):
            """data-gradient launch writing the activation gradient of tensor t (and, where the kernel can, the sums of the
            BatchNorm backward of the node that produced t: one pass over two tensors less per such layer)"""
            # the gradient of a max-pooled encoder output: this launch sees the pooled activation p = maxpool(relu(BN(y_enc))) (its own
            # input), so its epilogue can form the pooled part of that BatchNorm's backward sums in the activated form -- sum dp [p > 0],
            # sum dp p (bst_* with unit scale / rstd and zero shift / mean on the pooled tensor)
            kw.setdefault('tile_policy', self.tile_policy)        # (the dry-run probes below must ask for the tile family the launch will get)
            if (POOL_SUMS and not kw.get('accumulate') and t.node.op == 'pool' and len(consumers[t.id]) == 1 and t.id not in gact
                    and len(vals[t.id].srcs) == 1 and vals[t.id].affine is None):
                ent = act_sums_for(t.node.inputs[0])
                if ent is not None and vals[t.id].srcs[0][1] == kw['cout'] == t.node.inputs[0].channels:
                    one, zero = ident_vecs(kw['cout'])
                    kwp = dict(kw, bst=dict(y=vals[t.id].srcs[0][0].data_ptr(), ld=kw['cout'], scale=_fp(one), shift=_fp(zero), mean=_fp(zero),
                                            rstd=_fp(one), relu=1), stats=_fp(ent['buf']), stats_ld=kw['cout'])
                    probe = ops.make_conv_desc(**kwp)
                    if lib.satcv_conv2d_igemm_pipelined(C.byref(probe)) == 1:
                        ent['parts'].add('pool')
                        fn = self._conv_step(role='dgrad', **kwp)
                        fn.label += ' +poolsums'
                        return fn
            bt = None if kw.get('accumulate') else bst_target(t)
            # where it pays (measured per layer of the five-level U-Net, batch 64): the extra tile read hides under a
            # K loop of >= 1152 (3x3 x 128 channels) and under the 1x1 kernels.  The thin and middle layers are HBM-bound
            # themselves -- there the read costs the conv what the separate pass had cost -- and the thin ones run on the
            # persistent weights-stationary kernel, which does not carry the sums.
            kdepth = kw.get('kh', 1) * kw.get('kw', 1) * (kw.get('c0', 0) + (kw.get('c1', 0) or 0))
            # (128 -> 256 at 64 x 64, the one K = 1152 layer whose output is wider than its input, measured +49 us against -41 us)
            pays = FUSE_DGRAD_ALL or kdepth >= 2304 or (kdepth >= 1152 and kw['cout'] <= kw.get('c0', 0)) or kw.get('kh', 1) == 1
            if bt is not None and bt[1] == kw['cout'] and pays:
                sums = self._z(STAT_ROWS, 2, bt[1], dtype=torch.float64)
                kw2 = dict(kw, bst=bt[0], stats=_fp(sums), stats_ld=bt[1])
                probe = ops.make_conv_desc(**kw2)
                if lib.satcv_conv2d_igemm_pipelined(C.byref(probe)) == 1:
                    fused[t.id] = sums
                    fn = self._conv_step(role='dgrad', **kw2)
                    fn.label += ' +bnred'
                    return fn
            return self._conv_step(role='dgrad', **kw)

        def wgrad_step(r, dy, lddy, lay, cin_real, cout, hh, ww, k, dil, f=0, accum=0, last=False):
            nonlocal ws_need
            sa = self._src_args(r)
            if sa['x1'] is not None and sa['c0'] % 8:
                raise NotImplementedError(f'{lay.name}: training a convolution over a concatenation needs a first part of a multiple of 8 channels '
                                          f'(got {sa["c0"]}); inference has no such limit')
            d = ops.make_wgrad_desc(dy=dy, lddy=lddy, dw=rt.gptr(lay.name + '/kernel'), cin=cin_real, cout=cout, n=n, h=hh, w_=ww,
                                    dtype=dt, kh=k, kw=k, dil=dil, mode_dy=1 if f else 0, f=f if f else 1, transposed=1 if f else 0, accumulate=accum,
                                    whole_chip=1 if last else 0, **sa)
            nb = lib.satcv_conv2d_wgrad_workspace(C.byref(d))
            if nb < 0:
                raise RuntimeError(lib.satcv_last_error().decode())
            nvalid = (f * f if f else 1) * cout
            grp = None
            if DEFER and not accum and lay.name not in shared_layers and nvalid % 4 == 0 and not (k == 3 and dil > 1):
                d.defer_reduce = 1
                d._own_ws = nb
                rpending.append((d, 'w', 4 * k * k * cin_real * nvalid))
            elif RED3 and not accum and lay.name not in shared_layers and nvalid % 4 == 0 and not (k == 3 and dil > 1):
                d.defer_reduce = 1
                d._own_ws = nb
                grp = {'items': [(d, 'w')], 'tab': None}       # (a one-job table, built with the others once the workspaces are assigned)
                rgroups.append(grp)
            else:
                ws_need = max(ws_need, nb)
            wdescs.append(d)
            self.keep.append(d)
            label = f"wgrad k{k} d{dil} n{n} {hh}x{ww} {sa['c0']}+{sa['c1']}->{cout}{' convT f%d' % f if f else ''}"
            work = dict(kind='wgrad', taps=(f * f if f else k * k), px=n * hh * ww, cin=cin_real, cout=cout, esize=es)
            if self.side is None:
                fn = lambda st: check(lib.satcv_conv2d_wgrad(C.byref(d), st))
                fn.label, fn.work = label, work
                return fn
            # weight gradient and data gradient of a layer only share their INPUT (dy): run the weight gradients on a
            # second HIP stream so that their load/MFMA/store phases interleave with the main stream's kernels
            ev = torch.cuda.Event()
            side, sptr = self.side, C.c_void_p(self.side.cuda_stream)

            evw, evr = (torch.cuda.Event(), torch.cuda.Event()) if grp is not None else (None, None)

            def run(st, d=d, ev=ev, grp=grp, evw=evw, evr=evr):
                ev.record(torch.cuda.current_stream())
                side.wait_event(ev)
                check(lib.satcv_conv2d_wgrad(C.byref(d), sptr))
                if grp is not None:
                    t = grp['tab']
                    evw.record(side)
                    self.rstream.wait_event(evw)
                    check(lib.satcv_reduce_slabs_batched(t['jobs'].data_ptr(), t['prefix'].data_ptr(), t['n'], t['total'], C.c_void_p(self.rstream.cuda_stream)))
                    evr.record(self.rstream)
                    self._last_red_ev = evr
            run.label, run.work = label, work
            return run

        # data-parallel overlap: after every parameter-bearing node tell the gradient exchange which tail of the flat
        # gradient buffer is final (parallel.GradSync.ready_above)
        pending = {}                                   # layer name -> remaining visits
        for node in m.nodes:
            if node.layer is not None and any(ps.name in rt.offsets for ps in node.layer.specs):
                pending[node.layer.name] = pending.get(node.layer.name, 0) + 1
        layer_hi = {l.name: max(rt.offsets[ps.name] + ps.size for ps in l.specs if ps.name in rt.offsets)
                    for l in m.layers if l.name in pending}
        # a conv_batch_act node also writes the gamma / beta gradients of ITS BatchNormalization (a layer of its own, directly
        # above the convolution in the flat buffer): until the node has run, nothing below the end of that layer is final
        by_name = {l.name: l for l in m.layers}
        for node in m.nodes:
            bn = getattr(node.layer, 'bn_name', None) if node.op == 'cba' else None
            if bn in by_name and node.layer.name in layer_hi:
                his = [rt.offsets[ps.name] + ps.size for ps in by_name[bn].specs if ps.name in rt.offsets]
                if his:
                    layer_hi[node.layer.name] = max(layer_hi[node.layer.name], max(his))

        def flush_reduces(force=False):
            """schedule the batched slab sum of the launches recorded so far (on the weight-gradient stream, behind an event of the main
            stream: the fused thin-layer launches write their slabs there).  Returns False while a group is still being collected."""
            if not rpending:
                return True
            if not force and sum(b for _, _, b in rpending) < (16 << 20):
                return False
            grp = {'items': [(d, kind) for d, kind, _ in rpending], 'tab': None}
            rpending.clear()
            rgroups.append(grp)
            ev = torch.cuda.Event() if self.side is not None else None

            def flush(st, grp=grp, ev=ev):
                t = grp['tab']
                if self.side is not None:
                    ev.record(torch.cuda.current_stream())
                    self.side.wait_event(ev)
                    st = C.c_void_p(self.side.cuda_stream)
                check(lib.satcv_reduce_slabs_batched(t['jobs'].data_ptr(), t['prefix'].data_ptr(), t['n'], t['total'], st))
            flush.label = f"wgrad_reduce_batched {len(grp['items'])} layers"
            self.bwd.append(flush)
            return True

        def grads_ready(node):
            lay = node.layer
            if lay is None or lay.name not in pending:
                return
            pending[lay.name] -= 1
            if pending[lay.name] == 0:
                del pending[lay.name]
            if not flush_reduces(force=not pending):
                return                                  # (the gradients above `lo` are not final before their slabs are summed)
            lo = max((layer_hi[k] for k in pending), default=0)
            # ---- round 6: the optimizer step of the parameters in [lo, n) beside the REST of the backward pass.  When what is still pending is a
            # few per cent of the parameters (the three thin encoder blocks of get_unet_model hold 0.5 %), everything above `lo` is final: Adam
            # and the operand repack of that range run on the weight-gradient stream -- idle from here on: the thin layers' weight gradients
            # are part of the fused launches of the main stream -- while the main stream finishes the encoder's backward pass; the step's tail
            # then only updates [0, lo).  Single replica only (a gradient exchange must see every gradient first): Model.train_step_device
            # sets plan.early_opt per step.
            if EARLY_OPT and self.side is not None and self.eo_lo is None and pending and 0 < lo <= EARLY_OPT_FRAC * rt.n_train and lo % 4 == 0:
                self.eo_lo = lo
                evo = torch.cuda.Event()

                def early(st, lo=lo, evo=evo):
                    eo = self.early_opt
                    if eo is None:
                        return
                    evo.record(torch.cuda.current_stream())
                    self.side.wait_event(evo)
                    if self._last_red_ev is not None:
                        self.side.wait_event(self._last_red_ev)
                    sp = C.c_void_p(self.side.cuda_stream)
                    n = rt.n_train - lo
                    check(lib.satcv_adam_step_part(_fp(rt.pflat, lo), _fp(rt.gflat, lo), _fp(rt.adam_m, lo), _fp(rt.adam_v, lo), n, eo['beta_1'], eo['beta_2'],
                                                   eo['epsilon'], rt.adam_state.data_ptr(), _fp(rt.lr_mul, lo) if rt.lr_mul is not None else None, 0, sp))
                    rt.repack(lo, None, stream=sp)
                    self.eo_done = True
                early.label = f'early optimizer step [{lo}, {rt.n_train}) on the side stream'
                self.bwd.append(early)

            def ckpt(st, lo=lo):
                sync = getattr(m, '_sync_grads', None)
                if sync is not None and hasattr(sync, 'ready_above'):
                    if self._last_red_ev is not None:            # (gradients are final behind their slab sums on the third stream)
                        self.side.wait_event(self._last_red_ev)
                    sync.ready_above(rt.gflat, lo, self.side)
            self.bwd.append(ckpt)

        head_feed = {}                                 # tensor id -> the head launch that writes its gradient
        prev_node = None
        for node in reversed(m.nodes):
            if prev_node is not None:
                grads_ready(prev_node)                 # (the branches below `continue`: bookkeeping of the node just finished)
            prev_node = node
            op = node.op
            cx = ctx.get(id(node))
            if op == 'head':
                r = cx['r']
                (y, c) = r.srcs[0]
                dx = self._z(n, r.h, r.w, c)
                lay = node.layer
                hb = None
                ft = fuse_target(node.inputs[0]) if (cx['ncls'], c) in HEAD_FAST and r.affine else None
                if ft is not None:
                    hb = dict(mean=ft[0]['mean'], rstd=ft[0]['rstd'], sums=ft[0]['sums'], sums_ld=ft[0]['sums_ld'])
                    fused[node.inputs[0].id] = ft[1]
                hd = ops.make_head_desc(x=y.data_ptr(), ldx=c, cin=c, w=rt.pptr(lay.name + '/kernel'), b=rt.pptr(lay.name + '/bias'),
                                        ncls=cx['ncls'], activation=cx['act'], npix=n * r.h * r.w, dtype=dt,
                                        in_scale=_fp(r.affine['scale']) if r.affine else None,
                                        in_shift=_fp(r.affine['shift']) if r.affine else None,
                                        dlogits=self.dlogits.data_ptr(), dx=dx.data_ptr(), lddx=c,
                                        dw=rt.gptr(lay.name + '/kernel'), db=rt.gptr(lay.name + '/bias'), bnr=hb)
                self.keep.append(hd)
                self.bwd.append(lambda st, hd=hd: check(lib.satcv_head_bwd(C.byref(hd), st)))
                nb = lib.satcv_head_bwd_workspace(C.byref(hd))
                if nb > 0:            # per-workgroup partial rows + an ordered sum instead of float atomics: reproducible dW / db
                    part = self._z(nb // 4, dtype=torch.float32)
                    hd.partials = part.data_ptr()
                    self.bwd.append(lambda st, hd=hd: check(lib.satcv_head_bwd_finalize(C.byref(hd), st)))
                gact[node.inputs[0].id] = (dx, 0, c)
                # (the block under the head may form this gradient itself from the logit gradients: see the fused launch below)
                head_feed[node.inputs[0].id] = dict(hd=hd, ncls=cx['ncls'], c=c, w=rt.pptr(lay.name + '/kernel'), wname=lay.name + '/kernel',
                                                    fused_reduce=ft is not None)
            elif op == 'add_relu':
                # out = ReLU(BN(conv) + shortcut): the masked gradient g * (out > 0) belongs to BOTH addends.  It is formed in place
                # and handed to the branch's (linear) BN backward and to the shortcut; a tensor that already holds a gradient
                # (several consumers) receives it by addition.
                ty, tsc = node.inputs
                g = gact.get(node.outputs[0].id)
                if g is None or cx is None:
                    continue
                c, hh, ww, out = cx['c'], cx['h'], cx['w'], cx['out']
                if g[1] != 0 or g[2] != c:
                    raise NotImplementedError('residual-join gradient in a channel slice')
                gb, cnt = g[0], n * hh * ww * c
                self.bwd.append(lambda st, out=out, gb=gb, cnt=cnt: check(lib.satcv_relu_bwd(out.data_ptr(), gb.data_ptr(), cnt, dt, st)))
                for t in (ty, tsc):
                    prev = gact.get(t.id)
                    if prev is None:
                        gact[t.id] = (gb, 0, c)
                    else:
                        if prev[1] != 0 or prev[2] != c:
                            raise NotImplementedError('gradient fan-in into a channel slice')
                        self.bwd.append(lambda st, pb=prev[0], gb=gb, npx=n * hh * ww, c=c: check(lib.satcv_add_act(
                            pb.data_ptr(), None, None, gb.data_ptr(), None, None, 0, pb.data_ptr(), npx, c, dt, st)))
            elif op == 'cba' and not node.attrs.get('bn', True):
                # plain Conv2D: the gradient of its stored output is the incoming one as it is
                tin, tout = node.inputs[0], node.outputs[0]
                da = gact.get(tout.id)
                if da is None:
                    continue
                lay, r, cout = node.layer, cx['r'], cx['cout']
                if da[1] != 0 or da[2] != cout:
                    raise NotImplementedError('plain-convolution gradient in a channel slice')
                hh, ww, dyb = r.h, r.w, da[0]
                accum = 1 if lay.name in seen_layers else 0
                seen_layers.add(lay.name)
                bpart = self._z(max(lib.satcv_bias_grad_workspace(n * hh * ww, cout) // 4, 1), dtype=torch.float32)
                self.bwd.append(lambda st, dyb=dyb, cout=cout, npx=n * hh * ww, db=rt.gptr(lay.name + '/bias'), bpart=bpart: check(lib.satcv_bias_grad(
                    dyb.data_ptr(), cout, npx, cout, dt, db, bpart.data_ptr(), st)))
                pk = rt.packed[lay.name]
                self.bwd.append(wgrad_step(r, dyb.data_ptr(), cout, lay, pk['cin'], cout, hh, ww, cx['k'], cx['dil'], accum=accum))
                if tin.node.op != 'input':
                    cinp = r.c
                    prev = gact.get(tin.id)
                    if prev is not None and (prev[1] != 0 or prev[2] != cinp):
                        raise NotImplementedError('gradient fan-in into a channel slice')
                    gin = prev[0] if prev is not None else self._z(n, hh, ww, cinp)
                    self.bwd.append(dgrad_step(tin, x0=dyb.data_ptr(), c0=cout, w=pk['dgrad'].data_ptr(), y=gin.data_ptr(), ldy=cinp,
                                               n=n, h=hh, w_=ww, cout=cinp, cout_pad=rup(cinp, 32), kh=cx['k'], kw=cx['k'],
                                               dil=cx['dil'], dtype=dt, accumulate=1 if prev is not None else 0))
                    gact[tin.id] = (gin, 0, cinp)
            elif op == 'cba':
                tin, tout = node.inputs[0], node.outputs[0]
                da, dp = gact.get(tout.id), gpool.get(tout.id)
                graws = [gact[cn.outputs[0].id] for cn in consumers.get(tout.id, []) if cn.op == 'raw' and cn.outputs[0].id in gact]
                if da is None and dp is None:
                    if graws:
                        raise NotImplementedError('a convolution consumed only through its un-normalised output')
                    continue
                lay, r, y, aff, cout = node.layer, cx['r'], cx['y'], cx['aff'], cx['cout']
                yoff, ldy, aoff = cx['yoff'], cx['ldy'], cx['aoff']
                hh, ww = r.h, r.w
                pre = fused.get(tout.id)
                sums = pre if pre is not None else self._z(STAT_ROWS, 2, cout, dtype=torch.float64)
                accum = 1 if lay.name in seen_layers else 0
                seen_layers.add(lay.name)
                da_ptr = da[0].data_ptr() + da[1] * es if da is not None else None
                # ---- thin full- / half-resolution layers: ONE launch forms dy = BN-backward(g, y) in registers and uses the tile for
                # the data gradient AND the weight gradient (csrc/conv_bwd_fused.hip): 4 tensor passes instead of 7, no dy tensor
                fz = None
                if (getattr(rt.model, 'fuse_thin_bwd', True) and dt == ops.BF16 and da is not None and dp is None and not graws and not BIAS_NOISE
                        and tin.node.op != 'input' and gact.get(tin.id) is None and cx['k'] == 3 and cx['dil'] == 1 and da[2] == ldy
                        and lay.name not in shared_layers):      # (a shared layer's other visit may have its weight gradient in flight on the side stream)
                    pk = rt.packed[lay.name]
                    cinp = r.c
                    sa = self._src_args(r)
                    coef = self._z(2, cout, dtype=torch.float32)
                    gin = self._z(n, hh, ww, cinp)
                    fz = ops.make_bwdf_desc(g=da_ptr, yraw=y.data_ptr() + yoff * es, ldg=ldy, bn_scale=_fp(aff['scale'], aoff), bn_shift=_fp(aff['shift'], aoff),
                                            bn_mean=_fp(aff['mean'], aoff), bn_rstd=_fp(aff['rstd'], aoff), bn_coef=_fp(coef),
                                            linear=0 if node.attrs.get('relu', True) else 1, w_dgrad=pk['dgrad'].data_ptr(), dx=gin.data_ptr(), lddx=cinp,
                                            dw=rt.gptr(lay.name + '/kernel'), cin=pk['cin'], cout=cout, n=n, h=hh, w_=ww, dtype=dt, accumulate=accum, **sa)
                    # ... and, where the layer below is a BatchNorm + ReLU fed only by this gradient, the sums of ITS backward
                    # (what satcv_bn_bwd_reduce would compute in a pass over dx and that layer's raw output): bst_*
                    bt = bst_target(tin) if (sa['in_relu'] and sa['in_scale'] is not None) else None
                    sums_below = None
                    if bt is not None and bt[1] == cinp and bt[0].get('relu', 0) == 1:
                        sums_below = self._z(STAT_ROWS, 2, cinp, dtype=torch.float64)
                        fz.bst_sums, fz.bst_sums_ld, fz.bst_mean, fz.bst_rstd = _fp(sums_below), cinp, bt[0]['mean'], bt[0]['rstd']
                    # ... and, for the block under the 1 x 1 head, the gradient g itself: two logit gradients per pixel instead of the
                    # head's dx tensor written and read back (2 x 64 bytes per pixel)
                    hf = head_feed.get(tout.id)
                    if (hf is not None and getattr(rt.model, 'fuse_head_grad', True) and hf['ncls'] == 2 and hf['c'] == cout == 32 and cinp == 32
                            and hf['fused_reduce'] and pre is not None and len(consumers.get(tout.id, [])) == 1):
                        fz.hg_dlogits, fz.hg_w, fz.hg_ncls = self.dlogits.data_ptr(), hf['w'], 2
                        if lib.satcv_conv2d_bwd_fused_workspace(C.byref(fz)) < 0:
                            fz.hg_dlogits, fz.hg_w, fz.hg_ncls = None, None, 0
                        else:
                            hf['hd'].dx = None                      # satcv_head_bwd keeps its dW / db and the fused sums, stores no dx
                    nbf = lib.satcv_conv2d_bwd_fused_workspace(C.byref(fz))
                    if nbf < 0 and sums_below is not None:              # (the 64 -> 64 form does not carry the sums)
                        fz.bst_sums, fz.bst_sums_ld, fz.bst_mean, fz.bst_rstd = None, 0, None, None
                        sums_below = None
                        nbf = lib.satcv_conv2d_bwd_fused_workspace(C.byref(fz))
                    if nbf < 0:
                        fz = None
                    elif sums_below is not None:
                        fused[tin.id] = sums_below
                if fz is not None:
                    if DEFER and not accum and lay.name not in shared_layers:
                        fz.defer_reduce = 1
                        fz._own_ws = nbf
                        rpending.append((fz, 'f', 4 * 9 * pk['cin'] * cout))
                    else:
                        fused_ws_need = max(fused_ws_need, nbf)
                    fdescs.append(fz)
                    self.keep.append(fz)
                    cnt = float(n * hh * ww)
                    frozen_bn = lay.bn_name in self.frozen
                    dg_, db_ = rt.gptr(lay.bn_name + '/gamma'), rt.gptr(lay.bn_name + '/beta')
                    bd = ops.make_bnbwd_desc(yraw=y.data_ptr() + yoff * es, ldy=ldy, scale=_fp(aff['scale'], aoff), shift=_fp(aff['shift'], aoff),
                                             mean=_fp(aff['mean'], aoff), rstd=_fp(aff['rstd'], aoff), n=n, h=hh, w_=ww, c=cout, dtype=dt,
                                             da=da_ptr, ldda=da[2], sums=_fp(sums), sums_ld=cout, coef=_fp(coef), linear=fz.linear)
                    self.keep.append(bd)
                    if frozen_bn:
                        def fin(st, sums=sums, coef=coef):
                            sums.zero_(); coef.zero_()
                    else:
                        if pre is None:
                            red = lambda st, bd=bd: check(lib.satcv_bn_bwd_reduce(C.byref(bd), st))
                            red.label = f"bn_bwd_reduce n{n} {hh}x{ww} c{cout} f1"
                            red.work = dict(kind='bn_bwd_reduce', px=n * hh * ww, c=cout, esize=es)
                            self.bwd.append(red)
                        fin0 = lambda st, sums=sums, coef=coef, cnt=cnt, dg_=dg_, db_=db_, accum=accum, c_=cout: check(lib.satcv_bn_bwd_finalize(
                            _fp(sums), c_, c_, cnt, dg_, db_, _fp(coef), accum, st))
                        if self.sync_bn:
                            def fin(st, fin0=fin0, sums=sums):
                                parallel.allreduce_mean_(sums)
                                fin0(st)
                        else:
                            fin = fin0
                    self.bwd.append(fin)
                    fstep = lambda st, fz=fz: check(lib.satcv_conv2d_bwd_fused(C.byref(fz), st))
                    fstep.label = f"bwd_fused k3 n{n} {hh}x{ww} {sa['c0']}+{sa['c1']}->{cout}{' +bnred' if fz.bst_sums else ''}{' +headgrad' if fz.hg_dlogits else ''}"
                    fstep.work = dict(kind='bwd_fused', taps=9, px=n * hh * ww, cin=pk['cin'], cout=cout, esize=es)
                    self.bwd.append(fstep)
                    gact[tin.id] = (gin, 0, cinp)
                    self.dbg['dx:' + lay.name] = gin
                    self.dbg['dyparts:' + lay.name] = dict(g=da, y=y, yoff=yoff, ldy=ldy, aff=aff, aoff=aoff, cout=cout, coef=coef, linear=fz.linear)
                    if fz.hg_dlogits:
                        self.dbg['dyparts:' + lay.name]['head'] = (self.dlogits, head_feed[tout.id]['wname'])
                    self.dbg['_ctx:' + lay.name] = dict(da=da, dp=dp, y=y, yoff=yoff, ldy=ldy, aff=aff, aoff=aoff, cout=cout)
                    continue
                dy = self._z(n, hh, ww, cout)
                red, fin, app = bn_bwd_steps(da_ptr, da[2] if da is not None else 0, dp[0].data_ptr() if dp is not None else None,
                                             cout, gpool_f.get(tout.id, 1), y.data_ptr() + yoff * es, ldy, aff, aoff, sums, 0, cout, cout, hh, ww,
                                             dy.data_ptr(), cout, rt.gptr(lay.name + '/bias') if BIAS_NOISE else None,
                                             rt.gptr(lay.bn_name + '/gamma'), rt.gptr(lay.bn_name + '/beta'), accum,
                                             linear=0 if node.attrs.get('relu', True) else 1, frozen=lay.bn_name in self.frozen,
                                             act=act_sums.get(tout.id) if (act_sums.get(tout.id) or {}).get('parts') else None)
                # ---- encoder blocks of the full- / half-resolution levels: the pooled BatchNorm apply + weight gradient (+ data gradient)
                # in ONE launch (satcv_conv2d_bwd_fused with dpool / amax): no dy tensor, no separate weight-gradient launch
                fzp = None
                amax = self.amax_of.get(tout.id)
                if (amax is not None and getattr(rt.model, 'fuse_pool_bwd', True) and dt == ops.BF16 and da is not None and dp is not None
                        and gpool_f.get(tout.id, 1) == 2 and not graws and not BIAS_NOISE and cx['k'] == 3 and cx['dil'] == 1 and da[2] == ldy
                        and dp[1] == 0 and lay.bn_name not in self.frozen and gact.get(tin.id) is None and lay.name not in shared_layers):
                    pkp = rt.packed[lay.name]
                    sa = self._src_args(r)
                    want_dx = tin.node.op != 'input'
                    cinp = r.c
                    ginp = self._z(n, hh, ww, cinp) if want_dx else None
                    fzp = ops.make_bwdf_desc(g=da_ptr, yraw=y.data_ptr() + yoff * es, ldg=ldy, bn_scale=_fp(aff['scale'], aoff), bn_shift=_fp(aff['shift'], aoff),
                                             bn_mean=_fp(aff['mean'], aoff), bn_rstd=_fp(aff['rstd'], aoff), bn_coef=_fp(app.coef), linear=0,
                                             w_dgrad=pkp['dgrad'].data_ptr() if want_dx else None, dx=ginp.data_ptr() if want_dx else None,
                                             lddx=cinp if want_dx else 0, dw=rt.gptr(lay.name + '/kernel'), cin=pkp['cin'], cout=cout, n=n, h=hh, w_=ww,
                                             dtype=dt, accumulate=accum, dpool=dp[0].data_ptr(), lddp=dp[2], amax=amax.data_ptr(), **sa)
                    # its input is the max-pooled output of the block below: the pooled part of THAT block's sums, activated form
                    ent = None
                    if (want_dx and POOL_SUMS and tin.node.op == 'pool' and len(consumers[tin.id]) == 1 and sa['in_scale'] is None
                            and len(vals[tin.id].srcs) == 1 and tin.node.inputs[0].channels == cinp):
                        ent = act_sums_for(tin.node.inputs[0])
                    if ent is not None:
                        fzp.bst_sums, fzp.bst_sums_ld, fzp.bst_act_form = _fp(ent['buf']), cinp, 1
                    nbp = lib.satcv_conv2d_bwd_fused_workspace(C.byref(fzp))
                    if nbp < 0:
                        fzp = None
                    elif ent is not None:
                        ent['parts'].add('pool')
                if fzp is not None:
                    if DEFER and not accum and lay.name not in shared_layers:
                        fzp.defer_reduce = 1
                        fzp._own_ws = nbp
                        rpending.append((fzp, 'f', 4 * 9 * pkp['cin'] * cout))
                    else:
                        fused_ws_need = max(fused_ws_need, nbp)
                    fdescs.append(fzp)
                    self.keep.append(fzp)
                    self.bwd += [fin] if (pre is not None or red is None) else [red, fin]
                    fstep = lambda st, fzp=fzp: check(lib.satcv_conv2d_bwd_fused(C.byref(fzp), st))
                    fstep.label = f"bwd_fused k3 n{n} {hh}x{ww} {sa['c0']}+{sa['c1']}->{cout} pooled{'' if want_dx else ' nodx'}{' +poolsums' if fzp.bst_sums else ''}"
                    fstep.work = dict(kind='bwd_fused', taps=9, px=n * hh * ww, cin=pkp['cin'], cout=cout, esize=es, nodx=not want_dx, pooled=True)
                    self.bwd.append(fstep)
                    if want_dx:
                        gact[tin.id] = (ginp, 0, cinp)
                        self.dbg['dx:' + lay.name] = ginp
                    self.dbg['dyparts:' + lay.name] = dict(g=da, y=y, yoff=yoff, ldy=ldy, aff=aff, aoff=aoff, cout=cout, coef=app.coef, linear=0,
                                                           dp=dp, amax=amax)
                    self.dbg['_ctx:' + lay.name] = dict(da=da, dp=dp, y=y, yoff=yoff, ldy=ldy, aff=aff, aoff=aoff, cout=cout)
                    continue
                self.bwd += [fin, app] if (pre is not None or red is None) else [red, fin, app]
                for gr in graws:            # consumers of the un-normalised output add their gradient to dy (and to the bias gradient)
                    if gr[1] != 0 or gr[2] != cout:
                        raise NotImplementedError('raw-output gradient in a channel slice')
                    bpart = self._z(max(lib.satcv_bias_grad_workspace(n * hh * ww, cout) // 4, 1), dtype=torch.float32)
                    self.bwd.append(lambda st, dy=dy, gb=gr[0], npx=n * hh * ww, cout=cout, db=rt.gptr(lay.name + '/bias'), bpart=bpart: (
                        check(lib.satcv_bias_grad(gb.data_ptr(), cout, npx, cout, dt, db, bpart.data_ptr(), st)),
                        check(lib.satcv_add_act(dy.data_ptr(), None, None, gb.data_ptr(), None, None, 0, dy.data_ptr(), npx, cout, dt, st))))
                self.dbg['dy:' + lay.name] = dy
                self.dbg['_ctx:' + lay.name] = dict(da=da, dp=dp, y=y, yoff=yoff, ldy=ldy, aff=aff, aoff=aoff, cout=cout)
                pk = rt.packed[lay.name]
                # (the weight gradient of a layer fed by a model input is the last launch of the backward pass: nothing runs beside it)
                wstep = wgrad_step(r, dy.data_ptr(), cout, lay, pk['cin'], cout, hh, ww, cx['k'], cx['dil'], accum=accum,
                                   last=tin.node.op == 'input' and len(m.inputs) == 1 and os.environ.get('SATCV_WGRAD_LAST_FULL', '1') != '0')
                if not WGRAD_LATE:
                    self.bwd.append(wstep)
                if tin.node.op != 'input':
                    cinp = r.c
                    prev = gact.get(tin.id)
                    if prev is not None and (prev[1] != 0 or prev[2] != cinp):
                        raise NotImplementedError('gradient fan-in into a channel slice')
                    gin = prev[0] if prev is not None else self._z(n, hh, ww, cinp)
                    self.bwd.append(dgrad_step(tin, x0=dy.data_ptr(), c0=cout, w=pk['dgrad'].data_ptr(), y=gin.data_ptr(), ldy=cinp,
                                               n=n, h=hh, w_=ww, cout=cinp, cout_pad=rup(cinp, 32), kh=cx['k'], kw=cx['k'],
                                               dil=cx['dil'], dtype=dt, accumulate=1 if prev is not None else 0))
                    gact[tin.id] = (gin, 0, cinp)
                    self.dbg['dx:' + lay.name] = gin
                if WGRAD_LATE:
                    self.bwd.append(wstep)
            elif op == 'dropout':
                tin, tout = node.inputs[0], node.outputs[0]
                g = gact.get(tout.id)
                if g is None or cx is None:
                    continue
                if g[1] != 0 or g[2] != cx['ctot']:
                    raise NotImplementedError('dropout gradient in a channel slice')
                gt, mask, ctot, hw = g[0], cx['mask'], cx['ctot'], cx['hw']
                mode = 0 if cx['spatial'] else 1
                self.bwd.append(lambda st, gt=gt, mask=mask, ctot=ctot, hw=hw, mode=mode: check(lib.satcv_dropout_apply(
                    gt.data_ptr(), ctot, None, None, 0, _fp(mask), ctot, mode, gt.data_ptr(), ctot, n, hw, ctot, dt, st)))
                gact[tin.id] = (gt, 0, ctot)
            elif op == 'concat':
                g = gact.get(node.outputs[0].id)
                if g is None:
                    continue
                off = 0
                for t in node.inputs:          # each branch sees its channel slice of the gradient
                    gact[t.id] = (g[0], g[1] + off, g[2])
                    off += t.channels
            elif op == 'pool':
                tin, tout = node.inputs[0], node.outputs[0]
                if tout.id in gact:
                    if gact[tout.id][1] != 0:
                        raise NotImplementedError('pooled gradient in a channel slice')
                    gpool[tin.id] = gact[tout.id]
                    gpool_f[tin.id] = cx['f']
            elif op == 'concat_bn_relu':
                ta, tb = node.inputs
                tout = node.outputs[0]
                g = gact.get(tout.id)
                if g is None:
                    continue
                ra, rb, aff, ca, cb = cx['ra'], cx['rb'], cx['aff'], cx['ca'], cx['cb']
                ctot = ca + cb
                hh, ww = ra.h, ra.w
                pre = fused.get(tout.id)
                sums = pre if pre is not None else self._z(STAT_ROWS, 2, ctot, dtype=torch.float64)
                dskip = self._z(n, hh, ww, ca)
                du = self._z(n, hh, ww, cb)
                bn = node.layer.name
                upl = tb.node.layer
                if g[2] != ctot:
                    raise NotImplementedError('decoder concat gradient in a channel slice')
                gptr_ = g[0].data_ptr() + g[1] * es
                if not (tb.node.op == 'convT' and BIAS_NOISE) and ca % 8 == 0:
                    # ONE reduce / finalize / apply over the whole concatenation: the gradient of the concatenated tensor is read
                    # once per pass in full lines (the two half launches each touched every line of it for half of its bytes)
                    # the skip is the activated output of an encoder block that is also max-pooled: this apply pass has dskip and that
                    # activation in registers -- it also forms the skip part of the encoder BatchNorm's backward sums (sk_sums)
                    sec = dict(yraw1=rb.srcs[0][0].data_ptr(), ldy1=cb, dy1=du.data_ptr(), lddy1=cb, c_split=ca)
                    # round 5: the `up` channels' part of this apply pass, the transposed convolution's data gradient and its weight gradient as ONE
                    # launch (satcv_convt_bwd_fused, built at the convT node below): the apply pass then covers the skip channels only
                    ctbf = None
                    cxt = ctx.get(id(tb.node)) if tb.node.op == 'convT' else None
                    if (CTBF and dt == ops.BF16 and cxt is not None and bn not in self.frozen and upl.name not in shared_layers and cxt['f'] == 2
                            and cb in CTBF_COUTS and len(consumers.get(tb.id, [])) == 1):
                        rT = cxt['r']
                        saT = self._src_args(rT)
                        probe = ops.make_ctbf_desc(g=gptr_ + ca * es, ldg=ctot, yup=rb.srcs[0][0].data_ptr(), ldy=cb, bn_scale=None, bn_shift=None, bn_mean=None,
                                                   bn_rstd=None, bn_c1=None, bn_c2=None, x=saT['x0'], ldx=saT['c0'], w_dgrad=rt.packed[upl.name]['dgrad'].data_ptr(),
                                                   w_npad=rup(rT.c, 32), dx=saT['x0'], lddx=rT.c, dw=rt.gptr(upl.name + '/kernel'), cin=rT.c, cout=cb, n=n,
                                                   h=rT.h, w_=rT.w, dtype=dt)
                        if saT['x1'] is None and saT['c0'] == rT.c == rt.packed[upl.name]['cin'] and lib.satcv_convt_bwd_fused_workspace(C.byref(probe)) > 0:
                            ctbf = dict(g=gptr_ + ca * es, ldg=ctot, yup=rb.srcs[0][0].data_ptr(), ldy=cb, aff=aff, aoff=ca, ctot=ctot)
                            sec['dy1'] = None
                    ent = None
                    if (bn not in self.frozen and ta.id not in gact and any(cn.op == 'pool' for cn in consumers[ta.id])
                            and len([cn for cn in consumers[ta.id] if cn.op != 'pool']) == 1 and ta.channels == ca):
                        ent = act_sums_for(ta)
                    if ent is not None:
                        sec.update(sk_sums=_fp(ent['buf']), sk_sums_ld=ca)
                        ent['parts'].add('skip')
                    r_, f_, a_ = bn_bwd_steps(gptr_, ctot, None, 0, 1, ra.srcs[0][0].data_ptr(), ca, aff, 0, sums, 0, ctot, ctot, hh, ww,
                                              dskip.data_ptr(), ca, None, rt.gptr(bn + '/gamma'), rt.gptr(bn + '/beta'), frozen=bn in self.frozen,
                                              second=sec)
                    if ent is not None:
                        a_.label += ' +skipsums'
                    if ctbf is not None:
                        a_.label += ' (skip half)'
                        a_.work = dict(kind='bn_bwd_apply', px=n * hh * ww, c=ca, esize=es)
                        ctbf['coef'] = a_.coef
                    self.bwd += [f_, a_] if (pre is not None or r_ is None) else [r_, f_, a_]
                    gact[ta.id] = (dskip, 0, ca)
                    self.dbg['dskip:' + bn] = dskip
                    self.dbg['du:' + bn] = du
                    graw[tb.id] = ctbf if ctbf is not None else du
                    continue
                ra_, fa_, aa_ = bn_bwd_steps(gptr_, ctot, None, 0, 1, ra.srcs[0][0].data_ptr(), ca, aff, 0, sums, 0, ctot, ca, hh, ww,
                                             dskip.data_ptr(), ca, None, rt.gptr(bn + '/gamma'), rt.gptr(bn + '/beta'), frozen=bn in self.frozen)
                rb_, fb_, ab_ = bn_bwd_steps(gptr_ + ca * es, ctot, None, 0, 1, rb.srcs[0][0].data_ptr(), cb, aff, ca, sums, ca, ctot, cb,
                                             hh, ww, du.data_ptr(), cb, rt.gptr(upl.name + '/bias') if (tb.node.op == 'convT' and BIAS_NOISE) else None,
                                             rt.gptr(bn + '/gamma') + 4 * ca, rt.gptr(bn + '/beta') + 4 * ca, frozen=bn in self.frozen)
                self.bwd += [fa_, fb_, aa_, ab_] if (pre is not None or ra_ is None) else [ra_, rb_, fa_, fb_, aa_, ab_]
                # the skip is the activated output of an encoder conv_batch_act block
                gact[ta.id] = (dskip, 0, ca)
                self.dbg['dskip:' + bn] = dskip
                self.dbg['du:' + bn] = du
                graw[tb.id] = du
            elif op == 'convT':
                tin, tout = node.inputs[0], node.outputs[0]
                du = graw.get(tout.id)
                if du is None:
                    continue
                lay, r, cout, f = node.layer, cx['r'], cx['cout'], cx['f']
                pk = rt.packed[lay.name]
                if isinstance(du, dict):
                    # fused: BatchNorm-backward apply of the `up` channels + data gradient + weight gradient (csrc/convt_bwd_fused.hip)
                    cb_, sa = du, self._src_args(r)
                    cinp = r.c
                    gin = self._z(n, r.h, r.w, cinp)
                    af_, ao_, ct_ = cb_['aff'], cb_['aoff'], cb_['ctot']
                    fz = ops.make_ctbf_desc(g=cb_['g'], ldg=cb_['ldg'], yup=cb_['yup'], ldy=cb_['ldy'], bn_scale=_fp(af_['scale'], ao_), bn_shift=_fp(af_['shift'], ao_),
                                            bn_mean=_fp(af_['mean'], ao_), bn_rstd=_fp(af_['rstd'], ao_), bn_c1=_fp(cb_['coef'], ao_), bn_c2=_fp(cb_['coef'], ct_ + ao_),
                                            x=sa['x0'], ldx=sa['c0'], in_scale=sa['in_scale'], in_shift=sa['in_shift'], in_relu=sa['in_relu'],
                                            w_dgrad=pk['dgrad'].data_ptr(), w_npad=rup(cinp, 32), dx=gin.data_ptr(), lddx=cinp, dw=rt.gptr(lay.name + '/kernel'),
                                            cin=cinp, cout=cout, n=n, h=r.h, w_=r.w, dtype=dt)
                    bt = bst_target(tin) if (sa['in_relu'] and sa['in_scale'] is not None) else None
                    if bt is not None and bt[1] == cinp and bt[0].get('relu', 0) == 1:
                        sums_below = self._z(STAT_ROWS, 2, cinp, dtype=torch.float64)
                        fz.bst_sums, fz.bst_sums_ld, fz.bst_mean, fz.bst_rstd = _fp(sums_below), cinp, bt[0]['mean'], bt[0]['rstd']
                        fused[tin.id] = sums_below
                    nbc = lib.satcv_convt_bwd_fused_workspace(C.byref(fz))
                    if nbc <= 0:
                        raise RuntimeError('convt_bwd_fused: the probe accepted this shape, the final descriptor did not')
                    wsc = self._z(max(nbc // 4, 1), dtype=torch.float32)
                    fz.workspace, fz.workspace_bytes = wsc.data_ptr(), nbc
                    self.keep.append(fz)
                    cstep = lambda st, fz=fz: check(lib.satcv_convt_bwd_fused(C.byref(fz), st))
                    cstep.label = f"convt_bwd_fused n{n} {r.h}x{r.w} {cinp}<-4x{cout}{' +bnred' if fz.bst_sums else ''}"
                    cstep.work = dict(kind='convt_bwd_fused', px=n * r.h * r.w, cin=cinp, cout=cout, esize=es)
                    self.bwd.append(cstep)
                    gact[tin.id] = (gin, 0, cinp)
                    self.dbg['dx:' + lay.name] = gin
                    continue
                wstep = wgrad_step(r, du.data_ptr(), cout, lay, pk['cin'], cout, r.h, r.w, 1, 1, f=f)
                if not WGRAD_LATE:
                    self.bwd.append(wstep)
                cinp = r.c
                gin = self._z(n, r.h, r.w, cinp)
                self.bwd.append(dgrad_step(tin, x0=du.data_ptr(), c0=cout, w=pk['dgrad'].data_ptr(), y=gin.data_ptr(), ldy=cinp, n=n, h=r.h,
                                           w_=r.w, cout=cinp, cout_pad=rup(cinp, 32), kh=1, kw=1, dil=1, mode_in=1, f=f, dtype=dt))
                if WGRAD_LATE:
                    self.bwd.append(wstep)
                gact[tin.id] = (gin, 0, cinp)
                self.dbg['dx:' + lay.name] = gin
        if prev_node is not None:
            grads_ready(prev_node)
        if ws_need:
            ws = self._z(max(ws_need // 4, 1), dtype=torch.float32)
            for d in wdescs:
                if not d.defer_reduce:
                    d.workspace, d.workspace_bytes = ws.data_ptr(), ws_need
        if fused_ws_need:
            wsf = self._z(max(fused_ws_need // 4, 1), dtype=torch.float32)
            for d in fdescs:
                if not d.defer_reduce:
                    d.workspace, d.workspace_bytes = wsf.data_ptr(), fused_ws_need
        flush_reduces(force=True)
        if rgroups:
            from ._lib import ReduceJob
            own = [d for g in rgroups for d, _ in g['items']]
            arena = self._z(max(sum((d._own_ws + 255) // 256 * 256 for d in own) // 4, 1), dtype=torch.float32)
            off = 0
            for d in own:
                d.workspace, d.workspace_bytes = arena.data_ptr() + off, d._own_ws
                off += (d._own_ws + 255) // 256 * 256
            for g in rgroups:
                jobs, prefix, tot = [], [], 0
                for d, kind in g['items']:
                    j = ReduceJob()
                    check((lib.satcv_conv2d_wgrad_reduce_job if kind == 'w' else lib.satcv_conv2d_bwd_fused_reduce_job)(C.byref(d), C.byref(j)))
                    prefix.append(tot)
                    tot += int(lib.satcv_reduce_job_items(C.byref(j)))
                    jobs.append(j)
                arr = (ReduceJob * len(jobs))(*jobs)
                g['tab'] = dict(jobs=torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(rt.dev),
                                prefix=torch.tensor(prefix, dtype=torch.int64, device=rt.dev), n=len(jobs), total=tot)
        if self.side is not None:
            evj = torch.cuda.Event()

            evj2 = torch.cuda.Event()

            def join(st, evj=evj, evj2=evj2):          # the optimizer (main stream) must see every weight gradient
                evj.record(self.side)
                torch.cuda.current_stream().wait_event(evj)
                if self.rstream is not None:
                    evj2.record(self.rstream)
                    torch.cuda.current_stream().wait_event(evj2)
            self.bwd.append(join)

    # -- execution
    # (a training plan lets EVERY eligible deep 3x3 launch -- also the data gradients that carry no statistics -- run on the 16x16x32 tile;
    #  inference keeps the library default, which preserves bit-identical results across batch splits: csrc/conv_igemm_fast.hip.  Round 6:
    #  the choice travels in each launch's descriptor (satcv_conv_desc.tile_policy, set in _conv_step) -- no process-global option is touched)
    def _run(self, steps, st):
        for s in steps:
            s(st)

    def run_forward(self, st):
        self._run(self.fwd, st)

    def run_backward(self, st):
        self._run(self.bwd, st)

