"""The ConvLSTM2D model family of the reference on the HIP library: build_lstm_layers / build_lstm_layers2 / get_lstm_model /
get_lstm_autoencoder / get_hybrid_model / get_hierarchical_model (utils/model_tools.py:666-920, 1016-1060).

Unlike the U-Net plans (engine.Plan: one static launch list per input shape) these models are small, recurrent and multi-input, so
they run as an eager TAPE of C-ABI launches: every layer object has forward / backward methods that call the library (ops.py wrappers
around include/satcv.h) on device tensors; torch is the allocator.  No CPU path, no torch arithmetic on the data path.

Layout: a sequence (B, T, H, W, C) is ingested TIME-MAJOR, (T, B, H, W, Cpad) (satcv_ingest_seq), so that
  * the input convolution of ALL time steps of a ConvLSTM2D is ONE implicit-GEMM launch over T * B images (it does not depend on h),
  * a single step is a contiguous slice, and the recurrent weight gradient over steps 1 .. T-1 is one launch over (T-1) * B images.
Per step: recurrent 3x3 conv of h_{t-1} (satcv_conv2d_igemm) + the gate kernel (satcv_convlstm_gates_fwd).  BPTT: gate backward
(satcv_convlstm_gates_bwd) + the recurrent data gradient per step; weight / bias gradients once per layer after the loop.

Keras semantics (unpinned, see oracle/convlstm.py): gate order i, f, c, o; `activation=None` at every reference call site;
recurrent_activation is version dependent -- RECURRENT_ACTIVATION = 'hard_sigmoid' (Keras 2.x / TF 2.x default for ConvLSTM2D) or
'sigmoid' (Keras 3); unit_forget_bias; orthogonal recurrent initialiser; the recurrent convolution is never dilated.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import ops
from . import model_tools as mt
from ._lib import lib, check, F32, BF16, LstmGatesDesc, DenseDesc

RECURRENT_ACTIVATION = os.environ.get('SATCV_LSTM_RECURRENT_ACTIVATION', 'hard_sigmoid')
BN_EPS, BN_MOMENTUM = 1e-3, 0.99


def _dev():
    return torch.device('cuda', torch.cuda.current_device())


class _Params:
    """flat float32 parameters / gradients / Adam slots of one model (what engine.Runtime is to the U-Net graphs)"""

    def __init__(self):
        self.specs, self.n = {}, 0
        self.flat = self.grad = self.m = self.v = None
        self.state = None
        self.init = {}
        # MFMA operand images of the convolution kernels (kernel name -> forward / data-gradient image), valid for parameter version `ver`
        self.version = 0
        self._pk, self._pk_tab, self._pk_dtype = {}, None, None

    def bump(self):
        """the parameters changed (optimizer step, set_weights): every packed image is stale"""
        self.version += 1

    def get_packed(self, name, cin_pad, dtype):
        """forward and data-gradient operand images of Conv2D kernel `name` (ops.pack_weights layouts).  The images live in persistent buffers;
        when the parameters have changed since they were packed, ALL registered kernels are repacked by ONE launch (satcv_pack_weights_batched) --
        a training step packed every layer's kernels with a launch each (40 launches of ~5 us in a get_lstm_model step: profiles/r05_lstm_kernel_stats.csv),
        and inference repacked unchanged weights on every call."""
        if self._pk_dtype != dtype:
            self._pk, self._pk_tab, self._pk_dtype = {}, None, dtype
        e = self._pk.get(name)
        if e is None:
            k = self.p(name)
            fwd, dg = ops.pack_weights(k, cin_pad, dtype)
            self._pk[name] = e = dict(fwd=fwd, dg=dg, cin_pad=cin_pad, shape=tuple(k.shape), ver=self.version)
            self._pk_tab = None
            return fwd, dg
        if e['cin_pad'] != cin_pad:
            raise ValueError(f'{name}: packed with cin_pad {e["cin_pad"]}, asked for {cin_pad}')
        if e['ver'] != self.version:
            self._repack_all(dtype)
        return e['fwd'], e['dg']

    def prepare_capture(self):
        """called right before a training step is captured into a graph: the captured step must CONTAIN the repack launch -- replays update
        the parameters on the device and only bump `version` on the host, so a graph captured while the images happened to be fresh (an eager
        predict / evaluate after the last optimizer step) would multiply with the images of capture time for ever.  The job table is built
        here, outside the capture (a host-to-device copy)."""
        if self._pk:
            self._job_table()
        self.bump()

    def _job_table(self):
        from ._lib import PackJob
        if self._pk_tab is None:
            jobs = []
            for name, e in self._pk.items():
                kh, kw, cin, cout = e['shape']
                src = self.p(name).data_ptr()
                jobs.append(PackJob(src, e['fwd'].data_ptr(), 0, kh * kw, cin, cout, e['cin_pad'], ops.rup(cout, 32)))
                jobs.append(PackJob(src, e['dg'].data_ptr(), 1, kh * kw, cin, cout, ops.rup(cout, 16), ops.rup(cin, 32)))
            prefix, tot = [], 0
            for j in jobs:
                prefix.append(tot)
                tot += int(lib.satcv_pack_job_items(C.byref(j)))
            arr = (PackJob * len(jobs))(*jobs)
            self._pk_tab = dict(jobs=torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(_dev()),
                                prefix=torch.tensor(prefix, dtype=torch.int64, device=_dev()), n=len(jobs), total=tot)
        return self._pk_tab

    def _repack_all(self, dtype):
        t = self._job_table()
        check(lib.satcv_pack_weights_batched(t['jobs'].data_ptr(), t['prefix'].data_ptr(), t['n'], t['total'], dtype, ops.stream_ptr()))
        for e in self._pk.values():
            e['ver'] = self.version

    def add(self, name, shape, init, trainable=True):
        size = int(np.prod(shape))
        self.specs[name] = (self.n, tuple(shape), trainable)
        self.init[name] = init
        self.n += ops.rup(size, 4)
        return name

    def build(self):
        host = np.zeros(max(self.n, 4), np.float32)
        for name, (off, shape, _) in self.specs.items():
            host[off:off + int(np.prod(shape))] = np.asarray(self.init[name](), np.float32).reshape(-1)
        self.flat = torch.from_numpy(host).to(_dev())
        self.grad = torch.zeros_like(self.flat)
        self.m, self.v = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        self.state = torch.tensor([1e-3, 0.0, 1.0, 0.0], dtype=torch.float32, device=_dev())
        mul = np.ones(max(self.n, 4), np.float32)
        for name, (off, shape, tr) in self.specs.items():
            if not tr:
                mul[off:off + int(np.prod(shape))] = 0
        self.lr_mul = torch.from_numpy(mul).to(_dev())

    def p(self, name):
        off, shape, _ = self.specs[name]
        return self.flat[off:off + int(np.prod(shape))].view(shape)

    def g(self, name):
        off, shape, _ = self.specs[name]
        return self.grad[off:off + int(np.prod(shape))].view(shape)


def _glorot(rng, shape, fan_in, fan_out):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return lambda: rng.uniform(-lim, lim, size=shape).astype(np.float32)


def _orthogonal(rng, shape):
    def init():
        rows, cols = int(np.prod(shape[:-1])), shape[-1]
        a = rng.standard_normal((rows, cols))
        q, r = np.linalg.qr(a.T if rows < cols else a)
        q = q * np.sign(np.diag(r))
        q = q.T if rows < cols else q
        return q.reshape(shape).astype(np.float32)
    return init


class Act:
    """a device activation: tensor (N, H, W, Cpad) in the storage type, its real channel count, and the BatchNorm (+ ReLU) its consumers
    still have to apply (scale, shift, mean, rstd) -- the same raw-storage contract as the U-Net engine"""

    def __init__(self, t, c, bn=None, relu=False):
        self.t, self.c, self.bn, self.relu = t, c, bn, relu

    @property
    def scale(self):
        return self.bn[0] if self.bn else None

    @property
    def shift(self):
        return self.bn[1] if self.bn else None


# ------------------------------------------------------------------------------------------------ layers
class ConvLSTM2D:
    """layers.ConvLSTM2D(filters, [3, 3], padding='same', activation=None, dilation_rate=d, return_sequences=..., return_state=...)
    (utils/model_tools.py:690-700, 710-720)"""

    def __init__(self, params, rng, name, cin, filters, dilation=1, return_sequences=True, activation=None):
        self.P, self.name, self.cin, self.F, self.dil, self.rs = params, name, cin, filters, dilation, return_sequences
        self.act = 0 if activation in (None, 'linear') else 1
        k, F = 3, filters
        params.add(f'{name}/kernel', (k, k, cin, 4 * F), _glorot(rng, (k, k, cin, 4 * F), k * k * cin, k * k * 4 * F))
        params.add(f'{name}/recurrent_kernel', (k, k, F, 4 * F), _orthogonal(rng, (k, k, F, 4 * F)))

        def bias():
            b = np.zeros(4 * F, np.float32)
            b[F:2 * F] = 1.0                           # unit_forget_bias
            return b
        params.add(f'{name}/bias', (4 * F,), bias)
        if F % 8 or (4 * F) % 16:
            raise NotImplementedError('ConvLSTM2D filters must be a multiple of 8')

    def _pack(self, dtype):
        P = self.P
        cpad = ops.rup(self.cin, 16)
        self.wk, self.wk_d = P.get_packed(f'{self.name}/kernel', cpad, dtype)
        self.wr, self.wr_d = P.get_packed(f'{self.name}/recurrent_kernel', ops.rup(self.F, 16), dtype)

    def forward(self, x, T, B, training, dtype, want_stats=True, repeat=False):
        """x: Act, time-major (T * B, H, W, Cpad).  Returns the raw output Act ((T * B, ..) or (B, ..), F channels) and its BatchNorm
        statistics rows (sum, sum of squares of the stored h) for the BatchNormalization every reference call site applies next.
        repeat: x is ONE image set (B, H, W, C) presented at every step (tf.repeat(tf.expand_dims(encoded, 1), n_time, 1),
        utils/model_tools.py:831-832): its input convolution is computed once."""
        self._pack(dtype)
        F, P = self.F, self.P
        n, H, W, _ = x.t.shape
        assert n == (B if repeat else T * B)
        td = x.t.dtype
        dev = x.t.device
        Fp = ops.rup(F, 16)
        xg = ops.conv2d(x.t, self.wk, 4 * F, dil=self.dil, bias=P.p(f'{self.name}/bias'), in_scale=x.scale, in_shift=x.shift, in_relu=x.relu)
        hseq = torch.zeros(T * B, H, W, Fp, dtype=td, device=dev)
        cseq = torch.empty(T, B * H * W, F, dtype=torch.float32, device=dev)
        gates = torch.empty(T, B * H * W, 4 * F, dtype=td, device=dev) if training else None
        stats = ops.new_stats(Fp, dev) if want_stats else None
        rk = 0 if RECURRENT_ACTIVATION == 'hard_sigmoid' else 1
        npix = B * H * W
        for t in range(T):
            hg = ops.conv2d(hseq[(t - 1) * B:t * B], self.wr, 4 * F) if t > 0 else None
            d = LstmGatesDesc()
            d.xg, d.ldx = (xg if repeat else xg[t * B:(t + 1) * B]).data_ptr(), xg.shape[-1]
            d.hg, d.ldh_g = (hg.data_ptr(), hg.shape[-1]) if hg is not None else (None, 0)
            d.c_prev = cseq[t - 1].data_ptr() if t > 0 else None
            d.c_out, d.h_out, d.ldh = cseq[t].data_ptr(), hseq[t * B:(t + 1) * B].data_ptr(), Fp
            d.gates_out = gates[t].data_ptr() if gates is not None else None
            if stats is not None and (self.rs or t == T - 1):
                d.stats, d.stats_ld = stats.data_ptr(), Fp
            d.npix, d.filters, d.rec_act, d.act, d.dtype = npix, F, rk, self.act, ops.DTYPE_CODE[td]
            check(lib.satcv_convlstm_gates_fwd(C.byref(d), ops.stream_ptr()))
        self.ctx = dict(x=x, T=T, B=B, H=H, W=W, hseq=hseq, cseq=cseq, gates=gates, rk=rk, td=td, repeat=repeat)
        out = hseq if self.rs else hseq[(T - 1) * B:]
        self.h_last = hseq[(T - 1) * B:]
        return Act(out, F), stats, (T * B * H * W if self.rs else B * H * W)

    def backward(self, dout, dstate_h=None, need_dx=True):
        """dout: gradient of the returned tensor (storage type, F padded channels), dstate_h: gradient of the final hidden state taken
        through return_state.  Accumulates the three parameter gradients; returns the gradient of the (activated) input."""
        c = self.ctx
        T, B, H, W, F, P = c['T'], c['B'], c['H'], c['W'], self.F, self.P
        td, dev = c['td'], c['hseq'].device
        Fp = c['hseq'].shape[-1]
        npix = B * H * W
        dz = torch.empty(T * B, H, W, 4 * F, dtype=td, device=dev)
        dc = [torch.empty(npix, F, dtype=torch.float32, device=dev) for _ in range(2)]
        dh_rec = None
        for t in range(T - 1, -1, -1):
            srcs = []
            if self.rs:
                srcs.append(dout[t * B:(t + 1) * B])
            elif t == T - 1:
                srcs.append(dout)
            if dstate_h is not None and t == T - 1:
                srcs.append(dstate_h)
            if dh_rec is not None:
                srcs.append(dh_rec)
            assert len(srcs) <= 2                   # (the state gradient only joins at t = T - 1, where there is no recurrent one)
            d = LstmGatesDesc()
            if srcs:
                d.dh_a, d.lddh_a = srcs[0].data_ptr(), srcs[0].shape[-1]
            if len(srcs) > 1:
                d.dh_b, d.lddh_b = srcs[1].data_ptr(), srcs[1].shape[-1]
            if not srcs:                            # no gradient reaches h_t (return_sequences=False, t < T - 1 has dh_rec; only T = 1 corner)
                z = torch.zeros(B, H, W, Fp, dtype=td, device=dev)
                d.dh_a, d.lddh_a = z.data_ptr(), Fp
            d.dc_next = dc[(t + 1) & 1].data_ptr() if t < T - 1 else None
            d.gates_out = c['gates'][t].data_ptr()
            d.c_prev = c['cseq'][t - 1].data_ptr() if t > 0 else None
            d.c_out = c['cseq'][t].data_ptr()
            d.dz_out, d.lddz = dz[t * B:(t + 1) * B].data_ptr(), 4 * F
            d.dc_prev_out = dc[t & 1].data_ptr()
            d.npix, d.filters, d.rec_act, d.act, d.dtype = npix, F, c['rk'], self.act, ops.DTYPE_CODE[td]
            check(lib.satcv_convlstm_gates_bwd(C.byref(d), ops.stream_ptr()))
            dh_rec = ops.conv2d_dgrad(dz[t * B:(t + 1) * B], self.wr_d, F) if t > 0 else None
        x = c['x']
        # weight gradients: the input kernel over all T * B images at once, the recurrent kernel over steps 1 .. T-1 (h_{t-1} = hseq[t-1])
        dz_in = dz
        if c['repeat']:          # the same input at every step: its gradients are those of the SUM of dz over the steps (linearity)
            dz_in = dz[:B].clone()
            for t in range(1, T):
                check(lib.satcv_add_act(dz_in.data_ptr(), None, None, dz[t * B:(t + 1) * B].data_ptr(), None, None, 0, dz_in.data_ptr(), npix, 4 * F,
                                        ops.DTYPE_CODE[td], ops.stream_ptr()))
        ops.conv2d_wgrad(x.t, dz_in, self.cin, 4 * F, dil=self.dil, in_scale=x.scale, in_shift=x.shift, in_relu=x.relu, dw=P.g(f'{self.name}/kernel'))
        if T > 1:
            ops.conv2d_wgrad(c['hseq'][:(T - 1) * B], dz[B:], F, 4 * F, dw=P.g(f'{self.name}/recurrent_kernel'))
        nb = lib.satcv_bias_grad_workspace(T * B * H * W, 4 * F)
        ws = torch.empty(max(nb // 4, 1), dtype=torch.float32, device=dev)
        check(lib.satcv_bias_grad(dz.data_ptr(), 4 * F, T * B * H * W, 4 * F, ops.DTYPE_CODE[td], P.g(f'{self.name}/bias').data_ptr(), ws.data_ptr(), ops.stream_ptr()))
        if not need_dx:
            return None
        return ops.conv2d_dgrad(dz_in, self.wk_d, self.cin, dil=self.dil)


class BatchNorm:
    """layers.BatchNormalization() on a ConvLSTM2D output (statistics over batch, time and space), followed -- or not -- by ReLU"""

    def __init__(self, params, name, c):
        self.P, self.name, self.c = params, name, c
        params.add(f'{name}/gamma', (c,), lambda: np.ones(c, np.float32))
        params.add(f'{name}/beta', (c,), lambda: np.zeros(c, np.float32))
        params.add(f'{name}/moving_mean', (c,), lambda: np.zeros(c, np.float32), trainable=False)
        params.add(f'{name}/moving_var', (c,), lambda: np.ones(c, np.float32), trainable=False)

    def forward(self, a, stats, count, training, relu=True, bessel=True):
        P, n = self.P, self.name
        cp = a.t.shape[-1]

        def padded(v, fill):                         # per-channel vectors over the PADDED channel count (pad channels hold zeros: scale 1, shift 0)
            if cp == self.c:
                return v
            out = torch.full((cp,), fill, dtype=torch.float32, device=v.device)
            out[:self.c] = v
            return out
        if training:
            g, b = padded(P.p(f'{n}/gamma'), 1.0), padded(P.p(f'{n}/beta'), 0.0)
            mm, mv = padded(P.p(f'{n}/moving_mean'), 0.0), padded(P.p(f'{n}/moving_var'), 1.0)
            scale, shift, mean, rstd = ops.bn_finalize_train(stats, count, g, b, mm, mv, BN_EPS, BN_MOMENTUM, 1, bessel)
            if cp != self.c:
                P.p(f'{n}/moving_mean').copy_(mm[:self.c]); P.p(f'{n}/moving_var').copy_(mv[:self.c])
        else:
            scale, shift = ops.bn_affine_infer(padded(P.p(f'{n}/gamma'), 1.0), padded(P.p(f'{n}/beta'), 0.0),
                                               padded(P.p(f'{n}/moving_mean'), 0.0), padded(P.p(f'{n}/moving_var'), 1.0), BN_EPS)
            mean = rstd = None
        self.ctx = dict(a=a, relu=relu)
        out = Act(a.t, a.c, (scale, shift, mean, rstd), relu)
        self.out = out
        return out

    def backward(self, da):
        """da: gradient of the normalised (+ ReLU) tensor, storage type.  Returns the gradient of the raw input."""
        a, out = self.ctx['a'], self.out
        scale, shift, mean, rstd = out.bn
        y = a.t
        n, h, w_, cp = y.shape
        dev = y.device
        sums = ops.new_stats(cp, dev)
        coef = torch.empty(2, cp, dtype=torch.float32, device=dev)
        dgamma, dbeta = torch.empty(cp, dtype=torch.float32, device=dev), torch.empty(cp, dtype=torch.float32, device=dev)
        dy = torch.empty_like(y)
        d = ops.make_bnbwd_desc(yraw=y.data_ptr(), ldy=cp, scale=scale.data_ptr(), shift=shift.data_ptr(), mean=mean.data_ptr(), rstd=rstd.data_ptr(),
                                n=n, h=h, w_=w_, c=cp, dtype=ops.DTYPE_CODE[y.dtype], da=da.data_ptr(), ldda=da.shape[-1], sums=sums.data_ptr(),
                                sums_ld=cp, coef=coef.data_ptr(), dy=dy.data_ptr(), lddy_out=cp, linear=0 if self.ctx['relu'] else 1)
        st = ops.stream_ptr()
        check(lib.satcv_bn_bwd_reduce(C.byref(d), st))
        check(lib.satcv_bn_bwd_finalize(sums.data_ptr(), cp, cp, float(n * h * w_), dgamma.data_ptr(), dbeta.data_ptr(), coef.data_ptr(), 0, st))
        check(lib.satcv_bn_bwd_apply(C.byref(d), st))
        self.P.g(f'{self.name}/gamma').add_(dgamma[:self.c])
        self.P.g(f'{self.name}/beta').add_(dbeta[:self.c])
        return dy


class Dropout:
    """layers.Dropout(rate) (element-wise, utils/model_tools.py:699-700) / layers.SpatialDropout2D(rate) (whole feature maps,
    :899-900, 904-905) on the tape: training draws a counter-based mask (satcv_dropout_mask: 0 or 1 / (1 - rate)) and materialises
    act(x) * mask (satcv_dropout_apply applies the pending BatchNorm + ReLU on the way); inference passes the activation through."""

    def __init__(self, rate, spatial, seed):
        self.rate, self.spatial, self.seed, self.calls = float(rate), bool(spatial), int(seed), 0
        self.mask = None

    def forward(self, a, training):
        if not training or self.rate <= 0.0:
            self.mask = None
            return a
        n, h, w_, cp = a.t.shape
        rows = n if self.spatial else n * h * w_
        self.mask = torch.empty(rows, cp, dtype=torch.float32, device=a.t.device)
        self.calls += 1
        st = ops.stream_ptr()
        check(lib.satcv_dropout_mask(self.seed, self.calls << 8, self.rate, rows * cp, self.mask.data_ptr(), st))
        out = torch.empty_like(a.t)
        check(lib.satcv_dropout_apply(a.t.data_ptr(), cp, a.scale.data_ptr() if a.bn else None, a.shift.data_ptr() if a.bn else None, 1 if a.relu else 0,
                                      self.mask.data_ptr(), cp, 0 if self.spatial else 1, out.data_ptr(), cp, n, h * w_, cp, ops.DTYPE_CODE[out.dtype], st))
        return Act(out, a.c)

    def backward(self, g):
        if self.mask is None:
            return g
        n, h, w_, cp = g.shape
        out = torch.empty_like(g)
        check(lib.satcv_dropout_apply(g.data_ptr(), cp, None, None, 0, self.mask.data_ptr(), cp, 0 if self.spatial else 1, out.data_ptr(), cp, n, h * w_, cp,
                                      ops.DTYPE_CODE[g.dtype], ops.stream_ptr()))
        return out


class Dense1x1:
    """layers.Conv2D(cout, [1, 1]) over the concatenation of one or two sources, with softmax / sigmoid / linear / ReLU(max_value)
    (satcv_dense_small_fwd / _bwd); a source may sit on a coarser grid and be read through tf.image.resize(..., 'nearest')"""
    ACT = {'softmax': 0, 'sigmoid': 1, 'linear': 2, None: 2, 'relu': 3}

    def __init__(self, params, rng, name, cins, cout, activation='linear', max_value=0.0):
        self.P, self.name, self.cins, self.cout = params, name, list(cins), cout
        self.activation, self.max_value = self.ACT[activation], float(max_value or 0.0)
        cin = sum(cins)
        params.add(f'{name}/kernel', (1, 1, cin, cout), _glorot(rng, (1, 1, cin, cout), cin, cout))
        params.add(f'{name}/bias', (cout,), lambda: np.zeros(cout, np.float32))

    def _desc(self, srcs, out_hw):
        d = DenseDesc()
        d.nsrc = len(srcs)
        for i, (a, resize) in enumerate(srcs):
            s = d.src[i]
            s.x, s.ld, s.cin = a.t.data_ptr(), a.t.shape[-1], self.cins[i]
            s.dtype = F32 if a.t.dtype == torch.float32 else BF16
            if a.bn is not None:
                s.in_scale, s.in_shift, s.in_relu = a.scale.data_ptr(), a.shift.data_ptr(), 1 if a.relu else 0
            elif a.relu:                               # a plain pending ReLU (a linear head's output consumed through Activation('relu'))
                one, zero = self._ident(self.cins[i], a.t.device)
                s.in_scale, s.in_shift, s.in_relu = one.data_ptr(), zero.data_ptr(), 1
            if resize:
                s.hs, s.ws = a.t.shape[1], a.t.shape[2]
        d.w, d.b, d.cout = self.P.p(f'{self.name}/kernel').data_ptr(), self.P.p(f'{self.name}/bias').data_ptr(), self.cout
        d.activation, d.max_value = self.activation, self.max_value
        d.h, d.w_ = out_hw
        return d

    def _ident(self, c, dev):
        if getattr(self, '_id', None) is None or self._id[0].numel() < c:
            self._id = (torch.ones(c, dtype=torch.float32, device=dev), torch.zeros(c, dtype=torch.float32, device=dev))
        return self._id

    def forward(self, srcs, out_hw=None, want_classes=False):
        """srcs: list of (Act, resize flag).  Returns float32 (N, H, W, cout) (and int32 classes for softmax heads)."""
        a0 = srcs[0][0]
        nimg = a0.t.shape[0]
        H, W = out_hw if out_hw is not None else (a0.t.shape[1], a0.t.shape[2])
        d = self._desc(srcs, (H, W))
        out = torch.empty(nimg, H, W, self.cout, dtype=torch.float32, device=a0.t.device)
        classes = torch.empty(nimg, H, W, dtype=torch.int32, device=a0.t.device) if (want_classes and self.activation == 0) else None
        d.out, d.classes, d.npix = out.data_ptr(), classes.data_ptr() if classes is not None else None, nimg * H * W
        check(lib.satcv_dense_small_fwd(C.byref(d), ops.stream_ptr()))
        self.ctx = dict(srcs=srcs, hw=(H, W), out=out)
        return (out, classes) if want_classes else out

    def backward(self, dout, need_dx=(True, True)):
        """dout: dL/d out for linear / ReLU heads, dL/dlogits (from the loss kernel) for softmax / sigmoid heads.  Returns one gradient
        tensor per source (w.r.t. the source's ACTIVATED values, on the source's grid, in the source's storage type) or None."""
        srcs, (H, W), out = self.ctx['srcs'], self.ctx['hw'], self.ctx['out']
        d = self._desc(srcs, (H, W))
        if d.activation in (0, 1):
            d.activation = 2
        nimg = out.shape[0]
        dxs = []
        for i, (a, resize) in enumerate(srcs):
            if need_dx[i]:
                dx = torch.zeros(a.t.shape, dtype=a.t.dtype, device=a.t.device)
                d.src[i].dx, d.src[i].lddx, d.src[i].dx_dtype = dx.data_ptr(), dx.shape[-1], F32 if dx.dtype == torch.float32 else BF16
                dxs.append(dx)
            else:
                dxs.append(None)
        dz = torch.empty(nimg, H, W, self.cout, dtype=torch.float32, device=out.device)
        d.out, d.dout, d.dz_out, d.npix = out.data_ptr(), dout.data_ptr(), dz.data_ptr(), nimg * H * W
        d.dw, d.db = self.P.g(f'{self.name}/kernel').data_ptr(), self.P.g(f'{self.name}/bias').data_ptr()
        check(lib.satcv_dense_small_bwd(C.byref(d), ops.stream_ptr()))
        return dxs


# ------------------------------------------------------------------------------------------------ the two layer stacks
class LSTMLayers:
    """build_lstm_layers (utils/model_tools.py:666-717): ConvLSTM2D(64) 'conv_lstm' -> BatchNormalization 'batch_norm' -> ReLU ->
    ConvLSTM2D(64, dilation (3, 3), return_sequences) 'dilated_conv_lstm' -> BatchNormalization 'batch_norm2' -> ReLU"""

    def __init__(self, params, rng, n_channels, filters=64, return_sequences=False, prefix='', dropout=None):
        self.l1 = ConvLSTM2D(params, rng, prefix + 'conv_lstm', n_channels, filters, 1, True)
        self.bn1 = BatchNorm(params, prefix + 'batch_norm', filters)
        self.drop = Dropout(dropout, False, rng.integers(1 << 62)) if dropout else None       # layers.Dropout(dropout)(activated), :699-700
        self.l2 = ConvLSTM2D(params, rng, prefix + 'dilated_conv_lstm', filters, filters, 3, return_sequences)
        self.bn2 = BatchNorm(params, prefix + 'batch_norm2', filters)
        self.F = filters

    def forward(self, x, T, B, training, dtype):
        s1, st1, n1 = self.l1.forward(x, T, B, training, dtype)
        a1 = self.bn1.forward(s1, st1, n1, training)
        if self.drop is not None:
            a1 = self.drop.forward(a1, training)
        h2, st2, n2 = self.l2.forward(a1, T, B, training, dtype)
        return self.bn2.forward(h2, st2, n2, training)

    def backward(self, da2, need_dx=False):
        dh2 = self.bn2.backward(da2)
        da1 = self.l2.backward(dh2)
        if self.drop is not None:
            da1 = self.drop.backward(da1)
        ds1 = self.bn1.backward(da1)
        return self.l1.backward(ds1, need_dx=need_dx)


class LSTMLayers2:
    """build_lstm_layers2 (utils/model_tools.py:719-771): ConvLSTM2D(16, return_sequences, return_state) 'conv_lstm' -> BatchNormalization
    'batch_norm' -> ReLU -> ConvLSTM2D(16, dilation (3, 3), last state) 'dilated_conv_lstm' -> BatchNormalization 'batch_norm2';
    output = ReLU(state_h + normalized2) with state_h the FIRST layer's final hidden state."""

    def __init__(self, params, rng, n_channels, filters=16, prefix='', dropout=None):
        self.l1 = ConvLSTM2D(params, rng, prefix + 'conv_lstm', n_channels, filters, 1, True)
        self.bn1 = BatchNorm(params, prefix + 'batch_norm', filters)
        self.drop = Dropout(dropout, False, rng.integers(1 << 62)) if dropout else None       # layers.Dropout(dropout)(activated), :752-753
        self.l2 = ConvLSTM2D(params, rng, prefix + 'dilated_conv_lstm', filters, filters, 3, False)
        self.bn2 = BatchNorm(params, prefix + 'batch_norm2', filters)
        self.F = filters

    def forward(self, x, T, B, training, dtype):
        s1, st1, n1 = self.l1.forward(x, T, B, training, dtype)
        a1 = self.bn1.forward(s1, st1, n1, training)
        if self.drop is not None:
            a1 = self.drop.forward(a1, training)
        h2, st2, n2 = self.l2.forward(a1, T, B, training, dtype)
        z2 = self.bn2.forward(h2, st2, n2, training, relu=False)
        state_h = self.l1.h_last
        out = torch.empty_like(state_h)
        nb, H, W, cp = state_h.shape
        check(lib.satcv_add_act(z2.t.data_ptr(), z2.scale.data_ptr(), z2.shift.data_ptr(), state_h.data_ptr(), None, None, 1, out.data_ptr(),
                                nb * H * W, cp, ops.DTYPE_CODE[out.dtype], ops.stream_ptr()))
        self.out = out
        return Act(out, self.F)

    def backward(self, dout, need_dx=False):
        g = dout.clone()
        check(lib.satcv_relu_bwd(self.out.data_ptr(), g.data_ptr(), g.numel(), ops.DTYPE_CODE[g.dtype], ops.stream_ptr()))
        dh2 = self.bn2.backward(g)                    # the masked gradient serves both addends
        da1 = self.l2.backward(dh2)
        if self.drop is not None:
            da1 = self.drop.backward(da1)
        ds1 = self.bn1.backward(da1)
        return self.l1.backward(ds1, dstate_h=g, need_dx=need_dx)


def _ingest_seq(x, cpad, dtype):
    """(B, T, H, W, C) float32 host / device array -> time-major storage tensor (T * B, H, W, cpad)"""
    xt = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    xt = xt.to(_dev(), torch.float32).contiguous()
    B, T, H, W, Cc = xt.shape
    out = torch.empty(T * B, H, W, cpad, dtype=ops.TORCH_DTYPE[dtype], device=xt.device)
    check(lib.satcv_ingest_seq(xt.data_ptr(), out.data_ptr(), B, T, H, W, Cc, cpad, dtype, ops.stream_ptr()))
    return out, (B, T, H, W)


def _dev_f32(a):
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return t.to(_dev(), torch.float32).contiguous()


class _SeqModelBase:
    """compile / fit / predict / train_on_batch plumbing shared by the LSTM-family models (Adam through satcv_adam_step)"""

    def _finish(self):
        self.P.build()
        self.optimizer, self._loss = None, None
        self._metrics, self.metrics_names, self.stop_training = [], ['loss'], False
        self.compute_dtype = mt._DEFAULT_DTYPE
        self._graphs = {}

    @property
    def dtype_code(self):
        return BF16 if self.compute_dtype == 'bfloat16' else F32

    # ---- Keras surface of the reference's call sites (compile / fit / evaluate as in notebooks/UNET_G4G_2019_solar.ipynb:1206-1275 and
    # utils/model_tools.py:1162-1176).  Arguments this implementation does not act on are REFUSED, never swallowed.
    output_names = ('output',)

    def compile(self, optimizer='adam', loss=None, metrics=None, loss_weights=None, **kw):
        if kw:
            raise TypeError(f'{type(self).__name__}.compile: unsupported arguments {sorted(kw)}')
        if loss_weights is not None:
            raise NotImplementedError(f'{type(self).__name__}.compile: loss_weights (the outputs\' losses are added with weight 1, the Keras default)')
        self.optimizer = mt.Adam() if isinstance(optimizer, str) else optimizer
        spec = loss(mt._LossArg('y_true'), mt._LossArg('y_pred')) if callable(loss) else loss
        if not isinstance(spec, mt.LossSpec):
            raise ValueError('loss must be one of the model_tools loss functions (or a lambda wrapping one)')
        self._loss = spec
        ms = []
        for m_ in (metrics or []):
            if isinstance(m_, str) and m_ in ('accuracy', 'acc', 'categorical_accuracy'):
                ms.append(('accuracy', None))
            elif isinstance(m_, mt.MeanIoU):
                ms.append((m_.name, m_.num_classes))
            else:
                raise ValueError(f'{type(self).__name__}.compile: metric {m_!r} is not supported (accuracy, model_tools.MeanIoU)')
        self._metrics = ms
        outs = list(self.output_names)
        self.metrics_names = ['loss'] + ([f'{o}_loss' for o in outs] if len(outs) > 1 else [])
        for o in outs:
            self.metrics_names += [(f'{o}_{n}' if len(outs) > 1 else n) for n, _ in ms]
        self.stop_training = False
        self.__dict__.pop('_loss_w_dev', None)
        self._drop_graphs()
        self.P.state[0:1].fill_(self.optimizer._lr)

    @staticmethod
    def _metric_value(kind, ncls, conf):
        conf = conf.astype(np.float64)
        if kind == 'accuracy':
            return float(np.trace(conf) / max(conf.sum(), 1))
        tp = np.diag(conf)
        denom = conf.sum(0) + conf.sum(1) - tp
        valid = denom > 0
        return float((tp[valid] / denom[valid]).mean()) if valid.any() else 0.0

    def _loss_grad(self, out, y_true, activation):
        """fused device loss on the model output: returns (loss tensor, dL/dlogits for softmax / sigmoid heads, dL/dout for linear ones)"""
        if self._loss.weights is None:
            w = None
        else:
            w = self.__dict__.get('_loss_w_dev')
            if w is None or w.device != out.device:
                w = self._loss_w_dev = torch.as_tensor(np.asarray(self._loss.weights, dtype=np.float32)).to(out.device)
        y = y_true if isinstance(y_true, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(y_true, dtype=np.float32))
        y = y.to(out.device, torch.float32).contiguous()
        loss, dlog = ops.loss_fwd_bwd(self._loss.kind, out, y, w, activation=activation, eps=self._loss.eps)
        return loss, dlog

    # ---- hipGraph replay of a training step (SATCV_LSTM_GRAPH=0: always eager).  A ConvLSTM2D step is ~300 launches of a few
    # microseconds on the tape (per time step: recurrent convolution, gate kernel, their backward twins): the host, not the GPU, sets
    # the pace.  After two eager steps with the same shapes the third is captured (torch.cuda.graph: the tape's allocations come from
    # the graph's private pool) and later steps copy the batch into the static input tensors and replay.  The learning rate is
    # written outside the graph before every replay; everything else the step reads lives on the device.
    def _drop_graphs(self):
        self.__dict__['_graphs'] = {}

    def _graphed_step(self, step_fn, tensors):
        """step_fn(*device tensors) -> loss tensor (no host synchronisation inside); returns the loss tensor of this step"""
        if os.environ.get('SATCV_LSTM_GRAPH', '1') == '0' or getattr(self, '_no_graph', False):
            self.P.state[0:1].fill_(self.optimizer._lr)
            return step_fn(*tensors)
        graphs = self.__dict__.setdefault('_graphs', {})
        key = (self.compute_dtype,) + tuple((tuple(t.shape), t.dtype) for t in tensors)
        st = graphs.setdefault(key, {'n': 0})
        self.P.state[0:1].fill_(self.optimizer._lr)
        if 'g' not in st:
            st['n'] += 1
            if st['n'] <= 2 or st.get('off'):
                return step_fn(*tensors)
            try:
                st['in'] = [t.clone() for t in tensors]
                g = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                self.P.prepare_capture()        # (the graph must hold the repack launch: see _Params.prepare_capture)
                with torch.cuda.graph(g):
                    st['loss'] = step_fn(*st['in'])
                st['g'] = g
            except Exception as e:           # a step that cannot be captured stays eager (and says so once)
                import warnings
                warnings.warn(f'{type(self).__name__}: training step not captured ({e}); running eagerly')
                st['off'] = True
                st.pop('in', None)
                return step_fn(*tensors)
        for dst, src in zip(st['in'], tensors):
            dst.copy_(src, non_blocking=True)
        st['g'].replay()
        self.P.bump()                           # (the replayed step updated the parameters: packed images are stale for eager callers)
        return st['loss']

    def _adam(self, set_lr=True):
        P, opt = self.P, self.optimizer
        if set_lr:
            P.state[0:1].fill_(opt._lr)
        check(lib.satcv_adam_step(P.flat.data_ptr(), P.grad.data_ptr(), P.m.data_ptr(), P.v.data_ptr(), P.flat.numel(), opt.beta_1, opt.beta_2,
                                  opt.epsilon, P.state.data_ptr(), P.lr_mul.data_ptr(), ops.stream_ptr()))
        P.bump()

    def get_weights_dict(self):
        return {k: self.P.p(k).detach().cpu().numpy().copy() for k in self.P.specs}

    def set_weights_dict(self, d):
        for k, v in d.items():
            self.P.p(k).copy_(torch.as_tensor(np.asarray(v, np.float32)).to(self.P.flat.device).view(self.P.specs[k][1]))
        self.P.bump()

    # Model.save_weights / load_weights / evaluate call sites of the reference (utils/model_tools.py:1162-1196 and the notebooks):
    # the own .npz container; a model with a static-engine branch (the hybrid's U-Net) stores that branch under 'unet::<name>'
    def _branch_models(self):
        return {'unet': self.unet} if hasattr(self, 'unet') else {}

    @staticmethod
    def _refuse_h5(path, what):
        if str(path).endswith(('.h5', '.hdf5', '.keras')):
            raise NotImplementedError(f'{what}({path!r}): the ConvLSTM2D family is stored in the .npz container (one array per variable, keyed by its '
                                      f'Keras variable name); Keras HDF5 interchange covers the U-Net / ACNN models (model_tools.Model)')

    def save_weights(self, path):
        self._refuse_h5(path, 'save_weights')
        d = {k: v for k, v in self.get_weights_dict().items()}
        for tag, bm in self._branch_models().items():
            d.update({f'{tag}::{k}': v for k, v in bm.get_weights_dict().items()})
        np.savez(path if str(path).endswith('.npz') else str(path) + '.npz', **d)

    save = save_weights                    # Model.save call sites (ModelCheckpoint(save_weights_only=False)): the same container

    def load_weights(self, path, by_name=False, skip_mismatch=False):
        """the container is keyed by variable names, so `by_name` changes nothing; skip_mismatch (utils/model_tools.py:1162) skips variables that
        are absent from the file or have another shape instead of raising.  Returns the list of skipped names."""
        self._refuse_h5(path, 'load_weights')
        p = path if os.path.exists(path) else str(path) + '.npz'
        skipped = []
        with np.load(p, allow_pickle=False) as z:
            own = {k: z[k] for k in z.files if '::' not in k}
            take = {}
            for k, (_, shape, *_rest) in self.P.specs.items():
                if k in own and tuple(own[k].shape) == tuple(shape):
                    take[k] = own[k]
                elif skip_mismatch:
                    skipped.append(k)
                elif k not in own:
                    raise ValueError(f'{p}: no weights for {k!r}')
                else:
                    raise ValueError(f'{p}: {k!r} has shape {tuple(own[k].shape)}, the model expects {tuple(shape)}')
            self.set_weights_dict(take)
            for tag, bm in self._branch_models().items():
                bm.set_weights_dict({k.split('::', 1)[1]: z[k] for k in z.files if k.startswith(tag + '::')}, skip_mismatch=skip_mismatch)
        if skipped:
            import warnings
            warnings.warn(f'load_weights({p!r}, skip_mismatch=True): skipped {skipped}')
        return skipped

    def _eval_outputs(self, xb):
        """-> [(output tensor in the layout the loss kernel takes, activation tag, labels transform)] per model output"""
        raise NotImplementedError

    @staticmethod
    def _batches_of(x, y, batch_size, sample_weight=None):
        """arrays -> slices; a Sequence / iterable of (x, y[, sample_weight]) batches -> itself"""
        if y is not None:
            if sample_weight is not None:
                raise NotImplementedError('sample_weight: the fused device losses have no per-sample weights')
            multi = isinstance(x, (list, tuple))
            ymulti = isinstance(y, (list, tuple))
            n = (x[0] if multi else x).shape[0]
            bs = batch_size or 32
            for i in range(0, n, bs):
                yield ([a[i:i + bs] for a in x] if multi else x[i:i + bs]), ([a[i:i + bs] for a in y] if ymulti else y[i:i + bs])
            return
        it = (x[i] for i in range(len(x))) if hasattr(x, '__getitem__') and hasattr(x, '__len__') and not isinstance(x, (list, tuple, np.ndarray)) else iter(x)
        for batch in it:
            if len(batch) > 2 and batch[2] is not None and any(w is not None for w in (batch[2] if isinstance(batch[2], (list, tuple)) else [batch[2]])):
                raise NotImplementedError('the batch carries sample weights (LSTMAutoencoderGenerator(sample_weights=True), utils/processing.py:974-1049): '
                                          'the fused device losses have no per-sample weights -- build the generator with sample_weights=False')
            xb = batch[0]
            if isinstance(xb, (list, tuple)):
                xb = [a for a in xb if a is not None]
            yield xb, batch[1]

    def evaluate(self, x=None, y=None, batch_size=None, verbose=0, sample_weight=None, steps=None, return_dict=False, **kw):
        """Model.evaluate -> [loss, (per-output losses of a multi-output model,) metrics ...] aligned with metrics_names (the reference indexes
        it: utils/model_tools.py:1164-1168); a bare float when the loss is the only entry.  Inference mode: moving statistics, no dropout."""
        if kw:
            raise TypeError(f'{type(self).__name__}.evaluate: unsupported arguments {sorted(kw)}')
        if self._loss is None:
            raise RuntimeError('compile() the model before evaluate')
        nout = len(self.output_names)
        tot = np.zeros(1 + (nout if nout > 1 else 0))
        confs = [[np.zeros((nc or 1, nc or 1), np.int64) for _, nc in self._metrics] for _ in range(nout)]
        cnt = 0
        for i, (xb, yb) in enumerate(self._batches_of(x, y, batch_size, sample_weight)):
            if steps is not None and i >= steps:
                break
            outs = self._eval_outputs(xb)
            ys = list(yb) if isinstance(yb, (list, tuple)) else [yb]
            k = (xb[0] if isinstance(xb, (list, tuple)) else xb).shape[0]
            losses = []
            for oi, ((out, act, ytf), yy) in enumerate(zip(outs, ys)):
                yd = ytf(_dev_f32(yy))
                losses.append(float(self._loss_grad(out, yd, act)[0].item()))
                if self._metrics:
                    pred = out.float().argmax(-1).flatten().cpu().numpy()
                    true = yd.argmax(-1).flatten().cpu().numpy()
                    for mi, (kind, nc) in enumerate(self._metrics):
                        ncl = nc or out.shape[-1]
                        if confs[oi][mi].shape[0] != ncl:
                            confs[oi][mi] = np.zeros((ncl, ncl), np.int64)
                        np.add.at(confs[oi][mi], (true, pred), 1)
            tot[0] += sum(losses) * k
            if nout > 1:
                tot[1:] += np.asarray(losses) * k
            cnt += k
        vals = list(tot / max(cnt, 1))
        for oi in range(nout):
            vals += [self._metric_value(kind, nc, confs[oi][mi]) for mi, (kind, nc) in enumerate(self._metrics)]
        if return_dict:
            return dict(zip(self.metrics_names, vals))
        return vals if len(vals) > 1 else vals[0]

    def fit(self, x=None, y=None, batch_size=None, epochs=1, verbose=0, callbacks=None, validation_data=None, steps_per_epoch=None,
            validation_steps=None, initial_epoch=0, sample_weight=None, shuffle=False, **kw):
        """x: array(s) with y, or a Sequence / iterable of (x, y[, sample_weight]) batches (LSTMDataGenerator, LSTMAutoencoderGenerator,
        HybridDataGenerator).  callbacks (model_tools.ModelCheckpoint / TensorBoard) see `loss` and the `val_` entries of evaluate() on
        validation_data (arrays (x, y) or a batch iterable).  Arrays are taken in order (shuffle=True is refused: the reference fits from generators)."""
        if kw:
            raise TypeError(f'{type(self).__name__}.fit: unsupported arguments {sorted(kw)}')
        if shuffle:
            raise NotImplementedError(f'{type(self).__name__}.fit(shuffle=True): shuffle inside the generator (on_epoch_end), as the reference does')
        if self._loss is None:
            raise RuntimeError('compile() the model before fit')
        hist = mt.History()
        callbacks = list(callbacks or [])
        for cb in callbacks:
            cb.model = self
        self.stop_training = False
        for epoch in range(initial_epoch, epochs):
            tot, cnt = 0.0, 0
            for i, (xb, yb) in enumerate(self._batches_of(x, y, batch_size, sample_weight)):
                if steps_per_epoch is not None and i >= steps_per_epoch:
                    break
                k = (xb[0] if isinstance(xb, (list, tuple)) else xb).shape[0]
                tot, cnt = tot + self.train_on_batch(xb, yb) * k, cnt + k
            logs = {'loss': tot / max(cnt, 1)}
            if validation_data is not None:
                if isinstance(validation_data, tuple) and len(validation_data) in (2, 3) and not hasattr(validation_data[0], '__next__'):
                    v = self.evaluate(validation_data[0], validation_data[1], batch_size=batch_size, steps=validation_steps,
                                      sample_weight=validation_data[2] if len(validation_data) == 3 else None, return_dict=True)
                else:
                    v = self.evaluate(validation_data, steps=validation_steps, return_dict=True)
                logs.update({'val_' + k_: v_ for k_, v_ in v.items()})
            for k_, v_ in logs.items():
                hist.history[k_].append(v_)
            hist.epoch.append(epoch)
            if verbose:
                print(f'Epoch {epoch + 1}/{epochs} - ' + ' - '.join(f'{k_}: {v_:.4f}' for k_, v_ in logs.items()))
            for cb in callbacks:
                if hasattr(cb, 'on_epoch_end'):
                    cb.on_epoch_end(epoch, logs)
            if hasattr(x, 'on_epoch_end'):
                x.on_epoch_end()
            if self.stop_training:
                break
        return hist


class LSTMModel(_SeqModelBase):
    """get_lstm_model (utils/model_tools.py:773-808).  The reference body cannot run as coded (`layers.Input(n_time, None, None,
    n_channels)`, `activations(dense_layer)`: SURVEY Appendix B Q7); this is the network it describes: Input (n_time, None, None,
    n_channels) -> build_lstm_layers -> Conv2D(n_classes, [1, 1]) -> activation (default layers.ReLU(max_value=2.0))."""

    def __init__(self, n_channels, n_classes, n_time, activation='relu', max_value=2.0, dropout=None, seed=None):
        rng = np.random.default_rng(seed if seed is not None else mt._RNG.integers(1 << 31))
        self.P = _Params()
        self.n_channels, self.n_classes, self.n_time = n_channels, n_classes, n_time
        self.layers_ = LSTMLayers(self.P, rng, n_channels, dropout=dropout)
        self._no_graph = dropout is not None           # (a captured step would replay the mask drawn at capture time)
        self.dense = Dense1x1(self.P, rng, 'conv2d', [self.layers_.F], n_classes, activation, max_value)
        self._finish()

    def _forward(self, x, training):
        xt, (B, T, H, W) = _ingest_seq(x, ops.rup(self.n_channels, 16), self.dtype_code)
        if T != self.n_time:
            raise ValueError(f'model was built for {self.n_time} time steps, got {T}')
        feats = self.layers_.forward(Act(xt, self.n_channels), T, B, training, self.dtype_code)
        return self.dense.forward([(feats, False)])

    def predict(self, x, batch_size=None, verbose=0, **kw):
        n = x.shape[0]
        bs = batch_size or 32
        outs = [self._forward(x[i:i + bs], False).cpu().numpy() for i in range(0, n, bs)]
        return np.concatenate(outs, 0)

    def _eval_outputs(self, xb):
        return [(self._forward(xb, False), 'linear', lambda t: t)]

    def train_on_batch(self, x, y, sample_weight=None):
        if sample_weight is not None:
            raise NotImplementedError('sample_weight: the fused device losses have no per-sample weights')
        if self._loss is None:
            raise RuntimeError('compile() the model before fit/train')
        xd = (x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))).to(_dev(), torch.float32).contiguous()
        yd = (y if isinstance(y, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(y, dtype=np.float32))).to(_dev(), torch.float32).contiguous()
        return float(self._graphed_step(self._train_device, (xd, yd)).item())

    def _train_device(self, xd, yd):
        self.P.grad.zero_()
        out = self._forward(xd, True)
        loss, dout = self._loss_grad(out, yd, 'linear')
        (da,) = self.dense.backward(dout, need_dx=(True,))
        self.layers_.backward(da)
        self._adam(set_lr=False)
        return loss


def get_lstm_model(n_channels, n_classes, n_time, optim=None, metrics=None, loss=None, activation='relu', dropout=None, max_value=2.0):
    """utils/model_tools.py:773-808; compiled when `optim` and `loss` are given"""
    m = LSTMModel(n_channels, n_classes, n_time, activation=activation, max_value=max_value, dropout=dropout)
    if optim is not None and loss is not None:
        m.compile(optimizer=optim, loss=loss, metrics=metrics)
    return m


class HybridModel(_SeqModelBase):
    """get_hybrid_model (utils/model_tools.py:874-920): a U-Net branch (build_unet_layers, factors [3, 2, 2, 2]) and an LSTM branch
    (build_lstm_layers), each through Conv2D(n_classes, [1, 1], relu); the LSTM map is brought to the U-Net grid with
    tf.image.resize(..., 'nearest'), concatenated [lstm, unet] and classified by Conv2D(n_classes, [1, 1], softmax) 'probabilities'.
    Inputs [unet_input (B, H, W, C), lstm_input (B, T, h, w, c)].  The U-Net branch runs on the static engine (a model_tools.Model
    with a linear 1x1 head = the pre-activation of its `relu` dense layer), the rest on the tape."""

    def __init__(self, unet_dim, lstm_dim, n_classes, filters=(32, 64, 128, 256), factors=(3, 2, 2, 2), dropout=None, seed=None):
        rng = np.random.default_rng(seed if seed is not None else mt._RNG.integers(1 << 31))
        self.unet_dim, self.lstm_dim, self.n_classes = tuple(unet_dim), tuple(lstm_dim), n_classes
        inp = mt.Input(shape=(None, None, unet_dim[-1]))
        dec = mt.build_unet_layers(inp, list(filters), list(factors), dropout=dropout)
        if dropout is not None:                       # layers.SpatialDropout2D(dropout)(unet_output), :899-900
            dec = mt._dropout(dec, dropout, True)
        self.unet = mt.Model(inputs=inp, outputs=mt._Head(n_classes, 'linear', 'zeros', 'unet_dense')(dec))
        self.P = _Params()
        self.lstm = LSTMLayers(self.P, rng, lstm_dim[-1], dropout=dropout)
        self.lstm_drop = Dropout(dropout, True, rng.integers(1 << 62)) if dropout else None     # SpatialDropout2D on lstm_output, :904-905
        self.lstm_dense = Dense1x1(self.P, rng, 'lstm_dense', [self.lstm.F], n_classes, 'relu')
        self.fusion = Dense1x1(self.P, rng, 'probabilities', [n_classes, n_classes], n_classes, 'softmax')
        self._finish()

    def compile(self, optimizer='adam', loss=None, metrics=None, **kw):
        super().compile(optimizer, loss, metrics)
        self.unet.optimizer = self.optimizer
        self.unet.runtime.adam_state[0:1].fill_(self.optimizer._lr)

    def _forward(self, xs, training):
        xu, xl = xs
        self.unet.compute_dtype = self.compute_dtype
        n, h, w, _ = self.unet._shape_of(xu)
        plan = self.unet.runtime.plan(n, h, w, training)
        self.unet._stage_x(plan, xu)
        if training:
            plan.step_count += 1
        plan.run_forward(ops.stream_ptr())
        zu = plan.outputs[self.unet.outputs[0].id]                    # float32 (n, h, w, k): pre-activation of the U-Net's relu dense layer
        xt, (B, T, hh, ww) = _ingest_seq(xl, ops.rup(self.lstm_dim[-1], 16), self.dtype_code)
        feats = self.lstm.forward(Act(xt, self.lstm_dim[-1]), T, B, training, self.dtype_code)
        if self.lstm_drop is not None:
            feats = self.lstm_drop.forward(feats, training)
        zl = self.lstm_dense.forward([(feats, False)])                # float32 (B, hh, ww, k), already through its ReLU
        probs, classes = self.fusion.forward([(Act(zl, self.n_classes), True), (Act(zu, self.n_classes, relu=True), False)], out_hw=(h, w),
                                             want_classes=True)
        self._plan, self._zu = plan, zu
        return probs, classes

    def predict(self, x, batch_size=None, verbose=0, **kw):
        n = x[0].shape[0]
        bs = batch_size or 8
        outs = [self._forward([a[i:i + bs] for a in x], False)[0].cpu().numpy() for i in range(0, n, bs)]
        return np.concatenate(outs, 0)

    output_names = ('probabilities',)

    def _eval_outputs(self, xb):
        return [(self._forward(xb, False)[0], 'softmax', lambda t: t)]

    def train_on_batch(self, x, y, sample_weight=None):
        if sample_weight is not None:
            raise NotImplementedError('sample_weight: the fused device losses have no per-sample weights')
        if self._loss is None:
            raise RuntimeError('compile() the model before fit/train')
        rt = self.unet.runtime
        rt.ensure_adam()
        self.P.grad.zero_(); rt.gflat.zero_()
        probs, _ = self._forward(x, True)
        loss, dlog = self._loss_grad(probs, y, 'softmax')
        dzl, dau = self.fusion.backward(dlog, need_dx=(True, True))
        (dfeat,) = self.lstm_dense.backward(dzl, need_dx=(True,))
        if self.lstm_drop is not None:
            dfeat = self.lstm_drop.backward(dfeat)
        self.lstm.backward(dfeat)
        # U-Net branch: d relu(z_u) -> dz_u (mask in place) -> the plan's logit gradient -> its backward pass + Adam + repack
        check(lib.satcv_relu_bwd(self._zu.data_ptr(), dau.data_ptr(), dau.numel(), F32, ops.stream_ptr()))
        plan = self._plan
        plan.dlogits.view(-1).copy_(dau.view(-1))
        plan.run_backward(ops.stream_ptr())
        opt = self.optimizer
        rt.adam_state[0:1].fill_(opt._lr)
        check(lib.satcv_adam_step(rt.pflat.data_ptr(), rt.gflat.data_ptr(), rt.adam_m.data_ptr(), rt.adam_v.data_ptr(), rt.pflat.numel(), opt.beta_1,
                                  opt.beta_2, opt.epsilon, rt.adam_state.data_ptr(), None, ops.stream_ptr()))
        rt.repack()
        self.unet._weights_version = getattr(self.unet, '_weights_version', 0) + 1
        self._adam()
        return float(loss.item())


def get_hybrid_model(unet_dim, lstm_dim, n_classes, filters=[32, 64, 128, 256], factors=[3, 2, 2, 2], dropout=None, compile_model=False,
                     optim=None, metrics=None, loss=None):
    """utils/model_tools.py:874-920"""
    m = HybridModel(unet_dim, lstm_dim, n_classes, filters, factors, dropout)
    if compile_model:
        m.compile(optimizer=optim, loss=loss, metrics=metrics)
    return m


class LSTMAutoencoder(_SeqModelBase):
    """get_lstm_autoencoder (utils/model_tools.py:810-872): inputs [timeseries (B, T, H, W, C), sincos (B, H, W, 2)];
    encoded = build_lstm_layers2(timeseries); branch 1: encoded repeated n_time times -> ConvLSTM2D(32, return_sequences) 'lstm_decoder'
    -> TimeDistributed(Conv2D(n_classes, 1x1)) 'temporal_dense' -> activation; branch 2: concat([encoded, sincos]) -> Conv2D(n_classes,
    1x1) 'single_dense' -> activation.  Outputs [temporal (B, T, H, W, n_classes), single (B, H, W, n_classes)].  The reference leaves
    the model uncompiled; train_on_batch applies the compiled loss to BOTH outputs and adds the two (Keras' default for a list of outputs)."""

    def __init__(self, n_channels, n_time, n_classes, activation='relu', max_value=2.0, seed=None):
        rng = np.random.default_rng(seed if seed is not None else mt._RNG.integers(1 << 31))
        self.P = _Params()
        self.n_channels, self.n_time, self.n_classes = n_channels, n_time, n_classes
        self.enc = LSTMLayers2(self.P, rng, n_channels)
        self.dec = ConvLSTM2D(self.P, rng, 'lstm_decoder', self.enc.F, 32, 1, True)
        self.temporal = Dense1x1(self.P, rng, 'temporal_dense', [32], n_classes, activation, max_value)
        self.single = Dense1x1(self.P, rng, 'single_dense', [self.enc.F, 2], n_classes, activation, max_value)
        self._finish()

    def _forward(self, xs, training):
        x, sincos = xs
        xt, (B, T, H, W) = _ingest_seq(x, ops.rup(self.n_channels, 16), self.dtype_code)
        if T != self.n_time:
            raise ValueError(f'model was built for {self.n_time} time steps, got {T}')
        enc = self.enc.forward(Act(xt, self.n_channels), T, B, training, self.dtype_code)
        dseq, _, _ = self.dec.forward(enc, T, B, training, self.dtype_code, want_stats=False, repeat=True)
        tout = self.temporal.forward([(dseq, False)])                         # (T * B, H, W, k), time-major
        sc = sincos if isinstance(sincos, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(sincos, dtype=np.float32))
        sc = sc.to(_dev(), torch.float32).contiguous()
        sout = self.single.forward([(enc, False), (Act(sc, 2), False)])
        self._shape = (B, T, H, W)
        return tout, sout

    def predict(self, x, batch_size=None, verbose=0, **kw):
        tout, sout = self._forward(x, False)
        B, T, H, W = self._shape
        return [tout.view(T, B, H, W, -1).permute(1, 0, 2, 3, 4).contiguous().cpu().numpy(), sout.cpu().numpy()]

    output_names = ('temporal', 'single')

    def _eval_outputs(self, xb):
        tout, sout = self._forward(xb, False)
        B, T, H, W = self._shape
        return [(tout, 'linear', lambda t: t.permute(1, 0, 2, 3, 4).contiguous().view(T * B, H, W, -1)), (sout, 'linear', lambda t: t)]

    def train_on_batch(self, x, y, sample_weight=None):
        if sample_weight is not None and any(w is not None for w in (sample_weight if isinstance(sample_weight, (list, tuple)) else [sample_weight])):
            raise NotImplementedError('sample_weight: the fused device losses have no per-sample weights (LSTMAutoencoderGenerator(sample_weights=False))')
        if self._loss is None:
            raise RuntimeError('compile() the model before fit/train')
        tensors = tuple(_dev_f32(a) for a in x) + tuple(_dev_f32(a) for a in y)
        return float(self._graphed_step(self._train_device, tensors).item())

    def _train_device(self, xs, sincos, ty, sy):
        self.P.grad.zero_()
        tout, sout = self._forward([xs, sincos], True)
        B, T, H, W = self._shape
        ty = ty.permute(1, 0, 2, 3, 4).contiguous().view(T * B, H, W, -1)
        l1, d1 = self._loss_grad(tout, ty, 'linear')
        l2, d2 = self._loss_grad(sout, sy, 'linear')
        (ddec,) = self.temporal.backward(d1, need_dx=(True,))
        denc1 = self.dec.backward(ddec, need_dx=True)                          # gradient of `encoded` through the repeated sequence
        denc2, _ = self.single.backward(d2, need_dx=(True, False))
        check(lib.satcv_add_act(denc1.data_ptr(), None, None, denc2.data_ptr(), None, None, 0, denc1.data_ptr(), B * H * W, denc1.shape[-1],
                                ops.DTYPE_CODE[denc1.dtype], ops.stream_ptr()))
        self.enc.backward(denc1)
        self._adam(set_lr=False)
        return l1 + l2


def get_lstm_autoencoder(n_channels, n_time, n_classes, activation='relu', compile=False, optim=None, metrics=None, loss=None, max_value=2.0):
    """utils/model_tools.py:810-872"""
    m = LSTMAutoencoder(n_channels, n_time, n_classes, activation, max_value)
    if compile and optim is not None and loss is not None:
        m.compile(optimizer=optim, loss=loss, metrics=metrics)
    return m


def build_lstm_layers(params, rng, n_channels, return_sequences=False, dropout=None):
    """utils/model_tools.py:666-717 (layer stack object; the model builders above call it)"""
    return LSTMLayers(params, rng, n_channels, return_sequences=return_sequences, dropout=dropout)


def build_lstm_layers2(params, rng, n_channels, return_sequences=False, return_state=False, dropout=None):
    """utils/model_tools.py:719-771"""
    if return_sequences:
        raise NotImplementedError('build_lstm_layers2 is lowered as the reference calls it (return_sequences=False)')
    return LSTMLayers2(params, rng, n_channels, dropout=dropout)


# ------------------------------------------------------------------------------------------------ hierarchical model (ACNN + LSTM)
class ConvBN:
    """layers.Conv2D(filters, 3, padding='same', dilation_rate=d) -> layers.BatchNormalization(): one implicit-GEMM launch with the
    BatchNorm statistics in its epilogue; the normalisation (+ ReLU) stays pending for the consumers, as in the U-Net engine"""

    def __init__(self, params, rng, conv_name, bn_name, cin, cout, dilation=1, k=3):
        self.P, self.name, self.cin, self.cout, self.dil, self.k = params, conv_name, cin, cout, dilation, k
        params.add(f'{conv_name}/kernel', (k, k, cin, cout), _glorot(rng, (k, k, cin, cout), k * k * cin, k * k * cout))
        params.add(f'{conv_name}/bias', (cout,), lambda: np.zeros(cout, np.float32))
        self.bn = BatchNorm(params, bn_name, cout)

    def forward(self, x, training, dtype, relu):
        P = self.P
        self.wf, self.wd = P.get_packed(f'{self.name}/kernel', ops.rup(self.cin, 16), dtype)
        cp = ops.rup(self.cout, 16)
        stats = ops.new_stats(cp, x.t.device) if training else None
        y = ops.conv2d(x.t, self.wf, self.cout, kh=self.k, kw=self.k, dil=self.dil, bias=P.p(f'{self.name}/bias'), in_scale=x.scale, in_shift=x.shift,
                       in_relu=x.relu, stats=stats)
        self.x = x
        n, h, w, _ = y.shape
        return self.bn.forward(Act(y, self.cout), stats, n * h * w, training, relu=relu)

    def backward(self, da, need_dx=True):
        """da: gradient of the normalised (+ ReLU) output.  (The bias in front of a training-mode BatchNormalization has an exactly zero
        gradient: it is left at 0, as in the engine.)"""
        dy = self.bn.backward(da)
        x = self.x
        ops.conv2d_wgrad(x.t, dy, self.cin, self.cout, kh=self.k, kw=self.k, dil=self.dil, in_scale=x.scale, in_shift=x.shift, in_relu=x.relu,
                         dw=self.P.g(f'{self.name}/kernel'))
        return ops.conv2d_dgrad(dy, self.wd, self.cin, kh=self.k, kw=self.k, dil=self.dil) if need_dx else None


def _add_grads(a, b):
    """a += b (two gradient tensors of one activation with several consumers)"""
    n = a.numel() // a.shape[-1]
    check(lib.satcv_add_act(a.data_ptr(), None, None, b.data_ptr(), None, None, 0, a.data_ptr(), n, a.shape[-1], ops.DTYPE_CODE[a.dtype], ops.stream_ptr()))
    return a


class ACNN2Trunk:
    """build_acnn_layers2 (utils/model_tools.py:941-979): n_blocks x [Conv{l}_1 -> bn{l}_1 -> (+ previous sum, from block 1 on) -> ReLU{l}_1 ;
    DilateConv{l}_2 (rate 3) -> bn{l}_2 -> ReLU{l}_2].  forward returns the ReLU{l}_2 activations of every block (pending BatchNorm + ReLU);
    backward takes a gradient per tapped block."""

    def __init__(self, params, rng, nchannels, n_blocks=16, feature_num=16):
        self.n, self.F = n_blocks, feature_num
        self.c1, self.c2 = [], []
        for l in range(n_blocks):
            self.c1.append(ConvBN(params, rng, f'Conv{l}_1', f'bn{l}_1', nchannels if l == 0 else feature_num, feature_num))
            self.c2.append(ConvBN(params, rng, f'DilateConv{l}_2', f'bn{l}_2', feature_num, feature_num, dilation=3))

    def forward(self, x, training, dtype):
        feats, self.sums, fa = [], [], None
        f = x
        for l in range(self.n):
            normed = self.c1[l].forward(f, training, dtype, relu=(l == 0))
            y = normed.t
            n, h, w, cp = y.shape
            out = torch.empty_like(y)
            if l == 0:          # ReLU0_1(bn0_1(conv)): materialised, the next block adds it
                check(lib.satcv_bn_relu_pool(y.data_ptr(), normed.scale.data_ptr(), normed.shift.data_ptr(), out.data_ptr(), cp, None, None, 0, n, h, w, cp, 1,
                                             ops.DTYPE_CODE[y.dtype], ops.stream_ptr()))
            else:               # ReLU{l}_1(bn{l}_1(conv) + previous sum)
                check(lib.satcv_add_act(y.data_ptr(), normed.scale.data_ptr(), normed.shift.data_ptr(), fa.data_ptr(), None, None, 1, out.data_ptr(),
                                        n * h * w, cp, ops.DTYPE_CODE[y.dtype], ops.stream_ptr()))
            fa = out
            self.sums.append(out)
            f = self.c2[l].forward(Act(fa, self.F), training, dtype, relu=True)
            feats.append(f)
        return feats

    def backward(self, dfeats):
        """dfeats: {block index: gradient of its ReLU{l}_2 activation} (storage type); gradients of deeper consumers are added here"""
        dfa_next, dfeat = None, None            # gradient of the sum of block l + 1 w.r.t. the sum of block l; gradient of feature l from Conv{l+1}_1
        for l in range(self.n - 1, -1, -1):
            df = dfeats.get(l)
            if dfeat is not None:
                df = dfeat if df is None else _add_grads(df, dfeat)
            dsum = self.c2[l].backward(df) if df is not None else None           # gradient of the sum fa_l through DilateConv{l}_2
            if dfa_next is not None:
                dsum = dfa_next if dsum is None else _add_grads(dsum, dfa_next)
            if dsum is None:
                dfeat = dfa_next = None
                continue
            g = dsum                                  # through ReLU{l}_1: mask by the materialised sum; serves both addends
            check(lib.satcv_relu_bwd(self.sums[l].data_ptr(), g.data_ptr(), g.numel(), ops.DTYPE_CODE[g.dtype], ops.stream_ptr()))
            if l == 0:
                # block 0's ReLU belongs to its BatchNorm: the gradient was masked above, the BatchNorm backward runs in linear mode
                self.c1[0].bn.ctx['relu'] = False
                self.c1[0].backward(g, need_dx=False)
                dfeat = dfa_next = None
            else:
                dfeat = self.c1[l].backward(g, need_dx=True)               # gradient of feature l - 1 (the ReLU{l-1}_2 activation)
                dfa_next = g


class HierarchicalModel(_SeqModelBase):
    """get_hierarchical_model (utils/model_tools.py:1016-1060): get_acnn_model2's trunk with three softmax heads -- 'sub_probs' on the
    middle block's ReLU{(depth-1)//2}_2, 'acnn_probs' on the last block's, 'lstm_probs' on concat([nearest-resized build_lstm_layers output,
    last block's activation]).  Inputs [acnn_input (B, H, W, C), lstm_input (B, T, h, w, c)], outputs [sub_probs, acnn_probs, lstm_probs]."""

    def __init__(self, nclasses, acnn_nclasses, acnn_sub_nclasses, acnn_dim, lstm_dim, nfilters, depth, seed=None):
        rng = np.random.default_rng(seed if seed is not None else mt._RNG.integers(1 << 31))
        self.P = _Params()
        self.acnn_dim, self.lstm_dim, self.depth, self.mid = tuple(acnn_dim), tuple(lstm_dim), depth, (depth - 1) // 2
        self.trunk = ACNN2Trunk(self.P, rng, acnn_dim[-1], depth, nfilters)
        self.lstm = LSTMLayers(self.P, rng, lstm_dim[-1])
        self.sub = Dense1x1(self.P, rng, 'sub_probs', [nfilters], acnn_sub_nclasses, 'softmax')
        self.acnn = Dense1x1(self.P, rng, 'acnn_probs', [nfilters], acnn_nclasses, 'softmax')
        self.fuse = Dense1x1(self.P, rng, 'lstm_probs', [self.lstm.F, nfilters], nclasses, 'softmax')
        self._finish()

    def _forward(self, xs, training):
        xa, xl = xs
        xa = xa if isinstance(xa, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(xa, dtype=np.float32))
        x = ops.ingest_nhwc(xa.to(_dev(), torch.float32), ops.rup(self.acnn_dim[-1], 16), self.dtype_code)
        feats = self.trunk.forward(Act(x, self.acnn_dim[-1]), training, self.dtype_code)
        xt, (B, T, hh, ww) = _ingest_seq(xl, ops.rup(self.lstm_dim[-1], 16), self.dtype_code)
        lf = self.lstm.forward(Act(xt, self.lstm_dim[-1]), T, B, training, self.dtype_code)
        H, W = x.shape[1], x.shape[2]
        return [self.sub.forward([(feats[self.mid], False)]), self.acnn.forward([(feats[-1], False)]),
                self.fuse.forward([(lf, True), (feats[-1], False)], out_hw=(H, W))]

    def predict(self, x, batch_size=None, verbose=0, **kw):
        return [o.cpu().numpy() for o in self._forward(x, False)]

    output_names = ('sub_probs', 'acnn_probs', 'lstm_probs')

    def _eval_outputs(self, xb):
        return [(o, 'softmax', lambda t: t) for o in self._forward(xb, False)]

    def train_on_batch(self, x, y, sample_weight=None):
        if sample_weight is not None:
            raise NotImplementedError('sample_weight: the fused device losses have no per-sample weights')
        if self._loss is None:
            raise RuntimeError('compile() the model before fit/train')
        tensors = tuple(_dev_f32(a) for a in x) + tuple(_dev_f32(a) for a in y)
        return float(self._graphed_step(self._train_device, tensors).item())

    def _train_device(self, xa, xl, y0, y1, y2):
        self.P.grad.zero_()
        outs = self._forward([xa, xl], True)
        total, dl = None, []
        for o, yy in zip(outs, (y0, y1, y2)):
            loss, d = self._loss_grad(o, yy, 'softmax')
            total = loss if total is None else total + loss
            dl.append(d)
        (dmid,) = self.sub.backward(dl[0], need_dx=(True,))
        (dlast,) = self.acnn.backward(dl[1], need_dx=(True,))
        dlf, dlast2 = self.fuse.backward(dl[2], need_dx=(True, True))
        self.lstm.backward(dlf)
        dfe = {self.depth - 1: _add_grads(dlast, dlast2)}
        if self.mid == self.depth - 1:
            _add_grads(dfe[self.mid], dmid)
        else:
            dfe[self.mid] = dmid
        self.trunk.backward(dfe)
        self._adam(set_lr=False)
        return total


def get_hierarchical_model(nclasses, acnn_nclasses, acnn_sub_nclasses, acnn_dim, lstm_dim, nfilters, depth):
    """utils/model_tools.py:1016-1060"""
    return HierarchicalModel(nclasses, acnn_nclasses, acnn_sub_nclasses, acnn_dim, lstm_dim, nfilters, depth)
