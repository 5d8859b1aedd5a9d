"""FP8 (OCP e4m3fn) inference of the plain U-Net -- BASELINE config 5 ("1024x1024 large-tile sliding-window inference,
fp8 MFMA conv path").  The reference has no reduced-precision path (all arithmetic fp32); this is the
north-star's extension and is checked against the fp32 path of this build / the oracle by IoU.

Folded graph (inference only, `utils/model_tools.py:174-415` semantics):
  * every tensor in HBM is the ACTIVATED output relu(bn(conv)) stored as fp8 with one scale per tensor,
    value = q * fp8;  q comes from a calibration run of the bf16/fp32 plan (`Model.enable_fp8_inference`);
  * weights are fp8 with one scale per output channel;
  * BatchNorm (moving statistics), conv bias and all quantisation scales are folded into ONE per-channel multiplier and
    bias applied to the fp32 accumulator in the conv epilogue (`satcv_conv_desc.out_scale` / `bias`):
        out8 = Q( relu( acc * (q_in*w_scale*bn_scale/q_out) + (bn_scale*b + bn_shift)/q_out ) )
  * max-pool runs on the fp8 values directly (monotone); concat([skip, up]) -> BN -> ReLU is materialised: the
    transposed conv writes its half through its epilogue, the skip half is re-quantised by `satcv_affine_requant`.
MFMA: v_mfma_f32_32x32x16_fp8_fp8 (bf16 rate, half the operand bytes in HBM and LDS).
"""
import ctypes as C

import torch

from . import ops
import os

from ._lib import lib, check, FP8, FP8X, BF16

# layers whose input channels are a multiple of 64 run on the block-scaled K=64 MFMA (2x the bf16 rate)
USE_SCALED_MFMA = os.environ.get('SATCV_FP8_SCALED', '1') != '0'
FUSE_POOL = os.environ.get('SATCV_FUSE_POOL', '1') != '0'
HYBRID = os.environ.get('SATCV_FP8_HYBRID', '1') != '0'       # fp8 plans keep the full- and half-resolution levels in bf16 (Fp8Plan)
THIN_FP8 = os.environ.get('SATCV_FP8_THIN', '1') != '0'       # fp8 levels: thin 3x3 convs on the persistent thin-layer kernel, two-source concatenations

E4M3_MAX = 448.0
BN_EPS = 1e-3


def _rup(a, b):
    return (a + b - 1) // b * b


def _relu_amax(t, scale, shift):
    return float((t.float() * scale + shift).clamp_min_(0).amax())


def calibrate(model, x):
    """Per-tensor activation maxima of the folded graph from one run of the regular inference plan on `x`
    (ndarray / device tensor NHWC).  Returns {tensor id: q} with q = amax / 448."""
    n, h, w, _ = model._shape_of(x)
    plan = model._head_plan(n, h, w, False)
    model._stage_x(plan, x)
    plan.run_forward(ops.stream_ptr())
    torch.cuda.synchronize()
    q = {}

    def put(tid, amax):
        q[tid] = max(amax, 1e-12) / E4M3_MAX
    for node in model.nodes:
        cx = plan.node_ctx.get(id(node))
        if node.op == 'input':
            # (a resident float32 batch is read in place by the ingest kernel -- Model._stage_x sets plan.x_src and holds the tensor --
            #  so the staging tensor then holds zeros or an older batch: take the maximum of the tensor that was actually ingested)
            tid = node.outputs[0].id
            held = getattr(plan, '_x_hold', {}).get(tid) if plan.x_src.get(tid) else None
            put(tid, float((held if held is not None else plan.x_by_tid[tid]).abs().amax()))
        elif node.op == 'cba':
            y, aff, c = cx['y'], cx['aff'], cx['cout']
            if cx['yoff'] != 0 or cx['ldy'] != c:
                raise NotImplementedError('fp8 inference: ASPP / slot-mode concatenations are not lowered')
            put(node.outputs[0].id, _relu_amax(y, aff['scale'], aff['shift']))
        elif node.op == 'concat_bn_relu':
            ra, rb, aff, ca = cx['ra'], cx['rb'], cx['aff'], cx['ca']
            a = _relu_amax(ra.srcs[0][0], aff['scale'][:ca], aff['shift'][:ca])
            b = _relu_amax(rb.srcs[0][0], aff['scale'][ca:], aff['shift'][ca:])
            put(node.outputs[0].id, max(a, b))
    return q


def _single(v):
    if len(v) > 5:
        raise NotImplementedError('folded inference: only a conv block may consume the (skip, up) pair')
    return v


class _Ones:
    """scale table of the unquantised (bf16) folded graph"""

    def __getitem__(self, k):
        return 1.0


class Fp8Plan:
    """Static launch list of the folded fp8 forward for one (n, h, w)."""

    def __init__(self, model, n, h, w, q, first_bf16=True, store=FP8, hybrid=None):
        # store = BF16: the same folded graph with bf16 tensors and no quantisation (every scale 1) -- BatchNorm in the conv epilogues
        # instead of the consumers' loaders, activations written once
        # hybrid (fp8 store only, default on: SATCV_FP8_HYBRID=0 turns it off): the full- and half-resolution levels keep bf16 tensors and
        # run on the bf16 kernels, the levels below run in fp8.  The thin high-resolution layers are HBM-bound and have a persistent
        # weights-stationary kernel, fused pooling and a two-source loader only in bf16 (their fp8 forms went through the general tiled
        # kernel plus a requantisation pass per concatenation and were SLOWER than bf16: 1.32 vs 0.83 ms per batch of 64); the deep
        # layers are where e4m3 pays, on the block-scaled K = 64 MFMA at twice the bf16 rate.
        self.model, self.n, self.h, self.w, self.first_bf16, self.store = model, n, h, w, first_bf16, store
        self.hybrid = (HYBRID if hybrid is None else bool(hybrid)) and store == FP8
        self.q = q if store == FP8 else _Ones()
        self.tdt = torch.bfloat16 if store == BF16 else torch.float8_e4m3fn
        self.rt = model.runtime
        self.dev = self.rt.dev
        self.fwd, self.keep, self.outputs, self.x_by_tid = [], [], {}, {}
        self.esz = 2 if self.store == BF16 else 1
        self._build()

    def _hi(self, hh, ww):
        """True where a map of hh x ww pixels is stored in bf16: everywhere in the unquantised graph, the full- and half-resolution levels
        of the hybrid one."""
        return self.store == BF16 or (self.hybrid and hh * ww * 4 >= self.h * self.w)

    def _z(self, *shape, dtype=None):
        dtype = dtype or self.tdt
        t = torch.zeros(*shape, dtype=torch.uint8 if dtype == torch.float8_e4m3fn else dtype, device=self.dev)
        self.keep.append(t)
        return t

    def _f32(self, t):
        t = t.to(torch.float32).contiguous()
        self.keep.append(t)
        return t

    def _bn(self, name):
        rt = self.rt
        g, b = rt.get_param(name + '/gamma'), rt.get_param(name + '/beta')
        mm, mv = rt.get_param(name + '/moving_mean'), rt.get_param(name + '/moving_var')
        s = g / torch.sqrt(mv + BN_EPS)
        return s, b - mm * s

    def _pack(self, kernel, cin_pad, transposed, b16=False, thin=False):
        """operand image + per-output-channel scale of a Keras kernel: e4m3 with one scale per output channel, or bf16 (scale 1).
        thin: plain e4m3 items even where Cin % 64 == 0 -- the persistent thin-layer kernel's form (its layers are HBM-bound: the
        block-scaled MFMA's rate buys nothing there, the kernel's fused pooling / two-source loader do)."""
        if self.store == BF16 or b16:
            fwd, _ = ops.pack_weights(kernel.contiguous(), cin_pad, BF16, transposed=transposed, want_dgrad=False)
            self.keep.append(fwd)
            nout = kernel.shape[2] if transposed else kernel.shape[3]
            return fwd, torch.ones(nout, device=self.dev), BF16
        if transposed:                                  # (f, f, cout, cin)
            amax = kernel.abs().amax(dim=(0, 1, 3))
            wscale = amax.clamp_min(1e-12) / E4M3_MAX
            kq = kernel / wscale.view(1, 1, -1, 1)
        else:                                           # (kh, kw, cin, cout)
            amax = kernel.abs().amax(dim=(0, 1, 2))
            wscale = amax.clamp_min(1e-12) / E4M3_MAX
            kq = kernel / wscale.view(1, 1, 1, -1)
        dt = FP8X if (USE_SCALED_MFMA and cin_pad % 64 == 0 and not thin) else FP8
        fwd, _ = ops.pack_weights(kq.contiguous(), cin_pad, dt, transposed=transposed, want_dgrad=False)
        self.keep.append(fwd)
        return fwd, wscale, dt

    def _conv(self, dtype=None, **kw):
        dtype = dtype if dtype is not None else self.store
        d = ops.make_conv_desc(dtype=dtype, out_relu=1, **kw)
        self.keep.append(d)
        self.fwd.append(lambda st, d=d: check(lib.satcv_conv2d_igemm(C.byref(d), st)))

    def _requant(self, src, c, npix, scale, dt_in, dt_out, dst):
        """dst = Q(scale * src) channel-wise (a tensor changing storage between the bf16 and the fp8 levels of the hybrid graph)."""
        sc = self._f32(torch.full((c,), float(scale), device=self.dev))
        sh = self._f32(torch.zeros(c, device=self.dev))
        self.fwd.append(lambda st, src=src, dst=dst, sc=sc, sh=sh, c=c, npix=npix: check(
            lib.satcv_affine_requant(src.data_ptr(), c, sc.data_ptr(), sh.data_ptr(), 0, dst.data_ptr(), c, npix, c, dt_in, dt_out, st)))

    def _build(self):
        m, rt, n, q = self.model, self.rt, self.n, self.q
        B16 = torch.bfloat16
        consumers = {}
        for node in m.nodes:
            for t in node.inputs:
                consumers.setdefault(t.id, []).append(node)
        vals = {}                                        # tensor id -> (tensor [uint8 viewed as fp8, or bf16], channels, h, w, q [1.0 for bf16])
        prepooled = {}                                   # conv output tensor id -> its max-pooled tensor written by the conv epilogue
        cats = {}                                        # concat_bn_relu node id -> (cat tensor, ca, cb, q_cat, bn scale, bn shift)
        for node in m.nodes:
            op = node.op
            if op == 'input':
                t = node.outputs[0]
                cp = _rup(t.channels, 16)
                xin = self._z(n, self.h, self.w, t.channels, dtype=torch.float32)
                self.x_by_tid[t.id] = xin
                npix, cc = n * self.h * self.w, t.channels
                if self.first_bf16 or self.store == BF16 or self.hybrid:
                    # the input bands stay bf16 and the first conv block runs on the bf16 kernel: e4m3's 3 mantissa bits on
                    # the reflectances themselves were measured to flip ~1 % of confidently classified pixels
                    xb = self._z(n, self.h, self.w, cp, dtype=B16)
                    self.fwd.append(lambda st, xin=xin, xb=xb, npix=npix, cc=cc, cp=cp: check(
                        lib.satcv_ingest_nhwc(xin.data_ptr(), xb.data_ptr(), npix, cc, cp, BF16, st)))
                    vals[t.id] = (xb, cp, self.h, self.w, 1.0 if self._hi(self.h, self.w) else None)
                else:
                    x8 = self._z(n, self.h, self.w, cp)
                    inv = 1.0 / q[t.id]
                    self.fwd.append(lambda st, xin=xin, x8=x8, npix=npix, cc=cc, cp=cp, inv=inv: check(
                        lib.satcv_ingest_nhwc_scaled(xin.data_ptr(), x8.data_ptr(), npix, cc, cp, inv, FP8, st)))
                    vals[t.id] = (x8, cp, self.h, self.w, q[t.id])
            elif op == 'cba':
                tin, tout = node.inputs[0], node.outputs[0]
                x8, cin_s, hh, ww, qin = vals[tin.id][:5]
                dual = vals[tin.id][5] if len(vals[tin.id]) > 5 else None
                lay = node.layer
                if node.attrs.get('stride', 1) != 1 or not node.attrs.get('relu', True):
                    raise NotImplementedError('fp8 inference: strided / linear conv blocks are not lowered')
                kernel = rt.get_param(lay.name + '/kernel')
                cout = tout.channels
                if cout % 16 or cin_s != _rup(kernel.shape[2], 16):
                    raise NotImplementedError(f'{lay.name}: unsupported channel counts for the fp8 path')
                if qin is None:                       # bf16 input: bf16 conv (raw output), then BN + ReLU + quantisation in one pass
                    wb, _ = ops.pack_weights(kernel.contiguous(), cin_s, BF16, want_dgrad=False)
                    self.keep.append(wb)
                    yb = self._z(n, hh, ww, cout, dtype=B16)
                    d = ops.make_conv_desc(x0=x8.data_ptr(), c0=cin_s, w=wb.data_ptr(), bias=rt.pptr(lay.name + '/bias'), y=yb.data_ptr(), ldy=cout,
                                           n=n, h=hh, w_=ww, cout=cout, cout_pad=_rup(cout, 32), kh=node.attrs['k'], kw=node.attrs['k'],
                                           dil=node.attrs['dil'], dtype=BF16)
                    self.keep.append(d)
                    self.fwd.append(lambda st, d=d: check(lib.satcv_conv2d_igemm(C.byref(d), st)))
                    s, t_ = self._bn(lay.bn_name)
                    qo = q[tout.id]
                    rs, rsh = self._f32(s / qo), self._f32(t_ / qo)
                    y8 = self._z(n, hh, ww, cout)
                    npix = n * hh * ww
                    self.fwd.append(lambda st, yb=yb, y8=y8, rs=rs, rsh=rsh, cout=cout, npix=npix: check(
                        lib.satcv_affine_requant(yb.data_ptr(), cout, rs.data_ptr(), rsh.data_ptr(), 1, y8.data_ptr(), cout, npix, cout, BF16, FP8, st)))
                    vals[tout.id] = (y8, cout, hh, ww, qo)
                    continue
                b16 = self._hi(hh, ww)                   # this map's storage: bf16 (no quantisation, scale 1) or e4m3
                if b16 != (x8.dtype == B16):
                    raise NotImplementedError(f'{lay.name}: input and output of a conv block live on different sides of the bf16 / fp8 boundary')
                thin = (THIN_FP8 and not b16 and node.attrs['k'] == 3 and node.attrs['dil'] == 1 and cin_s in (32, 64) and cout in (32, 64)
                        and hh % 8 == 0 and ww % 32 == 0)
                if dual is not None and not b16 and not thin:
                    raise NotImplementedError(f'{lay.name}: a two-source fp8 conv outside the thin-layer kernel')
                w8, wscale, cdt = self._pack(kernel, cin_s, False, b16, thin)
                s, t_ = self._bn(lay.bn_name)
                qo = 1.0 if b16 else q[tout.id]
                oscale = self._f32(qin * wscale * s / qo)
                obias = self._f32((s * rt.get_param(lay.name + '/bias') + t_) / qo)
                ydt = B16 if b16 else None
                y8 = self._z(n, hh, ww, cout, dtype=ydt)
                k = node.attrs['k']
                # fuse the encoder block's MaxPooling2D into this conv's epilogue when the pipelined kernel takes the shape
                pool_kw = {}
                pnodes = [cn for cn in consumers.get(tout.id, []) if cn.op == 'pool']
                if len(pnodes) == 1 and hh % pnodes[0].attrs['f'] == 0 and ww % pnodes[0].attrs['f'] == 0 and FUSE_POOL:
                    fpool = pnodes[0].attrs['f']
                    p8 = self._z(n, hh // fpool, ww // fpool, cout, dtype=ydt)
                    pool_kw = dict(pool_y=p8.data_ptr(), pool_ld=cout, pool_f=fpool)
                src = dict(x0=x8.data_ptr(), c0=cin_s)
                if dual is not None:
                    src = dict(x0=x8.data_ptr(), c0=dual['c0'], x1=dual['x1'].data_ptr(), c1=dual['c1'], in_scale=dual['in_scale'].data_ptr(),
                               in_shift=dual['in_shift'].data_ptr(), in_relu=1)
                ckw = dict(w=w8.data_ptr(), bias=obias.data_ptr(), out_scale=oscale.data_ptr(), y=y8.data_ptr(), ldy=cout, n=n, h=hh, w_=ww, cout=cout,
                           cout_pad=_rup(cout, 32), kh=k, kw=k, dil=node.attrs['dil'], dtype=cdt, **src)
                if pool_kw:
                    probe = ops.make_conv_desc(out_relu=1, **ckw, **pool_kw)
                    if lib.satcv_conv2d_igemm_pipelined(C.byref(probe)):
                        ckw.update(pool_kw)
                        prepooled[tout.id] = (p8, cout, hh // fpool, ww // fpool, qo)
                self._conv(**ckw)
                if tout.id in prepooled and b16 and not self._hi(hh // fpool, ww // fpool):
                    # the pooled map belongs to the fp8 levels: quantise it (small: a quarter of the conv's output)
                    qp = q[tout.id]                      # (the scale calibrated for the un-pooled activation: max-pooling cannot exceed it)
                    pq = self._z(n, hh // fpool, ww // fpool, cout)
                    self._requant(p8, cout, n * (hh // fpool) * (ww // fpool), 1.0 / qp, BF16, FP8, pq)
                    prepooled[tout.id] = (pq, cout, hh // fpool, ww // fpool, qp)
                vals[tout.id] = (y8, cout, hh, ww, qo)
            elif op == 'pool':
                tin, tout = node.inputs[0], node.outputs[0]
                if tin.id in prepooled:                  # already produced by the conv's epilogue
                    vals[tout.id] = prepooled[tin.id]
                    continue
                x8, c, hh, ww, qin = _single(vals[tin.id])
                f = node.attrs['f']
                if hh % f or ww % f:
                    raise ValueError(f'input {self.h}x{self.w} is not divisible by the model downsampling')
                b16 = x8.dtype == B16
                p8 = self._z(n, hh // f, ww // f, c, dtype=B16 if b16 else None)
                self.fwd.append(lambda st, x8=x8, p8=p8, hh=hh, ww=ww, c=c, f=f, dt_=BF16 if b16 else self.store: check(
                    lib.satcv_maxpool(x8.data_ptr(), p8.data_ptr(), n, hh, ww, c, f, f, 0, dt_, st)))
                if b16 and not self._hi(hh // f, ww // f):
                    qp = q[tin.id]
                    pq = self._z(n, hh // f, ww // f, c)
                    self._requant(p8, c, n * (hh // f) * (ww // f), 1.0 / qp, BF16, FP8, pq)
                    p8, qin = pq, qp
                vals[tout.id] = (p8, c, hh // f, ww // f, qin)
            elif op == 'dropout':
                vals[node.outputs[0].id] = vals[node.inputs[0].id]           # identity at inference
            elif op == 'convT':
                tin, tout = node.inputs[0], node.outputs[0]
                cons = consumers.get(tout.id, [])
                if len(cons) != 1 or cons[0].op != 'concat_bn_relu' or cons[0].inputs[1] is not tout:
                    raise NotImplementedError('fp8 inference: a transposed conv must feed concat([skip, up]) -> BN -> ReLU')
                cat = cons[0]
                x8, cin_s, hh, ww, qin = _single(vals[tin.id])
                lay, f = node.layer, node.attrs['f']
                ca, cb = cat.inputs[0].channels, tout.channels
                if cb % 16 or ca % 16:
                    raise NotImplementedError(f'{lay.name}: unsupported channel counts for the fp8 path')
                b16 = self._hi(hh * f, ww * f)           # storage of the up-sampled map (and of the skip it is concatenated with)
                if b16 and x8.dtype != B16:
                    # the decoder climbs from the fp8 levels to the bf16 ones: de-quantise this block's input (a quarter of its output)
                    xb = self._z(n, hh, ww, cin_s, dtype=B16)
                    self._requant(x8, cin_s, n * hh * ww, qin, FP8, BF16, xb)
                    x8, qin = xb, 1.0
                kernel = rt.get_param(lay.name + '/kernel')
                w8, wscale, cdt = self._pack(kernel, cin_s, True, b16)
                s0, t0 = self._bn(cat.layer.name)
                qc = 1.0 if b16 else q[cat.outputs[0].id]
                oscale = self._f32(qin * wscale * s0[ca:] / qc)
                obias = self._f32((s0[ca:] * rt.get_param(lay.name + '/bias') + t0[ca:]) / qc)
                two_src = (THIN_FP8 and not b16 and ca + cb in (32, 64) and (hh * f) % 8 == 0 and (ww * f) % 32 == 0 and
                           all(cn.op == 'cba' and cn.attrs['k'] == 3 and cn.attrs['dil'] == 1 and cn.outputs[0].channels in (32, 64)
                               for cn in consumers.get(cat.outputs[0].id, [])))
                if b16 or two_src:
                    # the concatenation is not materialised -- the up-sampled half goes to its own tensor and the consumer conv reads
                    # (skip, up) as two sources, with the skip half's BN + ReLU (and, in fp8, its requantisation to the concatenation's
                    # scale) applied in its loader
                    cat8, ybase, ldy = None, self._z(n, hh * f, ww * f, cb, dtype=B16 if b16 else None), cb
                    yptr = ybase.data_ptr()
                else:
                    cat8 = self._z(n, hh * f, ww * f, ca + cb)
                    ybase, ldy, yptr = cat8, ca + cb, cat8.data_ptr() + ca * self.esz
                self._conv(x0=x8.data_ptr(), c0=cin_s, w=w8.data_ptr(), bias=obias.data_ptr(), out_scale=oscale.data_ptr(),
                           y=yptr, ldy=ldy, n=n, h=hh, w_=ww, cout=f * f * cb, cout_pad=_rup(f * f * cb, 32),
                           kh=1, kw=1, dil=1, mode_out=1, f=f, cstat=cb, dtype=cdt)
                cats[id(cat)] = (cat8 if cat8 is not None else ybase, ca, cb, qc, s0, t0, hh * f, ww * f, b16, two_src)
            elif op == 'concat_bn_relu':
                ta, tout = node.inputs[0], node.outputs[0]
                cat8, ca, cb, qc, s0, t0, hh, ww, b16, two_src = cats[id(node)]
                a8, c, ha, wa, qa = _single(vals[ta.id])
                if c != ca or (ha, wa) != (hh, ww):
                    raise ValueError(f'concatenation of a {ha}x{wa}x{c} skip with a {hh}x{ww} up-sampled map')
                if b16 or two_src:
                    if b16 != (a8.dtype == B16):
                        raise NotImplementedError('an up-sampled map concatenated with a skip of the other storage type')
                    insc = self._f32(torch.cat([s0[:ca] * qa / qc, torch.ones(cb, device=self.dev)]))
                    insh = self._f32(torch.cat([t0[:ca] / qc, torch.zeros(cb, device=self.dev)]))      # identity on the (already activated) up half
                    vals[tout.id] = (a8, ca + cb, hh, ww, qc, dict(x1=cat8, c0=ca, c1=cb, in_scale=insc, in_shift=insh))
                    continue
                rs = self._f32(s0[:ca] * qa / qc)
                rsh = self._f32(t0[:ca] / qc)
                npix = n * hh * ww
                self.fwd.append(lambda st, a8=a8, cat8=cat8, rs=rs, rsh=rsh, ca=ca, ld=ca + cb, npix=npix: check(
                    lib.satcv_affine_requant(a8.data_ptr(), ca, rs.data_ptr(), rsh.data_ptr(), 1, cat8.data_ptr(), ld, npix, ca, self.store, self.store, st)))
                vals[tout.id] = (cat8, ca + cb, hh, ww, qc)
            elif op == 'head':
                tin, tout = node.inputs[0], node.outputs[0]
                x8, c, hh, ww, qin = _single(vals[tin.id])
                lay = node.layer
                ncls = tout.channels
                act = {'softmax': 0, 'sigmoid': 1, 'linear': 2}[node.attrs['activation']]
                probs = self._z(n, hh, ww, ncls, dtype=torch.float32)
                classes = self._z(*((n, hh, ww) if act != 1 else (n, hh, ww, ncls)), dtype=torch.int32)
                sc = self._f32(torch.full((c,), qin, device=self.dev))
                sh = self._f32(torch.zeros(c, device=self.dev))
                hd = ops.make_head_desc(x=x8.data_ptr(), ldx=c, cin=c, w=rt.pptr(lay.name + '/kernel'), b=rt.pptr(lay.name + '/bias'),
                                        ncls=ncls, activation=act, npix=n * hh * ww, dtype=BF16 if x8.dtype == B16 else self.store,
                                        in_scale=sc.data_ptr(), in_shift=sh.data_ptr(),
                                        thresh=node.attrs.get('thresh', 0.5), probs=probs.data_ptr(), classes=classes.data_ptr())
                self.keep.append(hd)
                self.fwd.append(lambda st, hd=hd: check(lib.satcv_head_fwd(C.byref(hd), st)))
                self.outputs[tout.id] = probs
                self._classes = classes
            elif op == 'classes':
                self.outputs[node.outputs[0].id] = self._classes
            else:
                raise NotImplementedError(f'fp8 inference: op {op} is not lowered (plain U-Net graphs only)')

    def run_forward(self, st):
        for f in self.fwd:
            f(st)
