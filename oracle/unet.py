"""NumPy restatement of the reference U-Net graph AS CODED, forward + backward + Adam.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED (Keras ops).

Follows /root/reference/utils/model_tools.py:
  conv_batch_act :174-186   Conv2D(same) -> BatchNormalization -> ReLU
  conv_block     :211-240   AS CODED: cba1(inputs) executed twice, cba2 never used
                            (single conv per level; BN moving stats updated twice
                            per training step -- SURVEY.md Appendix B Q1/Q2)
  encoder_block  :262-286   conv_block -> MaxPooling2D(f, strides=f); returns (pooled, encoded)
  decoder_block  :288-318   Conv2DTranspose(k=s=f) -> concat([skip, up]) -> BN -> ReLU
                            -> (Conv3x3 -> BN -> ReLU) x 2
  build_unet_layers :321-379, get_unet_model :394-415 (1x1 softmax head 'probs',
                            argmax int32 'classes')
  DilatedSpatialPyramidPooling :533-574 (ASPP, image-pooling branch disabled)
Adam follows the Keras formulation (epsilon outside the bias correction,
SURVEY.md Appendix A).

store_dtype='bfloat16' (tests only): the same float64 arithmetic, but every tensor the
device path keeps in bf16 storage is rounded to bf16 (round-to-nearest-even) at the point
where the device stores it: the input tile, the MFMA weight images,
every raw conv / transposed-conv output (BatchNorm statistics are then those of the STORED
values), every activated tensor a consumer stages, every gradient tensor (head dx, dy,
dx, dskip, du).  Accumulations stay exact.  This separates "the kernels compute what the
reference computes on the values they are given" (tight) from the accumulated storage
rounding of a 30-BatchNorm chain (reported, loose).
"""
import numpy as np
from . import keras_ops as K

BN_EPS = 1e-3
BN_MOMENTUM = 0.99


def round_bf16(x):
    """round-to-nearest-even to bfloat16, returned in x's dtype (finite values)"""
    x = np.asarray(x)
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)).view(np.float32)
    return r.astype(x.dtype)


def _glorot(rng, shape, fan_in, fan_out, dtype):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(dtype)


def unet_param_specs(nclasses, nchannels, filters, factors):
    """Ordered (name, shape, kind) list; kind in {kernel, kernelT, bias, gamma, beta, mm, mv}."""
    assert len(filters) == len(factors), 'number of filters and factors must be equal'
    specs = []

    def conv(name, k, ci, co):
        specs.append((f'{name}.kernel', (k, k, ci, co), 'kernel'))
        specs.append((f'{name}.bias', (co,), 'bias'))

    def bn(name, c):
        specs.append((f'{name}.gamma', (c,), 'gamma'))
        specs.append((f'{name}.beta', (c,), 'beta'))
        specs.append((f'{name}.moving_mean', (c,), 'mm'))
        specs.append((f'{name}.moving_var', (c,), 'mv'))

    cin = nchannels
    for i, f in enumerate(filters):
        conv(f'enc{i}.conv', 3, cin, f)
        bn(f'enc{i}.bn', f)
        cin = f
    conv('center.conv', 3, cin, 2 * filters[-1])
    bn('center.bn', 2 * filters[-1])
    cin = 2 * filters[-1]
    for j in range(len(filters) - 1, -1, -1):
        f, s = filters[j], factors[j]
        specs.append((f'dec{j}.up.kernel', (s, s, f, cin), 'kernelT'))
        specs.append((f'dec{j}.up.bias', (f,), 'bias'))
        bn(f'dec{j}.bn0', 2 * f)
        conv(f'dec{j}.conv1', 3, 2 * f, f)
        bn(f'dec{j}.bn1', f)
        conv(f'dec{j}.conv2', 3, f, f)
        bn(f'dec{j}.bn2', f)
        cin = f
    conv('probs', 1, cin, nclasses)
    return specs


class UNetOracle:
    def __init__(self, nclasses, nchannels, filters=(32, 64, 128, 256, 512),
                 factors=(2, 2, 2, 2, 2), bias=None, dtype=np.float64, seed=0,
                 bessel=True, store_dtype=None):
        assert store_dtype in (None, 'bfloat16')
        self.q = round_bf16 if store_dtype == 'bfloat16' else (lambda v: v)
        self.store_sums_exact = store_dtype is not None
        self.nclasses, self.nchannels = nclasses, nchannels
        self.filters, self.factors = list(filters), list(factors)
        self.dtype, self.bessel = dtype, bessel
        rng = np.random.default_rng(seed)
        self.specs = unet_param_specs(nclasses, nchannels, self.filters, self.factors)
        self.params = {}
        for name, shape, kind in self.specs:
            if kind in ('kernel', 'kernelT'):
                rec = shape[0] * shape[1]
                self.params[name] = _glorot(rng, shape, shape[2] * rec, shape[3] * rec, dtype)
            elif kind == 'bias':
                if name == 'probs.bias':
                    # bias=None -> Keras falls back to glorot_uniform (Appendix B Q3)
                    self.params[name] = (np.full(shape, bias, dtype) if bias is not None
                                         else _glorot(rng, shape, shape[0], shape[0], dtype))
                else:
                    self.params[name] = np.zeros(shape, dtype)
            elif kind in ('gamma', 'mv'):
                self.params[name] = np.ones(shape, dtype)
            else:
                self.params[name] = np.zeros(shape, dtype)
        self.trainable = [n for n, _, k in self.specs if k not in ('mm', 'mv')]
        self.adam_m = {n: np.zeros_like(self.params[n]) for n in self.trainable}
        self.adam_v = {n: np.zeros_like(self.params[n]) for n in self.trainable}
        self.adam_t = 0
        self.cache = None

    # ------------------------------------------------------------------ blocks
    def _bn_fwd(self, name, x, training, updates):
        p = self.params
        g, b = p[f'{name}.gamma'], p[f'{name}.beta']
        if not training:
            return K.batchnorm_infer(x, g, b, p[f'{name}.moving_mean'], p[f'{name}.moving_var'], BN_EPS), None
        y, mean, var = K.batchnorm_train(x, g, b, BN_EPS)
        cnt = x.shape[0] * x.shape[1] * x.shape[2]
        for _ in range(updates):
            mm, mv = K.bn_update_moving(p[f'{name}.moving_mean'], p[f'{name}.moving_var'],
                                        mean, var, BN_MOMENTUM, cnt, self.bessel)
            p[f'{name}.moving_mean'], p[f'{name}.moving_var'] = mm, mv
        return y, (mean, var)

    def _cba_fwd(self, name, bnname, x, training, c, dilation=1, updates=1):
        p, q = self.params, self.q
        y = q(K.conv2d_same(x, q(p[f'{name}.kernel']), p[f'{name}.bias'], dilation))
        z, st = self._bn_fwd(bnname, y, training, updates)
        a_exact = K.relu(z)
        a = q(a_exact)                    # what a consumer's loader stages (the 1x1 head reads the fp32 value: c[name + ':exact'])
        c[name] = (x, y, st, a, dilation)
        c[name + ':exact'] = a_exact
        return a

    def _cba_bwd(self, name, bnname, da, c, g, da_sums=None):
        """da_sums (store_dtype mode only): the gradient the BatchNorm-backward SUMS are formed from when it is not the stored tensor --
        the block under the head: the head's backward kernel sums its float32 gradient, the apply step uses the bf16-rounded one."""
        p, q = self.params, self.q
        x, y, (mean, var), a, dilation = c[name]
        dz = K.relu_bwd(c[name + ':exact'], da)
        dy, g[f'{bnname}.gamma'], g[f'{bnname}.beta'] = K.batchnorm_train_bwd(
            y, p[f'{bnname}.gamma'], mean, var, dz, BN_EPS)
        if da_sums is not None:
            dzs = K.relu_bwd(c[name + ':exact'], da_sums)
            m = y.shape[0] * y.shape[1] * y.shape[2]
            rstd = 1.0 / np.sqrt(var + BN_EPS)
            xhat = (y - mean) * rstd
            dbeta, dgamma = dzs.sum(axis=(0, 1, 2)), (dzs * xhat).sum(axis=(0, 1, 2))
            dy = p[f'{bnname}.gamma'] * rstd * (dz - dbeta / m - xhat * dgamma / m)
            g[f'{bnname}.gamma'], g[f'{bnname}.beta'] = dgamma, dbeta
        dy = q(dy)
        dx, g[f'{name}.kernel'], g[f'{name}.bias'] = K.conv2d_same_bwd(x, q(p[f'{name}.kernel']), dy, dilation)
        dx = q(dx)
        self.dbg[f'dy:{name}'], self.dbg[f'dx:{name}'] = dy, dx
        return dx

    # ----------------------------------------------------------------- forward
    def forward(self, x, training=False, masks=None):
        """x (N,H,W,C).  Returns (probs, classes) -- get_unet_model outputs (:407).

        masks (training only): dropout masks with values 0 or 1/(1-rate), keyed by position --
        'pool0' SpatialDropout2D after the level-0 pool (:350-351), 'center' Dropout (:362-363),
        'dec0' SpatialDropout2D inside the last decoder block (:310-311, :375), 'final'
        SpatialDropout2D in front of the head (:401-402)."""
        x = self.q(np.asarray(x, self.dtype))
        p, c, q = self.params, {}, self.q
        masks = masks or {}
        c['masks'] = masks
        L = len(self.filters)
        h = x
        for i in range(L):
            # conv_block.call executes cba1 twice on the same input (:238-240): the
            # value is that of one execution, the BN moving stats update twice.
            a = self._cba_fwd(f'enc{i}.conv', f'enc{i}.bn', h, training, c, updates=2)
            c[f'enc{i}'] = a
            h = K.maxpool(a, self.factors[i])
            if i == 0 and 'pool0' in masks:
                h = q(h * masks['pool0'])
        h = self._cba_fwd('center.conv', 'center.bn', h, training, c, updates=2)
        if 'center' in masks:
            h = q(h * masks['center'])
        for j in range(L - 1, -1, -1):
            up = q(K.conv2d_transpose_ks(h, q(p[f'dec{j}.up.kernel']), p[f'dec{j}.up.bias']))
            cat = np.concatenate([c[f'enc{j}'], up], axis=-1)        # skip first (:307)
            z, st = self._bn_fwd(f'dec{j}.bn0', cat, training, 1)
            a0_exact = K.relu(z)
            a0 = q(a0_exact)
            c[f'dec{j}.up'] = (h, cat, st, a0_exact)
            if j == 0 and 'dec0' in masks:
                a0 = q(a0_exact * masks['dec0'])
            a1 = self._cba_fwd(f'dec{j}.conv1', f'dec{j}.bn1', a0, training, c)
            h = self._cba_fwd(f'dec{j}.conv2', f'dec{j}.bn2', a1, training, c)
        if 'final' in masks:
            h = q(h * masks['final'])
        else:
            h = c['dec0.conv2:exact']         # the head applies the BatchNorm + ReLU in fp32 registers: nothing is rounded in between
        logits = K.conv2d_same(h, p['probs.kernel'], p['probs.bias'])
        probs = K.softmax(logits)
        c['head'] = (h, probs)
        self.cache = c if training else None
        return probs, K.argmax_classes(probs)

    # ---------------------------------------------------------------- backward
    def backward(self, dprobs):
        """Gradients of all trainable params given dL/dprobs (after a training forward)."""
        p, c, g = self.params, self.cache, {}
        self.dbg = {}
        L = len(self.filters)
        h, probs = c['head']
        dlogits = K.softmax_bwd(probs, np.asarray(dprobs, self.dtype))
        dh, g['probs.kernel'], g['probs.bias'] = K.conv2d_same_bwd(h, p['probs.kernel'], dlogits)
        q = self.q
        dh_exact = dh
        dh = q(dh)
        masks = c.get('masks', {})
        if 'final' in masks:
            dh = dh * masks['final']
        dskip = {}
        for j in range(L):
            head_sums = dh_exact if (j == 0 and self.store_sums_exact and 'final' not in masks) else None
            da1 = self._cba_bwd(f'dec{j}.conv2', f'dec{j}.bn2', dh, c, g, da_sums=head_sums)
            da0 = self._cba_bwd(f'dec{j}.conv1', f'dec{j}.bn1', da1, c, g)
            if j == 0 and 'dec0' in masks:
                da0 = da0 * masks['dec0']
            hin, cat, (mean, var), a0 = c[f'dec{j}.up']
            dz = K.relu_bwd(a0, da0)
            dcat, g[f'dec{j}.bn0.gamma'], g[f'dec{j}.bn0.beta'] = K.batchnorm_train_bwd(
                cat, p[f'dec{j}.bn0.gamma'], mean, var, dz, BN_EPS)
            dcat = q(dcat)
            f = self.filters[j]
            dskip[j] = dcat[..., :f]
            self.dbg[f'dskip:dec{j}.bn0'], self.dbg[f'du:dec{j}.bn0'] = dcat[..., :f], dcat[..., f:]
            dh, g[f'dec{j}.up.kernel'], g[f'dec{j}.up.bias'] = K.conv2d_transpose_ks_bwd(
                hin, q(p[f'dec{j}.up.kernel']), dcat[..., f:])
            dh = q(dh)
            self.dbg[f'dx:dec{j}.up'] = dh
        if 'center' in masks:
            dh = dh * masks['center']
        dh = self._cba_bwd('center.conv', 'center.bn', dh, c, g)
        for i in range(L - 1, -1, -1):
            a = c[f'enc{i}']
            if i == 0 and 'pool0' in masks:
                dh = dh * masks['pool0']
            da = K.maxpool_bwd(a, self.factors[i], dh) + dskip[i]
            dh = self._cba_bwd(f'enc{i}.conv', f'enc{i}.bn', da, c, g)
        g['input'] = dh
        return g

    # -------------------------------------------------------------------- Adam
    def adam_step(self, grads, lr=9e-4, beta1=0.9, beta2=0.999, eps=1e-7):
        """Keras Adam: alpha_t = lr*sqrt(1-b2^t)/(1-b1^t); theta -= alpha_t*m/(sqrt(v)+eps)."""
        self.adam_t += 1
        t = self.adam_t
        alpha = lr * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
        for n in self.trainable:
            gr = grads[n]
            self.adam_m[n] = beta1 * self.adam_m[n] + (1 - beta1) * gr
            self.adam_v[n] = beta2 * self.adam_v[n] + (1 - beta2) * gr * gr
            self.params[n] = self.params[n] - alpha * self.adam_m[n] / (np.sqrt(self.adam_v[n]) + eps)


# ------------------------------------------------------------------------- ASPP
def aspp_param_specs(cin, f, name='ASPP'):
    specs = []
    for br, k in (('cba', 1), ('cba3_3', 3), ('cba3_6', 3), ('cba3_12', 3)):
        specs += [(f'{name}.{br}.conv.kernel', (k, k, cin, f), 'kernel'), (f'{name}.{br}.conv.bias', (f,), 'bias')]
        specs += [(f'{name}.{br}.bn.{s}', (f,), kd) for s, kd in
                  (('gamma', 'gamma'), ('beta', 'beta'), ('moving_mean', 'mm'), ('moving_var', 'mv'))]
    specs += [(f'{name}.cba3.conv.kernel', (1, 1, 4 * f, f), 'kernel'), (f'{name}.cba3.conv.bias', (f,), 'bias')]
    specs += [(f'{name}.cba3.bn.{s}', (f,), kd) for s, kd in
              (('gamma', 'gamma'), ('beta', 'beta'), ('moving_mean', 'mm'), ('moving_var', 'mv'))]
    return specs


def aspp_forward(params, x, training=False, name='ASPP'):
    """DilatedSpatialPyramidPooling.call (utils/model_tools.py:552-574):
    cba3(concat[cba(x), cba3_3(x), cba3_6(x), cba3_12(x)]); rates 3/6/12; the
    image-pooling branch is commented out (:568-570) and cba2 is unused."""
    def cba(br, inp, d):
        y = K.conv2d_same(inp, params[f'{name}.{br}.conv.kernel'], params[f'{name}.{br}.conv.bias'], d)
        g, b = params[f'{name}.{br}.bn.gamma'], params[f'{name}.{br}.bn.beta']
        if training:
            z, _, _ = K.batchnorm_train(y, g, b, BN_EPS)
        else:
            z = K.batchnorm_infer(y, g, b, params[f'{name}.{br}.bn.moving_mean'],
                                  params[f'{name}.{br}.bn.moving_var'], BN_EPS)
        return K.relu(z)
    outs = [cba('cba', x, 1), cba('cba3_3', x, 3), cba('cba3_6', x, 6), cba('cba3_12', x, 12)]
    return cba('cba3', np.concatenate(outs, axis=-1), 1)
