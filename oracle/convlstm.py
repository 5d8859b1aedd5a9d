"""NumPy restatement of tf.keras.layers.ConvLSTM2D as the reference calls it, forward + BPTT, and of the LSTM model builders.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: the arithmetic lives in TensorFlow / Keras (un-vendored,
un-pinned dependency of /root/reference/utils/model_tools.py:8-15); semantics follow the documented Keras ConvLSTM2D cell
(keras/layers/rnn/conv_lstm*.py, Keras 2.x / TF 2.x):

    x_i, x_f, x_c, x_o = conv(x_t, kernel[..., g F:(g + 1) F], padding, dilation_rate) + bias[g F:(g + 1) F]     g = i, f, c, o
    h_i, h_f, h_c, h_o = conv(h_{t-1}, recurrent_kernel[..., g F:(g + 1) F], 'same', NO dilation)                (no bias)
    i = rec_act(x_i + h_i);  f = rec_act(x_f + h_f);  c_t = f * c_{t-1} + i * act(x_c + h_c)
    o = rec_act(x_o + h_o);  h_t = o * act(c_t)
    kernel (kh, kw, Cin, 4F), recurrent_kernel (kh, kw, F, 4F), bias (4F) = zeros with ones on the forget gate (unit_forget_bias),
    h_0 = c_0 = 0; inputs (batch, time, rows, cols, channels).

The reference's call sites (utils/model_tools.py:690-728, 749-768, 834-843) pass `activation=None` (act = identity), 3x3 kernels,
padding 'same', dilation (3, 3) on the second layer, and leave `recurrent_activation` at its default -- which is VERSION DEPENDENT:
'hard_sigmoid' = clip(0.2 x + 0.5, 0, 1) in Keras 2.x (TF <= 2.15), 'sigmoid' in Keras 3.  Both are implemented; `rec_act` selects.
"""
import numpy as np
from . import keras_ops as K


def rec_act_fwd(z, kind):
    if kind == 'hard_sigmoid':
        return np.clip(0.2 * z + 0.5, 0.0, 1.0)
    if kind == 'sigmoid':
        return 1.0 / (1.0 + np.exp(-z))
    raise ValueError(kind)


def rec_act_bwd(z, y, kind):
    """d rec_act / dz given pre-activation z and value y"""
    if kind == 'hard_sigmoid':
        return 0.2 * ((z > -2.5) & (z < 2.5))
    return y * (1.0 - y)


def act_fwd(z, kind):
    return z if kind in (None, 'linear') else np.tanh(z)


def act_bwd(z, y, kind):
    return np.ones_like(z) if kind in (None, 'linear') else 1.0 - y * y


def convlstm_param_shapes(cin, filters, k=3):
    return {'kernel': (k, k, cin, 4 * filters), 'recurrent_kernel': (k, k, filters, 4 * filters), 'bias': (4 * filters,)}


def convlstm_init(rng, cin, filters, k=3, dtype=np.float64):
    """glorot_uniform kernel, orthogonal recurrent kernel (QR of a Gaussian, as keras.initializers.Orthogonal), unit forget bias"""
    lim = np.sqrt(6.0 / (k * k * cin + k * k * 4 * filters))
    kern = rng.uniform(-lim, lim, (k, k, cin, 4 * filters))
    a = rng.standard_normal((k * k * filters, 4 * filters))
    q, r = np.linalg.qr(a.T if a.shape[0] < a.shape[1] else a)
    q = q * np.sign(np.diag(r))
    q = q.T if a.shape[0] < a.shape[1] else q
    rec = q.reshape(k, k, filters, 4 * filters)
    bias = np.zeros(4 * filters)
    bias[filters:2 * filters] = 1.0
    return {'kernel': kern.astype(dtype), 'recurrent_kernel': rec.astype(dtype), 'bias': bias.astype(dtype)}


def convlstm_forward(x, p, dilation=1, activation=None, rec_act='hard_sigmoid', return_sequences=True):
    """x (B, T, H, W, Cin).  Returns (out, cache): out = (B, T, H, W, F) or the last hidden state (B, H, W, F);
    cache['h_last'], cache['c_last'] are the final states (return_state)."""
    B, T, H, W, _ = x.shape
    F = p['bias'].shape[0] // 4
    h = np.zeros((B, H, W, F), x.dtype)
    c = np.zeros((B, H, W, F), x.dtype)
    steps, hs = [], []
    for t in range(T):
        z = K.conv2d_same(x[:, t], p['kernel'], p['bias'], dilation) + K.conv2d_same(h, p['recurrent_kernel'], None, 1)
        zi, zf, zc, zo = (z[..., g * F:(g + 1) * F] for g in range(4))
        i, f, o = rec_act_fwd(zi, rec_act), rec_act_fwd(zf, rec_act), rec_act_fwd(zo, rec_act)
        g = act_fwd(zc, activation)
        c_new = f * c + i * g
        ac = act_fwd(c_new, activation)
        h_new = o * ac
        steps.append(dict(h_prev=h, c_prev=c, z=z, i=i, f=f, o=o, g=g, c=c_new, ac=ac))
        h, c = h_new, c_new
        hs.append(h)
    seq = np.stack(hs, axis=1)
    cache = dict(x=x, steps=steps, p=p, dilation=dilation, activation=activation, rec_act=rec_act, return_sequences=return_sequences,
                 h_last=h, c_last=c, F=F)
    return (seq if return_sequences else h), cache


def convlstm_backward(dout, cache, dh_last=None):
    """dout: gradient of the returned tensor ((B, T, H, W, F) or (B, H, W, F)); dh_last: additional gradient of the final hidden state
    (return_state consumers).  Returns (dx, grads) with grads for kernel / recurrent_kernel / bias."""
    x, steps, p, F = cache['x'], cache['steps'], cache['p'], cache['F']
    act, rk, dil = cache['activation'], cache['rec_act'], cache['dilation']
    B, T, H, W, _ = x.shape
    dk, drk, db = np.zeros_like(p['kernel']), np.zeros_like(p['recurrent_kernel']), np.zeros_like(p['bias'])
    dx = np.zeros_like(x)
    dh_next = np.zeros((B, H, W, F), x.dtype)
    dc_next = np.zeros((B, H, W, F), x.dtype)
    for t in range(T - 1, -1, -1):
        s = steps[t]
        dh = dh_next + (dout[:, t] if cache['return_sequences'] else (dout if t == T - 1 else 0.0))
        if dh_last is not None and t == T - 1:
            dh = dh + dh_last
        zi, zf, zc, zo = (s['z'][..., g * F:(g + 1) * F] for g in range(4))
        do = dh * s['ac']
        dc = dc_next + dh * s['o'] * act_bwd(s['c'], s['ac'], act)
        di, df, dg = dc * s['g'], dc * s['c_prev'], dc * s['i']
        dz = np.concatenate([di * rec_act_bwd(zi, s['i'], rk), df * rec_act_bwd(zf, s['f'], rk), dg * act_bwd(zc, s['g'], act),
                             do * rec_act_bwd(zo, s['o'], rk)], axis=-1)
        dxt, dkt, dbt = K.conv2d_same_bwd(x[:, t], p['kernel'], dz, dil)
        dx[:, t] = dxt
        dk += dkt
        db += dbt
        dh_next, drkt, _ = K.conv2d_same_bwd(s['h_prev'], p['recurrent_kernel'], dz, 1)
        drk += drkt
        dc_next = dc * s['f']
    return dx, {'kernel': dk, 'recurrent_kernel': drk, 'bias': db}


# ------------------------------------------------------------------------------------------ small layers around it
def bn5_train(x, gamma, beta, eps=1e-3):
    """BatchNormalization() on a (B, T, H, W, C) or (B, H, W, C) tensor: axis -1, statistics over every other axis"""
    ax = tuple(range(x.ndim - 1))
    mean, var = x.mean(axis=ax), x.var(axis=ax)
    return gamma * (x - mean) / np.sqrt(var + eps) + beta, mean, var


def bn5_train_bwd(x, gamma, mean, var, dy, eps=1e-3):
    ax = tuple(range(x.ndim - 1))
    m = x.size // x.shape[-1]
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (x - mean) * rstd
    dbeta, dgamma = dy.sum(axis=ax), (dy * xhat).sum(axis=ax)
    return gamma * rstd * (dy - dbeta / m - xhat * dgamma / m), dgamma, dbeta


def resize_nearest(x, oh, ow):
    """tf.image.resize(x, [oh, ow], method='nearest') (utils/model_tools.py:908, 1055).  TF 2.x's resize_images_v2 calls
    ResizeNearestNeighbor with half_pixel_centers=True: source index = min(floor((dst + 0.5) * in / out), in - 1)."""
    n, h, w, c = x.shape
    iy = np.minimum(np.floor((np.arange(oh) + 0.5) * (h / oh)).astype(np.int64), h - 1)
    ix = np.minimum(np.floor((np.arange(ow) + 0.5) * (w / ow)).astype(np.int64), w - 1)
    return x[:, iy][:, :, ix], (iy, ix)


def resize_nearest_bwd(dy, idx, h, w):
    iy, ix = idx
    n, oh, ow, c = dy.shape
    dx = np.zeros((n, h, w, c), dy.dtype)
    np.add.at(dx, (slice(None), iy[:, None], ix[None, :]), dy)
    return dx


class LSTMLayersOracle:
    """build_lstm_layers (utils/model_tools.py:666-717): ConvLSTM2D(64) -> BN -> ReLU -> ConvLSTM2D(64, dilation 3, last state) -> BN
    -> ReLU, and the 1x1 Conv2D + activation of get_lstm_model (:773-808; the body cannot run as coded -- Q7 -- this is the network it
    describes: Conv2D(n_classes, 1x1) followed by `activation`, default ReLU(max_value = 2.0))."""

    def __init__(self, n_channels, n_classes, filters=64, rec_act='hard_sigmoid', seed=0, head_max=2.0, dtype=np.float64):
        rng = np.random.default_rng(seed)
        self.F, self.rec_act, self.head_max = filters, rec_act, head_max
        self.p = {'l1': convlstm_init(rng, n_channels, filters, dtype=dtype), 'l2': convlstm_init(rng, filters, filters, dtype=dtype)}
        for b in ('bn1', 'bn2'):
            self.p[b] = {'gamma': np.ones(filters, dtype), 'beta': np.zeros(filters, dtype)}
        lim = np.sqrt(6.0 / (filters + n_classes))
        self.p['dense'] = {'kernel': rng.uniform(-lim, lim, (1, 1, filters, n_classes)).astype(dtype), 'bias': np.zeros(n_classes, dtype)}

    def features(self, x, mask1=None):
        """mask1: the layers.Dropout(dropout) between the two ConvLSTM2D layers (utils/model_tools.py:699-700) as a GIVEN mask
        (B, T, H, W, F) of 0 or 1 / (1 - rate) -- the test hands over the mask the device drew"""
        p = self.p
        s1, c1 = convlstm_forward(x, p['l1'], 1, None, self.rec_act, True)
        z1, m1, v1 = bn5_train(s1, p['bn1']['gamma'], p['bn1']['beta'])
        a1 = np.maximum(z1, 0)
        self.mask1 = mask1
        if mask1 is not None:
            a1_in = a1 * mask1
        else:
            a1_in = a1
        h2, c2 = convlstm_forward(a1_in, p['l2'], 3, None, self.rec_act, False)
        z2, m2, v2 = bn5_train(h2, p['bn2']['gamma'], p['bn2']['beta'])
        a2 = np.maximum(z2, 0)
        self.c = dict(c1=c1, s1=s1, st1=(m1, v1), a1=a1, c2=c2, h2=h2, st2=(m2, v2), a2=a2)
        return a2

    def features_bwd(self, da2):
        p, c, g = self.p, self.c, {}
        dz2 = da2 * (c['a2'] > 0)
        dh2, g['bn2.gamma'], g['bn2.beta'] = bn5_train_bwd(c['h2'], p['bn2']['gamma'], *c['st2'], dz2)
        da1, g2 = convlstm_backward(dh2, c['c2'])
        if getattr(self, 'mask1', None) is not None:
            da1 = da1 * self.mask1
        dz1 = da1 * (c['a1'] > 0)
        ds1, g['bn1.gamma'], g['bn1.beta'] = bn5_train_bwd(c['s1'], p['bn1']['gamma'], *c['st1'], dz1)
        dx, g1 = convlstm_backward(ds1, c['c1'])
        for k, v in g1.items():
            g['l1.' + k] = v
        for k, v in g2.items():
            g['l2.' + k] = v
        g['input'] = dx
        return g

    def forward(self, x, mask1=None):
        a2 = self.features(x, mask1)
        z = K.conv2d_same(a2, self.p['dense']['kernel'], self.p['dense']['bias'])
        out = np.clip(z, 0.0, self.head_max) if self.head_max is not None else np.maximum(z, 0)
        self.c['z'] = z
        return out

    def forward_infer(self, x, moving, eps=1e-3):
        """inference mode (Model.predict): each BatchNormalization uses its MOVING statistics, moving = {'bn1': (mean, var), 'bn2': (mean, var)}
        (Keras: gamma (x - moving_mean) / sqrt(moving_var + eps) + beta); no dropout"""
        p = self.p
        s1, _ = convlstm_forward(x, p['l1'], 1, None, self.rec_act, True)
        a1 = np.maximum(p['bn1']['gamma'] * (s1 - moving['bn1'][0]) / np.sqrt(moving['bn1'][1] + eps) + p['bn1']['beta'], 0)
        h2, _ = convlstm_forward(a1, p['l2'], 3, None, self.rec_act, False)
        a2 = np.maximum(p['bn2']['gamma'] * (h2 - moving['bn2'][0]) / np.sqrt(moving['bn2'][1] + eps) + p['bn2']['beta'], 0)
        z = K.conv2d_same(a2, p['dense']['kernel'], p['dense']['bias'])
        return np.clip(z, 0.0, self.head_max) if self.head_max is not None else np.maximum(z, 0)

    def backward(self, dout):
        z = self.c['z']
        mask = (z > 0) & ((z < self.head_max) if self.head_max is not None else True)
        dz = dout * mask
        da2, dk, db = K.conv2d_same_bwd(self.c['a2'], self.p['dense']['kernel'], dz)
        g = self.features_bwd(da2)
        g['dense.kernel'], g['dense.bias'] = dk, db
        return g


class LSTMAutoencoderOracle:
    """get_lstm_autoencoder (utils/model_tools.py:810-872) with build_lstm_layers2 (:719-771) as its encoder: forward of both outputs
    and the gradients of  sum(mse_4d-style upstream gradients)  through both branches."""

    def __init__(self, n_channels, n_time, n_classes, rec_act='hard_sigmoid', seed=0, head_max=2.0, dtype=np.float64):
        rng = np.random.default_rng(seed)
        F = 16
        self.F, self.T, self.rec_act, self.head_max = F, n_time, rec_act, head_max
        self.p = {'l1': convlstm_init(rng, n_channels, F, dtype=dtype), 'l2': convlstm_init(rng, F, F, dtype=dtype),
                  'dec': convlstm_init(rng, F, 32, dtype=dtype)}
        for b in ('bn1', 'bn2'):
            self.p[b] = {'gamma': np.ones(F, dtype), 'beta': np.zeros(F, dtype)}
        for name, cin in (('temporal', 32), ('single', F + 2)):
            lim = np.sqrt(6.0 / (cin + n_classes))
            self.p[name] = {'kernel': rng.uniform(-lim, lim, (1, 1, cin, n_classes)).astype(dtype), 'bias': np.zeros(n_classes, dtype)}

    def _head(self, a, name):
        z = K.conv2d_same(a, self.p[name]['kernel'], self.p[name]['bias'])
        return z, (np.clip(z, 0.0, self.head_max) if self.head_max is not None else np.maximum(z, 0))

    def forward(self, x, sincos):
        p = self.p
        B, T = x.shape[:2]
        s1, c1 = convlstm_forward(x, p['l1'], 1, None, self.rec_act, True)
        z1, m1, v1 = bn5_train(s1, p['bn1']['gamma'], p['bn1']['beta'])
        a1 = np.maximum(z1, 0)
        h2, c2 = convlstm_forward(a1, p['l2'], 3, None, self.rec_act, False)
        z2, m2, v2 = bn5_train(h2, p['bn2']['gamma'], p['bn2']['beta'])
        enc = np.maximum(c1['h_last'] + z2, 0)
        rep = np.repeat(enc[:, None], T, axis=1)
        dseq, cd = convlstm_forward(rep, p['dec'], 1, None, self.rec_act, True)
        zt, tout = self._head(dseq.reshape((B * T,) + dseq.shape[2:]), 'temporal')
        cat = np.concatenate([enc, sincos], -1)
        zs, sout = self._head(cat, 'single')
        self.c = dict(c1=c1, s1=s1, st1=(m1, v1), a1=a1, c2=c2, h2=h2, st2=(m2, v2), enc=enc, cd=cd, dseq=dseq, zt=zt, zs=zs, cat=cat, B=B, T=T)
        return tout.reshape((B, T) + tout.shape[1:]), sout

    def backward(self, dtout, dsout):
        p, c, g = self.p, self.c, {}
        B, T = c['B'], c['T']

        def head_bwd(z, dout, a, name):
            mask = (z > 0) & ((z < self.head_max) if self.head_max is not None else True)
            da, g[name + '.kernel'], g[name + '.bias'] = K.conv2d_same_bwd(a, p[name]['kernel'], dout * mask)
            return da
        ddseq = head_bwd(c['zt'], dtout.reshape(c['zt'].shape), c['dseq'].reshape((B * T,) + c['dseq'].shape[2:]), 'temporal')
        drep, gd = convlstm_backward(ddseq.reshape(c['dseq'].shape), c['cd'])
        denc = drep.sum(axis=1)
        dcat = head_bwd(c['zs'], dsout, c['cat'], 'single')
        denc = denc + dcat[..., :self.F]
        gm = denc * (c['enc'] > 0)
        dh2, g['bn2.gamma'], g['bn2.beta'] = bn5_train_bwd(c['h2'], p['bn2']['gamma'], *c['st2'], gm)
        da1, g2 = convlstm_backward(dh2, c['c2'])
        dz1 = da1 * (c['a1'] > 0)
        ds1, g['bn1.gamma'], g['bn1.beta'] = bn5_train_bwd(c['s1'], p['bn1']['gamma'], *c['st1'], dz1)
        _, g1 = convlstm_backward(ds1, c['c1'], dh_last=gm)
        for pre, gg in (('l1', g1), ('l2', g2), ('dec', gd)):
            for k, v in gg.items():
                g[f'{pre}.{k}'] = v
        return g


class HierarchicalOracle:
    """get_hierarchical_model (utils/model_tools.py:1016-1060): build_acnn_layers2 trunk (:941-979) + build_lstm_layers (:666-717) + the
    three softmax heads; forward of the three outputs and the gradients given dL/dlogits of each head."""

    def __init__(self, nclasses, acnn_nclasses, acnn_sub_nclasses, nchannels, lstm_channels, nfilters, depth, rec_act='hard_sigmoid', seed=0):
        rng = np.random.default_rng(seed)
        self.depth, self.mid, self.F = depth, (depth - 1) // 2, nfilters
        self.p = {}
        for l in range(depth):
            cin = nchannels if l == 0 else nfilters
            for nm, ci in ((f'Conv{l}_1', cin), (f'DilateConv{l}_2', nfilters)):
                lim = np.sqrt(6.0 / (9 * ci + 9 * nfilters))
                self.p[nm] = {'kernel': rng.uniform(-lim, lim, (3, 3, ci, nfilters)), 'bias': np.zeros(nfilters)}
            for nm in (f'bn{l}_1', f'bn{l}_2'):
                self.p[nm] = {'gamma': np.ones(nfilters), 'beta': np.zeros(nfilters)}
        self.lstm = LSTMLayersOracle(lstm_channels, 1, filters=64, rec_act=rec_act, seed=seed + 1)
        for nm, ci, co in (('sub_probs', nfilters, acnn_sub_nclasses), ('acnn_probs', nfilters, acnn_nclasses), ('lstm_probs', 64 + nfilters, nclasses)):
            lim = np.sqrt(6.0 / (ci + co))
            self.p[nm] = {'kernel': rng.uniform(-lim, lim, (1, 1, ci, co)), 'bias': np.zeros(co)}

    def forward(self, xa, xl):
        p, c = self.p, {}
        f, s = xa, None
        feats = []
        for l in range(self.depth):
            y1 = K.conv2d_same(f, p[f'Conv{l}_1']['kernel'], p[f'Conv{l}_1']['bias'])
            z1, m1, v1 = bn5_train(y1, p[f'bn{l}_1']['gamma'], p[f'bn{l}_1']['beta'])
            s_new = np.maximum(z1 if l == 0 else z1 + s, 0)
            y2 = K.conv2d_same(s_new, p[f'DilateConv{l}_2']['kernel'], p[f'DilateConv{l}_2']['bias'], 3)
            z2, m2, v2 = bn5_train(y2, p[f'bn{l}_2']['gamma'], p[f'bn{l}_2']['beta'])
            f_new = np.maximum(z2, 0)
            c[l] = dict(fin=f, y1=y1, st1=(m1, v1), s=s_new, y2=y2, st2=(m2, v2), f=f_new)
            f, s = f_new, s_new
            feats.append(f_new)
        lf = self.lstm.features(xl)
        up, idx = resize_nearest(lf, xa.shape[1], xa.shape[2])
        cat = np.concatenate([up, feats[-1]], -1)
        c.update(lf=lf, idx=idx, cat=cat)
        self.c = c
        head = lambda a, nm: K.softmax(K.conv2d_same(a, p[nm]['kernel'], p[nm]['bias']))
        return [head(feats[self.mid], 'sub_probs'), head(feats[-1], 'acnn_probs'), head(cat, 'lstm_probs')]

    def backward(self, dlogits):
        """dlogits: [dL/dlogits of sub_probs, acnn_probs, lstm_probs]"""
        p, c, g = self.p, self.c, {}
        F = self.F
        dmid, g['sub_probs.kernel'], g['sub_probs.bias'] = K.conv2d_same_bwd(c[self.mid]['f'], p['sub_probs']['kernel'], dlogits[0])
        dlast, g['acnn_probs.kernel'], g['acnn_probs.bias'] = K.conv2d_same_bwd(c[self.depth - 1]['f'], p['acnn_probs']['kernel'], dlogits[1])
        dcat, g['lstm_probs.kernel'], g['lstm_probs.bias'] = K.conv2d_same_bwd(c['cat'], p['lstm_probs']['kernel'], dlogits[2])
        dlf = resize_nearest_bwd(dcat[..., :64], c['idx'], c['lf'].shape[1], c['lf'].shape[2])
        for k, v in self.lstm.features_bwd(dlf).items():
            g['lstm.' + k] = v
        df = {self.depth - 1: dlast + dcat[..., 64:]}
        df[self.mid] = df.get(self.mid, 0) + dmid
        dfeat, ds_next = None, None
        for l in range(self.depth - 1, -1, -1):
            cl = c[l]
            d = df.get(l, 0) + (dfeat if dfeat is not None else 0)
            dz2 = d * (cl['f'] > 0)
            dy2, g[f'bn{l}_2.gamma'], g[f'bn{l}_2.beta'] = bn5_train_bwd(cl['y2'], p[f'bn{l}_2']['gamma'], *cl['st2'], dz2)
            ds, g[f'DilateConv{l}_2.kernel'], _ = K.conv2d_same_bwd(cl['s'], p[f'DilateConv{l}_2']['kernel'], dy2, 3)
            if ds_next is not None:
                ds = ds + ds_next
            gm = ds * (cl['s'] > 0)
            dy1, g[f'bn{l}_1.gamma'], g[f'bn{l}_1.beta'] = bn5_train_bwd(cl['y1'], p[f'bn{l}_1']['gamma'], *cl['st1'], gm)
            dfeat, g[f'Conv{l}_1.kernel'], _ = K.conv2d_same_bwd(cl['fin'], p[f'Conv{l}_1']['kernel'], dy1)
            ds_next = gm if l > 0 else None
        return g
