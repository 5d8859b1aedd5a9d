"""MI355X-native drop-in for the U-Net / ASPP path of the reference's utils/model_tools.py.

Same names, argument meaning and return conventions as
/root/reference/utils/model_tools.py (cited per symbol); underneath, every tensor op is a
hand-written HIP kernel reached through the C ABI (include/satcv.h).  There is no
TensorFlow and no CPU fallback: building a graph is pure Python, running it needs the
HIP library and a ROCm device.

Surface mirrored here
  losses         weighted_categorical_crossentropy :25, gen_dice :42, weighted_bce :96,
                 iou_loss :131, mse_4d :142
  blocks         conv_batch_act :174, conv_block :211 (AS CODED: one conv per level, BN moving
                 statistics updated twice), encoder_block :262, decoder_block :288
  builders       build_unet_layers :321, get_unet_model :394, DilatedSpatialPyramidPooling :533
  Model          compile / fit / evaluate / predict / save / load_weights / layers[i].trainable /
                 optimizer.learning_rate / metrics_names (call sites: utils/model_tools.py:1128-1176,
                 utils/prediction_tools.py:152,333; notebooks/UNET_G4G_2019_solar.ipynb:1206-1277)
"""
import ctypes as C
import json
import os
import time
from collections import defaultdict
from fractions import Fraction

import numpy as np
import torch

from . import engine as E
from . import ops
from ._lib import lib, check, F32, BF16

# ------------------------------------------------------------------------- globals
_DEFAULT_DTYPE = 'bfloat16'
_RNG = np.random.default_rng(0)
_UIDS = defaultdict(int)


def set_compute_dtype(name):
    """'bfloat16' (bf16 storage, MFMA bf16, fp32 accumulate) or 'float32' (exact fp32 MFMA)."""
    global _DEFAULT_DTYPE
    assert name in ('bfloat16', 'float32')
    _DEFAULT_DTYPE = name


def set_seed(seed):
    global _RNG, _SHUFFLE_RNG
    _RNG = np.random.default_rng(seed)
    _SHUFFLE_RNG = np.random.default_rng([int(seed) & 0xffffffff, 0x5f5])      # fit(shuffle=True) permutations


_SHUFFLE_RNG = np.random.default_rng(0x5f5)


def reset_uids():
    _UIDS.clear()


def _unique(base):
    """Keras-style automatic layer names: conv2d, conv2d_1, ..."""
    i = _UIDS[base]
    _UIDS[base] += 1
    return base if i == 0 else f'{base}_{i}'


def _glorot(shape, fan_in, fan_out):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return lambda: _RNG.uniform(-lim, lim, size=shape).astype(np.float32)


class _Constant:
    """tf.keras.initializers.Constant"""

    def __init__(self, value):
        self.value = value


# --------------------------------------------------------------------------- layers
class _LayerBase:
    def __init__(self, name):
        self.name = name
        self.trainable = True
        self.specs = []

    def _add(self, suffix, shape, kind, init):
        p = E.ParamSpec(f'{self.name}/{suffix}', shape, kind, init)
        self.specs.append(p)
        return p

    @property
    def weights(self):
        return [p.name for p in self.specs]


class _ConvParams(_LayerBase):
    """kernel + bias of a Conv2D / Conv2DTranspose (Keras defaults: glorot_uniform, zeros)."""

    def __init__(self, name, filters, kernel_size, transposed=False, bias_initializer='zeros'):
        super().__init__(name)
        self.filters, self.kernel_size, self.transposed = filters, kernel_size, transposed
        self.bias_initializer = bias_initializer
        self.bn_name = None
        self.built = False

    @property
    def kernel_name(self):
        return self.name + '/kernel'

    def build(self, cin):
        if self.built:
            return
        kh, kw = self.kernel_size
        f = self.filters
        shape = (kh, kw, f, cin) if self.transposed else (kh, kw, cin, f)
        rec = kh * kw
        self._add('kernel', shape, 'kernel', _glorot(shape, shape[2] * rec, shape[3] * rec))
        bi = self.bias_initializer
        if isinstance(bi, _Constant):
            init = lambda: np.full((f,), bi.value, np.float32)
        elif bi is None:
            # Keras add_weight default when the initializer argument is None
            init = _glorot((f,), f, f)
        else:
            init = lambda: np.zeros((f,), np.float32)
        self._add('bias', (f,), 'bias', init)
        self.built = True


class _BNParams(_LayerBase):
    """layers.BatchNormalization() defaults: axis -1, momentum 0.99, epsilon 1e-3."""

    def __init__(self, name=None):
        super().__init__(name or _unique('batch_normalization'))
        self.built = False

    def build(self, c):
        if self.built:
            return
        self._add('gamma', (c,), 'gamma', lambda: np.ones((c,), np.float32))
        self._add('beta', (c,), 'beta', lambda: np.zeros((c,), np.float32))
        self._add('moving_mean', (c,), 'moving_mean', lambda: np.zeros((c,), np.float32))
        self._add('moving_var', (c,), 'moving_var', lambda: np.ones((c,), np.float32))
        self.built = True


def Input(shape, name=None):
    """layers.Input(shape=[None, None, nchannels]) (utils/model_tools.py:397)."""
    node = E.Node('input', [])
    return node.out(int(shape[-1]), Fraction(1), name or _unique('input'))


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class conv_batch_act:
    """Single convolution -> batch norm -> activation layer stack (utils/model_tools.py:174-186).

    Lowered to ONE node: implicit-GEMM conv kernel with BN statistics in its epilogue; the BN
    affine + ReLU is applied by the consumer's loader."""

    def __init__(self, num_filters, kernel_size=(3, 3), dilation_rate=1, name='conv_batch_act', strides=1, activation='relu',
                 conv_name=None, bn_name=None, batch_norm=True, **kwargs):
        self.name = name
        ks = _pair(kernel_size)
        assert ks[0] == ks[1] and ks[0] % 2 == 1, 'kernel_size must be odd and square'
        self.strides = strides if isinstance(strides, int) else strides[0]      # extension (ResNet backbone): strided / linear variants
        self.activation = activation
        # extensions for the atrous CNN family: explicitly named layers, and a Conv2D that is not followed by BatchNormalization
        self.conv_layer = _ConvParams(conv_name or _unique('conv2d'), num_filters, ks, bias_initializer='zeros')
        self.bn_layer = _BNParams(bn_name) if batch_norm else None
        self.conv_layer.bn_name = self.bn_layer.name if batch_norm else None
        self.dilation_rate = dilation_rate if isinstance(dilation_rate, int) else dilation_rate[0]
        self.num_filters = num_filters
        self.bn_updates = 1

    @property
    def trainable(self):
        return self.conv_layer.trainable

    @trainable.setter
    def trainable(self, v):
        self.conv_layer.trainable = v
        if self.bn_layer is not None:
            self.bn_layer.trainable = v

    def _sublayers(self):
        return [l for l in (self.conv_layer, self.bn_layer) if l is not None]

    def __call__(self, inputs, bn_updates=None):
        self.conv_layer.build(inputs.channels)
        if self.bn_layer is not None:
            self.bn_layer.build(self.num_filters)
        node = E.Node('cba', [inputs], layer=self.conv_layer, k=self.conv_layer.kernel_size[0], dil=self.dilation_rate,
                      bn_updates=bn_updates or self.bn_updates, owner=self, stride=self.strides, relu=self.activation == 'relu',
                      bn=self.bn_layer is not None)
        return node.out(self.num_filters, inputs.down / self.strides)

    call = __call__


class conv_block:
    """U-Net convolution block (utils/model_tools.py:211-240).

    AS CODED the reference's call() runs `self.cba1(inputs)` twice and never uses cba2
    (:238-240): the block is ONE conv->BN->ReLU whose BN moving statistics are updated twice
    per training step.  cba2 is constructed (it consumes layer
    names, as in Keras) but never built.  `double_conv=True` gives the evidently intended
    conv->BN->ReLU x2 of the notebooks (notebooks/UNET_G4G_2019_solar.ipynb:1162-1169)."""

    def __init__(self, num_filters, kernel_size=(3, 3), dilation_rate=1, name='conv_block', double_conv=False, **kwargs):
        self.name = name
        self.cba1 = conv_batch_act(num_filters, kernel_size, dilation_rate)
        self.cba2 = conv_batch_act(num_filters, kernel_size, dilation_rate)
        self.double_conv = double_conv

    def _sublayers(self):
        return self.cba1._sublayers() + (self.cba2._sublayers() if self.double_conv else [])

    def __call__(self, inputs):
        if self.double_conv:
            return self.cba2(self.cba1(inputs))
        return self.cba1(inputs, bn_updates=2)

    call = __call__


class encoder_block:
    """U-Net downsampling encoder block conv -> max pool (utils/model_tools.py:262-286).
    Returns (pooled, encoded)."""

    def __init__(self, num_filters, kernel_size=(3, 3), dilation_rate=1, pool_size=(2, 2), name='encoder_block', **kwargs):
        self.name = name
        self.encoder = conv_block(num_filters, kernel_size, dilation_rate, **kwargs)
        ps = _pair(pool_size)
        assert ps[0] == ps[1]
        self.pool = ps[0]
        _unique('max_pooling2d')

    def _sublayers(self):
        return self.encoder._sublayers()

    def __call__(self, input):
        encoded = self.encoder(input)
        node = E.Node('pool', [encoded], f=self.pool)
        pooled = node.out(encoded.channels, encoded.down / self.pool)
        return pooled, encoded

    call = __call__


def _dropout(x, rate, spatial):
    node = E.Node('dropout', [x], rate=rate, spatial=spatial)
    return node.out(x.channels, x.down)


def decoder_block(input_tensor, concat_tensor, num_filters, up_size=(2, 2), dropout=None):
    """U-Net upsampling decoder block (utils/model_tools.py:288-318):
    Conv2DTranspose(k=s=up_size) -> concatenate([skip, up]) -> BN -> ReLU -> [SpatialDropout2D]
    -> (Conv3x3 -> BN -> ReLU) x 2.  The concat is never materialised."""
    us = _pair(up_size)
    assert us[0] == us[1]
    f = us[0]
    up = _ConvParams(_unique('conv2d_transpose'), num_filters, (f, f), transposed=True)
    up.build(input_tensor.channels)
    n_up = E.Node('convT', [input_tensor], layer=up, f=f)
    t_up = n_up.out(num_filters, input_tensor.down * f)
    _unique('concatenate')
    bn0 = _BNParams()
    bn0.build(concat_tensor.channels + num_filters)
    n_cat = E.Node('concat_bn_relu', [concat_tensor, t_up], layer=bn0)
    decoder = n_cat.out(concat_tensor.channels + num_filters, t_up.down)
    if dropout is not None:
        decoder = _dropout(decoder, dropout, True)
    for _ in range(2):
        cba = conv_batch_act(num_filters, (3, 3), 1, name='decoder_conv')
        cba.name = cba.conv_layer.name
        decoder = cba(decoder)
    return decoder



class DilatedSpatialPyramidPooling:
    """ASPP layer (utils/model_tools.py:533-574): cba3_1x1(concat[cba_1x1(x), cba3x3_d3(x), cba3x3_d6(x), cba3x3_d12(x)]).
    The image-pooling branch is commented out in the reference (:568-570) and cba2 is constructed but unused.
    The four branch convolutions write straight into channel slices of one tensor (no concatenate pass)."""

    def __init__(self, num_filters, name='ASPP', **kwargs):
        self.name = name
        self.cba = conv_batch_act(num_filters, kernel_size=(1, 1), dilation_rate=1)
        self.cba2 = conv_batch_act(num_filters, kernel_size=(1, 1), dilation_rate=1)
        self.cba3 = conv_batch_act(num_filters, kernel_size=(1, 1), dilation_rate=1)
        self.cba3_3 = conv_batch_act(num_filters, kernel_size=(3, 3), dilation_rate=3)
        self.cba3_6 = conv_batch_act(num_filters, kernel_size=(3, 3), dilation_rate=6)
        self.cba3_12 = conv_batch_act(num_filters, kernel_size=(3, 3), dilation_rate=12)

    def __call__(self, input):
        outs = [self.cba(input), self.cba3_3(input), self.cba3_6(input), self.cba3_12(input)]
        _unique('concatenate')
        node = E.Node('concat', outs)
        x = node.out(sum(t.channels for t in outs), input.down)
        return self.cba3(x)

    call = __call__


# ------------------------------------------------------------------ model construction
def build_unet_layers(input_tensor, filters=[32, 64, 128, 256, 512], factors=[2, 2, 2, 2, 2], dropout=None, double_conv=False):
    """Create U-Net layers (utils/model_tools.py:321-379)."""
    assert len(filters) == len(factors), 'number of filters and factors must be equal'
    levels = len(filters)
    net = {}
    encoder_pool = input_tensor
    for i, filt in enumerate(filters):
        factor = factors[i]
        encoder = encoder_block(filt, pool_size=(factor, factor), name=f'encoder_{i}', double_conv=double_conv)
        encoder_pool, encoded = encoder(encoder_pool)
        if i == 0 and dropout is not None:
            encoder_pool = _dropout(encoder_pool, dropout, True)          # :350-351
        net[f'encoder{i}'] = encoded
        net[f'encoder_pool{i}'] = encoder_pool
    conv = conv_block(filters[-1] * 2, double_conv=double_conv)
    center = conv(net[f'encoder_pool{levels - 1}'])
    decoder = _dropout(center, dropout, False) if dropout is not None else center   # :362-365
    for j in range(levels - 1, -1, -1):
        decoder = decoder_block(decoder, net[f'encoder{j}'], filters[j], up_size=(factors[j], factors[j]),
                                dropout=dropout if j == 0 else None)      # :373-377
    return decoder


class _Head:
    """layers.Conv2D(nclasses,(1,1),activation=...) fused with the argmax / threshold Lambda."""

    def __init__(self, nclasses, activation, bias_initializer, name):
        self.name = name
        self.params = _ConvParams(name, nclasses, (1, 1), bias_initializer=bias_initializer)
        self.activation = activation

    def __call__(self, x):
        self.params.build(x.channels)
        node = E.Node('head', [x], layer=self.params, activation=self.activation)
        return node.out(self.params.filters, x.down, self.name)


def _classes(probs, name, thresh=None):
    if thresh is not None:
        probs.node.attrs['thresh'] = float(thresh)
    node = E.Node('classes', [probs])
    return node.out(1, probs.down, name)


def get_unet_model(nclasses, nchannels, filters=[32, 64, 128, 256, 512], factors=[2, 2, 2, 2, 2], bias=None, dropout=None,
                   head_name: str = '', double_conv=False):
    """utils/model_tools.py:394-415: U-Net -> Conv2D(nclasses,1x1,softmax) 'probs' -> argmax int32
    f'{head_name}classes'.  Returns an UNCOMPILED Model(inputs, [probs, classes])."""
    bias_init = _Constant(bias) if bias is not None else None
    inputs = Input(shape=[None, None, nchannels])
    decoder = build_unet_layers(inputs, filters, factors, dropout=dropout, double_conv=double_conv)
    logit_input = _dropout(decoder, dropout, True) if dropout is not None else decoder
    probs = _Head(nclasses, 'softmax', bias_init, 'probs')(logit_input)
    classes = _classes(probs, f'{head_name}classes')
    model = Model(inputs=inputs, outputs=[probs, classes])
    model._builder = dict(fn='get_unet_model', nclasses=nclasses, nchannels=nchannels, filters=list(filters), factors=list(factors),
                          bias=bias, dropout=dropout, head_name=head_name, double_conv=double_conv)
    return model




# --------------------------------------------------------------------------- Siamese U-Net
def get_siamese_layers(input_a, input_b, filters=[32, 64, 128], factors=[2, 2, 2]):
    """utils/model_tools.py:576-636: shared-weight encoder on two dates, skips = concat([enc_b, enc_a]), shared ASPP on both
    pooled tensors, squeezed = concat([aspp_b, aspp_a]), then the ordinary decoder."""
    assert len(filters) == len(factors), 'filters and factors must be same length'
    levels = len(filters)
    net = {}
    pooled_a, pooled_b = input_a, input_b
    for i, filt in enumerate(filters):
        encoder = encoder_block(filt, pool_size=(factors[i], factors[i]), name=f'encoder_{i}')
        pooled_a, encoded_a = encoder(pooled_a)
        pooled_b, encoded_b = encoder(pooled_b)
        _unique('concatenate')
        n_cat = E.Node('concat', [encoded_b, encoded_a])
        net[f'encoder_{i}'] = n_cat.out(encoded_a.channels + encoded_b.channels, encoded_a.down)
    aspp = DilatedSpatialPyramidPooling(filters[-1] * 2)
    aspp_a = aspp(pooled_a)
    aspp_b = aspp(pooled_b)
    _unique('concatenate')
    n_sq = E.Node('concat', [aspp_b, aspp_a])
    decoder = n_sq.out(aspp_a.channels + aspp_b.channels, aspp_a.down)
    for j in range(levels - 1, -1, -1):
        decoder = decoder_block(decoder, net[f'encoder_{j}'], filters[j], up_size=(factors[j], factors[j]))
    return decoder


def make_siamese_unet(n_channels, filters, factors, bias=None, class_thresh=0.5):
    """utils/model_tools.py:638-663: two inputs (T2 image a, T1 image b), sigmoid 1x1 head 'probs', classes = probs > class_thresh."""
    bias_init = _Constant(bias) if bias is not None else None
    input_a = Input((None, None, n_channels))
    input_b = Input((None, None, n_channels))
    decoder = get_siamese_layers(input_a, input_b, filters=filters, factors=factors)
    probs = _Head(1, 'sigmoid', bias_init, 'probs')(decoder)
    classes = _classes(probs, 'classes', thresh=class_thresh)
    return Model(inputs=[input_a, input_b], outputs=[probs, classes])

# ------------------------------------------------------------- DeepLab-v3 (build-defined)
def _add_relu(y, shortcut):
    node = E.Node('add_relu', [y, shortcut])
    return node.out(y.channels, y.down)


def _bottleneck(x, width, stride=1, dilation=1, downsample=False):
    """ResNet-50 bottleneck: 1x1 -> 3x3 (stride / dilation) -> 1x1 (BN, no ReLU) + shortcut -> ReLU."""
    y = conv_batch_act(width, (1, 1))(x)
    y = conv_batch_act(width, (3, 3), dilation_rate=dilation, strides=stride)(y)
    y = conv_batch_act(4 * width, (1, 1), activation=None)(y)
    sc = conv_batch_act(4 * width, (1, 1), strides=stride, activation=None)(x) if downsample else x
    return _add_relu(y, sc)


def get_deeplabv3_model(nclasses, nchannels=4, aspp_filters=256, blocks=(3, 4, 6, 3), widths=(64, 128, 256, 512), head_name: str = ''):
    """DeepLab-v3 with a ResNet-50 backbone at output stride 16 and the reference's ASPP block (BASELINE config 3).

    The reference only NAMES this model (README.md:8): nothing of it exists under utils/.  It is therefore
    build-defined: ResNet-50 v1.5 (stride on the 3x3, last stage dilated by 2),
    a 4-band stem for NAIP RGBN tiles, DilatedSpatialPyramidPooling (utils/model_tools.py:533-574) on the
    stride-16 features, a 1x1 classifier, x16 bilinear up-sampling, softmax and argmax.  Inference only."""
    inputs = Input(shape=[None, None, nchannels])
    x = conv_batch_act(64, (7, 7), strides=2)(inputs)
    n_mp = E.Node('maxpool', [x], k=3, s=2, pad=1)
    x = n_mp.out(x.channels, x.down / 2)
    strides, dilations = (1, 2, 2, 1), (1, 1, 1, 2)
    for stage, (nb, wdt) in enumerate(zip(blocks, widths)):
        for b in range(nb):
            x = _bottleneck(x, wdt, stride=strides[stage] if b == 0 else 1, dilation=dilations[stage], downsample=(b == 0))
    x = DilatedSpatialPyramidPooling(aspp_filters)(x)
    logits = _Head(nclasses, 'linear', None, 'logits')(x)
    n_up = E.Node('upsample_head', [logits], factor=16, activation='softmax')
    probs = n_up.out(nclasses, Fraction(1), 'probs')
    classes = _classes(probs, f'{head_name}classes')
    model = Model(inputs=inputs, outputs=[probs, classes])
    model._infer_splitk = os.environ.get('SATCV_DEEPLAB_SPLITK', '1') == '1'
    model._builder = dict(fn='get_deeplabv3_model', nclasses=nclasses, nchannels=nchannels, aspp_filters=aspp_filters, blocks=list(blocks),
                          widths=list(widths), head_name=head_name)
    return model

def get_autoencoder(depth, optim=None, loss=None, mets=None, filters=[32, 64, 128, 256, 512], factors=[2, 2, 2, 2, 2]):
    """utils/model_tools.py:496-531: the five-level U-Net with a LINEAR 1x1 output 'continuous' (regression; the notebooks pair it
    with mse_4d).  The reference body calls `encoder_block(inputs, 32)` as if the layer class were a function and cannot run as
    coded; this is the network its comments describe, built from the same blocks as get_unet_model.  Compiled when `optim` is given."""
    inputs = Input(shape=[None, None, depth])
    decoder0 = build_unet_layers(inputs, filters, factors)
    preds = _Head(1, 'linear', 'zeros', 'continuous')(decoder0)
    model = Model(inputs=[inputs], outputs=[preds])
    if optim is not None:
        model.compile(optimizer=optim, loss={'continuous': loss} if loss is not None else None, metrics=mets)
    return model


# --------------------------------------------------------------------------- atrous CNN family
def _raw(t):
    """The output of a conv_batch_act block BEFORE its BatchNormalization."""
    node = E.Node('raw', [t])
    return node.out(t.channels, t.down)


def build_acnn_layers(input_tensor, depth, nfilters, nclasses):
    """utils/model_tools.py:922-939, as coded: from the second block on, `Conv2D_{l}_1` is applied to `feats` -- the OUTPUT OF THE
    PREVIOUS Conv2D, not of its BatchNormalization/ReLU -- and the `BN_{l}_2` / `relu_{l}_2` of every block but the last are dead
    code (functional-API layers that reach no output own no weights in the Keras model).  `depth` = 1 fails like the reference
    (`relu` is never bound)."""
    c0 = conv_batch_act(nfilters, (3, 3), conv_name='Conv2D_0_1', bn_name='BN_0')
    features_add = c0(input_tensor)
    feats = _raw(features_add)
    relu = None
    for layer in range(1, depth):
        norm = conv_batch_act(nfilters, (3, 3), activation=None, conv_name=f'Conv2D_{layer}_1', bn_name=f'BN_{layer}_1')(feats)
        features_add = _add_relu(norm, features_add)
        last = layer == depth - 1
        c2 = conv_batch_act(nfilters, (3, 3), dilation_rate=3, conv_name=f'Conv2D_{layer}_2', bn_name=f'BN_{layer}_2', batch_norm=last)
        relu = c2(features_add)
        feats = relu if not last else None
    if relu is None:
        raise NameError("name 'relu' is not defined")            # utils/model_tools.py:938 with depth < 2
    return _Head(nclasses, 'softmax', 'zeros', 'probabilities')(relu)


def build_acnn_layers2(feature_in, n_blocks=16, kernel_size=3, feature_num=16):
    """utils/model_tools.py:941-979: n_blocks x [Conv -> BN -> (+ previous sum) -> ReLU ; dilated(3) Conv -> BN -> ReLU]."""
    features, features_add = feature_in, None
    for layer in range(0, n_blocks):
        c1 = conv_batch_act(feature_num, kernel_size, activation='relu' if layer == 0 else None, conv_name=f'Conv{layer}_1', bn_name=f'bn{layer}_1')
        normed = c1(features)
        features_add = normed if layer == 0 else _add_relu(normed, features_add)
        features = conv_batch_act(feature_num, kernel_size, dilation_rate=3, conv_name=f'DilateConv{layer}_2', bn_name=f'bn{layer}_2')(features_add)
    return features


def get_acnn_model(nclasses, nfilters, nchannels, depth):
    """utils/model_tools.py:981-990: Model(inputs, softmax probabilities) -- a single output, no class map."""
    acnn_input = Input((None, None, nchannels))
    logits = build_acnn_layers(acnn_input, depth=depth, nfilters=nfilters, nclasses=nclasses)
    model = Model(inputs=acnn_input, outputs=logits)
    model._builder = dict(fn='get_acnn_model', nclasses=nclasses, nfilters=nfilters, nchannels=nchannels, depth=depth)
    return model


def get_acnn_model2(nclasses, nchannels, nfilters=16, depth=16):
    """utils/model_tools.py:992-1014."""
    input = Input((None, None, nchannels))
    features = build_acnn_layers2(feature_in=input, n_blocks=depth, kernel_size=3, feature_num=nfilters)
    logits = _Head(nclasses, 'softmax', 'zeros', 'probs')(features)
    m = Model(inputs=input, outputs=logits)
    m._builder = dict(fn='get_acnn_model2', nclasses=nclasses, nchannels=nchannels, nfilters=nfilters, depth=depth)
    return m


# ---------------------------------------------------------------------------- losses
class LossSpec:
    def __init__(self, kind, weights=None, eps=1e-6):
        self.kind, self.eps = kind, float(eps)
        self.weights = None if weights is None else np.asarray(weights, np.float32).reshape(-1)


class _LossArg:
    """placeholder handed to a user loss function so that it can describe itself"""

    def __init__(self, what):
        self.what = what


def _resident_f32(x, like):
    """True when a caller's batch tensor can be read in place of the plan's staging tensor: float32, contiguous, same shape and device."""
    return (x.dtype == torch.float32 and x.is_contiguous() and x.device == like.device and tuple(x.shape) == tuple(like.shape)
            and x.data_ptr() % 16 == 0)


def _y_ptr(plan):
    y = getattr(plan, 'y_src', None)
    return y.data_ptr() if y is not None else plan.y_true.data_ptr()


def _eager_loss(kind, y_true, y_pred, weights, activation='softmax', eps=1e-6):
    dev = torch.device('cuda')
    p = torch.as_tensor(np.asarray(y_pred, np.float32)).to(dev).contiguous()
    t = torch.as_tensor(np.asarray(y_true, np.float32)).to(dev).contiguous()
    w = None if weights is None else torch.as_tensor(np.asarray(weights, np.float32).reshape(-1)).to(dev)
    loss, _ = ops.loss_fwd_bwd(kind, p, t, w, activation, eps=eps)
    return float(loss.item())


def weighted_categorical_crossentropy(target, output, weights, axis=-1):
    """utils/model_tools.py:25-40 (reduced to its mean, as Keras does for a loss function)."""
    if isinstance(output, _LossArg):
        return LossSpec('weighted_categorical_crossentropy', weights)
    return _eager_loss('weighted_categorical_crossentropy', target, output, weights)


def weighted_bce(y_true, y_pred, pos_weight, logits=False):
    """utils/model_tools.py:96-112 (probability form; logits=True is not on the fused path)."""
    if logits:
        raise NotImplementedError('weighted_bce(logits=True): the model heads output probabilities')
    if isinstance(y_pred, _LossArg):
        return LossSpec('weighted_bce', [pos_weight])
    return _eager_loss('weighted_bce', y_true, y_pred, [pos_weight])


def gen_dice(y_true, y_pred, eps=1e-6, global_weights=None):
    """utils/model_tools.py:42-94: generalised Dice over (b, h*w, classes), mean over the batch.
    With global_weights=None the per-image class weights are 1/count^2 (eps where a class is absent) --
    the documented intent; the reference's own reduction axis (:80) does not broadcast."""
    if isinstance(y_pred, _LossArg):
        return LossSpec('gen_dice', global_weights if global_weights else None, eps)
    return _eager_loss('gen_dice', y_true, y_pred, global_weights if global_weights else None, eps=eps)


def iou_loss(true, pred):
    """utils/model_tools.py:131-140: 1 - sum(t*p) / sum(t + (1-t)*p) over the whole batch."""
    if isinstance(pred, _LossArg):
        return LossSpec('iou_loss')
    return _eager_loss('iou_loss', true, pred, None)


def mse_4d(y_true, y_pred, eps=1e-6):
    """utils/model_tools.py:142-166: mean squared error over the finite elements."""
    if isinstance(y_pred, _LossArg):
        return LossSpec('mse_4d')
    return _eager_loss('mse_4d', y_true, y_pred, None)


# ------------------------------------------------------------- optimizers / metrics
class _LR:
    def __init__(self, opt):
        self._opt = opt

    def numpy(self):
        return np.float32(self._opt._lr)

    def assign(self, v):
        self._opt._set_lr(float(v))

    def __float__(self):
        return float(self._opt._lr)


class Adam:
    """tf.keras.optimizers.Adam: epsilon 1e-7 outside the bias correction."""

    def __init__(self, learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7, **kwargs):
        self._lr = float(kwargs.get('lr', learning_rate))
        self.beta_1, self.beta_2, self.epsilon = beta_1, beta_2, epsilon
        self._rt = None

    def _set_lr(self, v):
        self._lr = v
        if self._rt is not None:
            self._rt.adam_state[0:1].fill_(v)

    @property
    def learning_rate(self):
        return _LR(self)

    @learning_rate.setter
    def learning_rate(self, v):
        self._set_lr(float(v))

    lr = learning_rate


class MeanIoU:
    """tf.keras.metrics.MeanIoU(num_classes): mean over classes of TP/(TP+FP+FN), on class ids."""

    def __init__(self, num_classes, name='mean_io_u'):
        self.num_classes, self.name = num_classes, name


class History:
    def __init__(self):
        self.history = defaultdict(list)
        self.epoch = []


class ModelCheckpoint:
    """tf.keras.callbacks.ModelCheckpoint (nb:1228-1235); `.best` is mutable as retrain_model needs (:1168)."""

    def __init__(self, filepath, monitor='val_loss', verbose=0, save_best_only=False, save_weights_only=False, mode='auto', **kw):
        self.filepath, self.monitor, self.verbose = filepath, monitor, verbose
        self.save_best_only, self.save_weights_only = save_best_only, save_weights_only
        if mode == 'auto':
            mode = 'max' if ('acc' in monitor or 'io_u' in monitor or 'iou' in monitor) else 'min'
        self.mode = mode
        self.best = -np.inf if mode == 'max' else np.inf
        self.model = None

    def on_epoch_end(self, epoch, logs):
        cur = logs.get(self.monitor)
        path = self.filepath.format(epoch=epoch + 1, **logs)
        if self.save_best_only:
            if cur is None:
                return
            better = cur > self.best if self.mode == 'max' else cur < self.best
            if not better:
                return
            self.best = cur
        (self.model.save_weights if self.save_weights_only else self.model.save)(path)


class TensorBoard:
    """tf.keras.callbacks.TensorBoard(log_dir): epoch scalars as real TensorBoard event files -- `<log_dir>/train` and
    `<log_dir>/validation`, tags `epoch_<name>` like Keras (notebooks/UNET_G4G_2019_solar.ipynb:1255) -- written without
    TensorFlow (tfrecord_io.EventFileWriter), plus one JSON line per epoch in `<log_dir>/scalars.jsonl`."""

    def __init__(self, log_dir='logs', **kw):
        self.log_dir = log_dir
        self.model = None
        self._writers = {}

    def _writer(self, which):
        from . import tfrecord_io
        if which not in self._writers:
            self._writers[which] = tfrecord_io.EventFileWriter(os.path.join(self.log_dir, which))
        return self._writers[which]

    def on_epoch_end(self, epoch, logs):
        os.makedirs(self.log_dir, exist_ok=True)
        with open(os.path.join(self.log_dir, 'scalars.jsonl'), 'a') as f:
            f.write(json.dumps(dict(epoch=epoch, **{k: float(v) for k, v in logs.items()})) + '\n')
        train = {f'epoch_{k}': float(v) for k, v in logs.items() if not k.startswith('val_')}
        val = {f'epoch_{k[4:]}': float(v) for k, v in logs.items() if k.startswith('val_')}
        if train:
            self._writer('train').scalars(epoch, train)
        if val:
            self._writer('validation').scalars(epoch, val)


# ----------------------------------------------------------------------------- Model
def _as_batches(x, y, batch_size, order=None):
    """ndarray pair / Sequence / iterable of (x, y) -> generator of batches (possibly endless).  order: optional permutation of the
    samples of array inputs (Keras' fit(shuffle=True) draws a new one every epoch)."""
    multi = isinstance(x, (list, tuple)) and len(x) > 0 and all(isinstance(a, (np.ndarray, torch.Tensor)) for a in x)
    if multi or y is not None or isinstance(x, (np.ndarray, torch.Tensor)):
        n = x[0].shape[0] if multi else x.shape[0]
        bs = batch_size or 32

        def take(a, i):
            if order is None:
                return a[i:i + bs]
            idx = order[i:i + bs]
            return a[torch.as_tensor(idx, device=a.device)] if isinstance(a, torch.Tensor) else a[idx]

        def gen():
            for i in range(0, n, bs):
                xb = [take(a, i) for a in x] if multi else take(x, i)
                yield (xb, take(y, i)) if y is not None else xb
        return gen(), (n + bs - 1) // bs
    if hasattr(x, '__getitem__') and hasattr(x, '__len__'):          # keras.utils.Sequence

        def gen():
            for i in range(len(x)):
                yield x[i]
        return gen(), len(x)
    return iter(x), None


def _prefetch_to_device(batches, dev):
    """Host (ndarray) batches of fit / evaluate -> float32 device tensors uploaded ONE BATCH AHEAD on a copy stream.

    A batch of 64 tiles is 67 MB of features + 34 MB of one-hot labels: staged on the compute stream the upload ran between two steps
    (host-fed training 6.1 k tiles/s against 8.1 k resident, tools/pcie_probe.py).  Here batch i + 1 is uploaded into the other of two
    device slots right after step i has been enqueued, so the DMA runs under step i's kernels; the consumer reads the resident tensors in
    place (Model._stage_x / _stage_y).  Batches that already live on the device, and anything that is not an (x, y) pair of arrays,
    pass through untouched.  SATCV_PREFETCH=0 turns it off."""
    it = iter(batches)
    try:
        first = next(it)
    except StopIteration:
        return

    def is_host_pair(b):
        if not (isinstance(b, (tuple, list)) and len(b) >= 2):
            return False
        xs = b[0] if isinstance(b[0], (list, tuple)) else [b[0]]
        return all(isinstance(a, np.ndarray) for a in xs) and isinstance(b[1], np.ndarray)

    if os.environ.get('SATCV_PREFETCH', '1') == '0' or not torch.cuda.is_available() or not is_host_pair(first):
        yield first
        yield from it
        return
    main = torch.cuda.current_stream()
    cs = torch.cuda.Stream()
    slots = [dict(x=None, y=None, done=None, free=None), dict(x=None, y=None, done=None, free=None)]

    def upload(b, s):
        xs = list(b[0]) if isinstance(b[0], (list, tuple)) else [b[0]]
        with torch.cuda.stream(cs):
            if s['free'] is not None:
                cs.wait_event(s['free'])             # the step that read this slot has run
            if s['x'] is None or len(s['x']) != len(xs) or any(tuple(t.shape) != a.shape for t, a in zip(s['x'], xs)):
                s['x'] = [torch.empty(a.shape, dtype=torch.float32, device=dev) for a in xs]
            if s['y'] is None or tuple(s['y'].shape) != b[1].shape:
                s['y'] = torch.empty(b[1].shape, dtype=torch.float32, device=dev)
            for t, a in zip(s['x'], xs):
                t.copy_(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)), non_blocking=True)
            s['y'].copy_(torch.from_numpy(np.ascontiguousarray(b[1], dtype=np.float32)), non_blocking=True)
            s['done'] = torch.cuda.Event()
            s['done'].record(cs)
        s['multi'] = isinstance(b[0], (list, tuple))
        s['rest'] = tuple(b[2:])

    cur = 0
    upload(first, slots[0])
    while True:
        s = slots[cur]
        main.wait_event(s['done'])
        yield ((s['x'] if s['multi'] else s['x'][0]), s['y']) + s['rest']
        s['free'] = torch.cuda.Event()
        s['free'].record(main)                       # (recorded after the consumer has enqueued its step)
        try:
            nb = next(it)
        except StopIteration:
            return
        if not is_host_pair(nb):                     # a mixed stream: hand the rest through as it is
            yield nb
            yield from it
            return
        cur ^= 1
        upload(nb, slots[cur])


class Model:
    """models.Model(inputs, outputs) with the subset of the Keras API the reference's callers use."""

    def __init__(self, inputs, outputs, name='model', dtype=None):
        self.inputs = list(inputs) if isinstance(inputs, (list, tuple)) else [inputs]
        self.outputs = list(outputs) if isinstance(outputs, (list, tuple)) else [outputs]
        self._single_output = not isinstance(outputs, (list, tuple))
        self.name = name
        self.nodes = E.topo_nodes(self.outputs)
        seen, self.layers, self.param_specs = set(), [], []
        for node in self.nodes:
            lay = node.layer
            subs = [lay] if lay is not None else []
            if node.op == 'cba':
                subs = [l for l in (lay, node.attrs['owner'].bn_layer) if l is not None]
            for l in subs:
                if id(l) not in seen:
                    seen.add(id(l))
                    self.layers.append(l)
                    self.param_specs += l.specs
        self.compute_dtype = dtype or _DEFAULT_DTYPE
        self.bn_bessel = True       # tf.keras (TF 2.x) BatchNormalization on NHWC input runs the fused kernel, whose running variance is the Bessel-corrected batch variance; False = Keras 3 / unfused behaviour
        self.fuse_head_bn_bwd = os.environ.get('SATCV_FUSE_HEAD_BN_BWD', '1') != '0'   # head backward also does the reduce pass of the last BN
        self.fuse_dgrad_bn_bwd = os.environ.get('SATCV_FUSE_DGRAD_BN_BWD', '1') != '0'  # data-gradient epilogues do the reduce pass of the BN below them
        self.wgrad_side_stream = os.environ.get('SATCV_WGRAD_STREAM', '1') != '0'      # weight gradients on a second HIP stream
        self.fuse_pool_bn_sums = os.environ.get('SATCV_FUSE_POOL_BN_SUMS', '1') != '0'  # encoder BN-backward sums formed by the producers of its gradients
        self.fuse_head_grad = os.environ.get('SATCV_FUSE_HEAD_GRAD', '1') != '0'      # the block under the head forms the head's data gradient in its loader
        self.fuse_pool_bwd = os.environ.get('SATCV_FUSE_POOL_BWD', '1') != '0'         # encoder blocks: pooled BN apply + weight (+ data) gradient in one launch
        self.fuse_thin_bwd = os.environ.get('SATCV_FUSE_THIN_BWD', '1') != '0'         # thin layers: BN-backward apply + data + weight gradient in one launch
        self.sync_bn = os.environ.get('SATCV_SYNC_BN', '0') == '1'      # data parallel: BatchNorm statistics over ALL replicas (parallel.py)
        self._rt = None
        self.optimizer, self._loss, self._metrics = None, None, []
        self.metrics_names = []
        self.stop_training = False
        self._builder = None
        self.output_names = [t.name for t in self.outputs]

    # ---- runtime
    @property
    def runtime(self):
        if self._rt is None:
            self._rt = E.Runtime(self, BF16 if self.compute_dtype == 'bfloat16' else F32)
            if self.optimizer is not None:
                self.optimizer._rt = self._rt
                self._rt.adam_state[0:1].fill_(self.optimizer._lr)
        return self._rt

    def count_params(self):
        return sum(p.size for p in self.param_specs)

    def get_layer(self, name):
        for l in self.layers:
            if l.name == name:
                return l
        raise ValueError(f'No such layer: {name}')

    def get_weights_dict(self):
        rt = self.runtime
        torch.cuda.synchronize()
        return {p.name: rt.get_param(p.name).detach().cpu().numpy().copy() for p in self.param_specs}

    def get_weights(self):
        """Keras list form: the variables in layer order (per layer: kernel, bias / gamma, beta, moving_mean, moving_variance)."""
        w = self.get_weights_dict()
        return [w[ps.name] for ps in self.param_specs]

    def set_weights(self, weights):
        weights = list(weights)
        if len(weights) != len(self.param_specs):
            raise ValueError(f'You called `set_weights(weights)` with a weight list of length {len(weights)}, but the model was expecting '
                             f'{len(self.param_specs)} weights.')
        self.set_weights_dict({ps.name: np.asarray(v) for ps, v in zip(self.param_specs, weights)})

    def set_weights_dict(self, d, skip_mismatch=False):
        rt = self.runtime
        for k, v in d.items():
            if k.endswith('/moving_variance'):               # tf.keras variable name (tools/keras_to_npz.py)
                k = k[:-len('moving_variance')] + 'moving_var'
            if k not in rt.specs:
                if skip_mismatch:
                    continue
                raise KeyError(k)
            if tuple(np.shape(v)) != rt.specs[k].shape:
                if skip_mismatch:
                    continue
                raise ValueError(f'{k}: shape {np.shape(v)} != {rt.specs[k].shape}')
            rt.set_param(k, v)
        rt.repack()
        if getattr(self, '_fp8_plans', None):
            self._fp8_plans = {}                             # fp8 weight images are re-quantised when the plans are rebuilt
        self._weights_version = getattr(self, '_weights_version', 0) + 1      # folded inference plans re-derive their arrays

    # ---- persistence.  Paths ending in .h5 / .hdf5 are Keras HDF5 files (hdf5_io.py: written and read without h5py / TensorFlow, in the
    # layer grouping tf.keras uses for this network so that the reference's own model can load them in layer order); anything else
    # is the own .npz container keyed '<layer>/<variable>' with Keras' auto-generated layer names.
    def _keras_layers(self):
        """[(top-level Keras layer name, [(variable name, array)])] in model order.  For get_unet_model graphs the encoder blocks
        and the centre block are the reference's custom layers (`encoder_{i}` / `conv_block`, utils/model_tools.py:348, 359) owning
        the six variables of their conv + BatchNormalization under nested scopes; every other layer is a plain Keras layer."""
        w = self.get_weights_dict()
        owner = {}
        try:
            sn = structural_names(self)
            for rn, pn in sn.items():
                blk = rn.split('.')[0]
                if blk.startswith('enc'):
                    owner[pn] = (f'encoder_{blk[3:]}', f'encoder_{blk[3:]}/conv_block/conv_batch_act/')
                elif blk == 'center':
                    owner[pn] = ('conv_block', 'conv_block/conv_batch_act/')
        except Exception:
            owner = {}
        groups, index = [], {}
        for lay in self.layers:
            for ps in lay.specs:
                top, prefix = owner.get(ps.name, (lay.name, ''))
                if top not in index:
                    index[top] = len(groups)
                    groups.append((top, []))
                var = ps.name.replace('/moving_var', '/moving_variance')
                groups[index[top]][1].append((f'{prefix}{var}:0', w[ps.name]))
        return groups

    @staticmethod
    def _is_h5_path(path):
        return str(path).endswith(('.h5', '.hdf5'))

    def save_weights(self, path):
        if self._is_h5_path(path):
            from . import hdf5_io
            return hdf5_io.write_keras_weights(path, self._keras_layers())
        np.savez(path if path.endswith('.npz') else path + '.npz', **self.get_weights_dict())

    def load_weights(self, path, by_name=False, skip_mismatch=False):
        """Own .npz container, or a Keras HDF5 file (`save_weights('x.h5')`, `save('x.h5')`, ModelCheckpoint .hdf5) read without
        h5py / TensorFlow (hdf5_io.py).  HDF5 semantics follow tf.keras (utils/model_tools.py:1162, 1196): by default the
        weight-bearing layers of the file are matched to the model's IN ORDER (names are ignored, shapes must agree);
        by_name=True matches variables by '<layer>/<variable>' name and ignores the rest, skip_mismatch then also skips
        variables whose shape differs."""
        from . import hdf5_io
        p = path if os.path.exists(path) else path + '.npz'
        if hdf5_io.is_hdf5(p):
            return self._load_keras_hdf5(hdf5_io.read_keras_weights(p), by_name, skip_mismatch)
        with np.load(p, allow_pickle=False) as z:
            self.set_weights_dict({k: z[k] for k in z.files if not k.startswith('__')}, skip_mismatch=skip_mismatch or by_name)

    def _load_keras_hdf5(self, layers, by_name, skip_mismatch):
        specs = {ps.name: ps for ps in self.param_specs}
        out = {}
        if by_name:
            for lname, ws in layers:
                for wn, arr in ws:
                    parts = wn.split(':')[0].split('/')
                    key = '/'.join(parts[-2:]) if len(parts) >= 2 else f'{lname}/{parts[-1]}'
                    key = key[:-len('moving_variance')] + 'moving_var' if key.endswith('/moving_variance') else key
                    if key not in specs:
                        continue
                    if tuple(arr.shape) != specs[key].shape:
                        if skip_mismatch:
                            continue
                        raise ValueError(f'{key}: shape {tuple(arr.shape)} in the file, {specs[key].shape} in the model')
                    out[key] = arr
        else:
            if skip_mismatch:
                raise ValueError('When calling model.load_weights, skip_mismatch can only be set to True when by_name is True.')
            arrays = [(lname, wn, arr) for lname, ws in layers for wn, arr in ws]
            mine = list(self.param_specs)
            if len(arrays) != len(mine):
                raise ValueError(f'the file holds {len(arrays)} weight arrays in {sum(1 for _, w in layers if w)} layers, the model '
                                 f'expects {len(mine)} in {len(self.layers)} layers')
            for (lname, wn, arr), ps in zip(arrays, mine):
                if tuple(arr.shape) != ps.shape:
                    raise ValueError(f'{lname}/{wn}: shape {tuple(arr.shape)} does not match {ps.name} {ps.shape} '
                                     '(weights are matched in layer order; use by_name=True for partial loads)')
                out[ps.name] = arr
        self.set_weights_dict({k: np.ascontiguousarray(v, dtype=np.float32) for k, v in out.items()})

    def save(self, path):
        rt = self.runtime
        if self._is_h5_path(path):
            from . import hdf5_io
            layers = self._keras_layers()
            cfg = {'class_name': 'Functional', 'config': {'name': self.name, 'layers': [{'name': n} for n, _ in layers]}}
            attrs = {'model_config': json.dumps(cfg), 'satcv_builder': json.dumps(self._builder or {}), 'keras_version': b'2.6.0', 'backend': b'tensorflow'}
            extra = None
            if rt.adam_m is not None:
                extra = {'optimizer_weights': [('satcv_adam/m:0', rt.adam_m.cpu().numpy()), ('satcv_adam/v:0', rt.adam_v.cpu().numpy()),
                                               ('satcv_adam/state:0', rt.adam_state.cpu().numpy())]}
            return hdf5_io.write_keras_weights(path, layers, root_attrs=attrs, model_weights_group=True, extra_groups=extra)
        d = self.get_weights_dict()
        d['__builder__'] = np.asarray(json.dumps(self._builder or {}))
        if rt.adam_m is not None:
            d['__adam_m__'], d['__adam_v__'] = rt.adam_m.cpu().numpy(), rt.adam_v.cpu().numpy()
            d['__adam_state__'] = rt.adam_state.cpu().numpy()
        np.savez(path if path.endswith('.npz') else path + '.npz', **d)

    # ---- compile
    def compile(self, optimizer='adam', loss=None, metrics=None, **kw):
        self.optimizer = Adam() if isinstance(optimizer, str) else optimizer
        if isinstance(loss, dict):           # Keras per-output form, e.g. loss={'logits': fn} (utils/model_tools.py:487, 526): one loss-bearing output
            unknown = [k for k in loss if k not in self.output_names]
            if unknown or len(loss) != 1:
                raise ValueError(f'loss dictionary must name exactly one of the outputs {self.output_names} (got {list(loss)})')
            loss = next(iter(loss.values()))
        if isinstance(metrics, dict):        # {'output name': [metrics]} -> flat list
            metrics = [mm for v in metrics.values() for mm in (v if isinstance(v, (list, tuple)) else [v])]
        if callable(loss):
            spec = loss(_LossArg('y_true'), _LossArg('y_pred'))
        elif isinstance(loss, LossSpec):
            spec = loss
        else:
            raise ValueError('loss must be one of the model_tools loss functions (or a lambda wrapping one)')
        if not isinstance(spec, LossSpec):
            raise ValueError('loss function did not resolve to a fused device loss')
        self._loss = spec
        self._metrics = list(metrics or [])
        names = ['loss']
        for mt in self._metrics:
            names.append(mt if isinstance(mt, str) else mt.name)
        self.metrics_names = names
        if self._rt is not None:
            self.optimizer._rt = self._rt
            self._rt.adam_state[0:1].fill_(self.optimizer._lr)

    # ---- inference
    def _head_plan(self, n, h, w, training):
        return self.runtime.plan(n, h, w, training)

    def enable_fp8_inference(self, calibration_tiles):
        """Switch predict / predict_on_device / predict_chips to the folded fp8 (e4m3) graph of fp8_infer.py.  The
        per-tensor activation scales come from one regular inference run on `calibration_tiles` (NHWC); the weights
        are re-quantised from the current fp32 parameters here, so call it again after training or load_weights."""
        from . import fp8_infer
        self._fp8_q = fp8_infer.calibrate(self, calibration_tiles)
        self._fp8_store = fp8_infer.FP8
        self._fp8_plans = {}
        return self

    def enable_folded_inference(self):
        """bf16 inference on the folded graph of fp8_infer.py (BatchNorm + bias in the conv epilogues, activations written once,
        no quantisation): plain U-Net graphs only."""
        from . import fp8_infer
        self._fp8_q, self._fp8_store, self._fp8_plans = {}, fp8_infer.BF16, {}
        return self

    def disable_fp8_inference(self):
        self._fp8_q = None
        self._fp8_plans = {}

    disable_folded_inference = disable_fp8_inference

    def _infer_plan(self, n, h, w):
        """inference launch list: the fp8 graph when enabled, else the folded bf16 graph (BatchNorm in the conv epilogues: 12 %
        faster than the training-style plan) for bf16 models it can lower, else the regular plan (fp32 parity mode, ASPP /
        Siamese / DeepLab graphs)."""
        from . import fp8_infer
        key = (n, h, w)
        if getattr(self, '_fp8_q', None) is not None:
            if key not in self._fp8_plans:
                self._fp8_plans[key] = fp8_infer.Fp8Plan(self, n, h, w, self._fp8_q, store=self._fp8_store)
            return self._fp8_plans[key]
        if self.compute_dtype == 'bfloat16' and getattr(self, '_folded_ok', True) and os.environ.get('SATCV_FOLDED_INFER', '1') != '0':
            plans = self.__dict__.setdefault('_folded_plans', {})
            ver = getattr(self, '_weights_version', 0)
            if key not in plans or plans[key].weights_version != ver:       # BN / bias / weight images are baked in at build time
                try:
                    plans[key] = fp8_infer.Fp8Plan(self, n, h, w, None, store=fp8_infer.BF16)
                    plans[key].weights_version = ver
                except NotImplementedError:
                    self._folded_ok = False
                    return self._head_plan(n, h, w, False)
            return plans[key]
        return self._head_plan(n, h, w, False)

    def _stage_x(self, plan, xb):
        xs = list(xb) if isinstance(xb, (list, tuple)) else [xb]
        if len(xs) != len(self.inputs):
            raise ValueError(f'model expects {len(self.inputs)} input array(s), got {len(xs)}')
        src = getattr(plan, 'x_src', None)
        for t, x in zip(self.inputs, xs):
            dst = plan.x_by_tid[t.id]
            if src is not None:
                src.pop(t.id, None)
            if isinstance(x, torch.Tensor):
                if src is not None and _resident_f32(x, dst):
                    # a float32 batch already resident on this device is read in place by the ingest kernel (no staging copy);
                    # the reference is held until the next batch is staged
                    src[t.id] = x.data_ptr()
                    plan._x_hold = getattr(plan, '_x_hold', {})
                    plan._x_hold[t.id] = x
                else:
                    dst.copy_(x.to(torch.float32), non_blocking=True)
            else:
                dst.copy_(torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)), non_blocking=True)

    def _shape_of(self, xb):
        """(n, h, w, c) of a batch (of the first input for multi-input models); Keras-style ValueError for anything that is not a
        non-empty NHWC batch with the channel counts the model was built for."""
        xs = list(xb) if isinstance(xb, (list, tuple)) else [xb]
        if len(xs) != len(self.inputs):
            raise ValueError(f'model expects {len(self.inputs)} input array(s), got {len(xs)}')
        for t, x in zip(self.inputs, xs):
            shp = tuple(x.shape)
            if len(shp) != 4:
                raise ValueError(f'input {t.name}: expected a 4-D (batch, height, width, channels) array, got shape {shp}')
            if shp[0] == 0:
                raise ValueError('Expect x to be a non-empty array or dataset.')
            if shp[3] != t.channels:
                raise ValueError(f'input {t.name}: expected {t.channels} channels (shape (None, None, None, {t.channels})), got shape {shp}')
            if shp != tuple(xs[0].shape[:3]) + (shp[3],):
                raise ValueError(f'inputs disagree in batch / spatial size: {tuple(xs[0].shape)} vs {shp}')
        return tuple(xs[0].shape)

    def predict_on_device(self, xb):
        """xb: (n,h,w,c) ndarray or device tensor -> list of device tensors [probs, classes] (no host sync)."""
        n, h, w, _ = self._shape_of(xb)
        plan = self._infer_plan(n, h, w)
        self._stage_x(plan, xb)
        sk = getattr(self, '_infer_splitk', False) and n * h * w <= 2 * 512 * 512      # (measured: +4 % on one 512 x 512 tile, -3 % on four)
        if sk:       # (build-defined DeepLab: single tiles leave the deep 1x1 launches on 16-128 workgroups; kernel choice is baked into a captured graph)
            lib.satcv_set_option(b'splitk', 2)
        try:
            if not self._replay_graph(plan):
                plan.run_forward(ops.stream_ptr())
        finally:
            if sk:
                lib.satcv_set_option(b'splitk', int(os.environ.get('SATCV_SPLITK', '0')))
        return [plan.outputs[t.id] for t in self.outputs]

    # ---- hipGraph replay of launch-bound inference plans (satcv_graph_begin / _end / _launch, include/satcv.h)
    # A plan of many short kernels -- DeepLab-v3 / ResNet-50 at batch 1: ~170 dependent launches of a few microseconds each -- is bound by
    # the HOST: a ctypes launch costs 10-20 us, the kernels 2-6 us.  Such a plan (at least SATCV_INFER_GRAPH_MIN launches, default 64) is
    # captured once per (plan, input address) after a warm-up run and replayed with ONE call; the U-Net plans (~50 launches, GPU-bound
    # even at batch 1) stay eager.  SATCV_INFER_GRAPH=0 turns it off, =2 forces it for every inference plan.
    def _replay_graph(self, plan):
        mode = int(os.environ.get('SATCV_INFER_GRAPH', '1'))
        fwd = getattr(plan, 'fwd', None)
        if mode == 0 or fwd is None or getattr(plan, 'training', False) or getattr(plan, 'dropouts', None):
            return False
        if mode < 2 and len(fwd) < int(os.environ.get('SATCV_INFER_GRAPH_MIN', '64')):
            return False
        cache = plan.__dict__.setdefault('_graphs', {})
        key = tuple(sorted((getattr(plan, 'x_src', None) or {}).items()))      # (a resident batch is read in place: its address is part of the graph)
        ent = cache.get(key)
        cur = torch.cuda.current_stream()
        if ent is None:
            cache[key] = 'warm'                  # first use of this key: an eager run (per-kernel one-time set-up must not happen inside a capture)
            return False
        if ent == 'warm':
            if len(cache) > 8:                   # callers that pass a fresh tensor every time: stop capturing for this plan
                for v in cache.values():
                    if v not in ('warm', 'off'):
                        lib.satcv_graph_destroy(v)
                cache.clear()
                cache[key] = 'off'
                return False
            side = self.__dict__.setdefault('_capture_stream', torch.cuda.Stream())     # (the legacy default stream cannot be captured)
            side.wait_stream(cur)
            sp = C.c_void_p(side.cuda_stream)
            plan.run_forward(sp)                 # per-stream set-up (the split-K slabs of this stream) outside the capture: the graph then
            #                                      holds the same launches, bit for bit, as the eager list
            if lib.satcv_graph_begin(sp) != 0:
                cache[key] = 'off'
                return False
            h = C.c_void_p()
            try:
                plan.run_forward(sp)
            finally:
                rc = lib.satcv_graph_end(sp, C.byref(h))
            if rc != 0 or not h.value:
                cache[key] = 'off'
                return False
            cache[key] = ent = h.value
            cur.wait_stream(side)                # (the eager pass on the capture stream writes the plan's buffers: finish it before the replay)
        if ent == 'off':
            return False
        check(lib.satcv_graph_launch(ent, C.c_void_p(cur.cuda_stream)))
        return True

    def predict(self, x, batch_size=None, verbose=0, steps=None, **kw):
        """Model.predict semantics used by the reference (utils/prediction_tools.py:152, 251, 333, 515):
        inference-mode forward; ndarray or iterable of batches; list of arrays in output order."""
        multi = len(self.inputs) > 1
        whole = isinstance(x, (np.ndarray, torch.Tensor)) or (multi and isinstance(x, (list, tuple)) and all(isinstance(a_, (np.ndarray, torch.Tensor)) for a_ in x))
        if whole:
            self._shape_of(x)                    # whole-array input: reject empty / mis-shaped data before anything is staged
        batches, nb = _as_batches(x, None, batch_size or 32)
        # results go D2H asynchronously into page-locked host arrays (one per output, grown as batches arrive): no per-batch
        # synchronisation and no final concatenation; the copy of batch i overlaps the staging and compute of batch i+1
        pinned, filled = None, 0
        for i, xb in enumerate(batches):
            if steps is not None and i >= steps:
                break
            if isinstance(xb, (tuple, list)) and not multi:
                xb = xb[0]
            res = self.predict_on_device(xb)
            nb_i = res[0].shape[0]
            if pinned is None:
                total = self._shape_of(x)[0] if whole else (steps or nb or 1) * nb_i
                rows = max(total, nb_i)
                big = sum(rows * r[0].numel() * r.element_size() for r in res) >= (16 << 20)      # page-locking small buffers costs more than it saves
                pinned = [torch.empty((rows,) + tuple(r.shape[1:]), dtype=r.dtype, pin_memory=big) for r in res]
            if filled + nb_i > pinned[0].shape[0]:               # iterable of unknown length: grow geometrically
                torch.cuda.current_stream().synchronize()
                grown = [torch.empty((2 * (filled + nb_i),) + tuple(p_.shape[1:]), dtype=p_.dtype, pin_memory=p_.is_pinned()) for p_ in pinned]
                for g_, p_ in zip(grown, pinned):
                    g_[:filled].copy_(p_[:filled])
                pinned = grown
            for p_, r in zip(pinned, res):
                p_[filled:filled + nb_i].copy_(r, non_blocking=p_.is_pinned())
            filled += nb_i
            # the plan's output buffers are rewritten by the next batch: that launch is stream-ordered after the copy above
        if pinned is None:
            raise ValueError('Expect x to be a non-empty array or dataset.')
        torch.cuda.current_stream().synchronize()
        arrays = [p_[:filled].numpy() for p_ in pinned]
        return arrays[0] if self._single_output else arrays

    __call__ = predict_on_device

    # ---- training
    def _apply_trainable(self):
        rt = self.runtime
        frozen = [l for l in self.layers if not l.trainable]
        if not frozen:
            rt.lr_mul = None
            return
        mul = torch.ones_like(rt.pflat)
        for l in frozen:
            for p in l.specs:
                if p.name in rt.offsets:
                    mul[rt.offsets[p.name]:rt.offsets[p.name] + p.size] = 0
        rt.lr_mul = mul

    def _loss_launch(self, plan, st, grad_scale=1.0):
        rt = self.runtime
        h = plan.head
        if getattr(self, '_loss_w_spec', None) is not self._loss:
            self._loss_w = None if self._loss.weights is None else torch.as_tensor(self._loss.weights).to(rt.dev)
            self._loss_w_spec = self._loss
        kind = ops.LOSS_KINDS[self._loss.kind]
        if kind in (0, 2) and self._loss_w is not None and self._loss_w.numel() != h['ncls']:
            raise ValueError(f'{self._loss.kind} needs one weight per class')
        if kind >= 2:
            if not hasattr(plan, 'loss_ws'):
                plan.loss_ws = torch.zeros(plan.n * 3 * h['ncls'], dtype=torch.float32, device=rt.dev)
            check(lib.satcv_loss_global_fwd_bwd(kind, h['probs'].data_ptr(), _y_ptr(plan),
                                                self._loss_w.data_ptr() if self._loss_w is not None else None, h['ncls'], h['act'], plan.n,
                                                h['r'].h * h['r'].w, self._loss.eps, grad_scale, plan.loss_ws.data_ptr(),
                                                plan.loss_buf.data_ptr(), plan.dlogits.data_ptr(), st))
            return
        check(lib.satcv_loss_fwd_bwd(kind, h['probs'].data_ptr(), _y_ptr(plan), self._loss_w.data_ptr(), h['ncls'], h['act'],
                                     plan.n * h['r'].h * h['r'].w, grad_scale, plan.loss_buf.data_ptr(), plan.dlogits.data_ptr(), st))

    def _stage_y(self, plan, yb):
        if not hasattr(plan, 'y_true'):
            h = plan.head
            plan.y_true = torch.zeros(plan.n, h['r'].h, h['r'].w, h['ncls'], dtype=torch.float32, device=self.runtime.dev)
        plan.y_src = None
        if isinstance(yb, torch.Tensor):
            if _resident_f32(yb, plan.y_true):
                plan.y_src = yb                  # read in place by the loss / confusion kernels
            else:
                plan.y_true.copy_(yb.to(torch.float32), non_blocking=True)
        else:
            plan.y_true.copy_(torch.from_numpy(np.ascontiguousarray(yb, dtype=np.float32)), non_blocking=True)

    def train_step_device(self, xb, yb, sync_grads=None):
        """One optimisation step; returns the device loss scalar (no host sync).
        sync_grads: optional callable(flat_grad_tensor) run between backward and Adam
        (data-parallel all-reduce, see parallel.py)."""
        if self._loss is None:
            raise RuntimeError('compile() the model before fit/train')
        rt = self.runtime
        rt.ensure_adam()
        n, h, w, _ = self._shape_of(xb)
        plan = self._head_plan(n, h, w, True)
        if getattr(self, '_frozen_applied', None) != plan.frozen:        # `layer.trainable` changed since the update masks were built
            self._apply_trainable()
            self._frozen_applied = plan.frozen
        self._stage_x(plan, xb)
        self._stage_y(plan, yb)
        st = ops.stream_ptr()
        check(lib.satcv_zero2(rt.gflat.data_ptr(), rt.gflat.numel() * 4 // 16 * 16, plan.loss_buf.data_ptr(), 4, st))
        if rt.gflat.numel() % 4:
            rt.gflat[rt.gflat.numel() // 4 * 4:].zero_()
        plan.step_count += 1                     # fresh dropout masks every step
        plan.run_forward(st)
        self._loss_launch(plan, st)
        opt = self.optimizer
        # (single replica: the bulk of the optimizer step may run inside the backward pass, engine.Plan `early`)
        plan.eo_done = False
        plan.early_opt = dict(beta_1=opt.beta_1, beta_2=opt.beta_2, epsilon=opt.epsilon) if sync_grads is None else None
        plan.run_backward(st)
        if sync_grads is not None:
            sync_grads(rt.gflat)
        if plan.eo_done:
            lo = plan.eo_lo                  # [lo, n) was updated and repacked on the side stream (joined at the end of the backward pass)
            check(lib.satcv_adam_step_part(rt.pflat.data_ptr(), rt.gflat.data_ptr(), rt.adam_m.data_ptr(), rt.adam_v.data_ptr(), lo,
                                           opt.beta_1, opt.beta_2, opt.epsilon, rt.adam_state.data_ptr(),
                                           rt.lr_mul.data_ptr() if rt.lr_mul is not None else None, 1, st))
            rt.repack(0, lo)
        else:
            check(lib.satcv_adam_step(rt.pflat.data_ptr(), rt.gflat.data_ptr(), rt.adam_m.data_ptr(), rt.adam_v.data_ptr(), rt.pflat.numel(),
                                      opt.beta_1, opt.beta_2, opt.epsilon, rt.adam_state.data_ptr(),
                                      rt.lr_mul.data_ptr() if rt.lr_mul is not None else None, st))
            rt.repack()
        self._weights_version = getattr(self, '_weights_version', 0) + 1
        return plan

    def train_on_batch(self, x, y):
        plan = self.train_step_device(x, y, getattr(self, '_sync_grads', None))
        return float(plan.loss_buf.item())

    def _metric_values(self, conf, loss):
        vals = [loss]
        conf = conf.astype(np.float64)
        for mt in self._metrics:
            if isinstance(mt, str):
                vals.append(float(np.trace(conf) / max(conf.sum(), 1)))
            else:
                tp = np.diag(conf)
                denom = conf.sum(0) + conf.sum(1) - tp
                valid = denom > 0
                vals.append(float((tp[valid] / denom[valid]).mean()) if valid.any() else 0.0)
        return vals

    def _run_epoch(self, batches, steps, train):
        rt = self.runtime
        loss_sum = torch.zeros(1, dtype=torch.float32, device=rt.dev)          # sum of batch loss x batch size: Keras weights the epoch mean by samples
        conf, cnt = None, 0
        sync = getattr(self, '_sync_grads', None)
        if steps is not None:
            import itertools as _it
            batches = _it.islice(batches, steps)     # (never pull -- or upload -- a batch past the step limit: generators may be endless)
        for i, b in enumerate(_prefetch_to_device(batches, rt.dev)):
            if steps is not None and i >= steps:
                break
            xb, yb = b[0], b[1]
            if train:
                plan = self.train_step_device(xb, yb, sync)
            else:
                n, h, w, _ = self._shape_of(xb)
                plan = self._head_plan(n, h, w, False)
                self._stage_x(plan, xb)
                self._stage_y(plan, yb)
                st = ops.stream_ptr()
                if not hasattr(plan, 'loss_buf'):
                    plan.loss_buf = torch.zeros(1, dtype=torch.float32, device=rt.dev)
                    plan.dlogits = torch.zeros(plan.n * plan.head['r'].h * plan.head['r'].w, plan.head['ncls'], dtype=torch.float32, device=rt.dev)
                plan.loss_buf.zero_()
                plan.run_forward(st)
                self._loss_launch(plan, st)
            loss_sum += plan.loss_buf * float(plan.n)
            hd = plan.head
            if self._metrics and hd['act'] == 0:
                if conf is None:
                    conf = torch.zeros(hd['ncls'], hd['ncls'], dtype=torch.int64, device=rt.dev)
                check(lib.satcv_confusion(hd['classes'].data_ptr(), _y_ptr(plan), hd['ncls'], hd['classes'].numel(),
                                          conf.data_ptr(), ops.stream_ptr()))
            cnt += plan.n
        loss = float(loss_sum.item()) / max(cnt, 1)
        c = conf.cpu().numpy() if conf is not None else np.zeros((1, 1))
        return self._metric_values(c, loss)

    def fit(self, x=None, y=None, batch_size=None, epochs=1, verbose=1, callbacks=None, validation_data=None, steps_per_epoch=None,
            validation_steps=None, initial_epoch=0, shuffle=True, **kw):
        """Model.fit as the notebooks call it (notebooks/UNET_G4G_2019_solar.ipynb:1267-1275).  shuffle (Keras default True) applies to
        array inputs only, as in Keras: a new sample permutation every epoch (seeded by set_seed); generators / Sequences / datasets are
        consumed in their own order.  shuffle='batch' permutes whole batches.  Data-parallel callers shard the arrays BEFORE fit():
        every rank draws the same permutation from the shared seed, so ranks given the same full array would train on identical batches."""
        if isinstance(x, (np.ndarray, torch.Tensor)):
            self._shape_of(x)
            if y is not None and len(y) != len(x):
                raise ValueError(f'x and y hold different numbers of samples: {len(x)} vs {len(y)}')
        self._apply_trainable()
        hist = History()
        callbacks = list(callbacks or [])
        for cb in callbacks:
            cb.model = self
        it = None
        for epoch in range(initial_epoch, epochs):
            t0 = time.time()
            if it is None or steps_per_epoch is None:
                order = None
                if shuffle and (isinstance(x, (np.ndarray, torch.Tensor)) or (isinstance(x, (list, tuple)) and y is not None)):
                    nsamp = x[0].shape[0] if isinstance(x, (list, tuple)) else x.shape[0]
                    if shuffle == 'batch':          # Keras: shuffle in batch-sized chunks (whole batches change places, their contents do not)
                        bs_ = batch_size or 32
                        order = np.concatenate([np.arange(b * bs_, min((b + 1) * bs_, nsamp)) for b in _SHUFFLE_RNG.permutation(-(-nsamp // bs_))])
                    else:
                        order = _SHUFFLE_RNG.permutation(nsamp)
                it, _ = _as_batches(x, y, batch_size, order)
            vals = self._run_epoch(it, steps_per_epoch, True)
            logs = dict(zip(self.metrics_names, vals))
            if validation_data is not None:
                if isinstance(validation_data, tuple) and len(validation_data) == 2 and isinstance(validation_data[0], (np.ndarray, torch.Tensor)):
                    vit, _ = _as_batches(validation_data[0], validation_data[1], batch_size)
                else:
                    vit, _ = _as_batches(validation_data, None, batch_size)
                vvals = self._run_epoch(vit, validation_steps, False)
                logs.update({'val_' + k: v for k, v in zip(self.metrics_names, vvals)})
            for k, v in logs.items():
                hist.history[k].append(v)
            hist.epoch.append(epoch)
            if verbose:
                print(f'Epoch {epoch + 1}/{epochs} - {time.time() - t0:.1f}s - ' + ' - '.join(f'{k}: {v:.4f}' for k, v in logs.items()))
            for cb in callbacks:
                if hasattr(cb, 'on_epoch_end'):
                    cb.on_epoch_end(epoch, logs)
            if hasattr(x, 'on_epoch_end'):
                x.on_epoch_end()
            if self.stop_training:
                break
        return hist

    def evaluate(self, x=None, y=None, batch_size=None, verbose=0, steps=None, **kw):
        """Model.evaluate -> list aligned with metrics_names (utils/model_tools.py:1164-1168)."""
        it, _ = _as_batches(x, y, batch_size)
        vals = self._run_epoch(it, steps, False)
        return vals if len(vals) > 1 else vals[0]

    def summary(self):
        print(f'Model "{self.name}": {len(self.layers)} weighted layers, {self.count_params():,} parameters, dtype {self.compute_dtype}')
        for l in self.layers:
            print(f'  {l.name:40s} ' + ', '.join(f'{p.name.split("/")[-1]}{p.shape}' for p in l.specs))


def unet_config_from_keras_weights(layers):
    """Arguments of get_unet_model recovered from the weight shapes of a Keras file of that network (the model_config JSON of a
    `.h5` names custom layers that only the reference's own module can rebuild): filters = output widths of the encoder blocks,
    factors = kernel sizes of the transposed convolutions (up_size == pool size, utils/model_tools.py:350-372), nclasses /
    nchannels from the head and the first kernel."""
    kernels = [(ln, wn, a.shape) for ln, ws in layers for wn, a in ws if a.ndim == 4]
    ups = [sh for ln, wn, sh in kernels if 'transpose' in wn or 'transpose' in ln]
    L = len(ups)
    if L == 0 or len(kernels) != 3 * L + 2 + L or kernels[-1][2][:2] != (1, 1):
        raise ValueError('the weight file does not have the layer structure of get_unet_model (encoder blocks, centre, transposed '
                         'convolution + two conv blocks per level, 1x1 head)')
    filters = [kernels[i][2][3] for i in range(L)]
    factors = [ups[L - 1 - i][0] for i in range(L)]
    return dict(nclasses=kernels[-1][2][3], nchannels=kernels[0][2][2], filters=filters, factors=factors)


def load_model(path, custom_objects=None, compile=False):
    """models.load_model for files written by Model.save (own .npz container) and for Keras HDF5 files of a get_unet_model network
    (architecture recovered from the weight shapes, weights loaded in layer order)."""
    from . import hdf5_io
    p = path if os.path.exists(path) else path + '.npz'
    if hdf5_io.is_hdf5(p):
        layers = hdf5_io.read_keras_weights(p)
        with hdf5_io.File(p) as f:
            cfg = json.loads(f.attrs['satcv_builder']) if 'satcv_builder' in f.attrs else {}
            opt = {k: f['optimizer_weights/satcv_adam/' + k + ':0'].read() for k in ('m', 'v', 'state')} if 'optimizer_weights/satcv_adam/m:0' in f else None
        builders = {fn.__name__: fn for fn in (get_unet_model, get_deeplabv3_model, get_acnn_model, get_acnn_model2)}
        reset_uids()
        if cfg.get('fn') in builders:            # written by Model.save of this build: any of its network families
            m = builders[cfg.pop('fn')](**cfg)
        else:                                    # a tf.keras file of the reference's get_unet_model
            m = get_unet_model(**unet_config_from_keras_weights(layers))
        m._load_keras_hdf5(layers, False, False)
        if opt is not None:
            rt = m.runtime
            rt.ensure_adam()
            rt.adam_m.copy_(torch.from_numpy(opt['m'])); rt.adam_v.copy_(torch.from_numpy(opt['v'])); rt.adam_state.copy_(torch.from_numpy(opt['state']))
        return m
    with np.load(p, allow_pickle=False) as z:
        cfg = json.loads(str(z['__builder__']))
        builders = {f.__name__: f for f in (get_unet_model, get_deeplabv3_model, get_acnn_model, get_acnn_model2)}
        if cfg.get('fn') not in builders:
            raise ValueError(f'file does not describe a network of {sorted(builders)}')
        fn = builders[cfg.pop('fn')]
        reset_uids()
        m = fn(**cfg)
        m.set_weights_dict({k: z[k] for k in z.files if not k.startswith('__')})
        if '__adam_m__' in z.files:
            rt = m.runtime
            rt.ensure_adam()
            rt.adam_m.copy_(torch.from_numpy(z['__adam_m__']))
            rt.adam_v.copy_(torch.from_numpy(z['__adam_v__']))
            rt.adam_state.copy_(torch.from_numpy(z['__adam_state__']))
    return m


def retrain_model(model_file, checkpoint, eval_data, metric, weights_file=None, by_name=False, skip_mismatch=False, custom_objects=None,
                  lr=None, freeze=None):
    """utils/model_tools.py:1128-1176: load a saved model (a path -- own container or Keras .h5 -- or a Model object), optionally a
    separate weights file (`by_name` / `skip_mismatch` as in Model.load_weights), evaluate it on `eval_data`, seed
    `checkpoint.best` with the current value of `metric`, set the learning rate and optionally freeze all but the last layer.
    Returns (model, checkpoint) like the reference.  `freeze`: `m.layers` holds the WEIGHTED layers here, so `layers[:-1]` keeps the
    `probs` head trainable -- the documented intent ("freeze all but the last layer").  In tf.keras the last entry of `model.layers`
    of a `get_unet_model` network is the weightless `classes` Lambda (utils/model_tools.py:406), so the reference's identical slice
    (:1174-1175) freezes the head as well and leaves nothing to train; set `m.get_layer('probs').trainable = False` for that.
    A model restored from a file carries no loss: pass
    custom_objects={'compile': dict(optimizer=..., loss=..., metrics=[...])} to compile it here (tf.keras restores that from the
    file's training_config, which names Python functions this build cannot import)."""
    m = load_model(model_file, custom_objects=custom_objects) if isinstance(model_file, (str, os.PathLike)) else model_file
    if weights_file:
        if str(weights_file).startswith('https'):
            m = get_blob_weights(m=m, hdf5_url=weights_file, by_name=by_name, skip_mismatch=skip_mismatch)
        else:
            m.load_weights(weights_file, by_name=by_name, skip_mismatch=skip_mismatch)
    if custom_objects and 'compile' in custom_objects:
        m.compile(**custom_objects['compile'])
    if m._loss is None:
        raise RuntimeError("retrain_model: the model has no loss; pass custom_objects={'compile': {...}} or compile() it first")
    evalMetrics = m.evaluate(x=eval_data, verbose=1)
    evalMetrics = evalMetrics if isinstance(evalMetrics, list) else [evalMetrics]
    metrics = m.metrics_names
    print(metrics)
    checkpoint.best = evalMetrics[metrics.index(metric)]
    if lr:
        m.optimizer.learning_rate = lr
    if freeze:
        for layer in m.layers[:-1]:
            layer.trainable = False
    return m, checkpoint


def normalize_confusion_matrix(arr):
    """utils/model_tools.py:1111-1126: rows (label categories) scaled to sum to one, rounded to 4 decimals."""
    arr = np.asarray(arr)
    return np.around(arr / arr.sum(axis=1)[:, np.newaxis], decimals=4)


# --------------------------------------------------------------------------- chunk prediction (Dask map_overlap callers)
_BLOB_MODELS = {}        # (absolute path, mtime[, weights path, mtime]) -> Model; the reference downloads and rebuilds the model PER CHUNK


def _blob_path(url):
    """Local path of a model / weights file.  The Azure download (BlobClient, utils/model_tools.py:1225-1236) is storage plumbing
    outside this build: fetch the blob first and pass the path (or a file:// URL)."""
    if url.startswith('file://'):
        url = url[len('file://'):]
    elif '://' in url:
        raise RuntimeError(f'{url.split("://")[0]} URLs are not fetched by this build: download the blob and pass the local path')
    return url if os.path.exists(url) or not os.path.exists(url + '.npz') else url + '.npz'


def get_blob_model(h5_url=None, hdf5_url=None, custom_objects=None):
    """utils/model_tools.py:1204-1269 for files written by Model.save; the loaded model is cached per (path, mtime)."""
    url = h5_url or hdf5_url
    if not url:
        print('must provide a url to either an .h5 or .hdf5 file')
        return None
    p = _blob_path(url)
    key = (os.path.abspath(p), os.path.getmtime(p))
    if key not in _BLOB_MODELS:
        _BLOB_MODELS[key] = load_model(p, custom_objects=custom_objects, compile=False)
    return _BLOB_MODELS[key]


def get_blob_weights(m, hdf5_url=None, by_name=False, skip_mismatch=False):
    """utils/model_tools.py:1178-1202: load a separate weights file into an existing model."""
    m.load_weights(_blob_path(hdf5_url), by_name=by_name, skip_mismatch=skip_mismatch)
    return m


def predict_chunk(data, model_blob_url, weights_blob_url=None, custom_objects=None):
    """utils/model_tools.py:1271-1304: predictions for ONE (C, H, W) chunk of a Dask / xarray mosaic -> np.squeeze(pred[0]).
    (The reference passes `model_blob_url=` / `weights_blob_url=` to get_blob_model, whose parameters are `h5_url` / `hdf5_url`:
    a TypeError as coded.  Here the model file is `model_blob_url`, optionally overlaid with the weights of `weights_blob_url`;
    both stay cached between chunks instead of being fetched and rebuilt for each one.)"""
    print('input shape', data.shape)
    m = get_blob_model(h5_url=model_blob_url, custom_objects=custom_objects)
    if weights_blob_url:
        wp = _blob_path(weights_blob_url)
        wkey = (os.path.abspath(wp), os.path.getmtime(wp))
        if getattr(m, '_blob_weights_key', None) != wkey:
            get_blob_weights(m, wp)
            m._blob_weights_key = wkey
    hwc = np.moveaxis(data, 0, -1)
    nhwc = np.expand_dims(hwc, axis=0)          # the model expects 4-D data
    pred = m.predict(nhwc)
    logits = np.squeeze(pred[0])
    print('logits shape', logits.shape)
    return logits


def structural_names(model):
    """Map structural weight names (enc{i}.conv.kernel, enc{i}.bn.gamma, center.*, dec{j}.up.*,
    dec{j}.bn0.*, dec{j}.conv1/2.*, dec{j}.bn1/2.*, probs.*) to this model's parameter names, for a
    network built by get_unet_model / build_unet_layers (as coded, single conv per level).
    Used for weight interchange with a Keras model of the same topology."""
    cbas = [n for n in model.nodes if n.op == 'cba']
    ups = [n for n in model.nodes if n.op == 'convT']
    cats = [n for n in model.nodes if n.op == 'concat_bn_relu']
    head = [n for n in model.nodes if n.op == 'head'][0]
    L = len(ups)
    out = {}

    def conv(prefix, node):
        out[f'{prefix}.kernel'] = node.layer.name + '/kernel'
        out[f'{prefix}.bias'] = node.layer.name + '/bias'

    def bn(prefix, lname):
        for s in ('gamma', 'beta', 'moving_mean', 'moving_var'):
            out[f'{prefix}.{s}'] = f'{lname}/{s}'

    assert len(cbas) == 3 * L + 1, 'not an as-coded U-Net'
    for i in range(L):
        conv(f'enc{i}.conv', cbas[i])
        bn(f'enc{i}.bn', cbas[i].attrs['owner'].bn_layer.name)
    conv('center.conv', cbas[L])
    bn('center.bn', cbas[L].attrs['owner'].bn_layer.name)
    for q in range(L):
        j = L - 1 - q
        conv(f'dec{j}.up', ups[q])
        bn(f'dec{j}.bn0', cats[q].layer.name)
        for r in (1, 2):
            node = cbas[L + 1 + 2 * q + (r - 1)]
            conv(f'dec{j}.conv{r}', node)
            bn(f'dec{j}.bn{r}', node.attrs['owner'].bn_layer.name)
    conv('probs', head)
    return out


# ---- ConvLSTM2D family (utils/model_tools.py:666-920, 1016-1109): the builders live in lstm_tools.py (tape executor, time-major tensors) and
# are reachable under the reference's module name; resolved on first use because lstm_tools imports this module
_LSTM_NAMES = ('build_lstm_layers', 'build_lstm_layers2', 'get_lstm_model', 'get_lstm_autoencoder', 'get_hybrid_model',
               'get_hierarchical_model', 'ConvLSTM2D')


def __getattr__(name):
    if name in _LSTM_NAMES:
        from . import lstm_tools
        return getattr(lstm_tools, name)
    raise AttributeError(f'module {__name__!r} has no attribute {name!r}')
