"""PyTorch-CPU restatement of the same U-Net graph (independent cross-check + CPU baseline).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Two uses:
  * tests/test_oracle_cpu.py: autograd of this graph must agree with the
    hand-written NumPy forward/backward of oracle/unet.py (two independent
    implementations of the Keras semantics of utils/model_tools.py:174-415);
  * bench.py `cpu_baseline` (kind "port"): TensorFlow is absent on the bench
    host, so the reference's TF-CPU path is stood in for by this oneDNN-backed
    graph, fp32, all host cores.
Semantic deltas handled here: Keras BN eps 1e-3, biased
batch variance, HWIO kernels -> OIHW, Conv2DTranspose (kh,kw,Cout,Cin) ->
torch (Cin,Cout,kh,kw), concat order skip-first, Keras Adam epsilon placement.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3


def params_to_torch(params, dtype=torch.float32, requires_grad=True):
    out = {}
    for k, v in params.items():
        t = torch.tensor(np.asarray(v), dtype=dtype)
        if requires_grad and not (k.endswith('moving_mean') or k.endswith('moving_var')):
            t.requires_grad_(True)
        out[k] = t
    return out


def _conv(x, k, b, dilation=1):
    kh = k.shape[0]
    pad = dilation * (kh - 1) // 2
    return F.conv2d(x, k.permute(3, 2, 0, 1), b, padding=pad, dilation=dilation)


def _bn(x, p, name, training):
    g, b = p[f'{name}.gamma'], p[f'{name}.beta']
    if training:
        mean = x.mean(dim=(0, 2, 3), keepdim=True)
        var = x.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
    else:
        mean = p[f'{name}.moving_mean'].view(1, -1, 1, 1)
        var = p[f'{name}.moving_var'].view(1, -1, 1, 1)
    return (x - mean) / torch.sqrt(var + BN_EPS) * g.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)


def unet_forward(p, x_nhwc, filters, factors, training=False):
    """Returns (probs NHWC, classes NHW int32).  p: dict of torch tensors (oracle names)."""
    x = x_nhwc.permute(0, 3, 1, 2)
    L = len(filters)
    skips = []
    h = x
    for i in range(L):
        a = F.relu(_bn(_conv(h, p[f'enc{i}.conv.kernel'], p[f'enc{i}.conv.bias']), p, f'enc{i}.bn', training))
        skips.append(a)
        h = F.max_pool2d(a, factors[i], factors[i])
    h = F.relu(_bn(_conv(h, p['center.conv.kernel'], p['center.conv.bias']), p, 'center.bn', training))
    for j in range(L - 1, -1, -1):
        kt = p[f'dec{j}.up.kernel']                      # (s,s,Cout,Cin)
        up = F.conv_transpose2d(h, kt.permute(3, 2, 0, 1), p[f'dec{j}.up.bias'], stride=factors[j])
        cat = torch.cat([skips[j], up], dim=1)
        a0 = F.relu(_bn(cat, p, f'dec{j}.bn0', training))
        a1 = F.relu(_bn(_conv(a0, p[f'dec{j}.conv1.kernel'], p[f'dec{j}.conv1.bias']), p, f'dec{j}.bn1', training))
        h = F.relu(_bn(_conv(a1, p[f'dec{j}.conv2.kernel'], p[f'dec{j}.conv2.bias']), p, f'dec{j}.bn2', training))
    logits = _conv(h, p['probs.kernel'], p['probs.bias'])
    probs = torch.softmax(logits, dim=1).permute(0, 2, 3, 1)
    return probs, torch.argmax(probs, dim=-1).to(torch.int32)


def weighted_cce_mean(target, output, weights):
    """mean of utils/model_tools.py:25-40."""
    w = torch.as_tensor(weights, dtype=output.dtype).view(1, -1)
    o = output / output.sum(-1, keepdim=True)
    o = torch.clamp(o, 1e-7, 1 - 1e-7)
    return (-(w * target * torch.log(o)).sum(-1)).mean()


def weighted_bce_mean(y_true, y_pred, pos_weight):
    yp = torch.clamp(y_pred, 0.00001, 0.99999)
    return (y_true * -torch.log(yp) * pos_weight + (1 - y_true) * -torch.log(1 - yp)).mean()


def keras_adam_(p, grads, m, v, t, lr=9e-4, b1=0.9, b2=0.999, eps=1e-7):
    alpha = lr * (1 - b2 ** t) ** 0.5 / (1 - b1 ** t)
    with torch.no_grad():
        for k, g in grads.items():
            m[k].mul_(b1).add_(g, alpha=1 - b1)
            v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
            p[k].sub_(alpha * m[k] / (v[k].sqrt() + eps))


def aspp_net_forward(p, x_nhwc, training=False):
    """Small encoder -> ASPP -> decoder -> softmax head used to check the ASPP block
    (DilatedSpatialPyramidPooling, utils/model_tools.py:533-574) inside a trainable graph.
    Parameter names: enc.*, aspp.{cba,cba3_3,cba3_6,cba3_12,cba3}.*, up.*, bn0.*, conv1.*, conv2.*, probs.*"""
    x = x_nhwc.permute(0, 3, 1, 2)

    def cba(name, t, d=1):
        return F.relu(_bn(_conv(t, p[f'{name}.kernel'], p[f'{name}.bias'], d), p, f'{name}.bn', training))
    enc = cba('enc', x)
    pooled = F.max_pool2d(enc, 2, 2)
    br = [cba('aspp.cba', pooled), cba('aspp.cba3_3', pooled, 3), cba('aspp.cba3_6', pooled, 6), cba('aspp.cba3_12', pooled, 12)]
    a = cba('aspp.cba3', torch.cat(br, dim=1))
    up = F.conv_transpose2d(a, p['up.kernel'].permute(3, 2, 0, 1), p['up.bias'], stride=2)
    cat = torch.cat([enc, up], dim=1)
    a0 = F.relu(_bn(cat, p, 'bn0', training))
    h = cba('conv2', cba('conv1', a0))
    logits = _conv(h, p['probs.kernel'], p['probs.bias'])
    probs = torch.softmax(logits, dim=1).permute(0, 2, 3, 1)
    return probs, torch.argmax(probs, dim=-1).to(torch.int32)


def deeplab_forward(plist, head, x_nhwc, blocks=(3, 4, 6, 3), widths=(64, 128, 256, 512)):
    """Build-defined DeepLab-v3 / ResNet-50 (output stride 16) + the reference's ASPP, inference mode.
    Mirrors satellite_computervision_amd.model_tools.get_deeplabv3_model: `plist` holds one dict
    (kernel HWIO, bias, gamma, beta, moving_mean, moving_var) per conv+BN in graph order, `head` = (kernel, bias)."""
    it = iter(plist)

    def cbn(t, stride=1, dilation=1, relu=True):
        q = next(it)
        k = q['kernel']
        pad = dilation * (k.shape[0] - 1) // 2
        y = F.conv2d(t, k.permute(3, 2, 0, 1), q['bias'], stride=stride, padding=pad, dilation=dilation)
        y = (y - q['moving_mean'].view(1, -1, 1, 1)) / torch.sqrt(q['moving_var'].view(1, -1, 1, 1) + BN_EPS) * q['gamma'].view(1, -1, 1, 1) \
            + q['beta'].view(1, -1, 1, 1)
        return F.relu(y) if relu else y
    x = x_nhwc.permute(0, 3, 1, 2)
    x = cbn(x, stride=2)
    x = F.max_pool2d(x, 3, 2, 1)
    strides, dil = (1, 2, 2, 1), (1, 1, 1, 2)
    for stage, nb in enumerate(blocks):
        for b in range(nb):
            s = strides[stage] if b == 0 else 1
            y = cbn(x)
            y = cbn(y, stride=s, dilation=dil[stage])
            y = cbn(y, relu=False)
            sc = cbn(x, stride=s, relu=False) if b == 0 else x
            x = F.relu(y + sc)
    br = [cbn(x), cbn(x, dilation=3), cbn(x, dilation=6), cbn(x, dilation=12)]
    x = cbn(torch.cat(br, dim=1))
    logits = F.conv2d(x, head[0].permute(3, 2, 0, 1), head[1])
    logits = F.interpolate(logits, scale_factor=16, mode='bilinear', align_corners=False)
    probs = torch.softmax(logits, dim=1).permute(0, 2, 3, 1)
    return probs, torch.argmax(probs, dim=-1).to(torch.int32)


def siamese_forward(p, xa_nhwc, xb_nhwc, filters, factors, training=False):
    """Siamese U-Net of utils/model_tools.py:576-663 (shared encoder + shared ASPP, sigmoid head).
    Parameter names: enc{i}.*, aspp.{cba,cba3_3,cba3_6,cba3_12,cba3}.*, dec{j}.{up,bn0,conv1,conv2}.*, probs.*"""
    def cba(name, t, d=1):
        return F.relu(_bn(_conv(t, p[f'{name}.kernel'], p[f'{name}.bias'], d), p, f'{name}.bn', training))

    def aspp(t):
        br = [cba('aspp.cba', t), cba('aspp.cba3_3', t, 3), cba('aspp.cba3_6', t, 6), cba('aspp.cba3_12', t, 12)]
        return cba('aspp.cba3', torch.cat(br, dim=1))
    a, b = xa_nhwc.permute(0, 3, 1, 2), xb_nhwc.permute(0, 3, 1, 2)
    skips = []
    for i in range(len(filters)):
        ea, eb = cba(f'enc{i}', a), cba(f'enc{i}', b)
        skips.append(torch.cat([eb, ea], dim=1))                      # Concatenate([encoded_b, encoded_a]) (:608)
        a, b = F.max_pool2d(ea, factors[i], factors[i]), F.max_pool2d(eb, factors[i], factors[i])
    h = torch.cat([aspp(b), aspp(a)], dim=1)                          # Concatenate([aspp_b, aspp_a]) (:624)
    for j in range(len(filters) - 1, -1, -1):
        up = F.conv_transpose2d(h, p[f'dec{j}.up.kernel'].permute(3, 2, 0, 1), p[f'dec{j}.up.bias'], stride=factors[j])
        a0 = F.relu(_bn(torch.cat([skips[j], up], dim=1), p, f'dec{j}.bn0', training))
        h = cba(f'dec{j}.conv2', cba(f'dec{j}.conv1', a0))
    probs = torch.sigmoid(_conv(h, p['probs.kernel'], p['probs.bias'])).permute(0, 2, 3, 1)
    return probs


def _cbn(p, conv, bn, t, training, d=1):
    """Conv2D `conv` then BatchNormalization `bn` (Keras '<layer>/<variable>' parameter names)."""
    return _bn(_conv(t, p[f'{conv}/kernel'], p[f'{conv}/bias'], d), {k.replace('/', '.'): v for k, v in p.items() if k.startswith(bn + '/')}, bn, training)


def acnn_forward(p, x_nhwc, depth, training=False):
    """build_acnn_layers (utils/model_tools.py:922-939) as coded: Conv2D_{l}_1 consumes the previous Conv2D's output (not its
    BN/ReLU); only the last block's BN_{l}_2 / ReLU reaches the 'probabilities' head.  Parameter names are the Keras layer names."""
    x = x_nhwc.permute(0, 3, 1, 2)
    feats = _conv(x, p['Conv2D_0_1/kernel'], p['Conv2D_0_1/bias'])
    features_add = F.relu(_bn(feats, {k.replace('/', '.'): v for k, v in p.items()}, 'BN_0', training))
    for layer in range(1, depth):
        feats = _conv(feats, p[f'Conv2D_{layer}_1/kernel'], p[f'Conv2D_{layer}_1/bias'])
        norm = _bn(feats, {k.replace('/', '.'): v for k, v in p.items()}, f'BN_{layer}_1', training)
        features_add = F.relu(norm + features_add)
        feats = _conv(features_add, p[f'Conv2D_{layer}_2/kernel'], p[f'Conv2D_{layer}_2/bias'], 3)
    relu = F.relu(_bn(feats, {k.replace('/', '.'): v for k, v in p.items()}, f'BN_{depth - 1}_2', training))
    return torch.softmax(_conv(relu, p['probabilities/kernel'], p['probabilities/bias']), dim=1).permute(0, 2, 3, 1)


def acnn2_forward(p, x_nhwc, depth, training=False):
    """get_acnn_model2 / build_acnn_layers2 (utils/model_tools.py:941-1014)."""
    q = {k.replace('/', '.'): v for k, v in p.items()}
    features = x_nhwc.permute(0, 3, 1, 2)
    features_add = None
    for layer in range(depth):
        normed = _bn(_conv(features, p[f'Conv{layer}_1/kernel'], p[f'Conv{layer}_1/bias']), q, f'bn{layer}_1', training)
        features_add = F.relu(normed if layer == 0 else normed + features_add)
        features = F.relu(_bn(_conv(features_add, p[f'DilateConv{layer}_2/kernel'], p[f'DilateConv{layer}_2/bias'], 3), q, f'bn{layer}_2', training))
    return torch.softmax(_conv(features, p['probs/kernel'], p['probs/bias']), dim=1).permute(0, 2, 3, 1)
