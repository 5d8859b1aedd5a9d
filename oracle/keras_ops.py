"""NumPy restatement of the tf.keras layer arithmetic the reference's hot path calls.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED for these
ops: the arithmetic lives in TensorFlow (un-vendored, un-pinned dependency of
/root/reference/utils/model_tools.py:8-15); semantics follow the documented
Keras defaults.  Cross-checked against PyTorch-CPU in
tests/test_oracle_cpu.py.

All tensors are NHWC, kernels HWIO (Conv2D) / HWOI (Conv2DTranspose), dtype is
whatever the caller passes (float64 for golden vectors, float32 for speed).
Every forward has a hand-written backward so the oracle can check the HIP
training path without autograd.
"""
import numpy as np


# --------------------------------------------------------------------------- conv
def _im2col(x, kh, kw, d):
    """'same' zero padding, stride 1, dilation d -> (N,H,W,kh*kw*C)."""
    n, h, w, c = x.shape
    ph, pw = d * (kh - 1) // 2, d * (kw - 1) // 2
    xp = np.zeros((n, h + 2 * ph, w + 2 * pw, c), dtype=x.dtype)
    xp[:, ph:ph + h, pw:pw + w, :] = x
    cols = np.empty((n, h, w, kh * kw, c), dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            cols[:, :, :, i * kw + j, :] = xp[:, i * d:i * d + h, j * d:j * d + w, :]
    return cols.reshape(n, h, w, kh * kw * c)


def conv2d_same(x, k, b=None, dilation=1):
    """layers.Conv2D(f, k, padding='same', dilation_rate=d) -- utils/model_tools.py:178.

    Cross-correlation, HWIO kernel, symmetric zero pad d*(k-1)/2, bias added.
    """
    kh, kw, ci, co = k.shape
    assert kh % 2 == 1 and kw % 2 == 1
    cols = _im2col(x, kh, kw, dilation)
    y = cols.reshape(-1, kh * kw * ci) @ k.reshape(kh * kw * ci, co)
    y = y.reshape(x.shape[0], x.shape[1], x.shape[2], co)
    if b is not None:
        y = y + b
    return y


def conv2d_same_bwd(x, k, dy, dilation=1):
    """Returns (dx, dk, db) of conv2d_same."""
    kh, kw, ci, co = k.shape
    n, h, w, _ = x.shape
    cols = _im2col(x, kh, kw, dilation).reshape(-1, kh * kw * ci)
    dyf = dy.reshape(-1, co)
    dk = (cols.T @ dyf).reshape(kh, kw, ci, co)
    db = dyf.sum(0)
    # dx = 'same' correlation of dy with the spatially flipped, io-swapped kernel
    kf = k[::-1, ::-1, :, :].transpose(0, 1, 3, 2)
    dx = conv2d_same(dy, np.ascontiguousarray(kf), None, dilation)
    return dx, dk, db


# ---------------------------------------------------------------------- batch norm
def batchnorm_train(x, gamma, beta, eps=1e-3):
    """layers.BatchNormalization() in training mode -- utils/model_tools.py:179.

    axis=-1; biased batch variance over (N,H,W); returns (y, mean, var).
    """
    mean = x.mean(axis=(0, 1, 2))
    var = x.var(axis=(0, 1, 2))
    xhat = (x - mean) / np.sqrt(var + eps)
    return gamma * xhat + beta, mean, var


def batchnorm_infer(x, gamma, beta, moving_mean, moving_var, eps=1e-3):
    return gamma * (x - moving_mean) / np.sqrt(moving_var + eps) + beta


def batchnorm_train_bwd(x, gamma, mean, var, dy, eps=1e-3):
    """Returns (dx, dgamma, dbeta) for training-mode BN."""
    m = x.shape[0] * x.shape[1] * x.shape[2]
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (x - mean) * rstd
    dbeta = dy.sum(axis=(0, 1, 2))
    dgamma = (dy * xhat).sum(axis=(0, 1, 2))
    dx = gamma * rstd * (dy - dbeta / m - xhat * dgamma / m)
    return dx, dgamma, dbeta


def bn_update_moving(moving_mean, moving_var, mean, var, momentum=0.99,
                     count=None, bessel=False):
    """moving <- moving*momentum + batch*(1-momentum).  `bessel` selects the
    TF2 fused-BN behaviour (unbiased variance in the moving average); default
    biased (Keras 3) -- SURVEY.md Appendix A, version-dependent, unpinned."""
    v = var * (count / max(count - 1.0, 1.0)) if bessel else var
    return (moving_mean * momentum + mean * (1 - momentum),
            moving_var * momentum + v * (1 - momentum))


# --------------------------------------------------------------------- activations
def relu(x):
    return np.maximum(x, 0)


def relu_bwd(y, dy):
    return dy * (y > 0)


def softmax(x):
    z = x - x.max(axis=-1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=-1, keepdims=True)


def softmax_bwd(p, dp):
    return p * (dp - (dp * p).sum(axis=-1, keepdims=True))


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def argmax_classes(p):
    """tf.cast(tf.math.argmax(x, -1), int32): ties -> lowest index (model_tools.py:406)."""
    return np.argmax(p, axis=-1).astype(np.int32)


# -------------------------------------------------------------------------- pooling
def maxpool(x, f):
    """layers.MaxPooling2D((f,f), strides=(f,f)) 'valid' -- utils/model_tools.py:281."""
    n, h, w, c = x.shape
    ho, wo = h // f, w // f
    xw = x[:, :ho * f, :wo * f, :].reshape(n, ho, f, wo, f, c)
    xw = xw.transpose(0, 1, 3, 2, 4, 5).reshape(n, ho, wo, f * f, c)
    return xw.max(axis=3)


def maxpool_bwd(x, f, dy):
    """Gradient routed to the FIRST arg-max of each window (row-major)."""
    n, h, w, c = x.shape
    ho, wo = h // f, w // f
    xw = x[:, :ho * f, :wo * f, :].reshape(n, ho, f, wo, f, c)
    xw = xw.transpose(0, 1, 3, 2, 4, 5).reshape(n, ho, wo, f * f, c)
    am = xw.argmax(axis=3)
    onehot = (np.arange(f * f).reshape(1, 1, 1, f * f, 1) == am[:, :, :, None, :])
    dw = onehot * dy[:, :, :, None, :]
    dw = dw.reshape(n, ho, wo, f, f, c).transpose(0, 1, 3, 2, 4, 5).reshape(n, ho * f, wo * f, c)
    dx = np.zeros_like(x)
    dx[:, :ho * f, :wo * f, :] = dw
    return dx


# ---------------------------------------------------------------- transposed conv
def conv2d_transpose_ks(x, k, b=None):
    """layers.Conv2DTranspose(f, s, strides=s, padding='same') -- model_tools.py:306.

    kernel (s,s,Cout,Cin); kernel==stride so windows do not overlap:
    out[n, y*s+i, x*s+j, o] = sum_c x[n,y,x,c] * k[i,j,o,c] + b[o].
    """
    s, s2, co, ci = k.shape
    assert s == s2
    n, h, w, _ = x.shape
    y = x.reshape(-1, ci) @ k.reshape(s * s * co, ci).T          # (NHW, s*s*co)
    y = y.reshape(n, h, w, s, s, co).transpose(0, 1, 3, 2, 4, 5).reshape(n, h * s, w * s, co)
    if b is not None:
        y = y + b
    return y


def conv2d_transpose_ks_bwd(x, k, dy):
    s, _, co, ci = k.shape
    n, h, w, _ = x.shape
    d = dy.reshape(n, h, s, w, s, co).transpose(0, 1, 3, 2, 4, 5).reshape(-1, s * s * co)
    dx = (d @ k.reshape(s * s * co, ci)).reshape(n, h, w, ci)
    dk = (d.T @ x.reshape(-1, ci)).reshape(s, s, co, ci)
    db = dy.sum(axis=(0, 1, 2))
    return dx, dk, db
