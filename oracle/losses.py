"""NumPy restatement of the reference's loss functions, with gradients w.r.t. y_pred.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows
/root/reference/utils/model_tools.py:25-166 line by line; the tf.* primitives
used there (clip_by_value, reduce_mean, weighted_cross_entropy_with_logits) are
restated from their documented definitions.  PARITY
UNPINNED (no reference tests); gradients cross-checked by torch autograd in
tests/test_oracle_cpu.py.

Each function returns (loss_scalar, dloss/dy_pred).  Where the reference
returns a per-pixel tensor (weighted CCE) the scalar is its mean, which is what
Keras' `compile(loss=fn)` reduces it to.
"""
import numpy as np

K_EPSILON = 1e-7          # tf.keras.backend.epsilon()


def weighted_categorical_crossentropy(target, output, weights, axis=-1):
    """utils/model_tools.py:25-40.  Returns (mean loss, grad, per-pixel loss)."""
    w = np.asarray(weights, dtype=output.dtype).reshape(1, -1)
    s = output.sum(axis=axis, keepdims=True)
    o = output / s                                              # :35
    lo, hi = K_EPSILON, 1.0 - K_EPSILON
    oc = np.clip(o, lo, hi)                                     # :39
    per_pixel = -(w * target * np.log(oc)).sum(axis=axis)       # :40
    npix = per_pixel.size
    inside = (o >= lo) & (o <= hi)
    g_o = -(w * target / oc) * inside / npix                    # d mean / d o
    grad = (g_o - (g_o * o).sum(axis=axis, keepdims=True)) / s  # through o = output/sum
    return per_pixel.mean(), grad, per_pixel


def weighted_bce(y_true, y_pred, pos_weight, logits=False):
    """utils/model_tools.py:96-112."""
    if logits:
        x, z, q = y_pred, y_true, pos_weight
        # tf.nn.weighted_cross_entropy_with_logits (documented stable form)
        l = 1 + (q - 1) * z
        bce = (1 - z) * x + l * (np.log1p(np.exp(-np.abs(x))) + np.maximum(-x, 0))
        sig = 1.0 / (1.0 + np.exp(-x))
        grad = ((1 - z) - l * (1 - sig)) / bce.size
        return bce.mean(), grad
    lo, hi = 0.00001, 0.99999
    yp = np.clip(y_pred, lo, hi)                                # :110
    bce = y_true * -np.log(yp) * pos_weight + (1 - y_true) * -np.log(1 - yp)   # :111
    inside = (y_pred >= lo) & (y_pred <= hi)
    grad = (-y_true * pos_weight / yp + (1 - y_true) / (1 - yp)) * inside / bce.size
    return bce.mean(), grad                                     # :112


def gen_dice(y_true, y_pred, eps=1e-6, global_weights=None):
    """utils/model_tools.py:42-94.

    Note: with global_weights=None the reference reduces `counts` over axis -1
    of the (b, h*w, classes) tensor (:80), which yields a (b, h*w) weight that
    cannot broadcast against the (b, classes) sums at :90 -- the batch-wise
    branch raises as coded.  The oracle implements the documented intent
    ("count how many of each class are present in each image": axis=1).
    """
    b, h, w_, c = y_true.shape
    t = y_true.reshape(b, h * w_, c)
    p = y_pred.reshape(b, h * w_, c)
    if global_weights:
        wts = np.asarray(global_weights, dtype=y_pred.dtype).reshape(1, c)
    else:
        counts = t.sum(axis=1)
        with np.errstate(divide='ignore'):
            wts = 1.0 / (counts ** 2)
        wts = np.where(np.isfinite(wts), wts, eps)              # :83
    multed = (t * p).sum(axis=1)                                # :86
    summed = (t + p).sum(axis=1)                                # :87
    num = (wts * multed).sum(axis=-1)                           # :90
    den = (wts * summed).sum(axis=-1)                           # :91
    dices = 1.0 - 2.0 * num / den                               # :92
    wb = np.broadcast_to(wts, (b, c))[:, None, :]
    grad = -2.0 * wb * (t * den[:, None, None] - num[:, None, None]) / (den[:, None, None] ** 2) / b
    return dices.mean(), grad.reshape(y_pred.shape)


def iou_loss(true, pred):
    """utils/model_tools.py:131-140."""
    inter = (true * pred).sum()
    union = (true + (1 - true) * pred).sum()
    grad = -(true * union - inter * (1 - true)) / union ** 2
    return 1.0 - inter / union, grad


def mse_4d(y_true, y_pred, eps=1e-6):
    """utils/model_tools.py:142-166: mean of squared error over finite elements."""
    diff = np.square(y_pred - y_true)
    finite = np.isfinite(diff)
    cnt = finite.sum()
    loss = diff[finite].mean()
    grad = np.where(finite, 2.0 * (y_pred - np.where(finite, y_true, 0)) / cnt, 0.0)
    return loss, grad
