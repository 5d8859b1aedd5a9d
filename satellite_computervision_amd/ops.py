"""Operator-level Python wrappers over the C ABI (torch tensors are only the device-memory
container: every wrapper passes raw `data_ptr()`s and the current HIP stream to libsatcv).

Each wrapper names the Keras layer call site of /root/reference/utils/model_tools.py that
the underlying kernel replaces.  There is no CPU path: tensors must live on a ROCm device.
"""
import ctypes as C
import torch

from . import _lib
from ._lib import lib, check, ConvDesc, WgradDesc, BnBwdDesc, HeadDesc, F32, BF16, FP8, FP8X, STAT_ROWS

TORCH_DTYPE = {F32: torch.float32, BF16: torch.bfloat16, FP8: torch.float8_e4m3fn, FP8X: torch.float8_e4m3fn}
DTYPE_CODE = {torch.float32: F32, torch.bfloat16: BF16, torch.float8_e4m3fn: FP8}


def rup(a, b):
    return (a + b - 1) // b * b


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda, 'satcv ops need device tensors (no CPU fallback)'
    return C.c_void_p(t.data_ptr())


def new_stats(c, device):
    return torch.zeros(STAT_ROWS, 2, c, dtype=torch.float64, device=device)      # satcv_stat_t rows


# ------------------------------------------------------------------ data movement
def ingest_nhwc(x_f32, cpad, dtype):
    """f32 NHWC -> storage dtype NHWC with channels zero-padded to cpad."""
    n, h, w, c = x_f32.shape
    x_f32 = x_f32.contiguous()
    out = torch.empty(n, h, w, cpad, dtype=TORCH_DTYPE[dtype], device=x_f32.device)
    check(lib.satcv_ingest_nhwc(ptr(x_f32), ptr(out), n * h * w, c, cpad, dtype, stream_ptr()))
    return out


def ingest_chw(planes, scale, cpad, dtype):
    """planar (n,c,h,w) uint8/uint16(int16 view)/float32 -> storage NHWC, value*scale."""
    n, c, h, w = planes.shape
    kind = {torch.uint8: 0, torch.int16: 1, torch.uint16: 1, torch.float32: 2}[planes.dtype]
    out = torch.empty(n, h, w, cpad, dtype=TORCH_DTYPE[dtype], device=planes.device)
    check(lib.satcv_ingest_chw(ptr(planes.contiguous()), kind, float(scale), ptr(out), n, c, h, w, cpad, dtype, stream_ptr()))
    return out


def packed_sizes(kh, kw, cin, cout, cin_pad, transposed):
    """(elements of fwd pack, elements of dgrad pack, N pad of fwd, N pad of dgrad)."""
    taps = kh * kw
    if not transposed:
        nf, nd = rup(cout, 32), rup(cin, 32)
        return taps * cin_pad * nf, taps * rup(cout, 16) * nd, nf, nd
    nf, nd = rup(taps * cout, 32), rup(cin, 32)
    return cin_pad * nf, rup(taps * cout, 16) * nd, nf, nd


def pack_weights(kernel_f32, cin_pad, dtype, transposed=False, want_dgrad=True, out_fwd=None, out_dgrad=None):
    """Keras kernel (kh,kw,cin,cout) [or Conv2DTranspose (kh,kw,cout,cin)] -> MFMA operand images."""
    if transposed:
        kh, kw, cout, cin = kernel_f32.shape
    else:
        kh, kw, cin, cout = kernel_f32.shape
    ef, ed, _, _ = packed_sizes(kh, kw, cin, cout, cin_pad, transposed)
    td = TORCH_DTYPE[dtype]
    fwd = out_fwd if out_fwd is not None else torch.empty(ef, dtype=td, device=kernel_f32.device)
    dg = out_dgrad if out_dgrad is not None else (torch.empty(ed, dtype=td, device=kernel_f32.device) if want_dgrad else None)
    check(lib.satcv_pack_weights(ptr(kernel_f32.contiguous()), ptr(fwd), ptr(dg), kh, kw, cin, cout, cin_pad,
                                 1 if transposed else 0, dtype, stream_ptr()))
    return fwd, dg


# --------------------------------------------------------------------------- conv
def make_conv_desc(*, x0, c0, w, y, ldy, n, h, w_, cout, cout_pad, dtype, x1=None, c1=0, in_scale=None, in_shift=None,
                   in_relu=0, bias=None, stats=None, stats_ld=0, kh=3, kw=3, dil=1, mode_in=0, mode_out=0, f=1,
                   cstat=None, out_relu=0, accumulate=0, stride=1, hin=0, win=0, out_scale=None, pool_y=None, pool_ld=0, pool_f=0, bst=None, tile_policy=0):
    d = ConvDesc()
    d.tile_policy = int(tile_policy)
    d.x0, d.x1, d.c0, d.c1 = x0, x1, c0, c1
    d.in_scale, d.in_shift, d.in_relu = in_scale, in_shift, int(in_relu)
    d.w, d.bias, d.y, d.ldy = w, bias, y, ldy
    d.stats, d.stats_ld = stats, stats_ld
    d.n, d.h, d.w_ = n, h, w_
    d.cout, d.cout_pad = cout, cout_pad
    d.kh, d.kw, d.dil = kh, kw, dil
    d.mode_in, d.mode_out, d.f = mode_in, mode_out, f
    d.cstat = cstat if cstat is not None else cout
    d.out_relu, d.dtype, d.accumulate = int(out_relu), dtype, int(accumulate)
    d.stride, d.hin, d.win, d.out_scale = stride, hin, win, out_scale
    d.pool_y, d.pool_ld, d.pool_f = pool_y, pool_ld, pool_f
    if bst:       # fused BatchNorm-backward reduce of the layer whose activation gradient this launch writes (satcv.h: bst_*)
        d.bst_y, d.bst_ld, d.bst_y1, d.bst_ld1, d.bst_split = bst['y'], bst['ld'], bst.get('y1'), bst.get('ld1', 0), bst.get('split', 0)
        d.bst_scale, d.bst_shift, d.bst_mean, d.bst_rstd, d.bst_relu = bst['scale'], bst['shift'], bst['mean'], bst['rstd'], int(bst.get('relu', 1))
    return d


def _p(t):
    return t.data_ptr() if t is not None else None


def conv2d(x, w_packed, cout, *, kh=3, kw=3, dil=1, bias=None, x1=None, in_scale=None, in_shift=None, in_relu=False,
           stats=None, out=None, out_relu=False, stride=1, bst=None, out_scale=None, pool_y=None, pool_f=0):
    """layers.Conv2D(cout,(kh,kw),padding='same',dilation_rate=dil) (utils/model_tools.py:178) on
    NHWC storage tensors; optional fused input BatchNorm-affine+ReLU and output sum/sumsq.  stride > 1 (ResNet backbone of the
    build-defined DeepLab): symmetric padding dil*(k-1)/2, output (h-1)//stride+1."""
    n, hin, win, c0 = x.shape
    h, w_ = ((hin - 1) // stride + 1, (win - 1) // stride + 1) if stride > 1 else (hin, win)
    dtype = DTYPE_CODE[x.dtype]
    c1 = x1.shape[-1] if x1 is not None else 0
    cpad = rup(cout, 32)
    y = out if out is not None else torch.empty(n, h, w_, rup(cout, 16), dtype=x.dtype, device=x.device)
    d = make_conv_desc(x0=_p(x), c0=c0, x1=_p(x1), c1=c1, w=_p(w_packed), y=_p(y), ldy=y.shape[-1], n=n, h=h, w_=w_,
                       cout=cout, cout_pad=cpad, dtype=dtype, in_scale=_p(in_scale), in_shift=_p(in_shift),
                       in_relu=in_relu, bias=_p(bias), stats=_p(stats), stats_ld=stats.shape[-1] if stats is not None else 0,
                       kh=kh, kw=kw, dil=dil, out_relu=out_relu, stride=stride, hin=hin if stride > 1 else 0, win=win if stride > 1 else 0,
                       out_scale=_p(out_scale), pool_y=_p(pool_y), pool_ld=pool_y.shape[-1] if pool_y is not None else 0, pool_f=pool_f,
                       bst={k: (_p(v) if torch.is_tensor(v) else v) for k, v in bst.items()} if bst else None)
    if bst and lib.satcv_conv2d_igemm_pipelined(C.byref(d)) != 1:
        raise ValueError('conv2d: this shape cannot carry the fused BatchNorm-backward reduce (bst); run satcv_bn_bwd_reduce separately')
    check(lib.satcv_conv2d_igemm(C.byref(d), stream_ptr()))
    return y


def conv2d_transpose(x, w_packed, cout, f, *, bias=None, in_scale=None, in_shift=None, in_relu=False, stats=None, out=None,
                     out_scale=None, out_relu=False):
    """layers.Conv2DTranspose(cout, f, strides=f, padding='same') (utils/model_tools.py:306)."""
    n, h, w_, c0 = x.shape
    dtype = DTYPE_CODE[x.dtype]
    y = out if out is not None else torch.empty(n, h * f, w_ * f, rup(cout, 16), dtype=x.dtype, device=x.device)
    d = make_conv_desc(x0=_p(x), c0=c0, w=_p(w_packed), y=_p(y), ldy=y.shape[-1], n=n, h=h, w_=w_, cout=f * f * cout,
                       cout_pad=rup(f * f * cout, 32), dtype=dtype, in_scale=_p(in_scale), in_shift=_p(in_shift),
                       in_relu=in_relu, bias=_p(bias), stats=_p(stats), stats_ld=stats.shape[-1] if stats is not None else 0,
                       kh=1, kw=1, dil=1, mode_out=1, f=f, cstat=cout, out_scale=_p(out_scale), out_relu=out_relu)
    check(lib.satcv_conv2d_igemm(C.byref(d), stream_ptr()))
    return y


def conv2d_dgrad(dy, w_dgrad, cin, *, kh=3, kw=3, dil=1, out=None, accumulate=False):
    """Data gradient of Conv2D: 'same' correlation of dy with the flipped, in/out-swapped kernel."""
    n, h, w_, cy = dy.shape
    dtype = DTYPE_CODE[dy.dtype]
    dx = out if out is not None else torch.empty(n, h, w_, rup(cin, 16), dtype=dy.dtype, device=dy.device)
    d = make_conv_desc(x0=_p(dy), c0=cy, w=_p(w_dgrad), y=_p(dx), ldy=dx.shape[-1], n=n, h=h, w_=w_, cout=cin,
                       cout_pad=rup(cin, 32), dtype=dtype, kh=kh, kw=kw, dil=dil, accumulate=accumulate)
    check(lib.satcv_conv2d_igemm(C.byref(d), stream_ptr()))
    return dx


def conv2d_transpose_dgrad(dy, w_dgrad, cin, cout, f, *, out=None, stats=None, bst=None):
    """Data gradient of Conv2DTranspose(k==s): space-to-depth gather of dy, 1x1 GEMM.  bst (with stats): the fused BatchNorm-backward
    sums of the layer whose activation gradient this launch writes (satcv.h: bst_*)."""
    n, hf, wf, cy = dy.shape
    h, w_ = hf // f, wf // f
    dtype = DTYPE_CODE[dy.dtype]
    assert cy == cout, 'dy channel stride must equal cout'
    dx = out if out is not None else torch.empty(n, h, w_, rup(cin, 16), dtype=dy.dtype, device=dy.device)
    d = make_conv_desc(x0=_p(dy), c0=cy, w=_p(w_dgrad), y=_p(dx), ldy=dx.shape[-1], n=n, h=h, w_=w_, cout=cin,
                       cout_pad=rup(cin, 32), dtype=dtype, kh=1, kw=1, dil=1, mode_in=1, f=f,
                       stats=_p(stats), stats_ld=stats.shape[-1] if stats is not None else 0,
                       bst={k: (_p(v) if torch.is_tensor(v) else v) for k, v in bst.items()} if bst else None)
    check(lib.satcv_conv2d_igemm(C.byref(d), stream_ptr()))
    return dx


def make_wgrad_desc(*, x0, c0, dy, lddy, dw, cin, cout, n, h, w_, dtype, x1=None, c1=0, in_scale=None, in_shift=None,
                    in_relu=0, kh=3, kw=3, dil=1, mode_dy=0, f=1, transposed=0, workspace=None, workspace_bytes=0, accumulate=0, whole_chip=0, defer_reduce=0):
    d = WgradDesc()
    d.x0, d.x1, d.c0, d.c1 = x0, x1, c0, c1
    d.in_scale, d.in_shift, d.in_relu = in_scale, in_shift, int(in_relu)
    d.dy, d.lddy, d.dw, d.cin, d.cout = dy, lddy, dw, cin, cout
    d.n, d.h, d.w_, d.kh, d.kw, d.dil = n, h, w_, kh, kw, dil
    d.mode_dy, d.f, d.transposed = mode_dy, f, transposed
    d.workspace, d.workspace_bytes, d.dtype, d.accumulate = workspace, workspace_bytes, dtype, int(accumulate)
    d.whole_chip = int(whole_chip)
    d.defer_reduce = int(defer_reduce)
    return d


def make_bwdf_desc(*, g, yraw, ldg, bn_scale, bn_shift, bn_mean, bn_rstd, bn_coef, x0, c0, w_dgrad, dx, lddx, dw, cin, cout, n, h, w_, dtype,
                   linear=0, x1=None, c1=0, in_scale=None, in_shift=None, in_relu=0, kh=3, kw=3, dil=1, workspace=None, workspace_bytes=0,
                   accumulate=0, bst_sums=None, bst_sums_ld=0, bst_mean=None, bst_rstd=None, bst_act_form=0, dpool=None, lddp=0, amax=None,
                   hg_dlogits=None, hg_w=None, hg_ncls=0):
    from ._lib import BwdfDesc
    d = BwdfDesc()
    d.g, d.yraw, d.ldg = g, yraw, ldg
    d.bn_scale, d.bn_shift, d.bn_mean, d.bn_rstd, d.bn_coef, d.linear = bn_scale, bn_shift, bn_mean, bn_rstd, bn_coef, int(linear)
    d.x0, d.x1, d.c0, d.c1 = x0, x1, c0, c1
    d.in_scale, d.in_shift, d.in_relu = in_scale, in_shift, int(in_relu)
    d.w_dgrad, d.dx, d.lddx, d.dw, d.cin, d.cout = w_dgrad, dx, lddx, dw, cin, cout
    d.n, d.h, d.w_, d.kh, d.kw, d.dil = n, h, w_, kh, kw, dil
    d.workspace, d.workspace_bytes, d.dtype, d.accumulate = workspace, workspace_bytes, dtype, int(accumulate)
    d.bst_sums, d.bst_sums_ld, d.bst_mean, d.bst_rstd, d.bst_act_form = bst_sums, bst_sums_ld, bst_mean, bst_rstd, int(bst_act_form)
    d.dpool, d.lddp, d.amax = dpool, lddp, amax
    d.hg_dlogits, d.hg_w, d.hg_ncls = hg_dlogits, hg_w, int(hg_ncls)
    return d


def conv_bwd_fused(g, yraw, scale, shift, mean, rstd, coef, x, w_dgrad, cin, cout, *, x1=None, in_scale=None, in_shift=None, in_relu=False,
                   linear=False, accumulate_into=None, bst=None, dpool=None, amax=None, want_dx=True, head=None):
    """BatchNorm-backward apply + data gradient + weight gradient of a thin conv -> BN -> ReLU block in one launch
    (satcv_conv2d_bwd_fused).  dpool / amax: the pooled form (encoder blocks).  head=(dlogits (npix, 2) fp32, w (cout, 2) fp32) with g=None:
    the block under the 1 x 1 head, g formed in the loader.  Returns (dx, dw) -- dx None with want_dx=False --, or None when the shape is
    outside the kernel's limits."""
    n, h, w_, c0 = x.shape
    c1 = x1.shape[-1] if x1 is not None else 0
    dx = torch.empty(n, h, w_, c0 + c1, dtype=x.dtype, device=x.device) if want_dx else None
    dw = accumulate_into if accumulate_into is not None else torch.empty(3, 3, cin, cout, dtype=torch.float32, device=x.device)
    d = make_bwdf_desc(g=_p(g), yraw=_p(yraw), ldg=g.shape[-1] if g is not None else yraw.shape[-1], bn_scale=_p(scale), bn_shift=_p(shift), bn_mean=_p(mean), bn_rstd=_p(rstd),
                       bn_coef=_p(coef), x0=_p(x), c0=c0, x1=_p(x1), c1=c1, in_scale=_p(in_scale), in_shift=_p(in_shift), in_relu=in_relu,
                       w_dgrad=_p(w_dgrad), dx=_p(dx), lddx=c0 + c1, dw=_p(dw), cin=cin, cout=cout, n=n, h=h, w_=w_, dtype=DTYPE_CODE[x.dtype],
                       linear=linear, accumulate=accumulate_into is not None, dpool=_p(dpool), lddp=dpool.shape[-1] if dpool is not None else 0,
                       amax=_p(amax), **(dict(hg_dlogits=_p(head[0]), hg_w=_p(head[1]), hg_ncls=head[0].shape[-1]) if head else {}),
                       **(dict(bst_sums=_p(bst['sums']), bst_sums_ld=bst['sums'].shape[-1], bst_mean=_p(bst.get('mean')), bst_rstd=_p(bst.get('rstd')),
                               bst_act_form=bst.get('act_form', 0)) if bst else {}))
    nb = lib.satcv_conv2d_bwd_fused_workspace(C.byref(d))
    if nb < 0:
        return None
    ws = torch.empty(max(nb // 4, 1), dtype=torch.float32, device=x.device)
    d.workspace, d.workspace_bytes = ws.data_ptr(), nb
    check(lib.satcv_conv2d_bwd_fused(C.byref(d), stream_ptr()))
    return dx, dw


def conv2d_wgrad(x, dy, cin, cout, *, kh=3, kw=3, dil=1, x1=None, in_scale=None, in_shift=None, in_relu=False,
                 transposed_f=0, dw=None):
    """Kernel gradient, Keras layout: (kh,kw,cin,cout), or (f,f,cout,cin) for Conv2DTranspose."""
    n, h, w_, c0 = x.shape
    dtype = DTYPE_CODE[x.dtype]
    c1 = x1.shape[-1] if x1 is not None else 0
    f = transposed_f
    if dw is None:
        shape = (f, f, cout, cin) if f else (kh, kw, cin, cout)
        dw = torch.empty(shape, dtype=torch.float32, device=x.device)
    d = make_wgrad_desc(x0=_p(x), c0=c0, x1=_p(x1), c1=c1, dy=_p(dy), lddy=dy.shape[-1], dw=_p(dw), cin=cin, cout=cout,
                        n=n, h=h, w_=w_, dtype=dtype, in_scale=_p(in_scale), in_shift=_p(in_shift), in_relu=in_relu,
                        kh=1 if f else kh, kw=1 if f else kw, dil=dil, mode_dy=1 if f else 0, f=f if f else 1,
                        transposed=1 if f else 0)
    nbytes = lib.satcv_conv2d_wgrad_workspace(C.byref(d))
    if nbytes < 0:
        raise _lib.SatcvError(lib.satcv_last_error().decode())
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=x.device)
    d.workspace, d.workspace_bytes = ws.data_ptr(), nbytes
    check(lib.satcv_conv2d_wgrad(C.byref(d), stream_ptr()))
    return dw


# ------------------------------------------------------------------------ batch norm
def bn_finalize_train(stats, count, gamma, beta, moving_mean, moving_var, eps=1e-3, momentum=0.99, updates=1, bessel=False):
    c = gamma.numel()
    dev = gamma.device
    scale, shift, mean, rstd = (torch.empty(c, dtype=torch.float32, device=dev) for _ in range(4))
    check(lib.satcv_bn_finalize_train(ptr(stats), stats.shape[-1], c, float(count), ptr(gamma), ptr(beta), eps, momentum,
                                      updates, int(bessel), ptr(moving_mean), ptr(moving_var), ptr(scale), ptr(shift),
                                      ptr(mean), ptr(rstd), stream_ptr()))
    return scale, shift, mean, rstd


def bn_affine_infer(gamma, beta, moving_mean, moving_var, eps=1e-3):
    c = gamma.numel()
    scale = torch.empty(c, dtype=torch.float32, device=gamma.device)
    shift = torch.empty_like(scale)
    check(lib.satcv_bn_affine_infer(ptr(gamma), ptr(beta), ptr(moving_mean), ptr(moving_var), eps, c, ptr(scale), ptr(shift), stream_ptr()))
    return scale, shift


def bn_relu_pool(yraw, scale, shift, f, want_act=True, want_pool=True, stats=None):
    """Activation('relu')(BatchNormalization()(y)) + MaxPooling2D(f, strides=f) (utils/model_tools.py:179-180, 281)."""
    n, h, w_, c = yraw.shape
    act = torch.empty_like(yraw) if want_act else None
    pooled = torch.empty(n, h // f, w_ // f, c, dtype=yraw.dtype, device=yraw.device) if want_pool else None
    check(lib.satcv_bn_relu_pool(ptr(yraw), ptr(scale), ptr(shift), ptr(act), 0, ptr(pooled), ptr(stats),
                                 stats.shape[-1] if stats is not None else 0, n, h, w_, c, f, DTYPE_CODE[yraw.dtype], stream_ptr()))
    return act, pooled


def bn_relu_pool_amax(yraw, scale, shift, f):
    """bn_relu_pool that also returns the arg-max byte of every pooling window (satcv_bn_relu_pool_amax)."""
    n, h, w_, c = yraw.shape
    act = torch.empty_like(yraw)
    pooled = torch.empty(n, h // f, w_ // f, c, dtype=yraw.dtype, device=yraw.device)
    amax = torch.empty(n, h // f, w_ // f, c, dtype=torch.uint8, device=yraw.device)
    check(lib.satcv_bn_relu_pool_amax(ptr(yraw), ptr(scale), ptr(shift), ptr(act), 0, ptr(pooled), ptr(amax), None, 0, n, h, w_, c, f,
                                      DTYPE_CODE[yraw.dtype], stream_ptr()))
    return act, pooled, amax


def make_bnbwd_desc(*, yraw, ldy, scale, shift, mean, rstd, n, h, w_, c, dtype, da=None, ldda=0, dpool=None, lddp=0, f=1,
                    sums=None, sums_ld=0, coef=None, dy=None, lddy_out=0, dbias=None, linear=0, yraw1=None, ldy1=0, dy1=None, lddy1=0,
                    c_split=0, sk_sums=None, sk_sums_ld=0):
    d = BnBwdDesc()
    d.sk_sums, d.sk_sums_ld = sk_sums, sk_sums_ld
    d.linear = linear
    d.yraw1, d.ldy1, d.dy1, d.lddy1, d.c_split = yraw1, ldy1, dy1, lddy1, c_split
    d.da, d.ldda, d.dpool, d.lddp, d.f = da, ldda, dpool, lddp, f
    d.yraw, d.ldy = yraw, ldy
    d.scale, d.shift, d.mean, d.rstd = scale, shift, mean, rstd
    d.sums, d.sums_ld, d.coef = sums, sums_ld, coef
    d.dy, d.lddy_out, d.dbias = dy, lddy_out, dbias
    d.n, d.h, d.w_, d.c, d.dtype = n, h, w_, c, dtype
    return d


def bn_relu_bwd(yraw, scale, shift, mean, rstd, da=None, dpool=None, f=1, want_dbias=False):
    """Backward of relu(BN_train(y)) given grad of the activation (and/or of its max-pool).
    Returns (dy, dgamma, dbeta, dbias)."""
    n, h, w_, c = yraw.shape
    dev, dtype = yraw.device, DTYPE_CODE[yraw.dtype]
    sums = new_stats(c, dev)
    coef = torch.empty(2, c, dtype=torch.float32, device=dev)
    dgamma = torch.empty(c, dtype=torch.float32, device=dev)
    dbeta = torch.empty_like(dgamma)
    dy = torch.empty_like(yraw)
    dbias = torch.zeros(c, dtype=torch.float32, device=dev) if want_dbias else None
    d = make_bnbwd_desc(yraw=_p(yraw), ldy=c, scale=_p(scale), shift=_p(shift), mean=_p(mean), rstd=_p(rstd), n=n, h=h, w_=w_,
                        c=c, dtype=dtype, da=_p(da), ldda=da.shape[-1] if da is not None else 0, dpool=_p(dpool),
                        lddp=dpool.shape[-1] if dpool is not None else 0, f=f, sums=_p(sums), sums_ld=c, coef=_p(coef),
                        dy=_p(dy), lddy_out=c, dbias=_p(dbias))
    check(lib.satcv_bn_bwd_reduce(C.byref(d), stream_ptr()))
    check(lib.satcv_bn_bwd_finalize(ptr(sums), c, c, float(n * h * w_), ptr(dgamma), ptr(dbeta), ptr(coef), 0, stream_ptr()))
    check(lib.satcv_bn_bwd_apply(C.byref(d), stream_ptr()))
    return dy, dgamma, dbeta, dbias


def bn_relu_bwd_concat(y0, y1, scale, shift, mean, rstd, da):
    """Backward of relu(BN_train(concat([y0, y1]))) (decoder_block, utils/model_tools.py:307-309) in one pass per kernel over the
    gradient `da` of the concatenation.  Returns (dy0, dy1, dgamma, dbeta)."""
    n, h, w_, c0 = y0.shape
    c1 = y1.shape[-1]
    c = c0 + c1
    dev, dtype = y0.device, DTYPE_CODE[y0.dtype]
    sums = new_stats(c, dev)
    coef = torch.empty(2, c, dtype=torch.float32, device=dev)
    dgamma = torch.empty(c, dtype=torch.float32, device=dev)
    dbeta = torch.empty_like(dgamma)
    dy0, dy1 = torch.empty_like(y0), torch.empty_like(y1)
    d = make_bnbwd_desc(yraw=_p(y0), ldy=c0, scale=_p(scale), shift=_p(shift), mean=_p(mean), rstd=_p(rstd), n=n, h=h, w_=w_, c=c, dtype=dtype,
                        da=_p(da), ldda=c, sums=_p(sums), sums_ld=c, coef=_p(coef), dy=_p(dy0), lddy_out=c0,
                        yraw1=_p(y1), ldy1=c1, dy1=_p(dy1), lddy1=c1, c_split=c0)
    check(lib.satcv_bn_bwd_reduce(C.byref(d), stream_ptr()))
    check(lib.satcv_bn_bwd_finalize(ptr(sums), c, c, float(n * h * w_), ptr(dgamma), ptr(dbeta), ptr(coef), 0, stream_ptr()))
    check(lib.satcv_bn_bwd_apply(C.byref(d), stream_ptr()))
    return dy0, dy1, dgamma, dbeta


# ------------------------------------------------------------------------------ head
def make_head_desc(*, x, ldx, cin, w, b, ncls, activation, npix, dtype, in_scale=None, in_shift=None, thresh=0.5,
                   probs=None, classes=None, dlogits=None, dx=None, lddx=0, dw=None, db=None, bnr=None, partials=None):
    d = HeadDesc()
    d.partials = partials
    d.x, d.ldx, d.cin = x, ldx, cin
    d.in_scale, d.in_shift, d.w, d.b = in_scale, in_shift, w, b
    d.ncls, d.activation, d.thresh = ncls, activation, thresh
    d.probs, d.classes, d.dlogits = probs, classes, dlogits
    d.dx, d.lddx, d.dw, d.db = dx, lddx, dw, db
    d.npix, d.dtype = npix, dtype
    if bnr:
        for k, v in bnr.items():
            setattr(d, 'bnr_' + k, v)
    return d


def head_fwd(x, w, b, activation='softmax', in_scale=None, in_shift=None, thresh=0.5):
    """Conv2D(ncls,(1,1),activation) + argmax / threshold (utils/model_tools.py:405-406, 660-661)."""
    n, h, w_, c = x.shape
    cin, ncls = w.shape
    act = 0 if activation == 'softmax' else 1
    probs = torch.empty(n, h, w_, ncls, dtype=torch.float32, device=x.device)
    classes = torch.empty((n, h, w_) if act == 0 else (n, h, w_, ncls), dtype=torch.int32, device=x.device)
    d = make_head_desc(x=_p(x), ldx=c, cin=cin, w=_p(w), b=_p(b), ncls=ncls, activation=act, npix=n * h * w_,
                       dtype=DTYPE_CODE[x.dtype], in_scale=_p(in_scale), in_shift=_p(in_shift), thresh=thresh,
                       probs=_p(probs), classes=_p(classes))
    check(lib.satcv_head_fwd(C.byref(d), stream_ptr()))
    return probs, classes


def head_bwd(x, w, dlogits, in_scale=None, in_shift=None):
    n, h, w_, c = x.shape
    cin, ncls = w.shape
    dx = torch.empty(n, h, w_, c, dtype=x.dtype, device=x.device)
    if c > cin:
        dx.zero_()
    dw = torch.zeros(cin, ncls, dtype=torch.float32, device=x.device)
    db = torch.zeros(ncls, dtype=torch.float32, device=x.device)
    d = make_head_desc(x=_p(x), ldx=c, cin=cin, w=_p(w), b=_p(db), ncls=ncls, activation=0, npix=n * h * w_,
                       dtype=DTYPE_CODE[x.dtype], in_scale=_p(in_scale), in_shift=_p(in_shift), dlogits=_p(dlogits),
                       dx=_p(dx), lddx=c, dw=_p(dw), db=_p(db))
    check(lib.satcv_head_bwd(C.byref(d), stream_ptr()))
    return dx, dw, db


LOSS_KINDS = {'weighted_categorical_crossentropy': 0, 'weighted_bce': 1, 'gen_dice': 2, 'iou_loss': 3, 'mse_4d': 4}


_ACT_CODE = {'softmax': 0, 'sigmoid': 1, 'linear': 2}


def loss_fwd_bwd(kind, probs, y_true, weights, activation='softmax', grad_scale=1.0, eps=1e-6):
    """Returns (loss scalar tensor, dL/dlogits); activation 'linear': the outputs ARE the logits (regression heads)."""
    ncls = probs.shape[-1]
    npix = probs.numel() // ncls
    loss = torch.zeros(1, dtype=torch.float32, device=probs.device)
    dlogits = torch.empty_like(probs)
    if LOSS_KINDS[kind] >= 2:
        nimg = probs.shape[0]
        ws = torch.empty(nimg * 3 * ncls, dtype=torch.float32, device=probs.device)
        check(lib.satcv_loss_global_fwd_bwd(LOSS_KINDS[kind], ptr(probs), ptr(y_true.contiguous()), ptr(weights), ncls,
                                            _ACT_CODE[activation], nimg, npix // nimg, eps, grad_scale, ptr(ws), ptr(loss),
                                            ptr(dlogits), stream_ptr()))
        return loss, dlogits
    check(lib.satcv_loss_fwd_bwd(LOSS_KINDS[kind], ptr(probs), ptr(y_true.contiguous()), ptr(weights), ncls,
                                 _ACT_CODE[activation], npix, grad_scale, ptr(loss), ptr(dlogits), stream_ptr()))
    return loss, dlogits


def confusion(classes, y_true, ncls):
    conf = torch.zeros(ncls, ncls, dtype=torch.int64, device=classes.device)
    check(lib.satcv_confusion(ptr(classes), ptr(y_true.contiguous()), ncls, classes.numel(), ptr(conf), stream_ptr()))
    return conf


def adam_step(p, g, m, v, state, beta1=0.9, beta2=0.999, eps=1e-7, lr_mul=None):
    check(lib.satcv_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), beta1, beta2, eps, ptr(state), ptr(lr_mul), stream_ptr()))


def make_ctbf_desc(*, g, ldg, yup, ldy, bn_scale, bn_shift, bn_mean, bn_rstd, bn_c1, bn_c2, x, ldx, w_dgrad, w_npad, dx, lddx, dw, cin, cout, n, h, w_,
                   dtype, linear=0, in_scale=None, in_shift=None, in_relu=0, workspace=None, workspace_bytes=0, accumulate=0, defer_reduce=0,
                   bst_sums=None, bst_sums_ld=0, bst_mean=None, bst_rstd=None):
    from ._lib import CtbfDesc
    d = CtbfDesc()
    d.g, d.ldg, d.yup, d.ldy = g, ldg, yup, ldy
    d.bn_scale, d.bn_shift, d.bn_mean, d.bn_rstd, d.bn_c1, d.bn_c2, d.linear = bn_scale, bn_shift, bn_mean, bn_rstd, bn_c1, bn_c2, int(linear)
    d.x, d.ldx, d.in_scale, d.in_shift, d.in_relu = x, ldx, in_scale, in_shift, int(in_relu)
    d.w_dgrad, d.w_npad, d.dx, d.lddx, d.dw, d.cin, d.cout = w_dgrad, w_npad, dx, lddx, dw, cin, cout
    d.n, d.h, d.w_, d.f = n, h, w_, 2
    d.workspace, d.workspace_bytes, d.dtype, d.accumulate, d.defer_reduce = workspace, workspace_bytes, dtype, int(accumulate), int(defer_reduce)
    d.bst_sums, d.bst_sums_ld, d.bst_mean, d.bst_rstd = bst_sums, bst_sums_ld, bst_mean, bst_rstd
    return d


def convt_bwd_fused(g_cat, c_skip, yup, scale, shift, mean, rstd, coef, x, w_dgrad, cin, cout, *, in_scale=None, in_shift=None, in_relu=False,
                    linear=False, bst=None):
    """backward of Conv2DTranspose(k = s = 2) under concat -> BatchNorm -> ReLU for the `up` channels (satcv_convt_bwd_fused).  g_cat (n, 2h, 2w, c_skip +
    cout): gradient of the activated concatenation; the six vectors cover ALL its channels (coef = [c1 | c2]); returns (dx, dw (2, 2, cout, cin))."""
    n, h, w_ = x.shape[0], x.shape[1], x.shape[2]
    ctot = g_cat.shape[-1]
    dx = torch.empty(n, h, w_, cin, dtype=x.dtype, device=x.device)
    dw = torch.empty(2, 2, cout, cin, dtype=torch.float32, device=x.device)
    es = g_cat.element_size()

    def off(v):
        return v.data_ptr() + 4 * c_skip
    d = make_ctbf_desc(g=g_cat.data_ptr() + es * c_skip, ldg=ctot, yup=_p(yup), ldy=yup.shape[-1], bn_scale=off(scale), bn_shift=off(shift), bn_mean=off(mean),
                       bn_rstd=off(rstd), bn_c1=coef.data_ptr() + 4 * c_skip, bn_c2=coef.data_ptr() + 4 * (ctot + c_skip), x=_p(x), ldx=x.shape[-1],
                       w_dgrad=_p(w_dgrad), w_npad=rup(cin, 32), dx=_p(dx), lddx=cin, dw=_p(dw), cin=cin, cout=cout, n=n, h=h, w_=w_, dtype=DTYPE_CODE[x.dtype],
                       linear=linear, in_scale=_p(in_scale), in_shift=_p(in_shift), in_relu=in_relu,
                       **({} if bst is None else dict(bst_sums=_p(bst['sums']), bst_sums_ld=cin, bst_mean=_p(bst.get('mean')), bst_rstd=_p(bst.get('rstd')))))
    nb = lib.satcv_convt_bwd_fused_workspace(C.byref(d))
    if nb < 0:
        raise RuntimeError('convt_bwd_fused: shape outside the kernel limits')
    ws = torch.empty(max(nb // 4, 1), dtype=torch.float32, device=x.device)
    d.workspace, d.workspace_bytes = ws.data_ptr(), nb
    check(lib.satcv_convt_bwd_fused(C.byref(d), stream_ptr()))
    return dx, dw
