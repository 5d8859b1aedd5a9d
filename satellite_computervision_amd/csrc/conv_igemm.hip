// Implicit-GEMM convolution on MFMA for gfx950 (NHWC activations, K = channels x taps).
//
//   D[pixel][cout] = sum_{tap, ci} A[pixel + tap][ci] * W[tap][ci][cout]
//
// One workgroup computes BM output pixels (a TH x TW spatial tile, possibly spanning
// several small images) x BN output channels.  Per K-chunk of KC input channels the
// halo tile of the activations and all taps of the weight slab are staged in LDS;
// every tap is then an LDS-address offset of the same A image (no im2col buffer).
// MFMA: v_mfma_f32_32x32x16_bf16 (bf16 storage) or 8 x v_mfma_f32_32x32x2_f32 (fp32
// storage, exact fp32 products) on identical fragment addressing.
//
// LDS images (elements of T, 8-channel groups so that a lane's fragment is one
// 16-byte (bf16) read):
//   A: [KC/8][RL][P][8]      RL rows incl. halo (per image segment), P = padded pitch
//   B: [taps][KC/8][BN][8]
// The previous layer's BatchNorm affine + ReLU is applied to A while staging
// (zero padding is inserted AFTER the transform, as Keras pads the activated tensor).
#include "igemm_common.hpp"
#include <cstdlib>

// TW: tile width (pixels); WM x WN waves; each wave MT x NT MFMA tiles of 32x32; KS k-steps
// of 16 channels per chunk.
template <typename T, int TW, int WM, int WN, int MT, int NT, int KS>
__global__ __launch_bounds__(WM* WN * 64) void igemm_kernel(const IgemmArgs a) {
  constexpr int NTHREADS = WM * WN * 64;
  constexpr int BM = WM * MT * 32;
  constexpr int BN = WN * NT * 32;
  constexpr int TH = BM / TW;
  constexpr int KC = KS * 16;
  constexpr int SLOTS = KC / 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* ldsA = reinterpret_cast<T*>(smem_raw);
  const int a_elems = SLOTS * a.rl * a.pitch * 8;
  T* ldsB = ldsA + a_elems;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, hh = lane >> 5;

  // ---- tile coordinates
  int bid = blockIdx.x;
  const int nt = bid % a.n_tiles;
  int mt = bid / a.n_tiles;
  const int tx = mt % a.tiles_x; mt /= a.tiles_x;
  const int ty = mt % a.tiles_y;
  const int grp = mt / a.tiles_y;
  const int n0 = grp * a.imgs;
  const int y0 = ty * TH;          // 0 in multi-image mode (tiles_y == 1)
  const int x0 = tx * TW;
  const int nbase = nt * BN;
  const int taps = a.kh * a.kw;
  const int cin = a.c0 + a.c1;

  // ---- per-lane A fragment base offsets (elements) for each M tile of this wave
  int a_off[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int q = (wm * MT + m) * 32 + r;          // tile-local pixel
    const int t = q / TW, cx = q % TW;
    const int k = t / a.rpi;
    // rows past the last image segment of the tile are never stored; keep their reads in range
    const int l0 = (k < a.imgs) ? k * a.seg + (t - k * a.rpi) * a.stride : 0;
    a_off[m] = (l0 * a.pitch + cx * a.stride) * 8;
  }
  const int slot_stride = a.rl * a.pitch * 8;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

  const int a_items = a.rl * a.cl * SLOTS;
  const int b_items = taps * SLOTS * BN;
  const T* wp = reinterpret_cast<const T*>(a.w);

  for (int chunk = 0; chunk < a.nchunks; ++chunk) {
    const int cg0 = chunk * KC;
    // ------------------------------------------------ stage A (activations, with transform)
    {
      // which source / spatial sub-position does this chunk read?
      const T* src; int cs, coff, si = 0, sj = 0;
      if (a.mode_in == 1) {                 // space-to-depth: K index = (i*f+j)*c0 + o
        const int ij = cg0 / a.c0;
        si = ij / a.f; sj = ij % a.f;
        src = reinterpret_cast<const T*>(a.x0); cs = a.c0; coff = cg0 - ij * a.c0;
      } else if (cg0 < a.c0) {
        src = reinterpret_cast<const T*>(a.x0); cs = a.c0; coff = cg0;
      } else {
        src = reinterpret_cast<const T*>(a.x1); cs = a.c1; coff = cg0 - a.c0;
      }
      for (int it = tid; it < a_items; it += NTHREADS) {
        const int slot = it % SLOTS;
        const int pix = it / SLOTS;
        const int c = pix % a.cl;
        const int L = pix / a.cl;
        const int k = L / a.seg;
        const int yy = L - k * a.seg - a.halh;
        const int n = n0 + k;
        const int y = y0 * a.stride + yy;          // input coordinates (stride 1: = output coordinates)
        const int x = x0 * a.stride + c - a.halw;
        float v[8];
        const int hlim = a.mode_in == 1 ? a.h : a.hs, wlim = a.mode_in == 1 ? a.w_ : a.ws;
        const bool valid = (n < a.n) && (y >= 0) && (y < hlim) && (x >= 0) && (x < wlim);
        if (valid) {
          size_t off;
          if (a.mode_in == 1)
            off = ((size_t)(n * a.hs + (y * a.f + si)) * a.ws + (x * a.f + sj)) * cs + coff + slot * 8;
          else
            off = ((size_t)(n * a.hs + y) * a.ws + x) * cs + coff + slot * 8;
          load8<T>(src + off, v);
          if (a.in_scale) {
            const float* sc = a.in_scale + cg0 + slot * 8;
            const float* sh = a.in_shift + cg0 + slot * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float t = v[e] * sc[e] + sh[e];
              v[e] = a.in_relu ? fmaxf(t, 0.f) : t;
            }
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = 0.f;
        }
        store8<T>(ldsA + ((slot * a.rl + L) * a.pitch + c) * 8, v);
      }
    }
    // ------------------------------------------------ stage B (weights)
    for (int it = tid; it < b_items; it += NTHREADS) {
      const int co = it % BN;
      const int run = it / BN;            // tap * SLOTS + slot
      const int slot = run % SLOTS;
      const int tap = run / SLOTS;
      const size_t goff = ((size_t)(tap * (cin / 8) + chunk * SLOTS + slot) * a.cout_pad + nbase + co) * 8;
      const T* g = wp + goff;
      T* l = ldsB + (size_t)(run * BN + co) * 8;
      if constexpr (std::is_same<T, bf16>::value) {
        *reinterpret_cast<uint4*>(l) = *reinterpret_cast<const uint4*>(g);
      } else {
        reinterpret_cast<uint4*>(l)[0] = reinterpret_cast<const uint4*>(g)[0];
        reinterpret_cast<uint4*>(l)[1] = reinterpret_cast<const uint4*>(g)[1];
      }
    }
    __syncthreads();
    // ------------------------------------------------ MFMA over taps x k-steps
    for (int ky = 0; ky < a.kh; ++ky) {
      for (int kx = 0; kx < a.kw; ++kx) {
        const int tap = ky * a.kw + kx;
        const int tap_off = (ky * a.dil * a.pitch + kx * a.dil) * 8;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const int slot = s * 2 + hh;
          FragT<T> af[MT], bf[NT];
#pragma unroll
          for (int m = 0; m < MT; ++m) af[m] = lds_frag<T>(ldsA + slot * slot_stride + a_off[m] + tap_off);
#pragma unroll
          for (int n = 0; n < NT; ++n)
            bf[n] = lds_frag<T>(ldsB + ((tap * SLOTS + slot) * BN + (wn * NT + n) * 32 + r) * 8);
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) mma32<T>(acc[m][n], af[m], bf[n]);
        }
      }
    }
    __syncthreads();
  }

  // ---------------------------------------------------------------- epilogue
  T* yp = reinterpret_cast<T*>(a.y);
  const int ho = a.mode_out ? a.h * a.f : a.h;
  const int wo = a.mode_out ? a.w_ * a.f : a.w_;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int cn = nbase + (wn * NT + n) * 32 + r;     // GEMM column
    const bool cvalid = cn < a.cout;
    const int cch = cvalid ? cn % a.cstat : 0;         // bias / stats / D2S channel
    const int ij = cvalid ? cn / a.cstat : 0;
    const float bv = (a.bias && cvalid) ? a.bias[cch] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
        const int q = (wm * MT + m) * 32 + row;
        const int t = q / TW, cx = q % TW;
        const int k = t / a.rpi;
        const int nimg = n0 + k;
        const int y = y0 + (t - k * a.rpi);
        const int x = x0 + cx;
        const bool pv = (k < a.imgs) && (nimg < a.n) && (y < a.h) && (x < a.w_);
        if (pv && cvalid) {
          float v = acc[m][n][i] + bv;
          if (a.out_relu) v = fmaxf(v, 0.f);
          const T tv = (T)v;
          size_t off;
          if (a.mode_out == 1) {
            const int oy = y * a.f + ij / a.f, ox = x * a.f + ij % a.f;
            off = ((size_t)(nimg * ho + oy) * wo + ox) * a.ldy + cch;
          } else {
            off = ((size_t)(nimg * ho + y) * wo + x) * a.ldy + cn;
          }
          const T tvv = a.accumulate ? (T)(a.accumulate == 2 ? fmaxf((float)yp[off] + (float)tv, 0.f) : (float)yp[off] + v) : tv;
          yp[off] = tvv;
          const float fv = (float)tvv;
          s1 += fv; s2 += fv * fv;
        }
      }
    }
    if (a.stats) {
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (hh == 0 && cvalid) {
        satcv_stat_t* row = a.stats + (size_t)((blockIdx.x + wave) % SATCV_STAT_ROWS) * 2 * a.stats_ld;
        atomicAdd(row + cch, (satcv_stat_t)s1);
        atomicAdd(row + a.stats_ld + cch, (satcv_stat_t)s2);
      }
    }
  }
}

// ------------------------------------------------------------------ host side
template <typename T, int TW, int WM, int WN, int MT, int NT, int KS>
static int launch_cfg(IgemmArgs& a, hipStream_t st) {
  constexpr int BM = WM * MT * 32, BN = WN * NT * 32, TH = BM / TW, KC = KS * 16;
  a.halh = a.dil * (a.kh - 1) / 2;
  a.halw = a.dil * (a.kw - 1) / 2;
  a.tiles_x = cdiv(a.w_, TW);
  if (a.h >= TH) { a.rpi = TH; a.imgs = 1; a.tiles_y = cdiv(a.h, TH); a.ngroups = a.n; }
  else { a.rpi = a.h; a.imgs = TH / a.h; a.tiles_y = 1; a.ngroups = cdiv(a.n, a.imgs); }
  a.seg = (a.rpi - 1) * a.stride + 1 + 2 * a.halh;
  a.rl = a.imgs * a.seg;
  a.cl = (TW - 1) * a.stride + 1 + 2 * a.halw;
  // pitch chosen so that the 16-lane groups of ds_read_b128 hit distinct 16-byte slots
  if (TW == 32) a.pitch = a.cl;
  else if (TW == 16) a.pitch = cdiv(a.cl, 16) * 16;
  else a.pitch = (a.cl <= 8) ? 8 : (cdiv(a.cl - 8, 16) * 16 + 8);
  a.n_tiles = cdiv(a.cout, BN);
  const int cin = a.c0 + a.c1;
  a.nchunks = cin / KC;
  if (cin % KC != 0) { satcv_set_error("igemm: K=%d not a multiple of chunk %d", cin, KC); return SATCV_ERR_INVALID; }
  if (a.x1 && (a.c0 % KC != 0)) { satcv_set_error("igemm: c0=%d not chunk aligned", a.c0); return SATCV_ERR_INVALID; }
  if (a.mode_in == 1 && (a.c0 % KC != 0)) { satcv_set_error("igemm: s2d c0=%d not chunk aligned", a.c0); return SATCV_ERR_INVALID; }
  if (a.mode_out == 1 && (a.cstat % BN != 0)) { satcv_set_error("igemm: d2s cstat=%d %% BN=%d", a.cstat, BN); return SATCV_ERR_INVALID; }
  if (a.cout_pad < a.n_tiles * BN) { satcv_set_error("igemm: cout_pad %d < %d", a.cout_pad, a.n_tiles * BN); return SATCV_ERR_INVALID; }
  const size_t lds = ((size_t)(KC / 8) * a.rl * a.pitch * 8 + (size_t)a.kh * a.kw * (KC / 8) * BN * 8) * sizeof(T);
  if (lds > 160 * 1024) { satcv_set_error("igemm: LDS %zu too large", lds); return SATCV_ERR_UNSUPPORTED; }
  auto kern = igemm_kernel<T, TW, WM, WN, MT, NT, KS>;
  { const int rc = satcv_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds); if (rc) return rc; }
  const long long blocks = (long long)a.ngroups * a.tiles_y * a.tiles_x * a.n_tiles;
  if (blocks <= 0 || blocks > 0x7fffffffLL) { satcv_set_error("igemm: bad grid %lld", blocks); return SATCV_ERR_INVALID; }
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(WM * WN * 64), lds, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { satcv_set_error("igemm launch: %s", hipGetErrorString(e)); return SATCV_ERR_HIP; }
  return SATCV_OK;
}

template <typename T, int TW>
static int launch_tw(IgemmArgs& a, hipStream_t st) {
  const int cin = a.c0 + a.c1;
  const int nspace = a.mode_out ? a.cstat : a.cout;      // BN must divide cstat in d2s mode
  const bool ks2 = (cin % 32 == 0) && (!a.x1 || a.c0 % 32 == 0) && (a.mode_in != 1 || a.c0 % 32 == 0) &&
                   (a.kh * a.kw == 1);                   // KC=32 only for 1x1 (LDS budget), else 16
  // preferred N tile first; if its LDS images do not fit (large dilation halo, fp32 storage) fall back to a
  // narrower N tile, whose weight slab is smaller
  int rc = SATCV_ERR_UNSUPPORTED;
  if (nspace >= 128 && nspace % 128 == 0)
    rc = ks2 ? launch_cfg<T, TW, 2, 2, 2, 2, 2>(a, st) : launch_cfg<T, TW, 2, 2, 2, 2, 1>(a, st);
  if (rc == SATCV_ERR_UNSUPPORTED && nspace >= 64 && nspace % 64 == 0)
    rc = ks2 ? launch_cfg<T, TW, 2, 2, 2, 1, 2>(a, st) : launch_cfg<T, TW, 2, 2, 2, 1, 1>(a, st);
  if (rc == SATCV_ERR_UNSUPPORTED)
    rc = ks2 ? launch_cfg<T, TW, 4, 1, 2, 1, 2>(a, st) : launch_cfg<T, TW, 4, 1, 2, 1, 1>(a, st);
  if (rc == SATCV_ERR_UNSUPPORTED)          // many taps (7x7 stem) in fp32: halve the pixel tile as well
    rc = launch_cfg<T, TW, 2, 1, 2, 1, 1>(a, st);
  if (rc == SATCV_ERR_UNSUPPORTED) satcv_set_error("igemm: no tile configuration fits the LDS for this shape");
  return rc;
}

template <typename T>
static int launch_t(IgemmArgs& a, hipStream_t st) {
  switch (igemm_pick_tw(a.w_)) {
    case 32: return launch_tw<T, 32>(a, st);
    case 16: return launch_tw<T, 16>(a, st);
    default: return launch_tw<T, 8>(a, st);
  }
}

void satcv_prof_begin(int kind, double flops, hipStream_t st);
void satcv_prof_end(int kind, hipStream_t st);

static int igemm_fill_args(const satcv_conv_desc* d, IgemmArgs& a) {
  SATCV_CHECK(d && d->x0 && d->w && d->y, "igemm: null pointer");
  SATCV_CHECK(d->c0 > 0 && d->c0 % 16 == 0 && d->c1 % 16 == 0, "igemm: channels must be multiples of 16 (c0=%d c1=%d)", d->c0, d->c1);
  SATCV_CHECK(d->dtype != SATCV_FP8X || (d->c0 % 64 == 0 && d->c1 % 64 == 0 && !d->in_scale), "igemm: the scaled fp8 path needs channels %% 64 == 0 and no input transform");
  SATCV_CHECK((d->c1 == 0) == (d->x1 == nullptr), "igemm: x1/c1 mismatch");
  SATCV_CHECK(d->kh >= 1 && d->kw >= 1 && (d->kh & 1) && (d->kw & 1) && d->dil >= 1, "igemm: bad taps");
  SATCV_CHECK(d->n > 0 && d->h > 0 && d->w_ > 0 && d->cout > 0, "igemm: bad dims");
  // (the kernels index pixels and elements of one tensor with 32-bit integers where it is safe to: refuse what would overflow them -- found by
  //  the UBSan run of the host side, tests/asan/host_abi_driver.c)
  SATCV_CHECK(satcv_pixels_ok(d->n, d->h, d->w_, d->f) && d->cout <= (1 << 20) && d->c0 <= (1 << 20) && d->c1 <= (1 << 20) &&
              d->kh <= 15 && d->kw <= 15 && d->dil <= (1 << 15) && d->f <= 16 && d->stride <= 16,
              "igemm: extents out of range (n*h*w must stay below 2^31 pixels, channels below 2^20)");
  SATCV_CHECK(d->cstat > 0, "igemm: cstat");
  SATCV_CHECK((!d->mode_in && !d->mode_out) || (d->f >= 2 && d->kh == 1 && d->kw == 1), "igemm: s2d/d2s need 1x1 taps and f>=2");
  SATCV_CHECK(!(d->accumulate && d->stats), "igemm: accumulate with statistics");
  SATCV_CHECK(!(d->mode_in && d->in_scale), "igemm: s2d source cannot carry an input transform");
  a.x0 = d->x0; a.x1 = d->x1; a.c0 = d->c0; a.c1 = d->c1;
  a.in_scale = d->in_scale; a.in_shift = d->in_shift; a.in_relu = d->in_relu;
  a.w = d->w; a.bias = d->bias; a.out_scale = d->out_scale; a.y = d->y; a.ldy = d->ldy;
  a.pool_y = d->pool_y; a.pool_ld = d->pool_ld; a.pool_f = d->pool_f;
  SATCV_CHECK(!d->pool_y || (d->pool_f >= 2 && !d->mode_out && !d->accumulate && d->h % d->pool_f == 0 && d->w_ % d->pool_f == 0 && d->pool_ld >= d->cout &&
                             d->cout % (d->dtype == SATCV_F32 ? 4 : (d->dtype == SATCV_BF16 ? 8 : 16)) == 0),
              "igemm: fused max-pool needs pool_f >= 2 dividing h and w_, plain output mode, whole 16-byte channel vectors");
  a.stats = d->stats; a.stats_ld = d->stats_ld;
  a.n = d->n; a.h = d->h; a.w_ = d->w_;
  a.hs = d->mode_in ? d->h * d->f : d->h; a.ws = d->mode_in ? d->w_ * d->f : d->w_;
  a.cout = d->cout; a.cout_pad = d->cout_pad;
  a.kh = d->kh; a.kw = d->kw; a.dil = d->dil;
  a.mode_in = d->mode_in; a.mode_out = d->mode_out; a.f = d->f;
  a.cstat = d->cstat; a.out_relu = d->out_relu; a.accumulate = d->accumulate;
  a.stride = d->stride > 1 ? d->stride : 1;
  if (a.stride > 1) {
    SATCV_CHECK(!d->mode_in && !d->mode_out && d->hin > 0 && d->win > 0 && d->h == (d->hin - 1) / a.stride + 1 && d->w_ == (d->win - 1) / a.stride + 1,
                "igemm: strided conv needs hin/win with h=(hin-1)/stride+1");
    a.hs = d->hin; a.ws = d->win;
  }
  a.dbg = 0;
  a.ksplit = 1; a.kslab = nullptr;
  a.bst_y = d->bst_y; a.bst_y1 = d->bst_y1; a.bst_ld = d->bst_ld; a.bst_ld1 = d->bst_ld1; a.bst_split = d->bst_y1 ? d->bst_split : 0;
  a.bst_scale = d->bst_scale; a.bst_shift = d->bst_shift; a.bst_mean = d->bst_mean; a.bst_rstd = d->bst_rstd; a.bst_relu = d->bst_relu;
  a.tile_policy = d->tile_policy;
  if (d->bst_y) {
    SATCV_CHECK(d->stats && d->bst_scale && d->bst_shift && d->bst_mean && d->bst_rstd, "igemm: bst_y needs stats and the four BatchNorm vectors");
    SATCV_CHECK(!d->accumulate && !d->mode_out && !d->out_relu && !d->pool_y && d->cstat == d->cout, "igemm: bst_y needs a plain, non-accumulating store");
    SATCV_CHECK(d->bst_ld >= (d->bst_y1 ? d->bst_split : d->cout) && (!d->bst_y1 || (d->bst_split > 0 && d->bst_split < d->cout && d->bst_ld1 >= d->cout - d->bst_split)),
                "igemm: bst_y channel strides");
  }
  if (a.mode_in == 1) {
    // K = f*f*c0 virtual channels gathered from one source
    SATCV_CHECK(!d->x1, "igemm: s2d with dual source");
    a.c1 = a.c0 * (d->f * d->f - 1);   // so that c0+c1 = K; x1 unused in s2d mode
  }
  return SATCV_OK;
}

static bool igemm_force_generic() {
  static const bool v = [] { const char* e = getenv("SATCV_IGEMM"); return e && e[0] == 'g'; }();
  return v;
}

extern "C" int satcv_conv2d_igemm_pipelined(const satcv_conv_desc* d) {
  IgemmArgs a;
  if (!d || igemm_fill_args(d, a) != SATCV_OK || igemm_force_generic()) return 0;
  if (a.kh == 3 && a.kw == 3 && a.stride == 1 && a.dil >= a.h && a.dil >= a.w_) { a.kh = a.kw = 1; a.dil = 1; }
  if (d->bst_y && a.dil == 3) {
    // thin dilated layers (atrous CNNs): without the fused sums the launch runs on the persistent weights-stationary kernel at 3-4 x the rate of the
    // tap-loop tile that could carry them -- the caller's separate reduce pass is the cheaper way (tools/family_time.py)
    IgemmArgs b = a;
    b.bst_y = nullptr; b.bst_y1 = nullptr;
    if (igemm_ws_launch(b, d->dtype, nullptr, true) == SATCV_OK) return 0;
  }
  return igemm_fast_launch(a, d->dtype, nullptr, true) == SATCV_OK ? 1 : 0;
}

extern "C" int satcv_conv2d_igemm(const satcv_conv_desc* d, void* stream) {
  IgemmArgs a;
  int rc = igemm_fill_args(d, a);
  if (rc) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const double flops = 2.0 * d->n * d->h * d->w_ * (double)d->cout * (double)(a.c0 + a.c1) * d->kh * d->kw;
  if (a.kh == 3 && a.kw == 3 && a.stride == 1 && a.dil >= a.h && a.dil >= a.w_) {
    // every off-centre tap of this dilated conv reads only zero padding: it IS the 1x1 conv of its centre tap
    const size_t esz = d->dtype == SATCV_BF16 ? 2 : 4;
    a.w = reinterpret_cast<const unsigned char*>(a.w) + (size_t)4 * ((a.c0 + a.c1) / 8) * a.cout_pad * 8 * esz;
    a.kh = a.kw = 1; a.dil = 1;
  }
  satcv_prof_begin(d->kh * d->kw > 1 ? 0 : 1, flops, st);
  rc = SATCV_ERR_UNSUPPORTED;
  if (!igemm_force_generic()) {
    rc = convt_thin_launch(a, d->dtype, st);                 // thin transposed convolutions: streaming kernel
    if (rc == SATCV_ERR_UNSUPPORTED) rc = convt_thin_dgrad_launch(a, d->dtype, st);      // ... and their data gradients
    if (rc == SATCV_ERR_UNSUPPORTED) rc = igemm_tr_launch(a, d->dtype, st, false);            // thin 3x3 layers: staging / matrix wave roles (round 5)
    if (rc == SATCV_ERR_UNSUPPORTED) rc = igemm_ws_launch(a, d->dtype, st, false);            // ... or the persistent weights-stationary kernel (64 -> 64, fp8)
    if (rc == SATCV_ERR_UNSUPPORTED) rc = igemm_fast_launch(a, d->dtype, st);
  }
  if (rc != SATCV_ERR_UNSUPPORTED) { /* launched (or failed hard) */ }
  else if (d->dtype == SATCV_FP8 || d->dtype == SATCV_FP8X) { satcv_set_error("igemm: this fp8 shape is outside the pipelined kernel's limits"); rc = SATCV_ERR_UNSUPPORTED; }
  else if (d->out_scale || d->pool_y || d->bst_y) { satcv_set_error("igemm: out_scale / pool_y / bst_y need the pipelined kernel"); rc = SATCV_ERR_UNSUPPORTED; }
  else if (d->dtype == SATCV_BF16) rc = launch_t<bf16>(a, st);
  else if (d->dtype == SATCV_F32) rc = launch_t<float>(a, st);
  else { satcv_set_error("igemm: bad dtype %d", d->dtype); rc = SATCV_ERR_INVALID; }
  satcv_prof_end(d->kh * d->kw > 1 ? 0 : 1, st);
  return rc;
}
