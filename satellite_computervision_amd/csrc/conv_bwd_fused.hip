// Fused backward of a THIN conv -> BatchNorm -> ReLU block (utils/model_tools.py:174-186 at decoder levels 0 and 1): ONE launch does what
// satcv_bn_bwd_apply + the data gradient + the weight gradient did in three.
//
// Why: on the full- and half-resolution levels every one of those three kernels already runs near the HBM rate it can reach, and together
// they move the layer's tensors 7 times (apply: g, y in, dy out; data gradient: dy in, dx out; weight gradient: dy, x in).  Only fusion
// removes traffic: here a workgroup loads g, y (the layer's output gradient and raw output) and x (its input) once per 8 x 32 tile,
// forms dy = scale * (g * mask - c1 - xhat * c2) in registers (never stored to HBM), and uses the dy tile twice from LDS:
//   * data gradient   dx[p][ci]       = sum_tap sum_co dy[p + 1 - tap][co] * W[tap][ci][co]     (A = dy halo tile, row reads; B = weights)
//   * weight gradient dW[tap][ci][co] += sum_q x[q][ci] * dy[q + 1 - tap][co]                   (A = x, B = dy, transposing reads)
// The weight gradient is summed over the interior pixels q of the tile against the SHIFTED dy halo (instead of over dy's interior
// against an x halo): the sum over all tiles covers every (x, dy) pair once, and the halo of dy is there for the data gradient anyway.
// 4 tensor passes (g, y, x in; dx out) instead of 7.  One LDS image of dy serves both products: [slot][row][pitch][8] planes, read by
// rows (ds_read_b128) for the data gradient and transposed (ds_read_b64_tr_b16) for the weight gradient; the plane stride is 64 B
// modulo 256 B, which keeps the four 64-byte segments of a transposing read on distinct banks.
// PERSISTENT workgroups: the data-gradient weights sit in LDS for the whole launch, the weight-gradient accumulators (each wave owns
// whole (ci-tile, co-tile, tap) products) live in registers across all tiles of the workgroup and leave as ONE fp32 slab per workgroup,
// summed by the weight-gradient reduce kernel in fixed order.  The next tile's loads are in flight during the MFMA phases.
#include "igemm_common.hpp"
#include <cstdlib>
#include <cstring>

// compile-time ablation for profiling builds (python -m satellite_computervision_amd.build -DBWDF_ABL=bits; SATCV_LIB selects the variant):
// 1 skip the MFMAs, 2 skip the dy arithmetic, 4 skip the dx stores, 8 skip the MFMA phases altogether (no fragment reads), 32 no global loads
#ifndef BWDF_ABL
#define BWDF_ABL 0
#endif
#define FABL(bit) ((BWDF_ABL & (bit)) != 0)

int wgrad_reduce_slabs(const float* ws, float* dw, int nslab, int taps, int kpad, int npad, int cin, int nvalid, int accumulate, hipStream_t st);   // conv_wgrad.hip
void reduce_job_fill(satcv_reduce_job* job, const float* ws, float* dw, int nslab, int taps, int kpad, int npad, int cin, int nvalid, int transposed, int accumulate);
void satcv_prof_begin(int kind, double flops, hipStream_t st);
void satcv_prof_end(int kind, hipStream_t st);

struct BwdfArgs {
  IgemmArgs e;                                   // the data gradient's output side, as the shared epilogue reads it (y = dx, ldy, n, h, w_, cout = CIN)
  const void* g; const void* yraw; int ldg;      // gradient w.r.t. the activated output, raw conv output (same channel stride)
  const float* bn_scale; const float* bn_shift; const float* bn_mean; const float* bn_rstd; const float* bn_coef; int bn_c, linear;
  const void* x0; const void* x1; int c0, c1;
  const float* in_scale; const float* in_shift; int in_relu;
  const void* w;                                 // data-gradient operand image [tap][COUT/8][CIN][8] (taps flipped: satcv_pack_weights mode 1)
  float* ws;                                     // [gridDim.x][9][CIN][COUT] partial weight gradients
  int tiles_x, tiles_y;
  // BatchNorm-backward sums of the layer BELOW (the BatchNorm + ReLU whose scale / shift are this launch's in_scale / in_shift): sum gm and
  // sum gm * xhat over the batch, gm = dx * [x > 0], into the same replica rows satcv_bn_bwd_reduce fills (satcv.h: bst_*)
  satcv_stat_t* bst_sums; int bst_ld; const float* bst_mean; const float* bst_rstd;
  int bst_act_form;                              // 1: leave the sums in the activated form (sum dx [x > 0], sum dx x): satcv_bn_bwd_finalize2 converts
  // POOL: the block's output was also max-pooled 2 x 2 -- g = da + (amax == position in the window ? dpool : 0)
  const void* dp; int lddp; const unsigned char* amax;
  // HG: the block feeds the 1 x 1 head directly -- g[p][c] = sum_k dlogits[p][k] w_head[c][k] is formed here from the 2 logit gradients of
  // a pixel (8 bytes) instead of being written by the head's backward kernel and read back (2 x 64 bytes per pixel)
  const float* hg_dl; const float* hg_w;
};

__device__ __forceinline__ bf16x4 tr_read4(const bf16* p) {
  short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(p));
  return __builtin_bit_cast(bf16x4, v);
}

// WPS = waves per SIMD the registers are budgeted for: 2 (256 registers; 4 waves x 2 workgroups per CU, or 8 waves x 1), or 1 for
// 64 -> 64 channels, whose 36 weight-gradient accumulator tiles + in-flight tile do not fit 256 registers at any wave count
// POOL: encoder blocks -- the gradient of the activated output is da (skip) + the 2 x 2 max-pool's gradient routed by the arg-max bytes
// the forward pooling kernel wrote.  NODG: no data gradient (the block is fed by the model input).  CINS: stored input channels when
// fewer than the 32 rows of an MFMA tile (16 for the first block: the upper x planes are zero)
template <int CIN, int COUT, int NW, int WPS, bool POOL = false, bool NODG = false, int CINS = CIN, bool HG = false>
__global__ __launch_bounds__(NW * 64, WPS) void bwd_fused_kernel(const BwdfArgs a, const int total_tiles) {
  static_assert(!HG || !POOL, "the head feeds a decoder block");
  typedef bf16 T;
  constexpr int TW = 32, TH = 8, BM = 256, RL = TH + 2, CL = TW + 2, PITCH = CL, EL = 8, NTHREADS = NW * 64;
  constexpr int SD = COUT / 8, SX = CIN / 8, SXR = CINS / 8;     // 16-byte channel slots of dy / x (SXR of them real)
  constexpr int DSTRIDE = RL * PITCH * EL;                       // 5,440 B = 64 B modulo 256 B
  constexpr int XSTRIDE = BM * EL + 32;                          // 4,096 B + 64 B
  static_assert((DSTRIDE * 2) % 256 == 64 && (XSTRIDE * 2) % 256 == 64, "plane strides must be 64 B modulo 256 B");
  constexpr int D_ITEMS = RL * CL * SD, DI = (D_ITEMS + NTHREADS - 1) / NTHREADS;
  constexpr int X_ITEMS = BM * SXR, XI = X_ITEMS / NTHREADS;
  static_assert(X_ITEMS % NTHREADS == 0 && NTHREADS % SD == 0 && NTHREADS % SXR == 0, "item -> thread mapping");
  constexpr int XPIX_STEP = NTHREADS / SXR;                       // interior pixels between two x items of a thread (a multiple of 32: same column)
  static_assert(XPIX_STEP % 32 == 0, "x items of a thread must share a column");
  constexpr int MT = 8 / NW, NT = CIN / 32;                      // data gradient: wave tile (MT x 32 pixels) x CIN
  // weight gradient: NTILE (ci-tile, co-tile, tap) products of 16 k-steps each.  Every wave runs FULL whole products; the REM left over are
  // cut along K -- each over WPP waves, KPW k-steps per wave, summed through LDS once per workgroup after the tile loop -- so that all waves
  // carry the same number of MFMAs per tile (with 9 products on 8 waves, wave 0 ran two and held the tile's barrier for the other seven)
#ifdef BWDF_NOBAL                                                              // (A/B build: whole products only, the surplus ones on the first waves)
  constexpr int NTILE = (CIN / 32) * (COUT / 32) * 9, FULL = (NTILE + NW - 1) / NW, REM = 0, WT = FULL;
#else
  constexpr int NTILE = (CIN / 32) * (COUT / 32) * 9, FULL = NTILE / NW, REM = NTILE % NW, WT = FULL + (REM ? 1 : 0);
#endif
  constexpr bool GUARD = FULL * NW > NTILE;
  constexpr int WPP = REM ? NW / REM : 1, KPW = REM ? 16 / WPP : 0;
  static_assert(REM == 0 || (NW % REM == 0 && 16 % WPP == 0 && KPW % 2 == 0), "left-over products must split evenly along K");
  constexpr int W_ITEMS = NODG ? 0 : 9 * SD * CIN;
  constexpr size_t R0_BYTES = ((size_t)(SD * DSTRIDE + SX * XSTRIDE) * sizeof(T) + 127) / 128 * 128;
  constexpr size_t W_BYTES = (size_t)W_ITEMS * EL * sizeof(T);
  // dx staging tile [256][CIN + 8]: over the dy planes when it fits there (the x planes stay intact: the fused sums of the layer below
  // read the activated x beside the staged dx), else a region of its own behind the tables
  constexpr size_t O_BYTES = (size_t)BM * (CIN + 8) * sizeof(T);
  constexpr bool O_ALIAS = O_BYTES <= (size_t)SD * DSTRIDE * sizeof(T);
  constexpr size_t TAB_BYTES = (size_t)(SD * 32 + SX * 16 + (HG ? 2 * COUT : 0)) * sizeof(float);
  static_assert(!NODG || O_ALIAS || true, "");
  constexpr size_t O_OFF = O_ALIAS ? 0 : (R0_BYTES + W_BYTES + TAB_BYTES + 127) / 128 * 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* ldsD = reinterpret_cast<T*>(smem_raw);                                   // dy halo planes
  T* ldsX = ldsD + SD * DSTRIDE;                                              // x interior planes
  T* ldsW = reinterpret_cast<T*>(smem_raw + R0_BYTES);                        // [tap][SD][CIN][8], resident
  float* tabD = reinterpret_cast<float*>(smem_raw + R0_BYTES + W_BYTES);      // [SD][4][8]: scale, shift, B, C of the BatchNorm backward
  float* tabX = tabD + SD * 32;                                               // [SX][2][8]: scale, shift of the input's BatchNorm
  float* tabH = tabX + SX * 16;                                               // HG: [SD][2][8]: the head's kernel, class 0 / class 1 of 8 channels

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const T* wp = reinterpret_cast<const T*>(a.w);

  // ---- once per workgroup: weights and the per-channel tables
  for (int it = tid; it < W_ITEMS; it += NTHREADS) lstore8<T>(ldsW + (size_t)it * EL, gload8<T>(wp + (size_t)it * EL));
  if constexpr (SXR < SX) {                                                   // zero planes of the padded input channels (never staged again)
    for (int it = tid; it < (SX - SXR) * BM; it += NTHREADS) lstore8<T>(ldsX + (SXR + it / BM) * XSTRIDE + (it % BM) * EL, zero8<T>());
  }
  for (int ch = tid; ch < COUT; ch += NTHREADS) {
    // dy = sc * (gm - c1 - xhat * c2), xhat = (y - mu) * rs   ==   sc * gm + B * y + C
    const float sc = a.bn_scale[ch], sh = a.bn_shift[ch], mu = a.bn_mean[ch], rs = a.bn_rstd[ch], c1 = a.bn_coef[ch], c2 = a.bn_coef[a.bn_c + ch];
    float* t = tabD + (ch >> 3) * 32 + (ch & 7);
    t[0] = sc; t[8] = sh; t[16] = -sc * c2 * rs; t[24] = sc * (c2 * rs * mu - c1);
  }
  if constexpr (HG) {
    for (int ch = tid; ch < COUT; ch += NTHREADS) {
      float* t = tabH + (ch >> 3) * 16 + (ch & 7);
      t[0] = a.hg_w[ch * 2]; t[8] = a.hg_w[ch * 2 + 1];                       // Keras (1, 1, cin, ncls) kernel: w[c][k]
    }
  }
  for (int ch = tid; ch < CINS; ch += NTHREADS) {
    float* t = tabX + (ch >> 3) * 16 + (ch & 7);
    t[0] = a.in_scale ? a.in_scale[ch] : 1.f; t[8] = a.in_scale ? a.in_shift[ch] : 0.f;
  }
  const float lin_lo = a.linear ? -INFINITY : 0.f;
  const bool xaff = a.in_scale != nullptr;
  const unsigned xrelu_lim = a.in_relu != 0 ? 0u : 0x80008000u;

  // ---- staged items of this thread (tile-invariant): dy halo items (g and y at one halo pixel, 8 channels), x interior items
  const int slot_d = tid % SD, slot_x = tid % SXR;
  // (halo row and column of an item packed in 10 bits, three items per register; the LDS destination and the source offset are
  //  re-derived where they are used, with 24-bit multiplies: the kernel sits at the 256-register cap of two waves per SIMD and every
  //  spilled value is reloaded by a scratch load that queues behind the prefetched tile on the in-order vmcnt counter)
  constexpr int DPK = (DI + 2) / 3;
  unsigned d_pk[DPK];
#pragma unroll
  for (int k = 0; k < DPK; ++k) d_pk[k] = 0;
#pragma unroll
  for (int j = 0; j < DI; ++j) {
    const int it = tid + j * NTHREADS;
    const int p = it / SD, c = p % CL, L = p / CL;
    d_pk[j / 3] |= (unsigned)(it < D_ITEMS ? (L << 6) | c : 0x3ff) << (10 * (j % 3));
  }
  auto d_item = [&](int j) -> int { return (int)((d_pk[j / 3] >> (10 * (j % 3))) & 0x3ffu); };      // 0x3ff: no such item
  const int g_rowstride = a.e.w_ * a.ldg;
  // g and y are addressed as (wave-uniform halo-origin pointer of the tile) + (unsigned 32-bit lane offset): no 64-bit lane pointers
  const unsigned gy_lane = (unsigned)slot_d * EL;
  const int xch0 = slot_x * EL;
  const bool xsecond = xch0 >= a.c0;
  const T* xsrc = xsecond ? reinterpret_cast<const T*>(a.x1) + (xch0 - a.c0) : reinterpret_cast<const T*>(a.x0) + xch0;
  const int xcs = xsecond ? a.c1 : a.c0;
  const int xq0 = tid / SXR;                                                  // first interior pixel of this thread
  const int x_l0 = slot_x * XSTRIDE + xq0 * EL;
  const int x_eoff0 = ((xq0 / TW) * a.e.w_ + (xq0 % TW)) * xcs, x_estep = (XPIX_STEP / TW) * a.e.w_ * xcs;

  // ---- fragment addressing
  int a_off[MT];                                                              // data gradient A operand: halo-image pixel of this lane's row
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int q = (wave * MT + m) * 32 + r;
    a_off[m] = ((q / TW) * PITCH + (q % TW)) * EL;
  }
  // transposing reads (weight gradient): a 16-lane group reads 4 pixels x 16 channels; lane -> (plane, half item, pixel)
  const int gi = lane >> 4, i16 = lane & 15;
  const int chb = 16 * (gi & 1) + 4 * (i16 & 3);
  const int pk = 8 * (gi >> 1) + (i16 >> 2);
  const int tr_x = (chb >> 3) * XSTRIDE + pk * EL + (chb & 7);
  const int tr_d = (chb >> 3) * DSTRIDE + pk * EL + (chb & 7);

  auto tile_origin = [&](int v, int& n0, int& y0, int& x0) {
    const int tx = v % a.tiles_x; v /= a.tiles_x;
    const int ty = v % a.tiles_y;
    n0 = v / a.tiles_y; y0 = ty * TH; x0 = tx * TW;
  };
  auto next_origin = [&](int& n0, int& y0, int& x0) {
    x0 += TW;
    if (x0 >= a.e.w_) { x0 = 0; y0 += TH; if (y0 >= a.e.h) { y0 = 0; ++n0; } }
  };
  Raw8<T> rg[HG ? 1 : DI], ry[DI], rx[XI];
  float2 rdl[HG ? DI : 1];                                                    // HG: the pixel's two logit gradients
  Raw8<T> rp[POOL ? DI : 1];                                                  // pooled gradient of the item's window
  uint2 ra[POOL ? DI : 1];                                                    // arg-max bytes of the window (8 channels)
  const int hp = a.e.h >> 1, wpool = a.e.w_ >> 1;
  unsigned valid = 0;
  auto issue_loads = [&](int n0, int y0, int x0) -> unsigned {
    unsigned vm = 0;
    const int ylo = 1 - y0, yhi = a.e.h - y0 + 1, xlo = 1 - x0, xhi = a.e.w_ - x0 + 1;      // limits of the halo row / column inside the image
    const long long pb = (long long)(n0 * a.e.h + y0 - 1) * a.e.w_ + (x0 - 1);                  // halo origin in pixels
    const long long bp = pb * a.ldg;                                                             // (may lie before the tensor: never dereferenced)
    const T* gb = reinterpret_cast<const T*>(a.g) + bp;
    const T* yb = reinterpret_cast<const T*>(a.yraw) + bp;
    const unsigned centre = (unsigned)(g_rowstride + a.ldg) + gy_lane;                         // the tile's first pixel: always inside
#pragma unroll
    for (int j = 0; j < DI; ++j) {
      const int pk_ = d_item(j), L = pk_ >> 6, c = pk_ & 63;
      const bool ok = pk_ != 0x3ff && L >= ylo && L < yhi && c >= xlo && c < xhi;
      vm |= (ok ? 1u : 0u) << j;
      unsigned off = ok ? (unsigned)(__mul24(L, g_rowstride) + __mul24(c, a.ldg)) + gy_lane : centre;   // (outside: a valid pixel, zeroed at the LDS store)
      asm volatile("" : "+v"(off));
      if constexpr (HG) {
        int poff = ok ? __mul24(L, a.e.w_) + c : a.e.w_ + 1;
        asm volatile("" : "+v"(poff));
        rdl[j] = reinterpret_cast<const float2*>(a.hg_dl)[pb + poff];
        ry[j] = gload8<T>(yb + off);
      } else if (!FABL(32)) { rg[j] = gload8<T>(gb + off); ry[j] = gload8<T>(yb + off); }
      else { rg[j] = zero8<T>(); ry[j] = zero8<T>(); rg[j].q[0].x = off; }
    }
    return vm;
  };
  // the pooled gradient and the arg-max bytes of every halo item's window (small, L2-friendly tensors): requested late -- after the
  // MFMA phases -- so that their 6 registers per item are not live beside the accumulators and fragments
  auto issue_pool = [&](int n0, int y0, int x0) {
    if constexpr (POOL) {
      const int ylo = 1 - y0, yhi = a.e.h - y0 + 1, xlo = 1 - x0, xhi = a.e.w_ - x0 + 1;
      const size_t pb = (size_t)(n0 * hp + (y0 >> 1)) * wpool + (x0 >> 1);
#pragma unroll
      for (int j = 0; j < DI; ++j) {
        const int pk_ = d_item(j), L = pk_ >> 6, c = pk_ & 63;
        const bool ok = pk_ != 0x3ff && L >= ylo && L < yhi && c >= xlo && c < xhi;
        // the window of halo pixel (L - 1, c - 1) relative to the tile's first window; outside items read that first window
        int po = ok ? __mul24((L - 1) >> 1, wpool) + ((c - 1) >> 1) : 0;
        asm volatile("" : "+v"(po));
        rp[j] = gload8<T>(reinterpret_cast<const T*>(a.dp) + (pb + po) * a.lddp + gy_lane);
        ra[j] = *reinterpret_cast<const uint2*>(a.amax + (pb + po) * COUT + gy_lane);
      }
    }
  };
  auto issue_x = [&](int n0, int y0, int x0) {
    const size_t bp = (size_t)(n0 * a.e.h + y0) * a.e.w_ + x0;
    const T* xb = xsrc + bp * xcs + x_eoff0;
#pragma unroll
    for (int j = 0; j < XI; ++j) { if (!FABL(32)) rx[j] = gload8<T>(xb + (size_t)j * x_estep); else { rx[j] = zero8<T>(); rx[j].q[0].x = (unsigned)(size_t)xb; } }
  };

  // ---- XCD-aware contiguous tile ranges (conv_igemm_ws.hip)
  const int G = gridDim.x;
  const int xcd = blockIdx.x & 7, nx = G >> 3, remx = G & 7;
  const int bid = (xcd < remx ? xcd * (nx + 1) : remx * (nx + 1) + (xcd - remx) * nx) + (blockIdx.x >> 3);
  const int per = total_tiles / G, extra = total_tiles % G;
  const int t_lo = bid * per + (bid < extra ? bid : extra), t_hi = t_lo + per + (bid < extra ? 1 : 0);

  // weight-gradient products of this wave: id = wave + t * NW  ->  (tap, ci-tile, co-tile); scalar LDS offsets of their operands
  f32x16 wacc[WT];
  int w_xo[WT], w_do[WT];
#pragma unroll
  for (int t = 0; t < WT; ++t) {
#pragma unroll
    for (int i = 0; i < 16; ++i) wacc[t][i] = 0.f;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int id = t < FULL ? wv + t * NW : FULL * NW + wv / WPP;
    const int tap = id % 9, pair = id / 9;
    const int ci_t = pair % (CIN / 32), co_t = pair / (CIN / 32);
    w_xo[t] = ci_t * 4 * XSTRIDE;
    w_do[t] = co_t * 4 * DSTRIDE + ((2 - tap / 3) * PITCH + (2 - tap % 3)) * EL;      // dy[q + 1 - tap] in halo coordinates: q + 2 - tap
    if (t >= FULL) {                                                          // this wave's K range of the shared product: k-steps k0 ... k0 + KPW - 1 (k0 even)
      const int k0 = (wv % WPP) * KPW;
      w_xo[t] += k0 * 16 * EL;
      w_do[t] += (k0 >> 1) * PITCH * EL;
    }
  }

  // (not in the one-wave-per-SIMD instantiation: its 120 registers of prefetched tile leave no room for 16 more accumulators)
  constexpr bool BST = !NODG && (WPS != 1 || POOL);      // (not in the dense one-wave-per-SIMD form; the pooled one has the room)
  float bs1[8], bs2[8];                                                       // fused sums of the layer below: this thread's 8 channels, all its tiles
#pragma unroll
  for (int e = 0; e < 8; ++e) { bs1[e] = 0.f; bs2[e] = 0.f; }
  int n0, y0, x0;
  if (t_lo < t_hi) { tile_origin(t_lo, n0, y0, x0); valid = issue_loads(n0, y0, x0); issue_x(n0, y0, x0); issue_pool(n0, y0, x0); }
  __syncthreads();                                                            // weights and tables are in LDS
  for (int t = t_lo; t < t_hi; ++t) {
    // ---- registers -> LDS: dy from (g, y) with the BatchNorm-backward coefficients, zero outside the image; x with its BatchNorm + ReLU
    {
      // this thread's 8 channels: scale, shift, B, C -- read once per tile (the opaque offset keeps the 32 values from being hoisted out
      // of the tile loop, where they would stay live through the MFMA phases)
      int toff = slot_d * 32;
      asm volatile("" : "+v"(toff));
      const float4* tp = reinterpret_cast<const float4*>(tabD + toff);
      float prm[32];
#pragma unroll
      for (int k = 0; k < 8; ++k) { const float4 v4 = tp[k]; prm[4 * k] = v4.x; prm[4 * k + 1] = v4.y; prm[4 * k + 2] = v4.z; prm[4 * k + 3] = v4.w; }
#pragma unroll
      for (int j = 0; j < DI; ++j) {
        float gv[8], yv[8];
        if constexpr (HG) {
          // the head's data gradient as satcv_head_bwd would have stored it: dl0 w[c][0] + dl1 w[c][1] in that order, rounded to bf16
          int hoff = slot_d * 16;
          asm volatile("" : "+v"(hoff));
          const float4* hp = reinterpret_cast<const float4*>(tabH + hoff);
          const float4 w00 = hp[0], w01 = hp[1], w10 = hp[2], w11 = hp[3];
          const float w0[8] = {w00.x, w00.y, w00.z, w00.w, w01.x, w01.y, w01.z, w01.w};
          const float w1[8] = {w10.x, w10.y, w10.z, w10.w, w11.x, w11.y, w11.z, w11.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) gv[e] = (float)(T)fmaf(rdl[j].y, w1[e], rdl[j].x * w0[e]);
        } else {
          unpack8<T>(rg[j], gv);
        }
        unpack8<T>(ry[j], yv);
        if constexpr (POOL) {
          float pv[8];
          unpack8<T>(rp[j], pv);
          const int pk2 = d_item(j);
          const unsigned sub = ((((pk2 >> 6) + 1) & 1) << 1) | (((pk2 & 63) + 1) & 1);      // position of halo pixel (L - 1, c - 1) in its 2 x 2 window
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const unsigned am = ((e < 4 ? ra[j].x : ra[j].y) >> (8 * (e & 3))) & 0xffu;
            gv[e] += am == sub ? pv[e] : 0.f;
          }
        }
        Raw8<T> v;
        if constexpr (FABL(2) && !HG) { v = rg[j]; v.q[0].x ^= ry[j].q[0].x; }
        else {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float act = yv[e] * prm[e] + prm[8 + e];
            const float gm = act > lin_lo ? gv[e] : 0.f;
            o[e] = (bf16)(prm[e] * gm + (prm[16 + e] * yv[e] + prm[24 + e]));
          }
          v.q[0] = __builtin_bit_cast(uint4, o);
        }
        v = select8<T>((valid >> j) & 1u, v);
        const int pk_ = d_item(j);
        if (pk_ != 0x3ff) lstore8<T>(ldsD + slot_d * DSTRIDE + __mul24((pk_ >> 6) * CL + (pk_ & 63), EL), v);
        __builtin_amdgcn_sched_barrier(0);                                    // (one item's unpacked values at a time)
      }
    }
    {
      float sc[8], sh[8];
      if (xaff) {
        int toff = slot_x * 16;
        asm volatile("" : "+v"(toff));
        const float4* tp = reinterpret_cast<const float4*>(tabX + toff);
        const float4 s0 = tp[0], s1 = tp[1], h0 = tp[2], h1 = tp[3];
        sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
        sh[0] = h0.x; sh[1] = h0.y; sh[2] = h0.z; sh[3] = h0.w; sh[4] = h1.x; sh[5] = h1.y; sh[6] = h1.z; sh[7] = h1.w;
      }
#pragma unroll
      for (int j = 0; j < XI; ++j) {
        Raw8<T> v = rx[j];
        if (xaff) v = affine8_lim(v, sc, sh, xrelu_lim);
        lstore8<T>(ldsX + x_l0 + j * (XPIX_STEP * EL), v);
      }
    }
    __syncthreads();
    // ---- the next tile's loads: in flight during this tile's MFMA phases and epilogue
    const int tn0 = n0, ty0 = y0, tx0 = x0;
    if (t + 1 < t_hi) { next_origin(n0, y0, x0); valid = issue_loads(n0, y0, x0); }

    // ---- weight gradient: K = the 256 interior pixels, 16 per step (half a tile row); x by transposing reads, dy likewise at the tap's shift
    // (product-major: the 16 k-steps of one (ci-tile, co-tile, tap) product run as one software-pipelined chain on its accumulator --
    //  the fragment reads of step k + 1 are issued before the MFMA of step k and pinned there, so the LDS latency is exposed once per
    //  product, not once per step; a wave's products are id = wave + tt * NW, only the last of which can fall beyond NTILE)
#pragma unroll
    for (int tt = 0; tt < FULL; ++tt) {
      if (!FABL(8) && (!GUARD || tt < FULL - 1 || __builtin_amdgcn_readfirstlane(wave) + (FULL - 1) * NW < NTILE)) {
        const T* xa = ldsX + w_xo[tt] + tr_x;
        const T* da = ldsD + w_do[tt] + tr_d;
        // (a real loop over groups of four k-steps = two tile rows, the prefetched fragments carried across its iterations: fully unrolled,
        //  the 48 MFMA blocks of a wave cost ~20 registers more than the kernel has; the read issued by the last step wraps to step 0
        //  and is discarded)
        bf16x4 fa[2][2], fb[2][2];
        auto read_k = [&](const T* xp, const T* dp, int u, int buf) {           // u: k-step inside the group
          const int xq = u * 16 * EL, dq = ((u >> 1) * PITCH + (u & 1) * 16) * EL;
          fa[buf][0] = tr_read4(xp + xq); fa[buf][1] = tr_read4(xp + xq + 4 * EL);
          fb[buf][0] = tr_read4(dp + dq); fb[buf][1] = tr_read4(dp + dq + 4 * EL);
        };
        read_k(xa, da, 0, 0);
#pragma unroll 1
        for (int grp = 0; grp < TH / 2; ++grp) {
          const T* xp = xa + grp * (64 * EL);
          const T* dp = da + grp * (2 * PITCH * EL);
          const int wrap = grp + 1 < TH / 2 ? 1 : 1 - TH / 2;                     // next group, or back to the first after the last
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            asm volatile("" ::: "memory");
            if (u < 3) read_k(xp, dp, u + 1, (u + 1) & 1);
            else read_k(xp + wrap * (64 * EL), dp + wrap * (2 * PITCH * EL), 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 afr = __builtin_shufflevector(fa[u & 1][0], fa[u & 1][1], 0, 1, 2, 3, 4, 5, 6, 7);
            const bf16x8 bfr = __builtin_shufflevector(fb[u & 1][0], fb[u & 1][1], 0, 1, 2, 3, 4, 5, 6, 7);
            if constexpr (!FABL(1)) wacc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr, wacc[tt], 0, 0, 0);
            else asm volatile("" :: "v"(afr), "v"(bfr));
          }
        }
      }
    }
    if constexpr (REM > 0) {
      if (!FABL(8)) {
        // this wave's share of a left-over product: KPW k-steps from its K offset (folded into w_xo / w_do), same one-step-ahead reads
        const T* xa = ldsX + w_xo[FULL] + tr_x;
        const T* da = ldsD + w_do[FULL] + tr_d;
        bf16x4 fa[2][2], fb[2][2];
        auto read_k = [&](int u, int buf) {
          const int xq = u * 16 * EL, dq = ((u >> 1) * PITCH + (u & 1) * 16) * EL;
          fa[buf][0] = tr_read4(xa + xq); fa[buf][1] = tr_read4(xa + xq + 4 * EL);
          fb[buf][0] = tr_read4(da + dq); fb[buf][1] = tr_read4(da + dq + 4 * EL);
        };
        read_k(0, 0);
#pragma unroll
        for (int u = 0; u < KPW; ++u) {
          asm volatile("" ::: "memory");
          if (u + 1 < KPW) read_k(u + 1, (u + 1) & 1);
          __builtin_amdgcn_sched_barrier(0);
          const bf16x8 afr = __builtin_shufflevector(fa[u & 1][0], fa[u & 1][1], 0, 1, 2, 3, 4, 5, 6, 7);
          const bf16x8 bfr = __builtin_shufflevector(fb[u & 1][0], fb[u & 1][1], 0, 1, 2, 3, 4, 5, 6, 7);
          if constexpr (!FABL(1)) wacc[FULL] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr, wacc[FULL], 0, 0, 0);
          else asm volatile("" :: "v"(afr), "v"(bfr));
        }
      }
    }
    // (the next tile's x items are requested here rather than with g and y: 16 registers less through the weight-gradient phase)
    if (t + 1 < t_hi) issue_x(n0, y0, x0);
    if (POOL && NODG && t + 1 < t_hi) issue_pool(n0, y0, x0);
    if constexpr (!NODG) {
    // ---- data gradient (after the weight gradient: its accumulators are then live only from here to the epilogue): 9 taps x COUT / 16 k-steps on the dy halo image (conv_igemm_ws.hip's loop with dy as the input)
    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
    {
      constexpr int KS = COUT / 16, STEPS = 9 * KS;
      FragT<T> af[2][MT], bf[2][NT];
      auto read_step = [&](int st, int buf) {
        const int tap = st / KS, ks = st % KS;
        const int tap_off = ((tap / 3) * PITCH + (tap % 3)) * EL;
        const int slot = ks * 2 + hh;
#pragma unroll
        for (int m = 0; m < MT; ++m) af[buf][m] = lds_frag<T>(ldsD + slot * DSTRIDE + a_off[m] + tap_off);
#pragma unroll
        for (int n = 0; n < NT; ++n) bf[buf][n] = lds_frag<T>(ldsW + ((tap * SD + slot) * CIN + n * 32 + r) * EL);
      };
      if constexpr (!FABL(8)) {
      read_step(0, 0);
#pragma unroll
      for (int st = 0; st < STEPS; ++st) {
        asm volatile("" ::: "memory");
        if (st + 1 < STEPS) read_step(st + 1, (st + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            if constexpr (!FABL(1)) mma32<T>(acc[m][n], af[st & 1][m], bf[st & 1][n]);
            else asm volatile("" :: "v"(af[st & 1][m].v), "v"(bf[st & 1][n].v));
          }
      }
      }
    }
    if (POOL && t + 1 < t_hi) issue_pool(n0, y0, x0);
    __syncthreads();                                                          // every wave is past its last fragment read: the tile region is free
    {
      // dx tile: accumulators -> bf16 -> LDS [pixel][CIN + 8] -> 16-byte row stores (the shared igemm_epilogue's interior path without
      // its statistics / bias / pooling branches: their live state cost this kernel ~30 registers it does not have)
      constexpr int OPITCH = CIN + 8, VPR = CIN / 8;
      T* ldsO = reinterpret_cast<T*>(smem_raw + O_OFF);
      T* op = ldsO + ((wave * MT) * 32 + 4 * hh) * OPITCH + r;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int i = 0; i < 16; ++i) op[(m * 32 + (i & 3) + 8 * (i >> 2)) * OPITCH + n * 32] = (T)acc[m][n][i];
      __syncthreads();
      const int vq = tid % VPR;                                               // this thread's 16-byte column group (NTHREADS % VPR == 0)
      T* yp = reinterpret_cast<T*>(a.e.y) + ((size_t)(tn0 * a.e.h + ty0) * a.e.w_ + tx0) * a.e.ldy + vq * 8;
      const unsigned row_pitch = (unsigned)a.e.w_ * a.e.ldy;
      const T* xact = ldsX + vq * XSTRIDE;                                    // this thread's channel group of the activated input
#pragma unroll 2
      for (int it = tid; it < BM * VPR; it += NTHREADS) {
        const int q = it / VPR;
        const uint4 dv = *reinterpret_cast<const uint4*>(ldsO + q * OPITCH + vq * 8);
        if constexpr (!FABL(4)) *reinterpret_cast<uint4*>(yp + (size_t)((q / TW) * row_pitch + (q % TW) * (unsigned)a.e.ldy)) = dv;
        else asm volatile("" :: "v"(dv.x));
        if (BST && a.bst_sums) {
          // sums of the BatchNorm backward below from the STORED gradient and the staged activation a = relu(sc * v + sh):
          // gm = dx * [a > 0];  sum gm * xhat = (sum dx * a - (sh + sc * mean) * sum gm) * rstd / sc   (a = 0 where the mask is 0)
          const bf16x8 d8 = __builtin_bit_cast(bf16x8, dv);
          const bf16x8 a8 = *reinterpret_cast<const bf16x8*>(xact + q * EL);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float df = (float)d8[e], af = (float)a8[e];
            bs1[e] += af > 0.f ? df : 0.f;
            bs2[e] += df * af;
          }
        }
      }
    }
    }
    __syncthreads();                                                          // staged output read out before the next tile is written
  }
  // ---- this workgroup's partial weight gradient: ws[block][tap][ci][co]
  if constexpr (REM > 0) {
    // the K slices of a left-over product: summed in wave order by the first wave of its group (the tile region is free: the loop ends with a barrier)
    float* red = reinterpret_cast<float*>(smem_raw);                          // [NW][16][64]
#pragma unroll
    for (int i = 0; i < 16; ++i) red[(wave * 16 + i) * 64 + lane] = wacc[FULL][i];
    __syncthreads();
    if (wave % WPP == 0) {
#pragma unroll 1
      for (int k = 1; k < WPP; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) wacc[FULL][i] += red[((wave + k) * 16 + i) * 64 + lane];
    }
  }
#pragma unroll
  for (int tt = 0; tt < WT; ++tt) {
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int id = tt < FULL ? wv + tt * NW : FULL * NW + wv / WPP;
    if ((tt < FULL && (!GUARD || id < NTILE)) || (tt >= FULL && wv % WPP == 0)) {
      const int tap = id % 9, pair = id / 9;
      const int ci_t = pair % (CIN / 32), co_t = pair / (CIN / 32);
      float* dst = a.ws + ((size_t)(blockIdx.x * 9 + tap) * CIN + ci_t * 32) * COUT + co_t * 32 + r;
#pragma unroll
      for (int i = 0; i < 16; ++i) dst[(size_t)((i & 3) + 8 * (i >> 2) + 4 * hh) * COUT] = wacc[tt][i];
    }
  }
  // ---- fused sums: threads of one channel group are summed through LDS in a fixed order; one pair of atomics per channel and workgroup
  if (BST && a.bst_sums && t_lo < t_hi) {
    constexpr int VPR = CIN / 8, NPER = NTHREADS / VPR;                       // threads per channel group
    float* red = reinterpret_cast<float*>(smem_raw);                          // [16][NTHREADS]: the tile region is free now
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[e * NTHREADS + tid] = bs1[e]; red[(8 + e) * NTHREADS + tid] = bs2[e]; }
    __syncthreads();
    if (tid < CIN) {
      const int vq = tid >> 3, e = tid & 7;                                   // channel tid = vq * 8 + e; its contributors: threads t with t % VPR == vq
      double t1 = 0.0, t2 = 0.0;
      for (int k = 0; k < NPER; ++k) { t1 += (double)red[e * NTHREADS + k * VPR + vq]; t2 += (double)red[(8 + e) * NTHREADS + k * VPR + vq]; }
      double s2 = t2;                                                         // activated form: the finalize converts
      if (!a.bst_act_form) {
        const double sc = (double)a.in_scale[tid], sh = (double)a.in_shift[tid], mu = (double)a.bst_mean[tid], rs = (double)a.bst_rstd[tid];
        s2 = sc != 0.0 ? (t2 - (sh + sc * mu) * t1) * (rs / sc) : 0.0;
      }
      satcv_stat_t* rowp = a.bst_sums + (size_t)(blockIdx.x % SATCV_STAT_ROWS) * 2 * a.bst_ld;
      atomicAdd(rowp + tid, (satcv_stat_t)t1);
      atomicAdd(rowp + a.bst_ld + tid, (satcv_stat_t)s2);
    }
  }
}

// ------------------------------------------------------------------ host side
template <int CIN, int COUT, int NW, bool NODG = false>
struct BwdfGeom {
  static constexpr int SD = COUT / 8, SX = CIN / 8;
  static constexpr size_t R0 = ((size_t)(SD * (10 * 34 * 8) + SX * (256 * 8 + 32)) * 2 + 127) / 128 * 128;
  static constexpr size_t O_BYTES = (size_t)256 * (CIN + 8) * 2;
  static constexpr bool O_ALIAS = O_BYTES <= (size_t)SD * (10 * 34 * 8) * 2;
  static constexpr size_t BASE = R0 + (NODG ? 0 : (size_t)9 * SD * CIN * 16) + (size_t)(SD * 32 + SX * 16 + 2 * COUT) * 4;
  static constexpr size_t LDS0 = O_ALIAS ? BASE : (BASE + 127) / 128 * 128 + O_BYTES;
  static constexpr size_t RED = (size_t)16 * NW * 64 * 4;                      // end-of-kernel reduction scratch of the fused sums
  static constexpr size_t LDS = LDS0 > RED ? LDS0 : RED;
};

static int g_ncu = 0;
static int bwdf_grid(size_t lds, int waves, int wps, long long total) {
  if (!g_ncu) {
    int dev = 0; hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return -1;
    g_ncu = p.multiProcessorCount;
  }
  int per_cu = (int)((160 * 1024) / lds);
  const int by_waves = 4 * wps / waves;                             // wps waves per SIMD
  if (per_cu > by_waves) per_cu = by_waves;
  if (per_cu < 1) per_cu = 1;
  long long grid = (long long)g_ncu * per_cu;
  if (grid > total) grid = total;
  return (int)grid;
}

static bool bwdf_shape_ok(const satcv_bwdf_desc* d, int& cin_s) {
  cin_s = d->c0 + d->c1;
  if (d->dtype != SATCV_BF16 || d->kh != 3 || d->kw != 3 || d->dil != 1) return false;
  if (d->x1 && d->c0 % 8 != 0) return false;
  if (d->h % 8 != 0 || d->w_ % 32 != 0) return false;               // whole 8 x 32 tiles
  if (d->ldg % 8 != 0 || ((uintptr_t)d->g % 16) != 0 || ((uintptr_t)d->yraw % 16) != 0) return false;
  if (d->hg_dlogits && (!d->hg_w || d->hg_ncls != 2 || d->dpool || ((uintptr_t)d->hg_dlogits % 8) != 0)) return false;
  if (d->dx && (d->lddx % 8 != 0 || ((uintptr_t)d->dx % 16) != 0)) return false;
  if ((long long)d->w_ * d->ldg >= (1 << 23)) return false;         // 24-bit multiplies of the halo offsets
  if (d->dpool) {
    // encoder blocks: 32 -> 64 channels with a data gradient, or the first block (16 stored channels -> 32) without one
    if (!d->amax || d->lddp % 8 != 0 || ((uintptr_t)d->dpool % 16) != 0 || ((uintptr_t)d->amax % 8) != 0 || d->x1) return false;
    if (d->dx) return cin_s == 32 && d->cin == 32 && d->cout == 64;
    return cin_s == 16 && d->cin <= 16 && d->cout == 32 && !d->bst_sums;
  }
  if (!d->dx) return false;
  if (d->bst_sums && cin_s == 64 && d->cout == 64) return false;    // (that instantiation does not carry the fused sums)
  if (!((cin_s == 32 && d->cout == 32) || (cin_s == 64 && d->cout == 32) || (cin_s == 64 && d->cout == 64))) return false;
  if (d->cin != cin_s) return false;                                // real == stored input channels (the slab has no padding rows)
  return true;
}

template <int CIN, int COUT, int NW, int WPS, bool POOL = false, bool NODG = false, int CINS = CIN, bool HG = false>
static int bwdf_launch(const satcv_bwdf_desc* d, hipStream_t st, bool query, int64_t* ws_bytes, satcv_reduce_job* job = nullptr) {
  using G = BwdfGeom<CIN, COUT, NW, NODG>;
  static_assert(G::LDS <= 160 * 1024, "tile + weights exceed the LDS");
  const long long total = (long long)d->n * (d->h / 8) * (d->w_ / 32);
  const int grid = bwdf_grid(G::LDS, NW, WPS, total);
  if (grid <= 0) { satcv_set_error("bwd_fused: device query failed"); return SATCV_ERR_HIP; }
  const size_t need = (size_t)grid * 9 * CIN * COUT * sizeof(float);
  if (job) { reduce_job_fill(job, d->workspace, d->dw, grid, 9, CIN, COUT, d->cin, COUT, 0, d->accumulate); return SATCV_OK; }
  if (query) { *ws_bytes = (int64_t)need; return SATCV_OK; }
  SATCV_CHECK((size_t)d->workspace_bytes >= need, "bwd_fused: workspace %lld < %zu", (long long)d->workspace_bytes, need);
  BwdfArgs a;
  memset(&a, 0, sizeof(a));
  a.e.y = d->dx; a.e.ldy = d->lddx; a.e.n = d->n; a.e.h = d->h; a.e.w_ = d->w_; a.e.cout = CIN; a.e.cout_pad = CIN; a.e.cstat = CIN;
  a.e.imgs = 1; a.e.rpi = 8; a.e.tiles_x = d->w_ / 32; a.e.tiles_y = d->h / 8; a.e.n_tiles = 1;
  a.g = d->g; a.yraw = d->yraw; a.ldg = d->ldg;
  a.bn_scale = d->bn_scale; a.bn_shift = d->bn_shift; a.bn_mean = d->bn_mean; a.bn_rstd = d->bn_rstd; a.bn_coef = d->bn_coef; a.bn_c = d->cout; a.linear = d->linear;
  a.x0 = d->x0; a.x1 = d->x1; a.c0 = d->c0; a.c1 = d->c1; a.in_scale = d->in_scale; a.in_shift = d->in_shift; a.in_relu = d->in_relu;
  a.w = d->w_dgrad; a.ws = d->workspace; a.tiles_x = d->w_ / 32; a.tiles_y = d->h / 8;
  a.bst_sums = d->bst_sums; a.bst_ld = d->bst_sums_ld; a.bst_mean = d->bst_mean; a.bst_rstd = d->bst_rstd; a.bst_act_form = d->bst_act_form;
  a.dp = d->dpool; a.lddp = d->lddp; a.amax = reinterpret_cast<const unsigned char*>(d->amax);
  a.hg_dl = d->hg_dlogits; a.hg_w = d->hg_w;
  auto kern = bwd_fused_kernel<CIN, COUT, NW, WPS, POOL, NODG, CINS, HG>;
  { const int rc = satcv_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), G::LDS); if (rc) return rc; }
  const double flops = (NODG ? 2.0 : 4.0) * d->n * d->h * d->w_ * (double)CINS * COUT * 9;          // (data gradient +) weight gradient
  satcv_prof_begin(3, flops, st);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), G::LDS, st, a, (int)total);
  satcv_prof_end(3, st);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { satcv_set_error("bwd_fused launch: %s", hipGetErrorString(e)); return SATCV_ERR_HIP; }
  if (d->defer_reduce) return SATCV_OK;              // the caller sums the slabs later (satcv_reduce_slabs_batched)
  return wgrad_reduce_slabs(d->workspace, d->dw, grid, 9, CIN, COUT, d->cin, COUT, d->accumulate, st);
}

static int bwdf_dispatch(const satcv_bwdf_desc* d, hipStream_t st, bool query, int64_t* ws_bytes, satcv_reduce_job* job = nullptr) {
  int cin_s;
  if (!bwdf_shape_ok(d, cin_s)) return SATCV_ERR_UNSUPPORTED;
  if (d->dpool) {
    // (32 -> 64 with the pooled gradient: 92 registers of prefetched items -- the 8-wave form needs 172 bytes of scratch; one wave per SIMD fits)
    if (d->dx) return bwdf_launch<32, 64, 4, 1, true, false>(d, st, query, ws_bytes, job);
    return bwdf_launch<32, 32, 8, 2, true, true, 16>(d, st, query, ws_bytes, job);
  }
  // (32 -> 32: 8 waves x 1 workgroup per CU measured equal to 4 waves x 2 workgroups and leaves registers for the fused sums; with the
  //  weight-gradient products balanced over the waves the 4-wave form -- 256 registers, 32 bytes of scratch -- measured 365 vs 346 us)
  if (cin_s == 32 && d->hg_dlogits) return bwdf_launch<32, 32, 8, 2, false, false, 32, true>(d, st, query, ws_bytes, job);
  if (d->hg_dlogits) return SATCV_ERR_UNSUPPORTED;
  if (cin_s == 32) return bwdf_launch<32, 32, 8, 2>(d, st, query, ws_bytes, job);
  if (d->cout == 32) return bwdf_launch<64, 32, 8, 2>(d, st, query, ws_bytes, job);
  return bwdf_launch<64, 64, 4, 1>(d, st, query, ws_bytes, job);
}

extern "C" int satcv_conv2d_bwd_fused_reduce_job(const satcv_bwdf_desc* d, satcv_reduce_job* job) {
  SATCV_CHECK(d && job && d->dw && d->workspace, "bwd_fused_reduce_job: null pointer");
  return bwdf_dispatch(d, nullptr, false, nullptr, job);
}

extern "C" int64_t satcv_conv2d_bwd_fused_workspace(const satcv_bwdf_desc* d) {
  int64_t nb = -1;
  if (!d || bwdf_dispatch(d, nullptr, true, &nb) != SATCV_OK) return -1;
  return nb;
}

extern "C" int satcv_conv2d_bwd_fused(const satcv_bwdf_desc* d, void* stream) {
  SATCV_CHECK(d && (d->g || d->hg_dlogits) && d->yraw && d->x0 && d->dw && d->workspace && (d->dx == nullptr || d->w_dgrad), "bwd_fused: null pointer");
  SATCV_CHECK(d->bn_scale && d->bn_shift && d->bn_mean && d->bn_rstd && d->bn_coef, "bwd_fused: BatchNorm coefficients missing");
  SATCV_CHECK((d->c1 == 0) == (d->x1 == nullptr) && d->n > 0 && d->h > 0 && d->w_ > 0, "bwd_fused: bad dims");
  SATCV_CHECK(!d->bst_sums || (d->bst_sums_ld >= d->c0 + d->c1 && (d->bst_act_form ? d->in_scale == nullptr
                               : (d->in_scale && d->in_shift && d->in_relu && d->bst_mean && d->bst_rstd))),
              "bwd_fused: the fused sums need the input's BatchNorm (in_scale / in_shift with ReLU, bst_mean / bst_rstd), or an activated input with bst_act_form");
  const int rc = bwdf_dispatch(d, reinterpret_cast<hipStream_t>(stream), false, nullptr);
  if (rc == SATCV_ERR_UNSUPPORTED) satcv_set_error("bwd_fused: shape outside the kernel's limits (ask satcv_conv2d_bwd_fused_workspace first)");
  return rc;
}
