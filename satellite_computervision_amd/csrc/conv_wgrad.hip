// Weight gradient of the implicit-GEMM convolutions on MFMA (gfx950).
//
//   dW[tap][ci][co] = sum_pixels  X[pixel + tap][ci] * dY[pixel][co]
//
// GEMM view: M = ci, N = co, K = pixels.  Both operands are stored pixel-major (NHWC), i.e.
// K is the SLOW dimension of both, so the MFMA fragments (8 consecutive k per lane) are
// gathered with the transposing LDS read ds_read_b64_tr_b16 (bf16) from row-major
// [pixel][channel] LDS images; fp32 storage uses v_mfma_f32_32x32x2_f32 with one dword per
// lane.  As in the forward kernel the 9 taps are address offsets of one halo image of X.
// Each workgroup owns a (ci-block, co-block) of dW for ALL taps, walks a strided subset of
// the pixel tiles (split-K) and writes an fp32 partial slab; a second kernel sums the slabs
// in fixed order (deterministic) into the Keras-layout gradient.
#include "common.hpp"
#include <cstdlib>
#include <cstring>

// compile-time ablation for profiling builds (-DSATCV_WABLATE=bits): 1 skip the global loads, 2 skip the MFMAs, 4 read every
// fragment from LDS offset 0 (no address arithmetic / bank pattern), 8 skip the LDS staging stores
#ifndef SATCV_WABLATE
#define SATCV_WABLATE 0
#endif
#define WABL(bit) ((SATCV_WABLATE & (bit)) != 0)
#ifdef SATCV_STAMP
// diagnostic build only: per-wave cycle sums of the phases of the pixel-tile loop (s_memtime), first 8 workgroups
__device__ unsigned long long g_wstamp[8][4][8];
#define WSTAMP(t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
extern "C" int satcv_debug_read_wstamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wstamp), sizeof(g_wstamp)) == hipSuccess ? 0 : -1;
}
#endif

struct WgradArgs {
  const void* x0; const void* x1; int c0, c1;
  const float* in_scale; const float* in_shift; int in_relu;
  const void* dy; int lddy;
  float* ws;
  int n, h, w_;
  int kh, kw, dil;
  int mode_dy, f, cout_t;      // cout_t: channels per sub-position of a transposed conv
  int kpad, npad;              // padded M (ci) and N (co) extents of the slab
  int cin_lim, n_lim;          // valid channels in X / valid N indices
  int tiles_x, tiles_y, ngroups, rpi, imgs, seg, rl, cl, halh, halw;
  int n_ci_blk, n_co_blk, nsplit, total_ptiles;
  int sy, sx;                  // extra source shift (one tap of a dilated conv handled as a shifted 1x1)
};

template <int TW, int NCI, int NCO, int NTAPS, typename T>
struct WgradGeom {
  static constexpr int CI_T = 32 * NCI, CO_T = 32 * NCO;
  // pixel pitch (elements): 64-byte rows are conflict-free for the transposing read; wider
  // rows are padded by 64 bytes so that 4 consecutive pixels land on distinct bank groups.
  static constexpr int PADE = 64 / (int)sizeof(T);
  static constexpr int XP = CI_T + (CI_T == 32 && sizeof(T) == 2 ? 0 : PADE);
  static constexpr int DP = CO_T + (CO_T == 32 && sizeof(T) == 2 ? 0 : PADE);
};

__device__ __forceinline__ bf16x4 tr_read(const bf16* p) {
  short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(p));
  return __builtin_bit_cast(bf16x4, v);
}

// NCI x NCO waves own distinct (ci, co) 32x32 tiles; NKS waves share a tile and split the
// k-steps (pixels) of every staged tile, each writing its own partial slab.
template <typename T, int TW, int NCI, int NCO, int NKS, int NTAPS, int PIX>
__global__ __launch_bounds__(NCI* NCO* NKS * 64, (TW == 8 ? 1 : 2)) void wgrad_kernel(const WgradArgs a) {
  using G = WgradGeom<TW, NCI, NCO, NTAPS, T>;
  constexpr int NTHREADS = NCI * NCO * NKS * 64;
  constexpr int BMPIX = PIX;               // pixels per staged tile: 128, or 256 for the thin layers (more bytes in flight)
  constexpr int CI_T = G::CI_T, CO_T = G::CO_T, XP = G::XP, DP = G::DP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* ldsX = reinterpret_cast<T*>(smem_raw);
  T* ldsD = ldsX + a.rl * a.cl * XP;
  int* tab = reinterpret_cast<int*>(ldsD + BMPIX * DP);
  int* ptab = tab + BMPIX;                 // halo pixel -> packed (image-in-tile, row, column), tile invariant

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wci = wave % NCI, wco = (wave / NCI) % NCO, wks = wave / (NCI * NCO);
  const int r = lane & 31, hh = lane >> 5;
  // 1-D grid; the (ci, co) blocks that read the SAME pixel tiles get ids congruent mod 8, i.e. run on one XCD and share its
  // L2 (each block re-reads dY / X: with the plain x-fastest order the re-reads of a tile came from 3-8 different L2s)
  int blk, sp;
  {
    const int nblk = a.n_ci_blk * a.n_co_blk, id = blockIdx.x;
    if ((a.nsplit & 7) == 0) { const int xcd = id & 7, j = id >> 3; blk = j % nblk; sp = (j / nblk) * 8 + xcd; }
    else { blk = id % nblk; sp = id / nblk; }
  }
  const int ci_blk = blk % a.n_ci_blk, co_blk = blk / a.n_ci_blk;
  const int ci0 = ci_blk * CI_T, co0 = co_blk * CO_T;

#ifdef SATCV_STAMP
  unsigned long long w0, w1, w2, w3, w4, w5, w6, zs[6] = {0, 0, 0, 0, 0, 0};
  WSTAMP(w0);
#endif
  f32x16 acc[NTAPS];
#pragma unroll
  for (int t = 0; t < NTAPS; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  for (int q = tid; q < BMPIX; q += NTHREADS) {
    const int t = q / TW, cx = q % TW;
    const int k = t / a.rpi;
    const int l0 = (k < a.imgs) ? k * a.seg + (t - k * a.rpi) : 0;
    tab[q] = (l0 * a.cl + cx) * XP;
  }

  for (int pix = tid; pix < a.rl * a.cl; pix += NTHREADS) {
    const int c = pix % a.cl, L = pix / a.cl;
    const int k = L / a.seg;
    const int yy = L - k * a.seg - a.halh;
    ptab[pix] = (k << 20) | ((yy + 64) << 10) | c;
  }
  __syncthreads();

  const T* dyp = reinterpret_cast<const T*>(a.dy);
  const int x_items = a.rl * a.cl * (CI_T / 8);
  constexpr int d_items = BMPIX * (CO_T / 8);
  constexpr int TH = BMPIX / TW;
  // register-staged items per thread: halo tile of a 3x3 / dilation-1 conv incl. the several-images-per-tile case
  constexpr int XMAXPIX = (TH + 2 * (TH / 4 > 1 ? TH / 4 : 1)) * (TW + 2);
  constexpr int XI = (XMAXPIX * (CI_T / 8) + NTHREADS - 1) / NTHREADS;
  constexpr int DI = (d_items + NTHREADS - 1) / NTHREADS;

  Raw8<T> xr[XI];
  Raw8<T> dr[DI];
  bool xv[XI];

  // issue every global load of tile `pt` back to back into registers (memory-level parallelism)
  auto load_tile = [&](int pt) {
    int m = pt;
    const int tx = m % a.tiles_x; m /= a.tiles_x;
    const int ty = m % a.tiles_y;
    const int grp = m / a.tiles_y;
    const int n0 = grp * a.imgs, y0 = ty * TH, x0 = tx * TW;
#pragma unroll
    for (int j = 0; j < XI; ++j) {
      const int it = tid + j * NTHREADS;
      xr[j] = zero8<T>(); xv[j] = false;
      if (it < x_items) {
        const int g = it % (CI_T / 8);
        const int pix = it / (CI_T / 8);
        const int pm = ptab[pix];
        const int c = pm & 1023, yy = ((pm >> 10) & 1023) - 64, k = pm >> 20;
        const int n = n0 + k, y = y0 + yy + a.sy, x = x0 + c - a.halw + a.sx;
        const int cg = ci0 + g * 8;
        if ((n < a.n) && (y >= 0) && (y < a.h) && (x >= 0) && (x < a.w_) && (cg < a.cin_lim)) {
          const T* src; int cs, coff;
          if (cg < a.c0) { src = reinterpret_cast<const T*>(a.x0); cs = a.c0; coff = cg; }
          else { src = reinterpret_cast<const T*>(a.x1); cs = a.c1; coff = cg - a.c0; }
          if (!WABL(1)) xr[j] = gload8<T>(src + ((size_t)(n * a.h + y) * a.w_ + x) * cs + coff);
          xv[j] = true;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < DI; ++j) {
      const int it = tid + j * NTHREADS;
      const int g = it % (CO_T / 8);
      const int q = it / (CO_T / 8);
      const int t = q / TW, cx = q % TW;
      const int k = (a.imgs == 1) ? 0 : t / a.rpi;
      const int nimg = n0 + k, y = y0 + (t - k * a.rpi), x = x0 + cx;
      const int cv = co0 + g * 8;
      dr[j] = zero8<T>();
      if ((it < d_items) && (k < a.imgs) && (nimg < a.n) && (y < a.h) && (x < a.w_) && (cv < a.n_lim)) {
        size_t off;
        if (a.mode_dy == 1) {
          const int ij = cv / a.cout_t, o = cv - ij * a.cout_t;
          off = ((size_t)(nimg * a.h * a.f + y * a.f + ij / a.f) * (a.w_ * a.f) + x * a.f + ij % a.f) * a.lddy + o;
        } else {
          off = ((size_t)(nimg * a.h + y) * a.w_ + x) * a.lddy + cv;
        }
        if (!WABL(1)) dr[j] = gload8<T>(dyp + off);
      }
    }
  };
  // BN affine + ReLU of the producing layer (in registers), then LDS
  auto store_tile = [&]() {
    if (WABL(8)) return;
#pragma unroll
    for (int j = 0; j < XI; ++j) {
      const int it = tid + j * NTHREADS;
      if (it < x_items) {
        const int g = it % (CI_T / 8);
        const int pix = it / (CI_T / 8);
        Raw8<T> v = xr[j];
        if (a.in_scale && xv[j]) v = affine8<T>(v, a.in_scale + ci0 + g * 8, a.in_shift + ci0 + g * 8, a.in_relu);
        lstore8<T>(ldsX + pix * XP + g * 8, v);
      }
    }
#pragma unroll
    for (int j = 0; j < DI; ++j) {
      const int it = tid + j * NTHREADS;
      if (it < d_items) lstore8<T>(ldsD + (it / (CO_T / 8)) * DP + (it % (CO_T / 8)) * 8, dr[j]);
    }
  };
  // fallback for halo tiles larger than the register budget (dilated taps): serial staging
  auto stage_generic = [&](int pt) {
    int m = pt;
    const int tx = m % a.tiles_x; m /= a.tiles_x;
    const int ty = m % a.tiles_y;
    const int grp = m / a.tiles_y;
    const int n0 = grp * a.imgs, y0 = ty * TH, x0 = tx * TW;
    for (int it = tid; it < x_items; it += NTHREADS) {
      const int g = it % (CI_T / 8);
      const int pix = it / (CI_T / 8);
      const int pm = ptab[pix];
      const int c = pm & 1023, yy = ((pm >> 10) & 1023) - 64, k = pm >> 20;
      const int n = n0 + k, y = y0 + yy + a.sy, x = x0 + c - a.halw + a.sx;
      const int cg = ci0 + g * 8;
      Raw8<T> v = zero8<T>();
      if ((n < a.n) && (y >= 0) && (y < a.h) && (x >= 0) && (x < a.w_) && (cg < a.cin_lim)) {
        const T* src; int cs, coff;
        if (cg < a.c0) { src = reinterpret_cast<const T*>(a.x0); cs = a.c0; coff = cg; }
        else { src = reinterpret_cast<const T*>(a.x1); cs = a.c1; coff = cg - a.c0; }
        v = gload8<T>(src + ((size_t)(n * a.h + y) * a.w_ + x) * cs + coff);
        if (a.in_scale) v = affine8<T>(v, a.in_scale + cg, a.in_shift + cg, a.in_relu);
      }
      lstore8<T>(ldsX + pix * XP + g * 8, v);
    }
    for (int it = tid; it < d_items; it += NTHREADS) {
      const int g = it % (CO_T / 8);
      const int q = it / (CO_T / 8);
      const int t = q / TW, cx = q % TW;
      const int k = (a.imgs == 1) ? 0 : t / a.rpi;
      const int nimg = n0 + k, y = y0 + (t - k * a.rpi), x = x0 + cx;
      const int cv = co0 + g * 8;
      Raw8<T> v = zero8<T>();
      if ((k < a.imgs) && (nimg < a.n) && (y < a.h) && (x < a.w_) && (cv < a.n_lim)) {
        size_t off;
        if (a.mode_dy == 1) {
          const int ij = cv / a.cout_t, o = cv - ij * a.cout_t;
          off = ((size_t)(nimg * a.h * a.f + y * a.f + ij / a.f) * (a.w_ * a.f) + x * a.f + ij % a.f) * a.lddy + o;
        } else {
          off = ((size_t)(nimg * a.h + y) * a.w_ + x) * a.lddy + cv;
        }
        v = gload8<T>(dyp + off);
      }
      lstore8<T>(ldsD + q * DP + g * 8, v);
    }
  };
  // 8 k-steps of 16 pixels on the staged tile
  auto mfma_phase = [&]() {
    for (int ks = wks; ks < BMPIX / 16; ks += NKS) {
      if constexpr (std::is_same<T, bf16>::value) {
        const int gi = lane >> 4, i16 = lane & 15;
        const int chb = 16 * (gi & 1) + 4 * (i16 & 3);
        const int qa = ks * 16 + 8 * (gi >> 1) + (i16 >> 2);
        const int qb = qa + 4;
        bf16x4 blo = tr_read(ldsD + qa * DP + wco * 32 + chb);
        bf16x4 bhi = tr_read(ldsD + qb * DP + wco * 32 + chb);
        bf16x8 bfr = __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7);
        const int xoa = tab[qa] + wci * 32 + chb;
        const int xob = tab[qb] + wci * 32 + chb;
        // the transposing reads of tap t+1 are issued before the MFMA of tap t (see conv_igemm_fast.hip)
        bf16x4 alo[2], ahi[2];
        alo[0] = tr_read(ldsX + xoa);
        ahi[0] = tr_read(ldsX + xob);
#pragma unroll
        for (int tap = 0; tap < NTAPS; ++tap) {
          if (tap + 1 < NTAPS) {
            const int ky = (tap + 1) / 3, kx = (tap + 1) % 3;
            const int toff = ((ky * a.dil) * a.cl + kx * a.dil) * XP;
            alo[(tap + 1) & 1] = tr_read(ldsX + (WABL(4) ? 0 : xoa + toff));
            ahi[(tap + 1) & 1] = tr_read(ldsX + (WABL(4) ? 64 : xob + toff));
          }
          __builtin_amdgcn_sched_barrier(0);
          bf16x8 afr = __builtin_shufflevector(alo[tap & 1], ahi[tap & 1], 0, 1, 2, 3, 4, 5, 6, 7);
          if (!WABL(2)) acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr, acc[tap], 0, 0, 0);
          else acc[tap][0] += (float)afr[0] + (float)bfr[0];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int q = ks * 16 + 2 * j + hh;
          const float b = ldsD[q * DP + wco * 32 + r];
          const int xo = tab[q] + wci * 32 + r;
#pragma unroll
          for (int tap = 0; tap < NTAPS; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            const int toff = (NTAPS == 1) ? 0 : ((ky * a.dil) * a.cl + kx * a.dil) * XP;
            acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(ldsX[xo + toff], b, acc[tap], 0, 0, 0);
          }
        }
      }
    }
  };

  if (x_items <= XI * NTHREADS) {
    // register-prefetch pipeline: loads of tile t+1 are in flight while tile t is multiplied
    if (sp < a.total_ptiles) {
      load_tile(sp);
      store_tile();
    }
    __syncthreads();
#ifdef SATCV_STAMP
    WSTAMP(w1);
    int ntl = 0;
#endif
    for (int pt = sp; pt < a.total_ptiles; pt += a.nsplit) {
      const int nxt = pt + a.nsplit;
#ifdef SATCV_STAMP
      WSTAMP(w2); ++ntl;
#endif
      if (nxt < a.total_ptiles) load_tile(nxt);
#ifdef SATCV_STAMP
      WSTAMP(w3);
#endif
      mfma_phase();
#ifdef SATCV_STAMP
      WSTAMP(w4);
#endif
      __syncthreads();
#ifdef SATCV_STAMP
      WSTAMP(w5);
#endif
      if (nxt < a.total_ptiles) {
        store_tile();
#ifdef SATCV_STAMP
        WSTAMP(w6);
        zs[3] += w6 - w5;
#endif
        __syncthreads();
      }
#ifdef SATCV_STAMP
      { unsigned long long w7; WSTAMP(w7); zs[0] += w3 - w2; zs[1] += w4 - w3; zs[2] += w5 - w4; zs[4] += w7 - (nxt < a.total_ptiles ? w6 : w5); }
#endif
    }
#ifdef SATCV_STAMP
    if (blockIdx.x < 8 && lane == 0) {
      for (int i = 0; i < 5; ++i) g_wstamp[blockIdx.x][wave][i] = zs[i] / (unsigned long long)max(ntl, 1);
      g_wstamp[blockIdx.x][wave][5] = w1 - w0;
      g_wstamp[blockIdx.x][wave][6] = ntl;
      WSTAMP(w2);
      g_wstamp[blockIdx.x][wave][7] = w2;
    }
#endif
  } else {
    for (int pt = sp; pt < a.total_ptiles; pt += a.nsplit) {
      __syncthreads();
      stage_generic(pt);
      __syncthreads();
      mfma_phase();
    }
  }
  // ---- combine the k-slice waves of a tile through LDS (fixed order), one tap at a time
  if constexpr (NKS > 1) {
    float* red = reinterpret_cast<float*>(smem_raw);          // staging buffers are dead now
    const int tile_id = wci + NCI * wco;                      // which (ci, co) tile this wave works on
#pragma unroll
    for (int tap = 0; tap < NTAPS; ++tap) {
      __syncthreads();
      if (wks > 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[((tile_id * (NKS - 1) + (wks - 1)) * 16 + i) * 64 + lane] = acc[tap][i];
      }
      __syncthreads();
      if (wks == 0) {
#pragma unroll
        for (int k = 0; k < NKS - 1; ++k)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[tap][i] += red[((tile_id * (NKS - 1) + k) * 16 + i) * 64 + lane];
      }
    }
  }
  // ---- partial slab: ws[sp][tap][ci][co]
  if (wks == 0) {
#pragma unroll
    for (int tap = 0; tap < NTAPS; ++tap) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ci = ci0 + wci * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
        const int co = co0 + wco * 32 + r;
        a.ws[((size_t)(sp * NTAPS + tap) * a.kpad + ci) * a.npad + co] = acc[tap][i];
      }
    }
  }
#ifdef SATCV_STAMP
  if (blockIdx.x < 8 && lane == 0) { unsigned long long w9; WSTAMP(w9); g_wstamp[blockIdx.x][wave][7] = w9 - g_wstamp[blockIdx.x][wave][7]; }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------
// Double-buffered form of the kernel above (the one the training step runs; the kernel above stays as the fallback for halo
// tiles beyond the register budget and for fp32).  s_memtime stamps of the single-buffered loop on 512 -> 512 channels at
// 16 x 16 (tools/wstamp_probe.py): of 10,400 cycles per pixel tile 3,400 went into issuing the next tile's loads (per-item
// table look-ups, divisions, bounds tests and 64-bit addresses, every tile anew), 1,800 into the LDS stores behind a barrier and
// only 4,800 into the MFMA phase (two workgroups per CU sharing the matrix pipe).  Here:
//   * everything tile-invariant about a staged item (LDS slot, halo coordinates, channel group, source, BatchNorm scale / shift
//     of its 8 channels) is computed ONCE per thread and kept in registers; per tile an item costs a bounds test and one address;
//   * no vector-memory instruction sits in a lane-dependent branch (items outside the image load pixel 0 and are zeroed by a select
//     when they go to LDS), so hipcc's counted s_waitcnt stay exact;
//   * two LDS stages, ONE barrier per pixel tile: while tile t is multiplied, tile t+1 moves from registers to the other stage and
//     tile t+2 is loaded into the registers it frees, one item between two groups of MFMAs (never a burst of loads: a burst keeps the
//     wave in its issue phase while the vector-memory pipe takes the instructions in, with the matrix pipe idle);
//   * 8 waves per workgroup (the k-steps of a staged tile are split over two or more wave groups, summed through LDS at the end
//     in fixed order), one workgroup per CU.
//
// NW > 1 (1x1 / transposed convolutions only): a wave owns NW co tiles of one ci tile, i.e. the workgroup covers a
// (32 NCI) x (32 NCO NW) block.  A 1x1 wave holds 16 accumulator registers per tile, so the 64 x 128 block of the NW = 1 form moved
// 49 KB from L2 to LDS per 8 MFMAs of a wave: the five transposed-conv gradients of the U-Net each re-read ~400 MB through L2 and took
// 80-95 us whatever their shape.  The 128 x 256 block (4 x 2 waves x 4 tiles, 64-pixel stages) halves those bytes per MFMA.
template <typename T, int TW, int NCI, int NCO, int NKS, int NTAPS, int PIX, int NW = 1>
__global__ __launch_bounds__(NCI* NCO* NKS * 64, 1) void wgrad_db_kernel(const WgradArgs a) {
  static_assert(NW == 1 || NTAPS == 1, "several co tiles per wave only for single-tap kernels");
  using G = WgradGeom<TW, NCI, NCO * NW, NTAPS, T>;
  constexpr int NACC = NTAPS * NW;
  constexpr int NTHREADS = NCI * NCO * NKS * 64;
  constexpr int BMPIX = PIX, TH = BMPIX / TW;
  constexpr int CI_T = G::CI_T, CO_T = G::CO_T, XP = G::XP, DP = G::DP;
  constexpr int GX = CI_T / 8, GD = CO_T / 8;
  static_assert(NTHREADS % GX == 0 && NTHREADS % GD == 0, "a thread's channel group must be loop-invariant");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int x_elems = a.rl * a.cl * XP;
  const int stage_elems = ((x_elems + BMPIX * DP) + 7) / 8 * 8;
  T* lds0 = reinterpret_cast<T*>(smem_raw);
  int* tab = reinterpret_cast<int*>(lds0 + 2 * stage_elems);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wci = wave % NCI, wco = (wave / NCI) % NCO, wks = wave / (NCI * NCO);
  const int r = lane & 31, hh = lane >> 5;
  int blk, sp;
  {
    const int nblk = a.n_ci_blk * a.n_co_blk, id = blockIdx.x;
    if ((a.nsplit & 7) == 0) { const int xcd = id & 7, j = id >> 3; blk = j % nblk; sp = (j / nblk) * 8 + xcd; }
    else { blk = id % nblk; sp = id / nblk; }
  }
  const int ci_blk = blk % a.n_ci_blk, co_blk = blk / a.n_ci_blk;
  const int ci0 = ci_blk * CI_T, co0 = co_blk * CO_T;

  f32x16 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  for (int q = tid; q < BMPIX; q += NTHREADS) {
    const int t = q / TW, cx = q % TW;
    const int k = t / a.rpi;
    const int l0 = (k < a.imgs) ? k * a.seg + (t - k * a.rpi) : 0;
    tab[q] = (l0 * a.cl + cx) * XP;
  }

  // ---- per-thread, tile-invariant description of the staged items
  const int x_items = a.rl * a.cl * GX;
  constexpr int d_items = BMPIX * GD;
  // (no halo without taps; the 256-pixel tile is only planned for maps of at least its height: one image, one halo)
  constexpr int XMAXPIX = NTAPS == 1 ? BMPIX : (PIX == 256 ? (TH + 2) * (TW + 2) : (TH + 2 * (TH / 4 > 1 ? TH / 4 : 1)) * (TW + 2));
  constexpr int XI = (XMAXPIX * GX + NTHREADS - 1) / NTHREADS;
  constexpr int DI = (d_items + NTHREADS - 1) / NTHREADS;
  constexpr int NU = XI + DI;
  const int gx = tid % GX, cgx = ci0 + gx * 8;
  const bool chv = cgx < a.cin_lim;
  const bool xsecond = chv && a.x1 != nullptr && cgx >= a.c0;
  const T* xsrc = xsecond ? reinterpret_cast<const T*>(a.x1) + (cgx - a.c0) : reinterpret_cast<const T*>(a.x0) + (chv ? cgx : 0);
  const int xcs = xsecond ? a.c1 : a.c0;
  // BatchNorm scale / shift of the producing layer for the CI_T channels of this block: a small LDS table read when an item is
  // written (16 registers per thread otherwise: the kernel sits at the 256-register cap of two waves per SIMD)
  float* ldsS = reinterpret_cast<float*>(tab + BMPIX);                     // [GX][16]: 8 scale + 8 shift per channel group
  const bool aff = a.in_scale != nullptr;
  if (aff && tid < GX * 16) {
    const int g = tid / 16, e = tid % 16, ch = ci0 + g * 8 + (e & 7);
    ldsS[tid] = ch < a.cin_lim ? (e < 8 ? a.in_scale[ch] : a.in_shift[ch]) : (e < 8 ? 1.f : 0.f);
  }
  // ReLU as a lower bound on the packed bf16 pair (signed 16-bit max: negative floats are negative integers): 0, or the most negative
  // value when the layer has no activation -- one packed instruction per channel pair, no per-element select on the runtime flag
  const unsigned relu_lim = a.in_relu != 0 ? 0u : 0x80008000u;
  const float relu_lo = a.in_relu != 0 ? 0.f : -INFINITY;
  // The staged items of a thread are tile-invariant up to the tile's base pixel: element offset from that pixel, LDS destination
  // (base + a compile-time multiple: NTHREADS is a multiple of the channel groups) and the halo coordinates packed for the per-tile
  // validity test.  The tile loop was VALU-bound on exactly this bookkeeping (337 vector instructions per 18 MFMAs on the thin
  // layers, a third of them in per-item (n, y, x) -> address arithmetic with quarter-rate 64-bit multiplies).
  constexpr int XSTEP = (NTHREADS / GX) * XP, DSTEP = (NTHREADS / GD) * DP;
  const int x_dst0 = (tid / GX) * XP + gx * 8;
  int x_rel[XI], x_eoff[XI];
#pragma unroll
  for (int j = 0; j < XI; ++j) {
    const int it = tid + j * NTHREADS;
    const int pix = it / GX;
    const int c = pix % a.cl, L = pix / a.cl;
    const int k = L / a.seg;
    const int yy = L - k * a.seg - a.halh;
    // bit 30: the item exists and its channels do; k | yy + 64 | c as before
    x_rel[j] = ((it < x_items && chv) ? (1 << 30) : 0) | (k << 20) | ((yy + 64) << 10) | c;
    x_eoff[j] = ((k * a.h + yy + a.sy) * a.w_ + (c - a.halw + a.sx)) * xcs;
  }
  const int gd = tid % GD, cv = co0 + gd * 8;
  const bool cvv = cv < a.n_lim;
  int d_ij = 0, d_o = cvv ? cv : 0;
  if (a.mode_dy == 1 && cvv) { d_ij = cv / a.cout_t; d_o = cv - d_ij * a.cout_t; }
  const int d_iy = a.mode_dy == 1 ? d_ij / a.f : 0, d_ix = a.mode_dy == 1 ? d_ij % a.f : 0;
  const int dfm = a.mode_dy == 1 ? a.f : 1;
  // dY: pixel (n, y, x) of the tile grid sits at ((n h + y) dfm + d_iy) (w dfm) + x dfm + d_ix of the stored gradient
  const T* dyp = reinterpret_cast<const T*>(a.dy) + ((size_t)d_iy * (a.w_ * dfm) + d_ix) * a.lddy + d_o;
  const int d_dst0 = (tid / GD) * DP + gd * 8;
  int d_rel[DI], d_eoff[DI];
#pragma unroll
  for (int j = 0; j < DI; ++j) {
    const int it = tid + j * NTHREADS;
    const int q = it / GD;
    const int t = q / TW, cx = q % TW;
    const int k = (a.imgs == 1) ? 0 : t / a.rpi;
    const int row = t - k * a.rpi;
    d_rel[j] = ((it < d_items && k < a.imgs && cvv) ? (1 << 30) : 0) | (k << 20) | (row << 10) | cx;
    d_eoff[j] = (((k * a.h + row) * dfm) * (a.w_ * dfm) + cx * dfm) * a.lddy;
  }

  Raw8<T> xr[XI], dr[DI];
  unsigned xmask = 0, dmask = 0;
  // per tile (wave-uniform): limits of the halo coordinates that fall inside the image, base element offsets of the two tensors
  int klim = 0, ylo = 0, yhi = 0, clo = 0, chi = 0, rhi = 0, xhi = 0;
  size_t xbase = 0, dbase = 0;
  auto tile_origin = [&](int pt) {
    int m = pt;
    const int tx = m % a.tiles_x; m /= a.tiles_x;
    const int ty = m % a.tiles_y;
    const int n0 = (m / a.tiles_y) * a.imgs, y0 = ty * TH, x0 = tx * TW;
    klim = a.n - n0;
    ylo = 64 - (y0 + a.sy); yhi = 64 + a.h - (y0 + a.sy);                 // on the stored yy + 64
    clo = a.halw - a.sx - x0; chi = a.w_ + a.halw - a.sx - x0;
    rhi = a.h - y0; xhi = a.w_ - x0;
    const size_t bp = ((size_t)n0 * a.h + y0) * a.w_ + x0;
    xbase = bp * xcs;
    dbase = (((size_t)n0 * a.h + y0) * dfm * (a.w_ * dfm) + (size_t)x0 * dfm) * a.lddy;
  };
  auto load_x = [&](int j) {
    const int c = x_rel[j] & 1023, yy = (x_rel[j] >> 10) & 1023, k = (x_rel[j] >> 20) & 1023;
    const bool ok = (x_rel[j] >> 30) && (k < klim) && (yy >= ylo) && (yy < yhi) && (c >= clo) && (c < chi);
    xmask = (xmask & ~(1u << j)) | ((ok ? 1u : 0u) << j);
    if (!WABL(1)) xr[j] = gload8<T>(xsrc + xbase + (ok ? x_eoff[j] : 0));
    else xr[j] = zero8<T>();
  };
  auto load_d = [&](int j) {
    const int cx = d_rel[j] & 1023, row = (d_rel[j] >> 10) & 1023, k = (d_rel[j] >> 20) & 1023;
    const bool ok = (d_rel[j] >> 30) && (k < klim) && (row < rhi) && (cx < xhi);
    dmask = (dmask & ~(1u << j)) | ((ok ? 1u : 0u) << j);
    if (!WABL(1)) dr[j] = gload8<T>(dyp + dbase + (ok ? d_eoff[j] : 0));
    else dr[j] = zero8<T>();
  };
  auto store_x = [&](int j, T* ldsX) {
    Raw8<T> v = xr[j];
    if (aff) {
      float sc[8], sh[8];
      {
        const float4* sp4 = reinterpret_cast<const float4*>(ldsS + gx * 16);
        const float4 s0 = sp4[0], s1 = sp4[1], h0 = sp4[2], h1 = sp4[3];
        sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
        sh[0] = h0.x; sh[1] = h0.y; sh[2] = h0.z; sh[3] = h0.w; sh[4] = h1.x; sh[5] = h1.y; sh[6] = h1.z; sh[7] = h1.w;
      }
      if constexpr (std::is_same<T, bf16>::value) {
        v = affine8_lim(v, sc, sh, relu_lim);
      } else {
        float* f = reinterpret_cast<float*>(&v.q[0]);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e] * sc[e] + sh[e], relu_lo);
      }
    }
    v = select8<T>((xmask >> j) & 1u, v);
    if (WABL(8)) { keep8<T>(v); return; }
    if (tid + j * NTHREADS < x_items) lstore8<T>(ldsX + x_dst0 + j * XSTEP, v);
  };
  auto store_d = [&](int j, T* ldsD) {
    const Raw8<T> v = select8<T>((dmask >> j) & 1u, dr[j]);
    if (WABL(8)) { keep8<T>(v); return; }
    if (tid + j * NTHREADS < d_items) lstore8<T>(ldsD + d_dst0 + j * DSTEP, v);
  };

  // ---- the tiles of this workgroup: sp, sp + nsplit, ...
  const int ntl = sp < a.total_ptiles ? (a.total_ptiles - 1 - sp) / a.nsplit + 1 : 0;
  __syncthreads();                                   // the scale / shift table is read by the first stores below
  if (ntl > 0) {
    tile_origin(sp);
#pragma unroll
    for (int j = 0; j < XI; ++j) load_x(j);
#pragma unroll
    for (int j = 0; j < DI; ++j) load_d(j);
#pragma unroll
    for (int j = 0; j < XI; ++j) store_x(j, lds0);
#pragma unroll
    for (int j = 0; j < DI; ++j) store_d(j, lds0 + x_elems);
    tile_origin(sp + (ntl > 1 ? 1 : 0) * a.nsplit);
#pragma unroll
    for (int j = 0; j < XI; ++j) load_x(j);
#pragma unroll
    for (int j = 0; j < DI; ++j) load_d(j);
  }
  __syncthreads();
  constexpr int KSW = (BMPIX / 16) / NKS;              // k-steps of 16 pixels per wave and tile
  constexpr int SLOTS = KSW * NACC;                    // MFMAs per wave and tile = places for a staging unit
  for (int i = 0; i < ntl; ++i) {
    T* ldsX = lds0 + (i & 1) * stage_elems;
    T* ldsD = ldsX + x_elems;
    T* othX = lds0 + ((i + 1) & 1) * stage_elems;
    T* othD = othX + x_elems;
    const bool do_store = i + 1 < ntl;
    // the registers hold tile i+1; its items go to the other stage one by one and are re-issued for tile i+2 (the last two
    // iterations re-load the last tile instead of branching around vector-memory instructions)
    tile_origin(sp + (i + 2 < ntl ? i + 2 : ntl - 1) * a.nsplit);
    auto side = [&](int slot) {
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        if ((u * SLOTS) / NU != slot) continue;
        if (u < XI) { if (do_store) store_x(u, othX); load_x(u); }
        else { if (do_store) store_d(u - XI, othD); load_d(u - XI); }
      }
    };
#pragma unroll
    for (int kk = 0; kk < KSW; ++kk) {
      const int ks = wks + kk * NKS;
      if constexpr (NW > 1) {
        // one X fragment, NW dY fragments (the mirror image of the tap loop below, where dY is shared and X moves)
        const int gi = lane >> 4, i16 = lane & 15;
        const int chb = 16 * (gi & 1) + 4 * (i16 & 3);
        const int qa = ks * 16 + 8 * (gi >> 1) + (i16 >> 2);
        const int qb = qa + 4;
        const bf16x4 alo = tr_read(ldsX + tab[qa] + wci * 32 + chb);
        const bf16x4 ahi = tr_read(ldsX + tab[qb] + wci * 32 + chb);
        const bf16x8 afr = __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7);
        const bf16* da = ldsD + qa * DP + wco * NW * 32 + chb;
        const bf16* db = ldsD + qb * DP + wco * NW * 32 + chb;
        bf16x4 blo[2], bhi[2];
        blo[0] = tr_read(da);
        bhi[0] = tr_read(db);
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          if (w + 1 < NW) {
            blo[(w + 1) & 1] = tr_read(da + (w + 1) * 32);
            bhi[(w + 1) & 1] = tr_read(db + (w + 1) * 32);
          }
          side(kk * NW + w);
          __builtin_amdgcn_sched_barrier(0);
          const bf16x8 bfr = __builtin_shufflevector(blo[w & 1], bhi[w & 1], 0, 1, 2, 3, 4, 5, 6, 7);
          if (!WABL(2)) acc[w] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr, acc[w], 0, 0, 0);
          else acc[w][0] += (float)afr[0] + (float)bfr[0];
        }
      } else if constexpr (std::is_same<T, bf16>::value) {
        const int gi = lane >> 4, i16 = lane & 15;
        const int chb = 16 * (gi & 1) + 4 * (i16 & 3);
        const int qa = ks * 16 + 8 * (gi >> 1) + (i16 >> 2);
        const int qb = qa + 4;
        bf16x4 blo = tr_read(ldsD + qa * DP + wco * 32 + chb);
        bf16x4 bhi = tr_read(ldsD + qb * DP + wco * 32 + chb);
        bf16x8 bfr = __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7);
        const int xoa = tab[qa] + wci * 32 + chb;
        const int xob = tab[qb] + wci * 32 + chb;
        bf16x4 alo[2], ahi[2];
        alo[0] = tr_read(ldsX + xoa);
        ahi[0] = tr_read(ldsX + xob);
#pragma unroll
        for (int tap = 0; tap < NTAPS; ++tap) {
          if (tap + 1 < NTAPS) {
            const int ky = (tap + 1) / 3, kx = (tap + 1) % 3;
            const int toff = (ky * (TW + 2) + kx) * XP;      // dilation 1 (the only plan of this kernel): cl = TW + 2, an instruction immediate
            alo[(tap + 1) & 1] = tr_read(ldsX + xoa + toff);
            ahi[(tap + 1) & 1] = tr_read(ldsX + xob + toff);
          }
          side(kk * NTAPS + tap);
          __builtin_amdgcn_sched_barrier(0);
          bf16x8 afr = __builtin_shufflevector(alo[tap & 1], ahi[tap & 1], 0, 1, 2, 3, 4, 5, 6, 7);
          if (!WABL(2)) acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr, acc[tap], 0, 0, 0);
          else acc[tap][0] += (float)afr[0] + (float)bfr[0];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int q = ks * 16 + 2 * j + hh;
          const float b = ldsD[q * DP + wco * 32 + r];
          const int xo = tab[q] + wci * 32 + r;
#pragma unroll
          for (int tap = 0; tap < NTAPS; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            const int toff = (NTAPS == 1) ? 0 : ((ky * a.dil) * a.cl + kx * a.dil) * XP;
            if (j == 0) side(kk * NTAPS + tap);
            acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(ldsX[xo + toff], b, acc[tap], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
  }
  // ---- combine the k-slice waves of a tile through LDS (fixed order), one tap at a time
  if constexpr (NKS > 1) {
    float* red = reinterpret_cast<float*>(smem_raw);          // staging buffers are dead now
    const int tile_id = wci + NCI * wco;
#pragma unroll
    for (int tap = 0; tap < NACC; ++tap) {
      __syncthreads();
      if (wks > 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[((tile_id * (NKS - 1) + (wks - 1)) * 16 + i) * 64 + lane] = acc[tap][i];
      }
      __syncthreads();
      if (wks == 0) {
#pragma unroll 1
        for (int k = 0; k < NKS - 1; ++k)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[tap][i] += red[((tile_id * (NKS - 1) + k) * 16 + i) * 64 + lane];
      }
    }
  }
  if (wks == 0) {
#pragma unroll
    for (int t = 0; t < NACC; ++t) {
      const int tap = NW > 1 ? 0 : t, w = NW > 1 ? t : 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ci = ci0 + wci * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
        const int co = co0 + (wco * NW + w) * 32 + r;
        a.ws[((size_t)(sp * NTAPS + tap) * a.kpad + ci) * a.npad + co] = acc[t][i];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Round 4: the deep 3x3 weight gradient on a (64 ci) x (128 co) x (9 taps) block per workgroup, dY by LDS-DMA.
//
// The (32 ci) x (128 co) block of wgrad_db_kernel staged 160 channel-rows per pixel for 32 x 128 x 9 products (4.3 bytes per kFLOP through
// the CU's vector-memory path, dY re-read by Cin / 32 blocks: 1.8-2.0x the algorithmic HBM traffic); its 64-ci form spilled because one
// more staged item per thread did not fit beside 144 accumulator registers.  Here dY -- 2/3 of the staged bytes, and the operand that
// needs no transform -- never touches a VGPR: every wave moves four 1-KB pieces per pixel tile with global_load_lds_dwordx4 straight into
// the other stage (no load registers, no ds_write, no per-item bookkeeping), which pays for the second ci tile:
//   * block 64 x 128 x 9: 8 waves = 2 (ci) x 4 (co) tiles of 32 x 32, every wave all 9 taps (144 accumulator registers) and ALL 8 k-steps of
//     a 128-pixel tile (no k-split, no LDS reduction at the end); 2.6 bytes per kFLOP staged, dY re-read by Cin / 64 blocks;
//   * dY image: 128 pixel rows of 256 B, UNPADDED (an LDS-DMA piece is 1 KB of contiguous LDS = 4 rows); the four 64-byte segments of a
//     row are XOR-swizzled by (row & 3) -- on the SOURCE address of the DMA and on the transposing reads -- so that the 4 rows x 64 B a
//     32-lane half reads fall on 64 distinct banks (the padded rows of the register-staged image did this with 64 B of padding);
//   * X (halo tile, BatchNorm + ReLU of the producing layer in registers) as before: register-staged, one item between two MFMAs;
//   * per tile: X items in the first third of the 72 MFMA slots, the four DMA pieces in the second third, `s_waitcnt vmcnt(0)` + ONE
//     barrier at the end (everything issued in an iteration has ~half an iteration to land; hipcc sees no DMA, so its own counted
//     waits for the X registers stay exact).
// Requires whole tiles (the DMA has no bounds handling), Cin % 64 == 0, Cout % 128 == 0, dilation 1, bf16; everything else stays on
// wgrad_db_kernel.
// Round 6, M16 = true (SATCV_WGRAD_M16=1): the same block on v_mfma_f32_16x16x32_bf16 -- K = 32 pixels per instruction, a 32 x 32 (ci, co) tile as 2 x 2
// blocks of 16 x 16 (the same 16 accumulator registers per tap).  A lane group g of 16 lanes holds the k indices 8 g ... 8 g + 7 of a fragment; they are fed
// the pixels 4 g ... 4 g + 3 and 16 + 4 g ... of the 32-pixel step (the same for X and dY: any k order is a valid contraction), so that the 32 lanes of a
// transposing read touch 8 CONSECUTIVE pixel rows x 32 bytes: conflict-free with X rows of 160 bytes (XP = 80: start banks 40 r mod 64) and with the dY rows'
// 32-byte granules XOR-ed by (row & 7) (the 32x32x16 form reads 4 rows x 64 bytes: 192-byte X rows, 64-byte segments XOR-ed by row & 3).
template <int TW, bool M16 = false>
__global__ __launch_bounds__(512, 1) void wgrad_dma_kernel(const WgradArgs a) {
  using T = bf16;
  constexpr int NCI = 2, NCO = 4, NTAPS = 9, NTHREADS = 512, BMPIX = 128, TH = BMPIX / TW;
  constexpr int CI_T = 32 * NCI, CO_T = 32 * NCO, XP = CI_T + (M16 ? 16 : 32), GX = CI_T / 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int x_elems = a.rl * a.cl * XP;
  const int stage_elems = x_elems + BMPIX * CO_T;
  T* lds0 = reinterpret_cast<T*>(smem_raw);
  int* tab = reinterpret_cast<int*>(lds0 + 2 * stage_elems);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wci = wave % NCI, wco = wave / NCI;
  const int r = lane & 31, hh = lane >> 5;
  int blk, sp;
  {
    const int nblk = a.n_ci_blk * a.n_co_blk, id = blockIdx.x;
    if ((a.nsplit & 7) == 0) { const int xcd = id & 7, j = id >> 3; blk = j % nblk; sp = (j / nblk) * 8 + xcd; }
    else { blk = id % nblk; sp = id / nblk; }
  }
  const int ci_blk = blk % a.n_ci_blk, co_blk = blk / a.n_ci_blk;
  const int ci0 = ci_blk * CI_T, co0 = co_blk * CO_T;

  f32x16 acc[NTAPS];                 // 32x32x16 form: one tile per tap
  f32x4 acc4[NTAPS][4];              // 16x16x32 form: 2 x 2 blocks per tap (block 2 c + b) -- only one of the two sets is live
#pragma unroll
  for (int t = 0; t < NTAPS; ++t) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) acc4[t][q] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  for (int q = tid; q < BMPIX; q += NTHREADS) {
    const int t = q / TW, cx = q % TW;
    const int k = t / a.rpi;
    const int l0 = (k < a.imgs) ? k * a.seg + (t - k * a.rpi) : 0;
    tab[q] = (l0 * a.cl + cx) * XP;
  }

  // ---- X: per-thread, tile-invariant description of the staged items (as in wgrad_db_kernel)
  const int x_items = a.rl * a.cl * GX;
  constexpr int XMAXPIX = (TH + 2 * (TH / 4 > 1 ? TH / 4 : 1)) * (TW + 2);
  constexpr int XI = (XMAXPIX * GX + NTHREADS - 1) / NTHREADS;
  constexpr int NPIECE = 4;                              // 1-KB dY pieces per wave and tile: 8 waves x 4 x 4 pixel rows = 128 pixels
  constexpr int NU = XI + NPIECE;
  const int gx = tid % GX, cgx = ci0 + gx * 8;
  const bool xsecond = a.x1 != nullptr && cgx >= a.c0;
  const T* xsrc = xsecond ? reinterpret_cast<const T*>(a.x1) + (cgx - a.c0) : reinterpret_cast<const T*>(a.x0) + cgx;
  const int xcs = xsecond ? a.c1 : a.c0;
  float* ldsS = reinterpret_cast<float*>(tab + BMPIX);                     // [GX][16]: 8 scale + 8 shift per channel group
  const bool aff = a.in_scale != nullptr;
  if (aff && tid < GX * 16) {
    const int g = tid / 16, e = tid % 16, ch = ci0 + g * 8 + (e & 7);
    ldsS[tid] = e < 8 ? a.in_scale[ch] : a.in_shift[ch];
  }
  const unsigned relu_lim = a.in_relu != 0 ? 0u : 0x80008000u;
  constexpr int XSTEP = (NTHREADS / GX) * XP;
  const int x_dst0 = (tid / GX) * XP + gx * 8;
  int x_rel[XI], x_eoff[XI];
#pragma unroll
  for (int j = 0; j < XI; ++j) {
    const int it = tid + j * NTHREADS;
    const int pix = it / GX;
    const int c = pix % a.cl, L = pix / a.cl;
    const int k = L / a.seg;
    const int yy = L - k * a.seg - a.halh;
    x_rel[j] = ((it < x_items) ? (1 << 30) : 0) | (k << 20) | ((yy + 64) << 10) | c;
    x_eoff[j] = ((k * a.h + yy) * a.w_ + (c - a.halw)) * xcs;
  }
  // ---- dY: byte offset of this lane's 16 bytes of piece j from the tile's first pixel.  Piece p = 4 wave + j holds the pixel rows
  // 4 p .. 4 p + 3; lane l writes chunk (l & 15) of row 4 p + (l >> 4) and fetches the chunk whose 64-byte segment is XOR-ed with that
  // row's low two bits (= l >> 4)
  unsigned d_voff[NPIECE];
#pragma unroll
  for (int j = 0; j < NPIECE; ++j) {
    const int q = (wave * NPIECE + j) * 4 + (lane >> 4);
    const int t = q / TW, cx = q % TW;
    const int k = (a.imgs == 1) ? 0 : t / a.rpi;
    const int row = t - k * a.rpi;
    // (M16: the 32-byte granule of the chunk XOR-ed with the row's low three bits -- row = 4 p + (lane >> 4), p = 4 wave + j)
    const int chunk = M16 ? (((((lane & 15) >> 1) ^ (4 * (j & 1) + (lane >> 4))) << 1) | (lane & 1)) : ((lane & 15) ^ ((lane >> 4) << 2));
    d_voff[j] = (unsigned)((((k * a.h + row) * a.w_ + cx) * a.lddy + chunk * 8) * (int)sizeof(T));
  }
  const T* dyp = reinterpret_cast<const T*>(a.dy) + co0;
  const unsigned lds_base = lds_addr_of(lds0);

  Raw8<T> xr[XI];
  unsigned xmask = 0;
  int klim = 0, ylo = 0, yhi = 0, clo = 0, chi = 0;
  size_t xbase = 0;
  auto tile_origin = [&](int pt, size_t& dbase) {
    int m = pt;
    const int tx = m % a.tiles_x; m /= a.tiles_x;
    const int ty = m % a.tiles_y;
    const int n0 = (m / a.tiles_y) * a.imgs, y0 = ty * TH, x0 = tx * TW;
    klim = a.n - n0;
    ylo = 64 - y0; yhi = 64 + a.h - y0;
    clo = a.halw - x0; chi = a.w_ + a.halw - x0;
    const size_t bp = ((size_t)n0 * a.h + y0) * a.w_ + x0;
    xbase = bp * xcs;
    dbase = bp * a.lddy;
  };
  auto load_x = [&](int j) {
    const int c = x_rel[j] & 1023, yy = (x_rel[j] >> 10) & 1023, k = (x_rel[j] >> 20) & 1023;
    const bool ok = (x_rel[j] >> 30) && (k < klim) && (yy >= ylo) && (yy < yhi) && (c >= clo) && (c < chi);
    xmask = (xmask & ~(1u << j)) | ((ok ? 1u : 0u) << j);
    xr[j] = gload8<T>(xsrc + xbase + (ok ? x_eoff[j] : 0));
  };
  auto store_x = [&](int j, T* ldsX) {
    Raw8<T> v = xr[j];
    if (aff) {
      float sc[8], sh[8];
      const float4* sp4 = reinterpret_cast<const float4*>(ldsS + gx * 16);
      const float4 s0 = sp4[0], s1 = sp4[1], h0 = sp4[2], h1 = sp4[3];
      sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
      sh[0] = h0.x; sh[1] = h0.y; sh[2] = h0.z; sh[3] = h0.w; sh[4] = h1.x; sh[5] = h1.y; sh[6] = h1.z; sh[7] = h1.w;
      v = affine8_lim(v, sc, sh, relu_lim);
    }
    v = select8<T>((xmask >> j) & 1u, v);
    if (tid + j * NTHREADS < x_items) lstore8<T>(ldsX + x_dst0 + j * XSTEP, v);
  };
  // piece j of the tile whose first pixel is at element offset dbase -> stage st
  auto dma_piece = [&](int j, size_t dbase, int st) {
    const unsigned dst = lds_base + (unsigned)((st * stage_elems + x_elems) * (int)sizeof(T)) + (unsigned)((wave * NPIECE + j) * 1024);
    lds_dma16(dyp + dbase, d_voff[j], __builtin_amdgcn_readfirstlane(dst));
  };

  const int ntl = sp < a.total_ptiles ? (a.total_ptiles - 1 - sp) / a.nsplit + 1 : 0;
  __syncthreads();                                   // the scale / shift table is read by the first stores below
  size_t dnext = 0, dtmp = 0;
  if (ntl > 0) {
    tile_origin(sp, dnext);
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) dma_piece(j, dnext, 0);
#pragma unroll
    for (int j = 0; j < XI; ++j) load_x(j);
#pragma unroll
    for (int j = 0; j < XI; ++j) store_x(j, lds0);
    tile_origin(sp + (ntl > 1 ? 1 : 0) * a.nsplit, dnext);      // registers: X of tile 1; dnext: dY of tile 1 (moved in iteration 0)
#pragma unroll
    for (int j = 0; j < XI; ++j) load_x(j);
  }
  dma_wait_all();
  __syncthreads();
  // fragment addresses (elements): dY rows 128 wide, segment swizzle = (pixel row & 3) = (i16 >> 2), a lane constant
  const int gi = lane >> 4, i16 = lane & 15;
  const int chb = 16 * (gi & 1) + 4 * (i16 & 3);
  const int d_lane = (8 * (gi >> 1) + (i16 >> 2)) * CO_T + ((wco ^ (i16 >> 2)) * 32) + chb;
  const int q_lane = 8 * (gi >> 1) + (i16 >> 2);
  // M16: this lane's pixel inside a 32-pixel k-step (low half; the high half is + 16) and its 4-channel piece of a 16-channel block
  const int p16 = 4 * gi + (i16 >> 2), c4 = 4 * (i16 & 3);
  int d16[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) d16[b] = p16 * CO_T + (((wco * 2 + b) ^ (p16 & 7)) * 16) + c4;
  constexpr int SLOTS = (BMPIX / 16) * NTAPS;          // 72 MFMAs per wave and tile
  for (int i = 0; i < ntl; ++i) {
    T* ldsX = lds0 + (i & 1) * stage_elems;
    T* ldsD = ldsX + x_elems;
    T* othX = lds0 + ((i + 1) & 1) * stage_elems;
    const bool do_store = i + 1 < ntl;
    const size_t dcur = dnext;                         // dY of tile i + 1 (this iteration's DMA)
    tile_origin(sp + (i + 2 < ntl ? i + 2 : ntl - 1) * a.nsplit, dtmp);      // X loads of tile i + 2
    // (two compile-time halves of four k-steps: hipcc does not fully unroll 72 MFMAs, and with a run-time slot number every staging unit
    //  would be compiled into every slot behind a branch)  half 0: the X items, half 1: the DMA pieces
    auto half_tile = [&](auto HALF) {
      constexpr int H0 = decltype(HALF)::value * 4;
      // the dY fragment and the first X fragment of k-step kq + 1 are requested during the last tap of k-step kq: a k-step otherwise
      // opened with six reads and a wait in front of its first MFMA (8 exposed LDS latencies per tile instead of 2)
      bf16x4 nb_lo[2], nb_hi[2], na_lo[2], na_hi[2];
      int xoa_n[2], xob_n[2];
      auto first_reads = [&](int ks, int slot_) {
        nb_lo[slot_] = tr_read(ldsD + d_lane + ks * 16 * CO_T);
        nb_hi[slot_] = tr_read(ldsD + d_lane + (ks * 16 + 4) * CO_T);
        xoa_n[slot_] = tab[ks * 16 + q_lane] + wci * 32 + chb;
        xob_n[slot_] = tab[ks * 16 + q_lane + 4] + wci * 32 + chb;
        na_lo[slot_] = tr_read(ldsX + xoa_n[slot_]);
        na_hi[slot_] = tr_read(ldsX + xob_n[slot_]);
      };
      first_reads(H0, 0);
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) {
        const int cur_ = kq & 1;
        const bf16x8 bfr = __builtin_shufflevector(nb_lo[cur_], nb_hi[cur_], 0, 1, 2, 3, 4, 5, 6, 7);
        const int xoa = xoa_n[cur_], xob = xob_n[cur_];
        bf16x4 alo[2], ahi[2];
        alo[0] = na_lo[cur_];
        ahi[0] = na_hi[cur_];
#pragma unroll
        for (int tap = 0; tap < NTAPS; ++tap) {
          if (tap + 1 < NTAPS) {
            const int ky = (tap + 1) / 3, kx = (tap + 1) % 3;
            const int toff = (ky * (TW + 2) + kx) * XP;      // dilation 1: cl = TW + 2, an instruction immediate
            alo[(tap + 1) & 1] = tr_read(ldsX + xoa + toff);
            ahi[(tap + 1) & 1] = tr_read(ldsX + xob + toff);
          } else if (kq + 1 < 4) {
            first_reads(H0 + kq + 1, cur_ ^ 1);
          }
          const int slot = kq * NTAPS + tap;                 // 0 .. 35 inside the half
          if constexpr (decltype(HALF)::value == 0) {
#pragma unroll
            for (int u = 0; u < XI; ++u)
              if (slot == 2 + 8 * u) { if (do_store) store_x(u, othX); load_x(u); }
          } else {
#pragma unroll
            for (int u = 0; u < NPIECE; ++u)
              if (slot == 1 + 6 * u) { if (do_store) dma_piece(u, dcur, (i + 1) & 1); }
          }
          __builtin_amdgcn_sched_barrier(0);
          const bf16x8 afr = __builtin_shufflevector(alo[tap & 1], ahi[tap & 1], 0, 1, 2, 3, 4, 5, 6, 7);
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr, acc[tap], 0, 0, 0);
        }
      }
    };
    // ---- the 16x16x32 form of a half tile: two k-steps of 32 pixels, per step the two dY fragments once and per tap the two X fragments
    // (4 transposing reads) + 4 MFMAs; the same one-tap-ahead / one-step-ahead reads and the same placement of the staging units
    auto half_tile16 = [&](auto HALF) {
      constexpr int H0 = decltype(HALF)::value * 2;
      // (one pair of X fragment buffers for the whole half: a k-step has 9 taps, so the buffer a step starts on alternates -- index (tap + kq) & 1,
      //  all compile-time; the first X fragments of the next step are read into the buffer its tap 0 will use, during this step's last tap)
      // whole 8-element fragments (the two transposing reads of one land in adjacent registers).  dY: ONE buffer, read at the top of its k-step -- a second one
      // (+8 registers) put the kernel over 256 and its scratch reloads queue behind the prefetched tile on the in-order vmcnt counter
      bf16x8 nb[1][2], af[2][2];
      int xl, xh;
      auto b_reads = [&](int ks, int s_) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
          nb[s_][b] = __builtin_shufflevector(tr_read(ldsD + d16[b] + ks * 32 * CO_T), tr_read(ldsD + d16[b] + (ks * 32 + 16) * CO_T), 0, 1, 2, 3, 4, 5, 6, 7);
      };
      auto a_reads = [&](int buf, int toff) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
          af[buf][c] = __builtin_shufflevector(tr_read(ldsX + xl + 16 * c + toff), tr_read(ldsX + xh + 16 * c + toff), 0, 1, 2, 3, 4, 5, 6, 7);
      };
      auto x_offsets = [&](int ks) {
        xl = tab[ks * 32 + p16] + wci * 32 + c4;
        xh = tab[ks * 32 + 16 + p16] + wci * 32 + c4;
      };
      x_offsets(H0);
      a_reads(0, 0);
#pragma unroll
      for (int kq = 0; kq < 2; ++kq) {
        const int cur_ = 0;
        b_reads(H0 + kq, 0);
#pragma unroll
        for (int tap = 0; tap < NTAPS; ++tap) {
          const int buf = (tap + kq) & 1;
          if (tap + 1 < NTAPS) {
            const int ky = (tap + 1) / 3, kx = (tap + 1) % 3;
            a_reads(buf ^ 1, (ky * (TW + 2) + kx) * XP);
          } else if (kq + 1 < 2) {
            x_offsets(H0 + kq + 1);
            a_reads(buf ^ 1, 0);
          }
          const int slot = kq * NTAPS + tap;                 // 0 .. 17 inside the half
          if constexpr (decltype(HALF)::value == 0) {
#pragma unroll
            for (int u = 0; u < XI; ++u)
              if (slot == 1 + 4 * u) { if (do_store) store_x(u, othX); load_x(u); }
          } else {
#pragma unroll
            for (int u = 0; u < NPIECE; ++u)
              if (slot == 4 * u) { if (do_store) dma_piece(u, dcur, (i + 1) & 1); }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              acc4[tap][c * 2 + b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[buf][c], nb[cur_][b], acc4[tap][c * 2 + b], 0, 0, 0);
            }
          }
        }
      }
    };
    static_assert(XI <= 4, "staging units of a half tile: slots 1, 5, 9, 13 of 18");
    if constexpr (M16) {
      half_tile16(std::integral_constant<int, 0>{});
      half_tile16(std::integral_constant<int, 1>{});
    } else {
      half_tile(std::integral_constant<int, 0>{});
      half_tile(std::integral_constant<int, 1>{});
    }
    dnext = dtmp;
    dma_wait_all();
    __syncthreads();
  }
#pragma unroll
  for (int tap = 0; tap < NTAPS; ++tap) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      // M16: register 4 (2 c + b) + rr of lane (gi, i16) is row 16 c + 4 gi + rr, column 16 b + i16 of the (ci, co) tile
      const int ci = M16 ? ci0 + wci * 32 + 16 * (i >> 3) + 4 * gi + (i & 3) : ci0 + wci * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
      const int co = M16 ? co0 + wco * 32 + 16 * ((i >> 2) & 1) + i16 : co0 + wco * 32 + r;
      a.ws[((size_t)(sp * NTAPS + tap) * a.kpad + ci) * a.npad + co] = M16 ? acc4[tap][i >> 2][i & 3] : acc[tap][i];
    }
  }
}

// dw (Keras layout) = sum over slabs in a fixed order (deterministic): 64 outputs x 4 split lanes
// per block, each lane sums every 4th slab, LDS combines the 4 partial sums.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nslab, int taps, int kpad,
                                                           int npad, int cin, int nvalid, int transposed, int accumulate) {
  __shared__ float part[4][64];
  const long long total = (long long)taps * cin * nvalid;
  const size_t slab = (size_t)taps * kpad * npad;
  const int ol = threadIdx.x & 63, sl = threadIdx.x >> 6;
  for (long long base = (long long)blockIdx.x * 64; base < total; base += (long long)gridDim.x * 64) {
    const long long it = base + ol;
    float s = 0.f;
    int co = 0, ci = 0, tap = 0;
    if (it < total) {
      co = (int)(it % nvalid);
      ci = (int)((it / nvalid) % cin);
      tap = (int)(it / ((long long)nvalid * cin));
      const float* p = ws + ((size_t)tap * kpad + ci) * npad + co;
      float s0 = 0.f, s1 = 0.f;
      int sp = sl;
      for (; sp + 4 < nslab; sp += 8) { s0 += p[(size_t)sp * slab]; s1 += p[(size_t)(sp + 4) * slab]; }
      if (sp < nslab) s0 += p[(size_t)sp * slab];
      s = s0 + s1;
    }
    part[sl][ol] = s;
    __syncthreads();
    if (sl == 0 && it < total) {
      const float r = (part[0][ol] + part[1][ol]) + (part[2][ol] + part[3][ol]);
      float* dst = transposed ? dw + (size_t)co * cin + ci : dw + ((size_t)tap * cin + ci) * nvalid + co;
      *dst = accumulate ? *dst + r : r;
    }
    __syncthreads();
  }
}

// float4 form for the plain (HWIO) layout: 4 consecutive output channels per lane, the same slab order and the same association of
// the partial sums as above (bit-identical results), a quarter of the load instructions and four times the bytes in flight
__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nslab, int taps, int kpad,
                                                            int npad, int cin, int nvalid, int accumulate) {
  __shared__ float4 part[4][64];
  const int nv4 = nvalid / 4;
  const long long total = (long long)taps * cin * nv4;
  const size_t slab = (size_t)taps * kpad * npad;
  const int ol = threadIdx.x & 63, sl = threadIdx.x >> 6;
  for (long long base = (long long)blockIdx.x * 64; base < total; base += (long long)gridDim.x * 64) {
    const long long it = base + ol;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int co = 0, ci = 0, tap = 0;
    if (it < total) {
      co = (int)(it % nv4) * 4;
      ci = (int)((it / nv4) % cin);
      tap = (int)(it / ((long long)nv4 * cin));
      const float* p = ws + ((size_t)tap * kpad + ci) * npad + co;
      float4 s0 = s, s1 = s;
      int sp = sl;
      for (; sp + 4 < nslab; sp += 8) {
        const float4 a = *reinterpret_cast<const float4*>(p + (size_t)sp * slab), b = *reinterpret_cast<const float4*>(p + (size_t)(sp + 4) * slab);
        s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
      }
      if (sp < nslab) { const float4 a = *reinterpret_cast<const float4*>(p + (size_t)sp * slab); s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w; }
      s = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
    }
    part[sl][ol] = s;
    __syncthreads();
    if (sl == 0 && it < total) {
      const float4 a = part[0][ol], b = part[1][ol], c = part[2][ol], e = part[3][ol];
      float4 r = make_float4((a.x + b.x) + (c.x + e.x), (a.y + b.y) + (c.y + e.y), (a.z + b.z) + (c.z + e.z), (a.w + b.w) + (c.w + e.w));
      float4* dst = reinterpret_cast<float4*>(dw + ((size_t)tap * cin + ci) * nvalid + co);
      if (accumulate) { const float4 o = *dst; r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w; }
      *dst = r;
    }
    __syncthreads();
  }
}

// many slabs, few outputs (thin layers: 128-256 slabs of a 9 x 32 x 32 kernel): 16 lanes per float4 output instead of 4, four loads in
// flight per lane -- the 4-lane form walked 64 slabs per lane with two loads in flight and took ~40 us for 9 MB.  Fixed summation
// order (lane l sums slabs l, l + 16, ... in four interleaved accumulators; the 16 lane sums are added in increasing lane order).
__global__ __launch_bounds__(256) void wgrad_reduce16_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nslab, int taps, int kpad,
                                                             int npad, int cin, int nvalid, int accumulate) {
  __shared__ float4 part[16][16];
  const int nv4 = nvalid / 4;
  const long long total = (long long)taps * cin * nv4;
  const size_t slab = (size_t)taps * kpad * npad;
  const int ol = threadIdx.x & 15, sl = threadIdx.x >> 4;
  for (long long base = (long long)blockIdx.x * 16; base < total; base += (long long)gridDim.x * 16) {
    const long long it = base + ol;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int co = 0, ci = 0, tap = 0;
    if (it < total) {
      co = (int)(it % nv4) * 4;
      ci = (int)((it / nv4) % cin);
      tap = (int)(it / ((long long)nv4 * cin));
      const float* p = ws + ((size_t)tap * kpad + ci) * npad + co;
      float4 a0 = s, a1 = s, a2 = s, a3 = s;
      int sp = sl;
      for (; sp + 48 < nslab; sp += 64) {
        const float4 v0 = *reinterpret_cast<const float4*>(p + (size_t)sp * slab), v1 = *reinterpret_cast<const float4*>(p + (size_t)(sp + 16) * slab);
        const float4 v2 = *reinterpret_cast<const float4*>(p + (size_t)(sp + 32) * slab), v3 = *reinterpret_cast<const float4*>(p + (size_t)(sp + 48) * slab);
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
        a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
        a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
      }
      for (; sp < nslab; sp += 16) { const float4 v = *reinterpret_cast<const float4*>(p + (size_t)sp * slab); a0.x += v.x; a0.y += v.y; a0.z += v.z; a0.w += v.w; }
      s = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w));
    }
    part[sl][ol] = s;
    __syncthreads();
    if (sl == 0 && it < total) {
      float4 r = part[0][ol];
#pragma unroll
      for (int k = 1; k < 16; ++k) { const float4 v = part[k][ol]; r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
      float4* dst = reinterpret_cast<float4*>(dw + ((size_t)tap * cin + ci) * nvalid + co);
      if (accumulate) { const float4 o = *dst; r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w; }
      *dst = r;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ host side
struct WgradPlan { int tw, nci, nco, nw, nks, ntaps, nsplit, kpad, npad, n_ci_blk, n_co_blk, pix, db, dma; size_t ws_bytes; };
extern int g_opt_wgrad_db;        // api.hip: satcv_set_option("wgrad_db", ...)
extern int g_opt_wgrad_m16;       // api.hip: satcv_set_option("wgrad_m16", ...)

static bool wgrad_pix256() {
  static const bool on = [] { const char* e = getenv("SATCV_WGRAD_PIX256"); return !e || atoi(e) != 0; }();
  return on;
}

static int pick_tw_w(int w) {
  int best = 8, bestpad = cdiv(w, 8) * 8;
  const int cands[2] = {16, 32};
  for (int i = 0; i < 2; ++i) { int p = cdiv(w, cands[i]) * cands[i]; if (p <= bestpad) { best = cands[i]; bestpad = p; } }
  return best;
}

static int wgrad_plan(const satcv_wgrad_desc* d, WgradPlan& p) {
  if (!(d->n > 0 && d->h > 0 && d->w_ > 0 && d->c0 > 0 && d->c1 >= 0 && d->cout > 0 && d->dil >= 1 && d->f >= 0 && d->f <= 16 && d->dil <= (1 << 15) &&
        satcv_pixels_ok(d->n, d->h, d->w_, d->f) && d->c0 <= (1 << 20) && d->c1 <= (1 << 20) && d->cout <= (1 << 20))) {
    satcv_set_error("wgrad: extents out of range (n*h*w must stay below 2^31 pixels, channels below 2^20)");
    return SATCV_ERR_INVALID;
  }
  const int cinx = d->c0 + d->c1;
  const int nspace = d->mode_dy ? d->f * d->f * d->cout : d->cout;
  p.ntaps = d->kh * d->kw;
  if (!(p.ntaps == 1 || (d->kh == 3 && d->kw == 3))) { satcv_set_error("wgrad: only 1x1 and 3x3 taps"); return SATCV_ERR_UNSUPPORTED; }
  p.tw = pick_tw_w(d->w_);
  const bool db_ok = g_opt_wgrad_db != 0 && d->dil == 1 && d->dtype == SATCV_BF16;
  // 1x1 / transposed convolutions: 16 accumulator registers per (ci, co) tile, so the double-buffered kernel's 8 waves cover a
  // 64 x 128 block (each dY tile is re-read by half as many ci blocks: these launches are staging-bound, ~110 us whatever their size)
  // ... and, where the layer has them, a 128 x 256 block with four co tiles per wave (see wgrad_db_kernel)
  p.nw = 1;
  p.dma = 0;
  {
    // deep 3x3 layers: the (64 ci) x (128 co) block with dY by LDS-DMA (wgrad_dma_kernel; SATCV_WGRAD_DMA=0 keeps the (32 x 128) block)
    static const int dma_on = [] { const char* e = getenv("SATCV_WGRAD_DMA"); return e ? atoi(e) : 1; }();
    const int th_ = 128 / p.tw;
    const bool whole = d->w_ % p.tw == 0 && (d->h >= th_ ? d->h % th_ == 0 : (th_ % d->h == 0 && d->n % (th_ / d->h) == 0));
    if (dma_on && db_ok && p.ntaps == 9 && !d->mode_dy && cinx % 64 == 0 && nspace % 128 == 0 && whole && d->lddy % 8 == 0 &&
        ((uintptr_t)d->dy % 16) == 0 && (!d->x1 || d->c0 % 8 == 0) && d->cin == cinx)
      p.dma = 1;
  }
  if (p.dma) { p.nci = 2; p.nco = 4; }
  else if (p.ntaps == 1 && db_ok && cinx % 128 == 0 && nspace % 256 == 0 && g_opt_wgrad_db != 2) { p.nci = 4; p.nco = 2; p.nw = 4; }
  else if (p.ntaps == 1) { p.nci = (db_ok && cinx % 64 == 0) ? 2 : 1; p.nco = 4; }
  else if (nspace % 128 == 0) { p.nci = 1; p.nco = 4; }
  else if (nspace % 64 == 0) { p.nci = (cinx % 64 == 0) ? 2 : 1; p.nco = 2; }
  else { p.nci = (cinx % 64 == 0) ? 2 : 1; p.nco = 1; }
  p.nks = 4 / (p.nci * p.nco) > 0 ? 4 / (p.nci * p.nco) : 1;                 // 4 waves per workgroup (single-buffered kernel)
  if (p.dma) p.nks = 1;
  const int ci_t = 32 * p.nci, co_t = 32 * p.nco * p.nw;
  p.n_ci_blk = cdiv(cinx, ci_t); p.n_co_blk = cdiv(nspace, co_t);
  p.kpad = p.n_ci_blk * ci_t; p.npad = p.n_co_blk * co_t;
  // thin layers (a handful of (ci, co) blocks at full resolution) are bound by bytes in flight: stage 256 pixels per step
  // (only with 32-channel X blocks: the 256-pixel halo tile of a 64-channel block is 65 KB and leaves ONE workgroup per CU --
  //  measured 842 us on dec0.conv1 (64->32), the slowest kernel of the step)
  // (and only the 1 x 1 block form in bf16: every other 256-pixel instantiation spills into scratch, which is ruinous)
  p.pix = (p.ntaps == 9 && p.tw == 32 && p.nci == 1 && p.nco == 1 && d->dtype == SATCV_BF16 && p.n_ci_blk * p.n_co_blk <= 3 && d->h >= 8 && d->w_ >= 256 &&
           d->dil == 1 && wgrad_pix256()) ? 256 : 128;
  if (p.nw > 1) p.pix = 64;
  if (p.dma) p.pix = 128;
  const int th = p.pix / p.tw;
  const int tiles_x = cdiv(d->w_, p.tw);
  long long ptiles;
  if (d->h >= th) ptiles = (long long)d->n * cdiv(d->h, th) * tiles_x;
  else ptiles = (long long)cdiv(d->n, th / d->h) * tiles_x;
  // two workgroups are resident per CU (144 accumulator registers): 512 fill the chip exactly once -- a third one per CU only
  // started a second, half-empty round (thin layers: 257 -> 217 us, 450 -> 366 us).  Layers with many (ci, co) blocks keep 768:
  // their per-block pixel share would get too coarse to balance.
  const int nblk = p.n_ci_blk * p.n_co_blk;
  long long ns = cdiv(nblk >= 128 ? 768 : 512, nblk);
  // double-buffered kernel (8 waves, one workgroup per CU): 256 workgroups fill the chip once; each should walk >= 16 pixel tiles so
  // that its set-up and its slab write (~16k cycles) stay small beside the tile loop
  p.db = db_ok ? 1 : 0;
  // ... but only HALF the chip: a resident workgroup of this kernel (8 waves x ~250 registers) owns its CU for the kernel's whole
  // duration, and with all 256 CUs taken the main stream's next kernels -- the few-block BatchNorm finalize launches above all -- wait
  // for the weight gradient to END (bn_bwd_finalize: 5 -> 12-280 us in the step).  The weight gradients have slack (their stream is
  // busy 4 of 10 ms): 160 workgroups make each of them ~12 % slower and the step 1.5 % faster (10.29 -> 10.13 ms, A/B/A/B; 128: the
  // same step, weight gradients 25 % slower; 224: -1.0 %; 96: the weight gradients become the critical path, +3 %).
  // With the round-4 LDS-DMA kernel (11 % faster by itself) the balance moved: 128 workgroups 8.53 ms, 160 8.64 ms (A/B on one box,
  // medians of five 20-step regions, gpurun_out -> profiles/r04_ab_wgs.txt).  SATCV_WGRAD_WGS overrides.
  static const int db_wgs = [] { const char* e = getenv("SATCV_WGRAD_WGS"); const int v = e ? atoi(e) : 128; return v >= 8 ? v : 128; }();
  const int wgs = d->whole_chip ? 256 : db_wgs;
  if (p.db) ns = nblk >= wgs ? 1 : wgs / nblk;
  if (ns > ptiles) ns = ptiles;
  if (ns > 768) ns = 768;
  if (ns < 1) ns = 1;
  if (ns >= 8) ns -= ns % 8;                               // XCD-aware block order needs a multiple of 8
  p.nsplit = (int)ns;
  p.ws_bytes = (size_t)p.nsplit * p.ntaps * p.kpad * p.npad * sizeof(float);
  return SATCV_OK;
}

extern "C" int64_t satcv_conv2d_wgrad_workspace(const satcv_wgrad_desc* d) {
  WgradPlan p;
  if (!d || wgrad_plan(d, p) != SATCV_OK) return -1;
  size_t need = p.ws_bytes;
  if (p.ntaps == 9) {                       // per-tap fallback of strongly dilated convs
    satcv_wgrad_desc d1 = *d;
    d1.kh = d1.kw = 1;
    WgradPlan p1;
    if (wgrad_plan(&d1, p1) == SATCV_OK && p1.ws_bytes > need) need = p1.ws_bytes;
  }
  return (int64_t)need;
}

template <typename T, int TW, int NCI, int NCO, int NKS, int NTAPS, int PIX = 128>
static int wgrad_launch(const satcv_wgrad_desc* d, const WgradPlan& p, hipStream_t st, int sy = 0, int sx = 0) {
  using G = WgradGeom<TW, NCI, NCO, NTAPS, T>;
  constexpr int TH = PIX / TW;
  WgradArgs a;
  a.x0 = d->x0; a.x1 = d->x1; a.c0 = d->c0; a.c1 = d->c1;
  a.in_scale = d->in_scale; a.in_shift = d->in_shift; a.in_relu = d->in_relu;
  a.dy = d->dy; a.lddy = d->lddy; a.ws = d->workspace;
  a.n = d->n; a.h = d->h; a.w_ = d->w_; a.kh = NTAPS == 1 ? 1 : d->kh; a.kw = NTAPS == 1 ? 1 : d->kw; a.dil = d->dil;
  a.sy = sy; a.sx = sx;
  a.mode_dy = d->mode_dy; a.f = d->f; a.cout_t = d->cout;
  a.kpad = p.kpad; a.npad = p.npad;
  a.cin_lim = d->c0 + d->c1; a.n_lim = d->mode_dy ? d->f * d->f * d->cout : d->cout;
  a.halh = a.dil * (a.kh - 1) / 2; a.halw = a.dil * (a.kw - 1) / 2;
  a.tiles_x = cdiv(d->w_, TW);
  if (d->h >= TH) { a.rpi = TH; a.imgs = 1; a.tiles_y = cdiv(d->h, TH); a.ngroups = d->n; }
  else { a.rpi = d->h; a.imgs = TH / d->h; a.tiles_y = 1; a.ngroups = cdiv(d->n, a.imgs); }
  a.seg = a.rpi + 2 * a.halh; a.rl = a.imgs * a.seg; a.cl = TW + 2 * a.halw;
  a.n_ci_blk = p.n_ci_blk; a.n_co_blk = p.n_co_blk; a.nsplit = p.nsplit;
  a.total_ptiles = a.ngroups * a.tiles_y * a.tiles_x;
  size_t lds = ((size_t)a.rl * a.cl * G::XP + PIX * G::DP) * sizeof(T) + (PIX + (size_t)a.rl * a.cl) * sizeof(int);
  if (lds < 3 * 4096) lds = 3 * 4096;          // k-slice reduction scratch
  if (lds > 160 * 1024) return SATCV_ERR_UNSUPPORTED;
  auto kern = wgrad_kernel<T, TW, NCI, NCO, NKS, NTAPS, PIX>;
  { const int rc = satcv_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds); if (rc) return rc; }
  hipLaunchKernelGGL(kern, dim3(p.n_ci_blk * p.n_co_blk * p.nsplit), dim3(NCI * NCO * NKS * 64), lds, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { satcv_set_error("wgrad launch: %s", hipGetErrorString(e)); return SATCV_ERR_HIP; }
  return SATCV_OK;
}

template <typename T, int TW, int NCI, int NCO, int NKS, int NTAPS, int PIX = 128, int NW = 1>
static int wgrad_db_launch(const satcv_wgrad_desc* d, const WgradPlan& p, hipStream_t st, int sy = 0, int sx = 0) {
  using G = WgradGeom<TW, NCI, NCO * NW, NTAPS, T>;
  constexpr int TH = PIX / TW, NTHREADS = NCI * NCO * NKS * 64, GX = G::CI_T / 8;
  WgradArgs a;
  a.x0 = d->x0; a.x1 = d->x1; a.c0 = d->c0; a.c1 = d->c1;
  a.in_scale = d->in_scale; a.in_shift = d->in_shift; a.in_relu = d->in_relu;
  a.dy = d->dy; a.lddy = d->lddy; a.ws = d->workspace;
  a.n = d->n; a.h = d->h; a.w_ = d->w_; a.kh = NTAPS == 1 ? 1 : d->kh; a.kw = NTAPS == 1 ? 1 : d->kw; a.dil = d->dil;
  a.sy = sy; a.sx = sx;
  a.mode_dy = d->mode_dy; a.f = d->f; a.cout_t = d->cout;
  a.kpad = p.kpad; a.npad = p.npad;
  a.cin_lim = d->c0 + d->c1; a.n_lim = d->mode_dy ? d->f * d->f * d->cout : d->cout;
  a.halh = a.dil * (a.kh - 1) / 2; a.halw = a.dil * (a.kw - 1) / 2;
  a.tiles_x = cdiv(d->w_, TW);
  if (d->h >= TH) { a.rpi = TH; a.imgs = 1; a.tiles_y = cdiv(d->h, TH); a.ngroups = d->n; }
  else { a.rpi = d->h; a.imgs = TH / d->h; a.tiles_y = 1; a.ngroups = cdiv(d->n, a.imgs); }
  a.seg = a.rpi + 2 * a.halh; a.rl = a.imgs * a.seg; a.cl = TW + 2 * a.halw;
  a.n_ci_blk = p.n_ci_blk; a.n_co_blk = p.n_co_blk; a.nsplit = p.nsplit;
  a.total_ptiles = a.ngroups * a.tiles_y * a.tiles_x;
  // register-staged items per thread (same bound as in the kernel)
  constexpr int XMAXPIX = NTAPS == 1 ? PIX : (PIX == 256 ? (TH + 2) * (TW + 2) : (TH + 2 * (TH / 4 > 1 ? TH / 4 : 1)) * (TW + 2));
  constexpr int XI = (XMAXPIX * GX + NTHREADS - 1) / NTHREADS;
  if ((long long)a.rl * a.cl * GX > (long long)XI * NTHREADS) return SATCV_ERR_UNSUPPORTED;
  if (a.rl >= 1024 || a.cl >= 1024 || a.imgs >= 1024) return SATCV_ERR_UNSUPPORTED;
  if (NTAPS == 9 && (a.dil != 1 || a.cl != TW + 2)) return SATCV_ERR_UNSUPPORTED;       // the kernel's tap offsets are compile-time
  const size_t stage = (((size_t)a.rl * a.cl * G::XP + (size_t)PIX * G::DP) + 7) / 8 * 8;
  size_t lds = 2 * stage * sizeof(T) + (size_t)PIX * sizeof(int) + (size_t)GX * 16 * sizeof(float);
  const size_t red = (size_t)NCI * NCO * (NKS > 1 ? NKS - 1 : 0) * 16 * 64 * sizeof(float);
  if (lds < red) lds = red;
  if (lds > 160 * 1024) return SATCV_ERR_UNSUPPORTED;
  auto kern = wgrad_db_kernel<T, TW, NCI, NCO, NKS, NTAPS, PIX, NW>;
  { const int rc = satcv_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds); if (rc) return rc; }
  hipLaunchKernelGGL(kern, dim3(p.n_ci_blk * p.n_co_blk * p.nsplit), dim3(NTHREADS), lds, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { satcv_set_error("wgrad_db launch: %s", hipGetErrorString(e)); return SATCV_ERR_HIP; }
  return SATCV_OK;
}

template <int TW, bool M16 = false>
static int wgrad_dma_launch(const satcv_wgrad_desc* d, const WgradPlan& p, hipStream_t st) {
  constexpr int PIX = 128, TH = PIX / TW, NTHREADS = 512, XP = M16 ? 80 : 96, GX = 8;
  WgradArgs a;
  memset(&a, 0, sizeof(a));
  a.x0 = d->x0; a.x1 = d->x1; a.c0 = d->c0; a.c1 = d->c1;
  a.in_scale = d->in_scale; a.in_shift = d->in_shift; a.in_relu = d->in_relu;
  a.dy = d->dy; a.lddy = d->lddy; a.ws = d->workspace;
  a.n = d->n; a.h = d->h; a.w_ = d->w_; a.kh = 3; a.kw = 3; a.dil = 1;
  a.f = 1; a.cout_t = d->cout;
  a.kpad = p.kpad; a.npad = p.npad;
  a.cin_lim = d->c0 + d->c1; a.n_lim = d->cout;
  a.halh = 1; a.halw = 1;
  a.tiles_x = d->w_ / TW;
  if (d->h >= TH) { a.rpi = TH; a.imgs = 1; a.tiles_y = d->h / TH; a.ngroups = d->n; }
  else { a.rpi = d->h; a.imgs = TH / d->h; a.tiles_y = 1; a.ngroups = d->n / a.imgs; }
  a.seg = a.rpi + 2; a.rl = a.imgs * a.seg; a.cl = TW + 2;
  a.n_ci_blk = p.n_ci_blk; a.n_co_blk = p.n_co_blk; a.nsplit = p.nsplit;
  a.total_ptiles = a.ngroups * a.tiles_y * a.tiles_x;
  constexpr int XMAXPIX = (TH + 2 * (TH / 4 > 1 ? TH / 4 : 1)) * (TW + 2);
  constexpr int XI = (XMAXPIX * GX + NTHREADS - 1) / NTHREADS;
  if ((long long)a.rl * a.cl * GX > (long long)XI * NTHREADS) return SATCV_ERR_UNSUPPORTED;
  if (a.rl >= 1024 || a.cl >= 1024 || a.imgs >= 1024) return SATCV_ERR_UNSUPPORTED;
  // a lane's dY offset inside a tile is a 32-bit byte offset
  if ((long long)a.imgs * d->h * d->w_ * d->lddy * 2 >= (1ll << 31)) return SATCV_ERR_UNSUPPORTED;
  const size_t stage = (size_t)a.rl * a.cl * XP + (size_t)PIX * 128;
  const size_t lds = 2 * stage * sizeof(bf16) + (size_t)PIX * sizeof(int) + (size_t)GX * 16 * sizeof(float);
  if (lds > 160 * 1024) return SATCV_ERR_UNSUPPORTED;
  auto kern = wgrad_dma_kernel<TW, M16>;
  { const int rc = satcv_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds); if (rc) return rc; }
  hipLaunchKernelGGL(kern, dim3(p.n_ci_blk * p.n_co_blk * p.nsplit), dim3(NTHREADS), lds, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { satcv_set_error("wgrad_dma launch: %s", hipGetErrorString(e)); return SATCV_ERR_HIP; }
  return SATCV_OK;
}

// the double-buffered kernel (bf16): 8 waves = (ci, co) tiles x k-slices
template <typename T, int TW>
static int wgrad_db_cfg(const satcv_wgrad_desc* d, const WgradPlan& p, hipStream_t st, int sy, int sx) {
  if constexpr (std::is_same<T, bf16>::value) {
    if (p.dma) {
      // option wgrad_m16 (SATCV_WGRAD_M16, default 0): the 16x16x32 form -- correct, and measured 33 % SLOWER per launch (profiles/r06_ab_wgrad_m16.txt)
      return g_opt_wgrad_m16 ? wgrad_dma_launch<TW, true>(d, p, st) : wgrad_dma_launch<TW, false>(d, p, st);
    }      // (the plan's slab geometry is this kernel's: no other kernel can serve it)
  }
  if (p.ntaps == 1 && p.nw == 4) return wgrad_db_launch<T, TW, 4, 2, 1, 1, 64, 4>(d, p, st, sy, sx);
  if (p.ntaps == 1 && p.nci == 2) return wgrad_db_launch<T, TW, 2, 4, 1, 1>(d, p, st, sy, sx);
  if (p.ntaps == 1) return wgrad_db_launch<T, TW, 1, 4, 2, 1>(d, p, st, sy, sx);
  if (p.nci == 1 && p.nco == 4) return wgrad_db_launch<T, TW, 1, 4, 2, 9>(d, p, st);
  if (p.nci == 2 && p.nco == 2) return wgrad_db_launch<T, TW, 2, 2, 2, 9>(d, p, st);
  if constexpr (TW == 32) {
    if (p.pix == 256 && p.nci == 1 && p.nco == 1) return wgrad_db_launch<T, TW, 1, 1, 8, 9, 256>(d, p, st);
  }
  if (p.nci == 1 && p.nco == 2) return wgrad_db_launch<T, TW, 1, 2, 4, 9>(d, p, st);
  if (p.nci == 2 && p.nco == 1) return wgrad_db_launch<T, TW, 2, 1, 4, 9>(d, p, st);
  return wgrad_db_launch<T, TW, 1, 1, 8, 9>(d, p, st);
}

template <typename T, int TW>
static int wgrad_cfg(const satcv_wgrad_desc* d, const WgradPlan& p, hipStream_t st, int sy = 0, int sx = 0) {
  if constexpr (std::is_same<T, bf16>::value) {
    if (p.db) {
      const int rc = wgrad_db_cfg<T, TW>(d, p, st, sy, sx);
      if (rc != SATCV_ERR_UNSUPPORTED) return rc;
      if (p.ntaps == 1 && (p.nci != 1 || p.nw != 1)) { satcv_set_error("wgrad: block plan without a kernel"); return rc; }   // (the plan's slab geometry is the db kernel's)
    }
  }
  if (p.ntaps == 1) return wgrad_launch<T, TW, 1, 4, 1, 1>(d, p, st, sy, sx);
  if (p.nci == 1 && p.nco == 4) return wgrad_launch<T, TW, 1, 4, 1, 9>(d, p, st);
  if (p.nci == 2 && p.nco == 2) return wgrad_launch<T, TW, 2, 2, 1, 9>(d, p, st);
  if constexpr (TW == 32) {
    if constexpr (std::is_same<T, bf16>::value) {
      if (p.pix == 256 && p.nci == 1 && p.nco == 1) return wgrad_launch<T, TW, 1, 1, 4, 9, 256>(d, p, st);
    }
  }
  if (p.nci == 1 && p.nco == 2) return wgrad_launch<T, TW, 1, 2, 2, 9>(d, p, st);
  if (p.nci == 2 && p.nco == 1) return wgrad_launch<T, TW, 2, 1, 2, 9>(d, p, st);
  return wgrad_launch<T, TW, 1, 1, 4, 9>(d, p, st);
}
template <typename T>
static int wgrad_t(const satcv_wgrad_desc* d, const WgradPlan& p, hipStream_t st, int sy = 0, int sx = 0) {
  switch (p.tw) {
    case 32: return wgrad_cfg<T, 32>(d, p, st, sy, sx);
    case 16: return wgrad_cfg<T, 16>(d, p, st, sy, sx);
    default: return wgrad_cfg<T, 8>(d, p, st, sy, sx);
  }
}
static int wgrad_any(const satcv_wgrad_desc* d, const WgradPlan& p, hipStream_t st, int sy = 0, int sx = 0) {
  if (d->dtype == SATCV_BF16) return wgrad_t<bf16>(d, p, st, sy, sx);
  if (d->dtype == SATCV_F32) return wgrad_t<float>(d, p, st, sy, sx);
  satcv_set_error("wgrad: bad dtype");
  return SATCV_ERR_INVALID;
}
static int wgrad_reduce_launch(const satcv_wgrad_desc* d, const WgradPlan& p, float* dw, int nvalid, hipStream_t st) {
  const long long total = (long long)p.ntaps * d->cin * nvalid;
  if (!d->transposed && nvalid % 4 == 0 && p.npad % 4 == 0 && ((uintptr_t)dw % 16) == 0 && ((uintptr_t)d->workspace % 16) == 0) {
    if (p.nsplit >= 32) {
      int grid16 = (int)((total / 4 + 15) / 16); if (grid16 > 16384) grid16 = 16384;
      hipLaunchKernelGGL(wgrad_reduce16_kernel, dim3(grid16), dim3(256), 0, st, (const float*)d->workspace, dw, p.nsplit, p.ntaps, p.kpad, p.npad, d->cin, nvalid,
                         d->accumulate);
      hipError_t e16 = hipGetLastError();
      if (e16 != hipSuccess) { satcv_set_error("wgrad reduce launch: %s", hipGetErrorString(e16)); return SATCV_ERR_HIP; }
      return SATCV_OK;
    }
    int grid4 = (int)((total / 4 + 63) / 64); if (grid4 > 8192) grid4 = 8192;
    hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3(grid4), dim3(256), 0, st, (const float*)d->workspace, dw, p.nsplit, p.ntaps, p.kpad, p.npad, d->cin, nvalid,
                       d->accumulate);
    hipError_t e4 = hipGetLastError();
    if (e4 != hipSuccess) { satcv_set_error("wgrad reduce launch: %s", hipGetErrorString(e4)); return SATCV_ERR_HIP; }
    return SATCV_OK;
  }
  int grid = (int)((total + 63) / 64); if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid), dim3(256), 0, st, d->workspace, dw, p.nsplit, p.ntaps, p.kpad, p.npad, d->cin, nvalid,
                     d->transposed, d->accumulate);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { satcv_set_error("wgrad reduce launch: %s", hipGetErrorString(e)); return SATCV_ERR_HIP; }
  return SATCV_OK;
}

void satcv_prof_begin(int kind, double flops, hipStream_t st);
void satcv_prof_end(int kind, hipStream_t st);

// slab sum for other producers of [nslab][taps][kpad][npad] partial weight gradients (conv_bwd_fused.hip): plain HWIO layout
int wgrad_reduce_slabs(const float* ws, float* dw, int nslab, int taps, int kpad, int npad, int cin, int nvalid, int accumulate, hipStream_t st) {
  satcv_wgrad_desc d;
  memset(&d, 0, sizeof(d));
  d.workspace = const_cast<float*>(ws); d.cin = cin; d.accumulate = accumulate; d.transposed = 0;
  WgradPlan p;
  memset(&p, 0, sizeof(p));
  p.nsplit = nslab; p.ntaps = taps; p.kpad = kpad; p.npad = npad;
  return wgrad_reduce_launch(&d, p, dw, nvalid, st);
}

// ... and for the transposed-convolution layout (f, f, cout, cin) of convt_bwd_fused.hip: slabs [nslab][kpad = cin][npad = f f cout]
int wgrad_reduce_slabs_t(const float* ws, float* dw, int nslab, int kpad, int npad, int cin, int nvalid, int accumulate, hipStream_t st) {
  satcv_wgrad_desc d;
  memset(&d, 0, sizeof(d));
  d.workspace = const_cast<float*>(ws); d.cin = cin; d.accumulate = accumulate; d.transposed = 1;
  WgradPlan p;
  memset(&p, 0, sizeof(p));
  p.nsplit = nslab; p.ntaps = 1; p.kpad = kpad; p.npad = npad;
  return wgrad_reduce_launch(&d, p, dw, nvalid, st);
}

extern "C" int satcv_conv2d_wgrad(const satcv_wgrad_desc* d, void* stream) {
  SATCV_CHECK(d && d->x0 && d->dy && d->dw && d->workspace, "wgrad: null pointer");
  SATCV_CHECK(d->c0 > 0 && d->c0 % 8 == 0 && d->c1 % 8 == 0 && (d->c1 == 0) == (d->x1 == nullptr), "wgrad: bad source channels");
  SATCV_CHECK(d->cin > 0 && d->cin <= d->c0 + d->c1 && d->cout > 0, "wgrad: bad cin/cout");
  SATCV_CHECK(d->n > 0 && d->h > 0 && d->w_ > 0 && d->dil >= 1, "wgrad: bad dims");
  SATCV_CHECK(!d->mode_dy || (d->f >= 2 && d->kh == 1 && d->kw == 1 && d->cout % 8 == 0), "wgrad: transposed conv needs 1x1 taps, f>=2");
  SATCV_CHECK(!d->x1 || d->c0 % 8 == 0, "wgrad: dual source needs c0 %% 8 == 0");      // (a thread's 8-channel item never straddles the two sources)
  WgradPlan p;
  int rc = wgrad_plan(d, p); if (rc) return rc;
  SATCV_CHECK((size_t)d->workspace_bytes >= p.ws_bytes, "wgrad: workspace %lld < %zu", (long long)d->workspace_bytes, p.ws_bytes);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nvalid = d->mode_dy ? d->f * d->f * d->cout : d->cout;
  const double flops = 2.0 * d->n * d->h * d->w_ * (double)nvalid * (double)(d->c0 + d->c1) * d->kh * d->kw;
  satcv_prof_begin(2, flops, st);
  rc = wgrad_any(d, p, st);
  if (rc == SATCV_ERR_UNSUPPORTED && p.ntaps == 9) {
    if (d->defer_reduce) { satcv_prof_end(2, st); satcv_set_error("wgrad: defer_reduce is not available on the per-tap path of strongly dilated convolutions"); return SATCV_ERR_UNSUPPORTED; }
    // halo tile of a strongly dilated conv does not fit the LDS: treat every tap as a shifted 1x1 product
    satcv_wgrad_desc d1 = *d;
    d1.kh = d1.kw = 1;
    WgradPlan p1;
    rc = wgrad_plan(&d1, p1);
    if (rc == SATCV_OK && (size_t)d->workspace_bytes < p1.ws_bytes) { satcv_set_error("wgrad: workspace too small for the per-tap path"); rc = SATCV_ERR_INVALID; }
    for (int tap = 0; tap < 9 && rc == SATCV_OK; ++tap) {
      rc = wgrad_any(&d1, p1, st, (tap / 3 - 1) * d->dil, (tap % 3 - 1) * d->dil);
      if (rc == SATCV_OK) rc = wgrad_reduce_launch(&d1, p1, d->dw + (size_t)tap * d->cin * nvalid, nvalid, st);
    }
    satcv_prof_end(2, st);
    return rc;
  }
  satcv_prof_end(2, st);
  if (rc) return rc;
  if (d->defer_reduce) return SATCV_OK;            // the caller sums the slabs later (satcv_reduce_slabs_batched)
  return wgrad_reduce_launch(d, p, d->dw, nvalid, st);
}

// ------------------------------------------------------------------ deferred, batched slab sum (include/satcv.h)
// item = (float4 of 4 consecutive output channels, lane of its group): lane l sums slabs l, l + lanes, ... (four interleaved chains, added
// in a fixed order), the group's lanes are combined in increasing lane order.  One thread block walks items of possibly several jobs.
static int reduce_job_lanes(int nslab) { int l = 1; while (l < 16 && 2 * l <= nslab / 2) l *= 2; return l; }      // >= 2 slabs per lane
extern "C" int64_t satcv_reduce_job_items(const satcv_reduce_job* j) {
  if (!j || j->nvalid % 4 != 0 || j->lanes < 1) return -1;
  // rounded up to a multiple of 16 (the largest `lanes`): with every job's prefix a multiple of 16 a lane group never straddles a wave, whatever
  // the `lanes` of the jobs in front of it (the padded items are inactive in the kernel)
  const int64_t real = (int64_t)j->taps * j->cin * (j->nvalid / 4) * j->lanes;
  return (real + 15) / 16 * 16;
}
__global__ __launch_bounds__(256) void reduce_slabs_batched_kernel(const satcv_reduce_job* __restrict__ jobs, const long long* __restrict__ prefix, int njobs,
                                                                   long long total) {
  for (long long it0 = (long long)blockIdx.x * 256; it0 < total; it0 += (long long)gridDim.x * 256) {
    const long long it = it0 + threadIdx.x;
    const bool act0 = it < total;
    const long long itc = act0 ? it : total - 1;
    int ji = 0;
    for (int q = 1; q < njobs; ++q) ji += (prefix[q] <= itc) ? 1 : 0;      // (a few dozen jobs: a linear scan of an L1-resident table)
    const satcv_reduce_job j = jobs[ji];
    const long long real = (long long)j.taps * j.cin * (j.nvalid / 4) * j.lanes;      // (the job's item count is `real` rounded up to 16)
    const long long loc0 = itc - prefix[ji];
    const bool act = act0 && loc0 < real;
    const long long loc = loc0 < real ? loc0 : real - 1;
    const int P = j.lanes, l = (int)(loc % P);
    const long long o4 = loc / P;
    const int nv4 = j.nvalid / 4;
    const int co = (int)(o4 % nv4) * 4, ci = (int)((o4 / nv4) % j.cin), tap = (int)(o4 / ((long long)nv4 * j.cin));
    const size_t slab = (size_t)j.taps * j.kpad * j.npad;
    const float* p = j.ws + ((size_t)tap * j.kpad + ci) * j.npad + co;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    int sp = l;
    for (; sp + 3 * P < j.nslab; sp += 4 * P) {
      const float4 v0 = *reinterpret_cast<const float4*>(p + (size_t)sp * slab), v1 = *reinterpret_cast<const float4*>(p + (size_t)(sp + P) * slab);
      const float4 v2 = *reinterpret_cast<const float4*>(p + (size_t)(sp + 2 * P) * slab), v3 = *reinterpret_cast<const float4*>(p + (size_t)(sp + 3 * P) * slab);
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
      a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
      a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
      a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
    }
    for (; sp < j.nslab; sp += P) { const float4 v = *reinterpret_cast<const float4*>(p + (size_t)sp * slab); a0.x += v.x; a0.y += v.y; a0.z += v.z; a0.w += v.w; }
    float4 r = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w));
    // lanes of a group are consecutive threads of one wave (256 and 64 are multiples of every lanes value): ordered tree, lane 0 keeps it
    for (int o = 1; o < P; o <<= 1) {
      const float x = __shfl_down(r.x, o, 64), y = __shfl_down(r.y, o, 64), z = __shfl_down(r.z, o, 64), w = __shfl_down(r.w, o, 64);
      if ((l & (2 * o - 1)) == 0) { r.x += x; r.y += y; r.z += z; r.w += w; }
    }
    if (act && l == 0) {
      if (j.transposed) {
        float* dst = j.dw + (size_t)co * j.cin + ci;
        const float rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[(size_t)e * j.cin] = j.accumulate ? dst[(size_t)e * j.cin] + rr[e] : rr[e];
      } else {
        float4* dst = reinterpret_cast<float4*>(j.dw + ((size_t)tap * j.cin + ci) * j.nvalid + co);
        if (j.accumulate) { const float4 o = *dst; r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w; }
        *dst = r;
      }
    }
  }
}
extern "C" int satcv_reduce_slabs_batched(const satcv_reduce_job* jobs_dev, const int64_t* prefix_dev, int32_t njobs, int64_t total_items, void* stream) {
  SATCV_CHECK(jobs_dev && prefix_dev && njobs > 0 && njobs <= 4096 && total_items > 0, "reduce_slabs_batched: bad arguments");
  SATCV_CHECK(total_items % 16 == 0, "reduce_slabs_batched: the prefix table must be built from satcv_reduce_job_items (multiples of 16)");
  long long grid = (total_items + 255) / 256; if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(reduce_slabs_batched_kernel, dim3((unsigned)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), jobs_dev,
                     reinterpret_cast<const long long*>(prefix_dev), njobs, (long long)total_items);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { satcv_set_error("reduce_slabs_batched launch: %s", hipGetErrorString(e)); return SATCV_ERR_HIP; }
  return SATCV_OK;
}
void reduce_job_fill(satcv_reduce_job* job, const float* ws, float* dw, int nslab, int taps, int kpad, int npad, int cin, int nvalid, int transposed, int accumulate) {
  job->ws = ws; job->dw = dw; job->nslab = nslab; job->taps = taps; job->kpad = kpad; job->npad = npad; job->cin = cin; job->nvalid = nvalid;
  job->transposed = transposed; job->accumulate = accumulate; job->lanes = reduce_job_lanes(nslab); job->pad_ = 0;
}
extern "C" int satcv_conv2d_wgrad_reduce_job(const satcv_wgrad_desc* d, satcv_reduce_job* job) {
  SATCV_CHECK(d && job && d->dw && d->workspace, "wgrad_reduce_job: null pointer");
  WgradPlan p;
  int rc = wgrad_plan(d, p); if (rc) return rc;
  const int nvalid = d->mode_dy ? d->f * d->f * d->cout : d->cout;
  if (nvalid % 4 != 0) { satcv_set_error("wgrad_reduce_job: %d output channels are not a multiple of 4", nvalid); return SATCV_ERR_UNSUPPORTED; }
  reduce_job_fill(job, d->workspace, d->dw, p.nsplit, p.ntaps, p.kpad, p.npad, d->cin, nvalid, d->transposed, d->accumulate);
  return SATCV_OK;
}
