// Software-pipelined implicit-GEMM convolution (the hot variant of conv_igemm.hip).
//
// Thin-layer note (measured, round 1): the 16..64-channel full- and half-resolution layers are bound by bytes in flight, not by MFMA
// or LDS (2.0-2.9 TB/s algorithmic vs 5.4 TB/s of a copy kernel).  What helped: 4 waves/SIMD for the small-accumulator
// configurations (+10-15 %).  What did not: persistent workgroups with the next tile's loads in flight during the
// epilogue (the extra live state costs one wave/SIMD, net +-0; removed), 32-channel chunks, 256-pixel x 128-channel tiles,
// 8-wave thin tiles, de-phasing the workgroups of a CU with s_sleep.
//
// Same math, LDS images and MFMA fragment addressing as igemm_kernel, plus:
//   * the halo-tile gather table (LDS offset + source pixel per staged 16-byte item) is computed
//     ONCE per workgroup and kept in registers -- no integer divisions in the K loop;
//   * register-prefetch pipeline: the global loads of chunk c+1 are issued before the MFMAs of
//     chunk c and written to LDS after them (one LDS buffer, T14-style issue-early/write-late);
//   * the previous layer's BN affine + ReLU is applied in registers between load and LDS write;
//   * the output tile is staged through LDS and written with 16-byte coalesced stores
//     (NHWC rows; depth-to-space rows for the transposed conv);
//   * blockIdx is remapped so that the workgroups sharing an activation tile / neighbouring
//     halos run on the same XCD (private L2).
#include "igemm_common.hpp"
#include <cstdlib>
extern int g_opt_igemm_db, g_opt_igemm_thin, g_opt_igemm_sched, g_opt_splitk, g_opt_igemm_m16;      // api.hip: satcv_set_option

// compile-time ablation switches for profiling builds (-DSATCV_ABLATE=bits): 1 skip global stores, 2 skip MFMA,
// 4 skip activation loads, 8 skip weight loads, 16 skip the LDS fragment reads, 32 skip the whole epilogue, 64 skip the LDS stores,
// 128 every chunk loads the weights of chunk 0 (cache-resident: what is left is the cost of issuing them), 256 likewise the activations
#ifndef SATCV_ABLATE
#define SATCV_ABLATE 0
#endif
#define ABL(bit) ((SATCV_ABLATE & (bit)) != 0)
// diagnostic build only (-DSATCV_STAMP): per-wave cycle sums of the phases of the double-buffered K loop (s_memtime), for the first
// workgroups; read back with satcv_debug_read_stamps (not part of the product ABI, absent from normal builds)
#ifdef SATCV_STAMP
__device__ unsigned long long g_stamp[8][8][8];
#define STAMP(t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
extern "C" int satcv_debug_read_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp), sizeof(g_stamp)) == hipSuccess ? 0 : -1;
}
#endif
// TL (tap loop, TAPS == 1 geometry): a 3x3 conv whose dilation makes the halo tile many times larger than the tile itself (ASPP: 3 / 6 /
// 12 against 4 x 32 pixels) runs as 9 x Cin/32 K-chunks instead -- per tap the tile is gathered at its shifted position (no halo, zero
// outside the image) and multiplied with that tap's weight slab.
// DB (double buffer): two LDS stages and ONE barrier per K-chunk.  The registers hold chunk c+1 while chunk c is multiplied; at the top
// of iteration c they are written into the stage that chunk c-1 used (every wave is past the barrier that ended c-1) and re-issued
// for chunk c+2 at once, so a chunk's loads are in flight for a whole iteration (write-after-barrier / re-issue-immediately,
// cdna_hip_programming.md T14).  Used with the 256-pixel x 128-channel, 8-wave tile for the deep layers: per MFMA the 9-tap weight
// slab (77 % of the staged bytes, re-fetched from L2 by every workgroup) is amortised over twice the pixels of the 128 x 128 tile
// (20.7 instead of 37.7 staged bytes per MFMA-cycle and CU at full rate), which is what bounded that tile at ~0.9 PFLOP/s.
// WDMA (round 4, double-buffered tile only): the weight slab of a chunk -- 77 % of the staged bytes, and the operand that needs no
// transform -- moves by LDS-DMA (global_load_lds_dwordx4, 1-KB pieces, no staging registers, no ds_write) into a THREE-slot ring two
// chunks ahead; the activations keep the register path (fused input BatchNorm + ReLU, zero padding).
// SK: split-K instantiation (see fast_cfg); every other instantiation compiles exactly the single-pass code (the K range is the constant [0, nchunks):
// compiled in unconditionally the extra live state cost the 128 x 128 tile a wave per SIMD)
// M16 (round 5, double-buffered tiles with chunks of a multiple of 32 channels): the contraction runs on v_mfma_f32_16x16x32_bf16 -- K = 32 per
// instruction, a lane quarter (l >> 4) feeds 8 channels, so a K step reads FOUR slot planes of one tap; a wave's 64 x 64 output is 4 x 4
// blocks of 16 x 16 kept in the same 64 accumulator registers (igemm_common.hpp: AccMap).  Same matrix cycles and LDS bytes per FLOP as
// the 32x32x16 form; the chip holds a higher clock under this shape (profiles/r04_exp_mfma_16x16x32_timing.txt: -5...-8 % on every
// MFMA-bound launch), and a 32-channel chunk halves the barriers per MFMA.  The slot planes are padded to a multiple of 256 bytes: the four
// 16-lane groups of a ds_read_b128 then take 16 distinct 16-byte bank slots each (MI355X_MICROARCH.md, LDS table).
template <typename T, int TW, int WM, int WN, int MT, int NT, int KS, int TAPS, bool DYN, bool TL = false, bool DB = false, int WPS = 0, bool WDMA = false, bool SK = false, bool M16 = false>
// thin configurations (<= 32 accumulator registers) request 4 waves/SIMD; the scaled-fp8 fragments are 8 registers each, so
// that path asks for 2.  WPS overrides (tile-at-once configurations stage a whole tile through registers)
__global__ __launch_bounds__(WM* WN * 64, (WPS ? WPS : (MT * NT <= 2 ? (KTraits<T>::SUB == 2 ? 2 : 4) : 1))) void igemm_fast_kernel(const IgemmArgs a) {
  constexpr int NTHREADS = WM * WN * 64;
  constexpr int BM = WM * MT * 32;
  constexpr int BN = WN * NT * 32;
  constexpr int TH = BM / TW;
  constexpr int EL = KTraits<T>::EL, SUB = KTraits<T>::SUB;      // elements per staged item, items per lane half per MFMA
  // DYN = dilated 3x3 taps: halo width and LDS pitch come from the arguments.  Otherwise they are compile-time functions of the
  // tile width, which turns every tap / sub-slot offset of the fragment reads into an instruction immediate (no address VGPRs).
  constexpr int CLc = TW + (TAPS == 9 ? 2 : 0);
  // (M16: the 16 lanes of a fragment quarter read 16 consecutive pixels of ONE row when TW >= 16, so the row pitch is free: no padding)
  constexpr int PITCHc = igemm_pitch(TW, CLc, M16);
  static_assert(!M16 || (std::is_same<T, bf16>::value && DB && !SK && !DYN && KS % 2 == 0 && TW >= 16), "16x16x32 form: double-buffered bf16 tiles, chunks of 32 k channels, tiles at least 16 pixels wide");
  const int pitch = DYN ? a.pitch : PITCHc;
  const int cl = DYN ? a.cl : CLc;
  const int dil = DYN ? a.dil : 1;
  constexpr int KC = KS * 2 * SUB * EL;                          // channels per chunk (16 per K-step; 64 for scaled fp8)
  constexpr int SLOTS = KC / EL;
  // staged A items per thread: halo tile of a 3x3 / dilation-1 conv incl. the several-images-per-tile case
  constexpr int XMAXPIX = (TAPS == 1 && DB) ? BM : (TH + 2 * (TH / 4 > 1 ? TH / 4 : 1)) * (TW + 2);      // (the double-buffered single-tap tile stages no halo; the older single-tap forms keep their budget: sized exactly, their 16-channel forms needed 24 bytes of scratch)
  constexpr int AI = (XMAXPIX * SLOTS + NTHREADS - 1) / NTHREADS;
  constexpr int BI = WDMA ? 1 : (TAPS * SLOTS * BN + NTHREADS - 1) / NTHREADS;  // staged B items per thread (none with WDMA: a dummy of 1)
  static_assert(!WDMA || (DB && !TL && std::is_same<T, bf16>::value && BN * EL * (int)sizeof(T) == 2048), "weights by LDS-DMA: double-buffered bf16 tile of 128 output channels");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* ldsA = reinterpret_cast<T*>(smem_raw);
  // padding between the slot planes: ds_write_b128 is serviced in groups of 8 lanes over 32 banks (128 B).  With 4+ slots the
  // 8 lanes of a group store 2 pixels x 4 slots, so the plane stride must be 32 B modulo 128 B for the eight 16-byte stores to
  // fall on distinct banks (unpadded 1x1 / 32-channel-chunk planes are multiples of 512 B: measured 43 % bank-conflict cycles)
  const int plane = a.rl * pitch * EL;
  const int spad = M16 ? ((256 - (plane * (int)sizeof(T)) % 256) % 256) / (int)sizeof(T)
                       : (SLOTS > 2 ? ((32 - (plane * (int)sizeof(T)) % 128 + 128) % 128) / (int)sizeof(T) : 0);
  const int slot_stride = plane + spad;
  const int a_stage = SLOTS * slot_stride;                                    // A planes of one stage
  constexpr int b_slab = TAPS * SLOTS * BN * EL;                              // weight slab of one chunk
  // without WDMA a stage is [A planes | weight slab] (DB: two stages); with WDMA: [A stage 0 | A stage 1 | ring slot 0 | 1 | 2]
  T* ldsB = WDMA ? ldsA + 2 * a_stage : ldsA + a_stage;
  const int stage_elems = a_stage + b_slab;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, hh = lane >> 5;
#ifdef SATCV_STAMP
  unsigned long long k0, k1, k2, k3;
  STAMP(k0);
#endif

  // ---- XCD-aware tile id: blocks b and b+8 share an XCD (private L2); give each XCD a contiguous tile range so that the
  //      workgroups running side by side read neighbouring halos
  const int G = gridDim.x;
  int bid;
  {
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q = G >> 3, rem = G & 7;
    bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (orig >> 3);
  }
  // split-K: the ksplit workgroups of one tile have consecutive ids (one XCD: they read the same activation tile)
  int split = 0, c_begin = 0, c_end = a.nchunks;
  if constexpr (SK) {
    const int ksplit = a.ksplit > 1 ? a.ksplit : 1;
    split = bid % ksplit;
    bid /= ksplit;
    c_begin = (int)((long long)split * a.nchunks / ksplit); c_end = (int)((long long)(split + 1) * a.nchunks / ksplit);
  }
  const int nbase = (bid % a.n_tiles) * BN;
  const int cin = a.c0 + a.c1;
  const int fs = a.mode_in == 1 ? a.f : a.stride;      // source pixels per output pixel (space-to-depth gather / strided conv)
  const int slot_t = tid % SLOTS;                  // NTHREADS % SLOTS == 0: a thread's slot is fixed

  // ---- tile-independent gather tables (registers): LDS offset and halo coordinates of every staged 16-byte item
  int a_l[AI], a_pk[AI], a_p[AI];
  const int a_items = a.rl * cl * SLOTS;
#pragma unroll
  for (int j = 0; j < AI; ++j) {
    const int it = tid + j * NTHREADS;
    a_l[j] = -1; a_pk[j] = 0; a_p[j] = -1;
    if (it < a_items) {
      const int pix = it / SLOTS;
      const int c = pix % cl, L = pix / cl;          // (division by a constant unless DYN)
      const int k = (a.imgs == 1) ? 0 : L / a.seg;
      a_l[j] = slot_t * slot_stride + (L * pitch + c) * EL;
      a_pk[j] = (k << 24) | ((L - k * a.seg) << 12) | c;
    }
  }
  constexpr int b_items = TAPS * SLOTS * BN;
  int b_g[BI];
#pragma unroll
  for (int j = 0; j < BI; ++j) {
    const int it = min(tid + j * NTHREADS, b_items - 1);      // surplus items of the last round: a valid address, never stored
    const int co = it % BN, run = it / BN;
    b_g[j] = ((run / SLOTS) * (cin / EL) + (run % SLOTS)) * a.cout_pad + nbase + co;
  }

  // fragment addresses of this lane: one per 32-row tile (32x32x16: row l & 31, slot offset from the lane half added per step) or one per
  // 16-row block (16x16x32: row l & 15, the slot plane of the lane quarter folded in)
  constexpr int NAF = M16 ? 2 * MT : MT;
  int a_off[NAF];
#pragma unroll
  for (int m = 0; m < NAF; ++m) {
    const int q = M16 ? (wm * MT * 32 + m * 16 + (lane & 15)) : ((wm * MT + m) * 32 + r);
    const int t = q / TW, cx = q % TW;
    const int k = (a.imgs == 1) ? 0 : t / a.rpi;
    const int l0 = (k < a.imgs) ? k * a.seg + (t - k * a.rpi) : 0;
    a_off[m] = (l0 * pitch + cx) * EL + (M16 ? (lane >> 4) * slot_stride : 0);
  }
  const int b_lane16 = ((lane >> 4) * BN + wn * NT * 32 + (lane & 15)) * EL;      // (M16) weight fragment: slot of the lane quarter, column l & 15

  const T* wp = reinterpret_cast<const T*>(a.w);
  Raw8<T> ra[AI], rb[BI];
  float* ldsT = reinterpret_cast<float*>(WDMA ? ldsA + 2 * a_stage + 3 * b_slab : ldsA + (DB ? 2 : 1) * stage_elems);  // [2][cin] scale, shift (behind the stage(s))
  if (a.in_scale) {
    for (int i = tid; i < cin; i += NTHREADS) { ldsT[i] = a.in_scale[i]; ldsT[cin + i] = a.in_shift[i]; }      // (visible after the barrier that ends the prologue)
  }

  auto tile_origin = [&](int v, int& n0, int& y0, int& x0) {
    int mt = v / a.n_tiles;
    const int tx = mt % a.tiles_x; mt /= a.tiles_x;
    const int ty = mt % a.tiles_y;
    n0 = (mt / a.tiles_y) * a.imgs; y0 = ty * TH; x0 = tx * TW;
  };
  auto gather_pixels = [&](int n0, int y0, int x0, int oy = 0, int ox = 0) {      // source pixel of every staged item of the tile at (n0, y0, x0)
#pragma unroll
    for (int j = 0; j < AI; ++j) {
      const int k = a_pk[j] >> 24, yy = ((a_pk[j] >> 12) & 0xfff) - a.halh, c = (a_pk[j] & 0xfff) - a.halw;
      const int n = n0 + k, y = y0 + yy, x = x0 + c;            // output-grid coordinates (halo rows / columns may fall outside)
      const int sy = y * fs + oy, sx = x * fs + ox;             // source coordinates; (oy, ox) = tap offset of the tap-loop form
      a_p[j] = ((a_l[j] >= 0) && (n < a.n) && (y >= 0) && (y < a.h) && (x >= 0) && (x < a.w_) && (sy >= 0) && (sy < a.hs) && (sx >= 0) && (sx < a.ws))
                   ? (n * a.hs + sy) * a.ws + sx : -1;
    }
  };
  // No vector-memory instruction of the K loop sits inside a lane-dependent branch: hipcc's s_waitcnt bookkeeping turns conservative at
  // the joins of such branches and put an `s_waitcnt vmcnt(0)` between the activation loads and the weight loads of a chunk (the
  // activation latency was exposed once per chunk, measured: 1.1 us x 64 chunks on 1024 -> 512 channels at 16 x 16).  Items outside
  // the image load pixel 0 and are zeroed by a select when they are written to LDS (after the affine: padding is zero in the
  // ACTIVATED tensor); the surplus weight items of the last round re-load the last valid item.
  float4 rs[4];                                    // this thread's 8 scale + 8 shift values of the chunk in flight (fused input BatchNorm)
  // staging is cut into UNITS -- activation items, the scale / shift values, weight items -- so that the double-buffered loop can
  // spread them over the tap steps of the chunk being multiplied (one LDS store + one global load between two groups of MFMAs):
  // issued back to back, the 11 loads of a chunk kept every wave in its issue phase for 1,400-3,000 cycles per chunk while the
  // vector-memory pipe took them in, with the matrix pipe idle (s_memtime stamps: tools/stamp_probe.py)
  struct ChunkSrc { const T* src; int cs, coff, sadd, cg0s; size_t cadd; const float* scp; const float* shp; };
  auto chunk_src = [&](int chunk_) {
    ChunkSrc c;
    const int tap = TL ? chunk_ / a.cpt : 0;
    const int chunk = TL ? chunk_ - tap * a.cpt : chunk_;
    const int cg0 = chunk * KC;
    c.sadd = 0;
    if (a.mode_in == 1) {
      const int ij = cg0 / a.c0;
      c.sadd = (ij / a.f) * a.ws + (ij % a.f);
      c.src = reinterpret_cast<const T*>(a.x0); c.cs = a.c0; c.coff = cg0 - ij * a.c0;
    } else if (cg0 < a.c0) {
      c.src = reinterpret_cast<const T*>(a.x0); c.cs = a.c0; c.coff = cg0;
    } else {
      c.src = reinterpret_cast<const T*>(a.x1); c.cs = a.c1; c.coff = cg0 - a.c0;
    }
    c.cadd = ((size_t)chunk * SLOTS + (TL ? (size_t)tap * (cin / EL) : 0)) * a.cout_pad;
    // (a pointer select, not a branch: without an input transform the values are loaded from the weight image and never used)
    c.cg0s = cg0 + slot_t * EL;
    c.scp = a.in_scale ? a.in_scale + cg0 + slot_t * EL : reinterpret_cast<const float*>(a.w);
    c.shp = a.in_scale ? a.in_shift + cg0 + slot_t * EL : reinterpret_cast<const float*>(a.w);
    return c;
  };
  // double-buffered tap-loop form: the gather table changes with the tap while loads of an earlier tap are still held in registers, so an
  // item's validity is captured when it is loaded
  bool ra_ok[AI];
  auto load_a = [&](const ChunkSrc& c, int j) {
    const int p = a_p[j] < 0 ? 0 : a_p[j];
    if constexpr (TL && DB) ra_ok[j] = a_p[j] >= 0;
    if (!ABL(4)) ra[j] = gload8<T>(c.src + (size_t)(p + c.sadd) * c.cs + (ABL(256) ? 0 : c.coff) + slot_t * EL);
    else ra[j] = zero8<T>();
  };
  auto load_b = [&](const ChunkSrc& c, int j) {
    if (!ABL(8)) rb[j] = gload8<T>(wp + ((size_t)b_g[j] + (ABL(128) ? 0 : c.cadd)) * EL);
    else rb[j] = zero8<T>();
  };
  // from_lds (double-buffered loop, after its first barrier): the scale / shift vectors sit in an LDS table behind the two stages.  As
  // global loads these 4 x 16 bytes per thread and chunk were 32 of the 88 one-KB wave-loads a chunk pushes through the CU's
  // vector-memory path -- the path whose instruction rate, not whose bytes, bounds the loop -- and launches
  // without an input transform (the data gradients) issued them too, as dummy loads of the weight image
  auto load_p = [&](const ChunkSrc& c, bool from_lds = false) {
    if constexpr (KTraits<T>::EL == 8) {
      if (from_lds) {
        if (a.in_scale) {
          const float4* tp = reinterpret_cast<const float4*>(ldsT + c.cg0s);
          const float4* hp = reinterpret_cast<const float4*>(ldsT + cin + c.cg0s);
          rs[0] = tp[0]; rs[1] = tp[1]; rs[2] = hp[0]; rs[3] = hp[1];
        }
      } else {
        rs[0] = reinterpret_cast<const float4*>(c.scp)[0]; rs[1] = reinterpret_cast<const float4*>(c.scp)[1];
        rs[2] = reinterpret_cast<const float4*>(c.shp)[0]; rs[3] = reinterpret_cast<const float4*>(c.shp)[1];
      }
    }
  };
  auto store_a = [&](int chunk_, int boff, int j) {
    Raw8<T> v = ra[j];
    if (a.in_scale) {
      if constexpr (KTraits<T>::EL == 8) v = affine8r<T>(v, rs, a.in_relu);
      else {
        const int cg0 = (TL ? chunk_ % a.cpt : chunk_) * KC + slot_t * EL;
        v = affine8<T>(v, a.in_scale + cg0, a.in_shift + cg0, a.in_relu);
      }
    }
    if constexpr (TL && DB) v = select8<T>(ra_ok[j], v);
    else v = select8<T>(a_p[j] >= 0, v);
    if (ABL(64)) { keep8<T>(v); return; }
    if (a_l[j] >= 0) lstore8<T>(ldsA + boff + a_l[j], v);
  };
  auto store_b = [&](int boff, int j) {
    const int it = tid + j * NTHREADS;
    if (ABL(64)) { keep8<T>(rb[j]); return; }
    if (it < b_items) lstore8<T>(ldsB + boff + (size_t)it * EL, rb[j]);
  };
  // single-buffered instantiations keep the loader they were tuned with (lane-predicated loads, scale / shift read where an item is
  // written: the thin configurations sit exactly at the 128-register cap of four waves per SIMD, 16 more live registers spill)
  auto load_regs = [&](int chunk_) {
    if constexpr (DB) {
      const ChunkSrc c = chunk_src(chunk_);
      // (same issue order as the units of the double-buffered loop -- activations, scale / shift, weights -- so that the counted
      //  s_waitcnt of the first iteration and of the steady state agree)
#pragma unroll
      for (int j = 0; j < AI; ++j) load_a(c, j);
      load_p(c);
      if constexpr (!WDMA) {
#pragma unroll
        for (int j = 0; j < BI; ++j) load_b(c, j);
      }
    } else {
      const int tap = TL ? chunk_ / a.cpt : 0;
      const int chunk = TL ? chunk_ - tap * a.cpt : chunk_;
      const int cg0 = chunk * KC;
      const T* src; int cs, coff, sadd = 0;
      if (a.mode_in == 1) {
        const int ij = cg0 / a.c0;
        sadd = (ij / a.f) * a.ws + (ij % a.f);
        src = reinterpret_cast<const T*>(a.x0); cs = a.c0; coff = cg0 - ij * a.c0;
      } else if (cg0 < a.c0) {
        src = reinterpret_cast<const T*>(a.x0); cs = a.c0; coff = cg0;
      } else {
        src = reinterpret_cast<const T*>(a.x1); cs = a.c1; coff = cg0 - a.c0;
      }
#pragma unroll
      for (int j = 0; j < AI; ++j) {
        if (a_p[j] >= 0 && !ABL(4)) ra[j] = gload8<T>(src + (size_t)(a_p[j] + sadd) * cs + coff + slot_t * EL);
        else ra[j] = zero8<T>();
      }
      const size_t cadd = ((size_t)chunk * SLOTS + (TL ? (size_t)tap * (cin / EL) : 0)) * a.cout_pad;
#pragma unroll
      for (int j = 0; j < BI; ++j) {
        if (tid + j * NTHREADS < b_items && !ABL(8)) rb[j] = gload8<T>(wp + ((size_t)b_g[j] + cadd) * EL);
        else rb[j] = zero8<T>();
      }
    }
  };
  auto store_lds = [&](int chunk_, int boff = 0) {
    if constexpr (DB) {
#pragma unroll
      for (int j = 0; j < AI; ++j) store_a(chunk_, boff, j);
      if constexpr (!WDMA) {
#pragma unroll
        for (int j = 0; j < BI; ++j) store_b(boff, j);
      }
    } else {
      const int chunk = TL ? chunk_ % a.cpt : chunk_;
      const int cg0 = chunk * KC + slot_t * EL;
#pragma unroll
      for (int j = 0; j < AI; ++j) {
        if (a_l[j] >= 0) {
          Raw8<T> v = ra[j];
          if (a.in_scale && a_p[j] >= 0) {
#ifdef SATCV_RS_GLOBAL
            v = affine8<T>(v, a.in_scale + cg0, a.in_shift + cg0, a.in_relu);
#else
            // (from the LDS table once it is visible -- every chunk but the first, which is stored before the kernel's first barrier: as
            //  global loads inside the store phase their latency sat between the two barriers of every chunk)
            if (chunk_ > c_begin) v = affine8<T>(v, ldsT + cg0, ldsT + cin + cg0, a.in_relu);
            else v = affine8<T>(v, a.in_scale + cg0, a.in_shift + cg0, a.in_relu);
#endif
          }
          if (ABL(64)) { keep8<T>(v); continue; }
          lstore8<T>(ldsA + boff + a_l[j], v);
        }
      }
#pragma unroll
      for (int j = 0; j < BI; ++j) {
        const int it = tid + j * NTHREADS;
        if (ABL(64)) { keep8<T>(rb[j]); continue; }
        if (it < b_items) lstore8<T>(ldsB + boff + (size_t)it * EL, rb[j]);
      }
    }
  };

  // ---- WDMA: round r of the weight slab of chunk `chunk_` -> ring slot `ring`.  The slab is 36 (TAPS = 9) or 4 * KS (TAPS = 1) pieces of
  // 1 KB: piece p = half (p & 1) of the 2-KB row of (tap, slot) = ((p >> 1) / SLOTS, (p >> 1) % SLOTS), contiguous in the packed image
  // and in LDS; wave w moves pieces w, w + 8, ...  Everything but the lane's 16-byte offset is scalar.
  constexpr int NPIECES = TAPS * SLOTS * 2, NROUNDS = (NPIECES + 7) / 8;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned lds_ring = lds_addr_of(ldsB);
  auto dma_b = [&](int chunk_, int ring, int r) {
    if constexpr (WDMA) {
      const int p = wave_u + 8 * r;
      if (p < NPIECES) {
        const int run = p >> 1, tap = run / SLOTS, slot = run % SLOTS;
        const size_t off = ((size_t)(tap * (cin / EL) + chunk_ * SLOTS + slot) * a.cout_pad + nbase + (p & 1) * 64) * EL;
        lds_dma16(wp + off, (unsigned)lane * 16u, lds_ring + (unsigned)(ring * b_slab * (int)sizeof(T)) + (unsigned)p * 1024u);
      }
    }
  };

  int n0, y0, x0;
  tile_origin(bid, n0, y0, x0);
  if constexpr (TL) gather_pixels(n0, y0, x0, -a.halh_tl, -a.halw_tl);
  else gather_pixels(n0, y0, x0);
#ifdef SATCV_STAMP
  STAMP(k1);
#endif
  if constexpr (WDMA) {
#pragma unroll
    for (int r = 0; r < NROUNDS; ++r) dma_b(c_begin, c_begin % 3, r);
#pragma unroll
    for (int r = 0; r < NROUNDS; ++r) dma_b(c_end - c_begin > 1 ? c_begin + 1 : c_begin, (c_begin + 1) % 3, r);
  }
  if constexpr (TL && SK) {
    if (c_begin > 0) {                            // split-K of a tap-loop launch: this workgroup's first chunk may belong to a later tap
      const int tap0 = c_begin / a.cpt;
      gather_pixels(n0, y0, x0, (tap0 / a.kw) * a.dil - a.halh_tl, (tap0 % a.kw) * a.dil - a.halw_tl);
    }
  }
  int tap_g = TL ? c_begin / a.cpt : 0;             // (double-buffered tap-loop form) the tap the gather table a_p currently describes
  auto regather = [&](int chunk_) {
    if constexpr (TL && DB) {
      const int t = chunk_ / a.cpt;
      if (t != tap_g) {                             // wave-uniform; no vector-memory instruction inside
        tap_g = t;
        gather_pixels(n0, y0, x0, (t / a.kw) * a.dil - a.halh_tl, (t % a.kw) * a.dil - a.halw_tl);
      }
    }
  };
  load_regs(c_begin);
  store_lds(c_begin, DB ? (WDMA ? (c_begin & 1) * a_stage : (c_begin & 1) * stage_elems) : 0);
  if constexpr (DB) { if (c_end - c_begin > 1) { regather(c_begin + 1); load_regs(c_begin + 1); } }
  if constexpr (WDMA) dma_wait_all();
  __syncthreads();
#ifdef SATCV_STAMP
  STAMP(k2);
#endif
  {
    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
    // one K-chunk: software-pipelined fragment reads -- the LDS reads of step s+1 are issued before the MFMAs of step s (the
    // compiler otherwise waits for every read right before its MFMAs and the matrix pipe idles for the LDS latency)
    // rot: the tap step this wave starts a chunk with (the order of the steps is free).  In the 8-wave double-buffered tile the two
    // waves of a SIMD otherwise run the same stream in lockstep -- both in their fragment-read bursts, both at their MFMAs --; starting
    // waves 4-7 half a chunk later de-phases them (MI355X_MICROARCH.md, two waves per SIMD, item 9)
    // (rot is a wave-uniform run-time value: the tap offsets of a staggered wave are scalar additions instead of instruction immediates;
    //  two compile-time copies of the loop cost 200 bytes of scratch per lane)
    auto compute_chunk = [&](int boff, int boffB, auto&& side) {
      if constexpr (M16) {
        constexpr int K32 = KS / 2, STEPS = TAPS * K32;      // K = 32 per step: four slot planes of one tap
        FragT<T> af[2][2 * MT], bf[2][2 * NT];
        auto read_step = [&](int st, int buf) {
          const int tap = st / K32, k32 = st % K32;
          const int ky = TAPS == 1 ? 0 : tap / 3, kx = TAPS == 1 ? 0 : tap % 3;
          const int tap_off = (ky * pitch + kx) * EL;          // (dilation 1; compile-time)
#pragma unroll
          for (int m = 0; m < 2 * MT; ++m) af[buf][m] = lds_frag<T>(ldsA + boff + k32 * 4 * slot_stride + a_off[m] + tap_off);
#pragma unroll
          for (int n = 0; n < 2 * NT; ++n) bf[buf][n] = lds_frag<T>(ldsB + boffB + b_lane16 + ((tap * SLOTS + k32 * 4) * BN + n * 16) * EL);
        };
        read_step(0, 0);
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
          asm volatile("" ::: "memory");
          if (st + 1 < STEPS) read_step(st + 1, (st + 1) & 1);
          side(st);
          __builtin_amdgcn_sched_barrier(0);
          if (!ABL(2)) mma16_step<MT, NT>(acc, af[st & 1], bf[st & 1]);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) asm volatile("" : "+v"(acc[m][n]));
      } else {
      constexpr int STEPS = TAPS * KS;
      FragT<T> af[2][MT], bf[2][NT];
      auto read_step = [&](int st_, int buf) {
        // (compile-time step -> tap: the offsets below are instruction immediates.  A runtime rotation of the tap order for half of the
        //  waves -- tried against the issue imbalance between the older and the younger waves -- put 2-3 address instructions in front of
        //  every fragment read, ~150 per chunk, for no gain)
        const int st = st_;
        const int tap = st / KS, ks = st % KS;
        const int ky = TAPS == 1 ? 0 : tap / 3, kx = TAPS == 1 ? 0 : tap % 3;
        const int tap_off = (ky * dil * pitch + kx * dil) * EL;
        const int slot = (ks * 2 + hh) * SUB;
#pragma unroll
        for (int m = 0; m < MT; ++m) af[buf][m] = lds_frag<T>(ldsA + boff + (ABL(16) ? 0 : slot * slot_stride + a_off[m] + tap_off), slot_stride);
#pragma unroll
        for (int n = 0; n < NT; ++n) bf[buf][n] = lds_frag<T>(ldsB + boffB + ((tap * SLOTS + slot) * BN + (wn * NT + n) * 32 + r) * EL, BN * EL);
      };
      read_step(0, 0);
#pragma unroll
      for (int st = 0; st < STEPS; ++st) {
        asm volatile("" ::: "memory");              // IR-level fence: later steps' LDS reads must not be hoisted up here
        if (st + 1 < STEPS) read_step(st + 1, (st + 1) & 1);
        side(st);                                   // (double-buffered loop) this step's share of the staging of the next chunks
        __builtin_amdgcn_sched_barrier(0);          // keep the prefetch ahead of this step's MFMAs
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) { if (!ABL(2)) mma32<T>(acc[m][n], af[st & 1][m], bf[st & 1][n]); }
      }
      // pin this chunk's MFMAs before the barrier: nothing else orders them, and the compiler otherwise sinks all of them below
      // the LDS restaging of the next chunk, which keeps every fragment of the chunk live (seen with the scaled-fp8 form: spills)
      if constexpr (SUB == 2 || DB) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) asm volatile("" : "+v"(acc[m][n]));
      }
      }
    };
    if constexpr (DB) {
#ifdef SATCV_STAMP
      unsigned long long t0, t1, t2, t3, t4, s_wait = 0, s_store = 0, s_comp = 0, s_bar = 0;
#endif
      // units of one chunk's staging in program order: A items (store c+1, then re-issue for c+2), the scale / shift values (needed by
      // the A stores above, so re-issued after them), weight items; unit u runs in tap step u (the surplus in the last step)
      constexpr int NUNITS = AI + 1 + (WDMA ? NROUNDS : BI), STEPS_ = M16 ? TAPS * KS / 2 : TAPS * KS;
      for (int chunk = c_begin; chunk < c_end; ++chunk) {
        // A stages alternate; without WDMA the weight slab sits behind the A planes of the same stage, with it in ring slot chunk % 3
        const int cur = WDMA ? (chunk & 1) * a_stage : (chunk & 1) * stage_elems;
        const int oth = WDMA ? a_stage - cur : stage_elems - cur;
        const int curB = WDMA ? (chunk % 3) * b_slab : cur;
        const bool do_store = chunk + 1 < c_end, do_load = chunk + 2 < c_end;
        // (wave-uniform flags; the loads of the last two iterations re-read the last chunk instead of branching around vector-memory
        //  instructions inside the loop: see the note at the loaders)
        const int nxt2 = do_load ? chunk + 2 : c_end - 1;
        const ChunkSrc cs_ = chunk_src(nxt2);
        regather(nxt2);
#ifdef SATCV_STAMP
        STAMP(t0);
        STAMP(t1);
        STAMP(t2);
#endif
        auto side = [&](int st) {
#pragma unroll
          for (int u = 0; u < NUNITS; ++u) {
            // (M16: a step is 16 MFMAs of 16 cycles -- the units are spread evenly over the steps, in program order)
            if ((M16 ? (u * STEPS_) / NUNITS : (u < STEPS_ ? u : STEPS_ - 1)) != st) continue;
            if (u < AI) { if (do_store) store_a(chunk + 1, oth, u); load_a(cs_, u); }
#ifdef SATCV_RS_GLOBAL                                  // (A/B build: the scale / shift values by global loads, as before)
            else if (u == AI) load_p(cs_, false);
#else
            else if (u == AI) load_p(cs_, true);
#endif
            else if constexpr (WDMA) dma_b(nxt2, (chunk + 2) % 3, u - AI - 1);      // (slot (chunk + 2) % 3 was last read in iteration chunk - 1)
            else { if (do_store) store_b(oth, u - AI - 1); load_b(cs_, u - AI - 1); }
          }
        };
        compute_chunk(cur, curB, side);
#ifdef SATCV_STAMP
        STAMP(t3);
#endif
        if constexpr (WDMA) {
          // the slab of chunk + 1 (this wave's pieces, issued one iteration ago) must have landed before the barrier that lets every wave
          // read it: everything but what this iteration issued AFTER them -- AI activation loads and this wave's pieces of chunk + 2
          if (wave_u + 8 * (NROUNDS - 1) < NPIECES) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(AI + NROUNDS) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(AI + NROUNDS - 1) : "memory");
        }
        __syncthreads();
#ifdef SATCV_STAMP
        STAMP(t4);
        s_wait += t1 - t0; s_store += t2 - t1; s_comp += t3 - t2; s_bar += t4 - t3;
#endif
      }
      if constexpr (WDMA) { dma_wait_all(); __syncthreads(); }      // (the epilogue's staging tile aliases the ring: no piece may still be in flight)
#ifdef SATCV_STAMP
      if (blockIdx.x < 8 && lane == 0) {
        g_stamp[blockIdx.x][wave][0] = s_wait; g_stamp[blockIdx.x][wave][1] = s_store; g_stamp[blockIdx.x][wave][2] = s_comp; g_stamp[blockIdx.x][wave][3] = s_bar;
      }
#endif
    } else {
#ifdef SATCV_STAMP
    unsigned long long t0, t1, t2, t3, t4, t5, s_issue = 0, s_comp = 0, s_bar1 = 0, s_store = 0, s_bar2 = 0;
#endif
    for (int chunk = c_begin; chunk < c_end; ++chunk) {
      const bool more = chunk + 1 < c_end;
#ifdef SATCV_STAMP
      STAMP(t0);
#endif
      if (more) {
        if constexpr (TL) {
          if ((chunk + 1) % a.cpt == 0) {             // next chunk starts a new tap: its tile sits at another offset
            const int tap = (chunk + 1) / a.cpt;
            gather_pixels(n0, y0, x0, (tap / a.kw) * a.dil - a.halh_tl, (tap % a.kw) * a.dil - a.halw_tl);
          }
        }
        load_regs(chunk + 1);
      }
#ifdef SATCV_STAMP
      STAMP(t1);
#endif
      compute_chunk(0, 0, [](int) {});
#ifdef SATCV_STAMP
      STAMP(t2);
#endif
      __syncthreads();
#ifdef SATCV_STAMP
      STAMP(t3);
#endif
      if (more) {
        store_lds(chunk + 1);
#ifdef SATCV_STAMP
        STAMP(t4);
#endif
        __syncthreads();
      }
#ifdef SATCV_STAMP
      else { STAMP(t4); }
      STAMP(t5);
      s_issue += t1 - t0; s_comp += t2 - t1; s_bar1 += t3 - t2; s_store += t4 - t3; s_bar2 += t5 - t4;
#endif
    }
#ifdef SATCV_STAMP
    if (blockIdx.x < 8 && lane == 0) {
      g_stamp[blockIdx.x][wave][0] = s_issue; g_stamp[blockIdx.x][wave][1] = s_comp; g_stamp[blockIdx.x][wave][2] = s_bar1;
      g_stamp[blockIdx.x][wave][3] = s_store; g_stamp[blockIdx.x][wave][7] = s_bar2;
    }
#endif
    }
#ifdef SATCV_STAMP
    STAMP(k3);
#endif

    if constexpr (SK) {
      // partial tile -> kslab[split][pixel of the (n, h, w) grid][cout] (fp32, raw sums: bias / scale / ReLU / rounding / statistics are the
      // finish kernel's); lanes r = 32 consecutive channels = 128 contiguous bytes per accumulator row
      const size_t mtot = (size_t)a.n * a.h * a.w_;
      float* slab = a.kslab + (size_t)split * mtot * a.cout;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int q = (wm * MT + m) * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
          const int t = q / TW, cx = q % TW;
          const int k = (a.imgs == 1) ? 0 : t / a.rpi;
          const int nimg = n0 + k, y = y0 + (t - k * a.rpi), x = x0 + cx;
          if (!((k < a.imgs) && (nimg < a.n) && (y < a.h) && (x < a.w_))) continue;
          float* row = slab + ((size_t)(nimg * a.h + y) * a.w_ + x) * a.cout;
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const int cn = nbase + (wn * NT + n) * 32 + r;
            if (cn < a.cout) row[cn] = acc[m][n][i];
          }
        }
      return;
    } else
    igemm_epilogue<T, TW, WM, WN, MT, NT, ABL(1), !DYN, true, M16>(a, acc, n0, y0, x0, nbase, smem_raw);      // (the dilated-halo instantiations sit at their register cap)
#ifdef SATCV_STAMP
    {
      unsigned long long k4;
      STAMP(k4);
      if (blockIdx.x < 8 && lane == 0) {
        g_stamp[blockIdx.x][wave][4] = k1 - k0; g_stamp[blockIdx.x][wave][5] = k2 - k1; g_stamp[blockIdx.x][wave][6] = k4 - k3;
      }
    }
#endif
  }
}


// ------------------------------------------------------------------ split-K: workspace and finish kernel
// one fp32 scratch buffer per stream that has used split-K (grown on demand; allocated outside any capture: the first, eager run of a
// plan sizes it).  Returns nullptr when the allocation fails (the launch then runs unsplit).
#include <mutex>
static float* igemm_splitk_workspace(hipStream_t st, size_t bytes) {
  struct Slot { hipStream_t st; int dev; void* p; size_t n; bool captured; };
  static Slot slots[16];
  static int nslots = 0;
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  int dev = 0;
  (void)hipGetDevice(&dev);
  Slot* s = nullptr;
  for (int i = 0; i < nslots; ++i) if (slots[i].st == st && slots[i].dev == dev) s = &slots[i];
  if (!s) {
    if (nslots == 16) return nullptr;
    s = &slots[nslots++];
    *s = Slot{st, dev, nullptr, 0, false};
  }
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
  if (s->n < bytes) {
    if (capturing) return nullptr;      // never allocate inside a capture (callers run the launch list once on the capture stream first)
    // a buffer whose address sits in a captured graph stays alive when the stream later needs a larger one (a handful of growth events
    // per process: one per larger plan shape)
    if (s->p && !s->captured) { (void)hipStreamSynchronize(st); (void)hipFree(s->p); }
    s->p = nullptr; s->n = 0; s->captured = false;
    const size_t want = bytes + (bytes >> 2);
    if (hipMalloc(&s->p, want) != hipSuccess) { (void)hipGetLastError(); s->p = nullptr; return nullptr; }
    s->n = want;
  }
  if (capturing) s->captured = true;
  return reinterpret_cast<float*>(s->p);
}

// y = T(relu?(out_scale * sum_s slab[s] + bias)), statistics of the stored values: one thread per 8 channels of a pixel, a thread keeps its
// channel group over the grid-stride walk, per-block LDS sums, one pair of atomics per channel and block into the replica rows
template <typename T>
__global__ __launch_bounds__(256) void igemm_splitk_finish_kernel(const IgemmArgs a, long long mtot) {
  __shared__ float ssum[2][1024];
  const int G = a.cout / 8;
  const long long total = mtot * G;
  const bool st = a.stats != nullptr && a.cout <= 1024 && 256 % G == 0;
  if (st) { for (int i = threadIdx.x; i < 2 * 1024; i += blockDim.x) (&ssum[0][0])[i] = 0.f; __syncthreads(); }
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  const size_t slab = (size_t)mtot * a.cout;
  int g = 0;
  for (long long it = blockIdx.x * (long long)blockDim.x + threadIdx.x; it < total; it += (long long)gridDim.x * blockDim.x) {
    g = (int)(it % G);
    const long long p = it / G;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    for (int s = 0; s < a.ksplit; ++s) {
      const float4 u0 = *reinterpret_cast<const float4*>(a.kslab + s * slab + p * a.cout + g * 8);
      const float4 u1 = *reinterpret_cast<const float4*>(a.kslab + s * slab + p * a.cout + g * 8 + 4);
      v[0] += u0.x; v[1] += u0.y; v[2] += u0.z; v[3] += u0.w; v[4] += u1.x; v[5] += u1.y; v[6] += u1.z; v[7] += u1.w;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ch = (g * 8 + e) % a.cstat;
      float x = v[e] * (a.out_scale ? a.out_scale[ch] : 1.f) + (a.bias ? a.bias[ch] : 0.f);
      if (a.out_relu) x = fmaxf(x, 0.f);
      v[e] = x;
      const float r = round_to<T>(x);
      s1[e] += r; s2[e] += r * r;
    }
    store8<T>(reinterpret_cast<T*>(a.y) + p * a.ldy + g * 8, v);
  }
  if (st) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { atomicAdd(&ssum[0][g * 8 + e], s1[e]); atomicAdd(&ssum[1][g * 8 + e], s2[e]); }
    __syncthreads();
    for (int c = threadIdx.x; c < a.cout; c += blockDim.x) {
      satcv_stat_t* row = a.stats + (size_t)(blockIdx.x % SATCV_STAT_ROWS) * 2 * a.stats_ld;
      atomicAdd(row + c % a.cstat, (satcv_stat_t)ssum[0][c]);
      atomicAdd(row + a.stats_ld + c % a.cstat, (satcv_stat_t)ssum[1][c]);
    }
  }
}
template <typename T>
static int igemm_splitk_finish(const IgemmArgs& a, hipStream_t st) {
  if constexpr (sizeof(T) != 2) { return SATCV_ERR_UNSUPPORTED; }
  else {
    const long long mtot = (long long)a.n * a.h * a.w_;
    const int G = a.cout / 8;
    if (a.stats && !(a.cout <= 1024 && 256 % G == 0)) { satcv_set_error("igemm split-K: statistics need cout <= 1024 with 256 %% (cout / 8) == 0"); return SATCV_ERR_UNSUPPORTED; }
    long long grid = (mtot * G + 255) / 256;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(igemm_splitk_finish_kernel<T>, dim3((unsigned)grid), dim3(256), 0, st, a, mtot);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { satcv_set_error("igemm split-K finish launch: %s", hipGetErrorString(e)); return SATCV_ERR_HIP; }
    return SATCV_OK;
  }
}

// ------------------------------------------------------------------ host side
template <typename T, int TW, int WM, int WN, int MT, int NT, int KS, int TAPS, bool TL = false, bool DB = false, int WPS = 0, bool WDMA = false, bool SK = false, bool M16 = false>
static int fast_cfg(IgemmArgs& a, hipStream_t st, bool dry) {
  constexpr int EL = KTraits<T>::EL, SUB = KTraits<T>::SUB;
  constexpr int BM = WM * MT * 32, BN = WN * NT * 32, TH = BM / TW, KC = KS * 2 * SUB * EL, NTHREADS = WM * WN * 64;
  a.halh = TL ? 0 : a.dil * (a.kh - 1) / 2;
  a.halw = TL ? 0 : a.dil * (a.kw - 1) / 2;
  a.tiles_x = cdiv(a.w_, TW);
  if (a.h >= TH) { a.rpi = TH; a.imgs = 1; a.tiles_y = cdiv(a.h, TH); a.ngroups = a.n; }
  else { a.rpi = a.h; a.imgs = TH / a.h; a.tiles_y = 1; a.ngroups = cdiv(a.n, a.imgs); }
  a.seg = a.rpi + 2 * a.halh;
  a.rl = a.imgs * a.seg;
  a.cl = TW + 2 * a.halw;
  a.pitch = igemm_pitch(TW, a.cl, M16);
  a.n_tiles = cdiv(a.cout, BN);
  const int cin = a.c0 + a.c1;
  a.cpt = cin / KC;
  a.nchunks = TL ? a.kh * a.kw * a.cpt : a.cpt;
  a.halh_tl = a.dil * (a.kh - 1) / 2; a.halw_tl = a.dil * (a.kw - 1) / 2;
  if (TL && (a.mode_in != 0 || a.cpt < 1)) return SATCV_ERR_UNSUPPORTED;
  if (a.stride != 1 && (a.mode_in != 0 || a.mode_out != 0 || !(TL || TAPS == 1))) return SATCV_ERR_UNSUPPORTED;
  if (cin % KC != 0 || (a.x1 && a.c0 % KC != 0) || (a.mode_in == 1 && a.c0 % KC != 0)) return SATCV_ERR_UNSUPPORTED;
  // depth-to-space tiles: whole sub-pixel positions per tile, or whole tiles per sub-pixel position
  if (a.mode_out == 1 && !(a.cstat % BN == 0 || (BN % a.cstat == 0 && a.cstat % (16 / (int)sizeof(T)) == 0))) return SATCV_ERR_UNSUPPORTED;
  if (a.pool_y && (TH % a.pool_f != 0 || TW % a.pool_f != 0 || a.rpi % a.pool_f != 0)) return SATCV_ERR_UNSUPPORTED;    // pooling windows inside one tile
  if (a.cout_pad < a.n_tiles * BN) return SATCV_ERR_UNSUPPORTED;
  {
    constexpr int XMAXPIX = (TAPS == 1 && DB) ? BM : (TH + 2 * (TH / 4 > 1 ? TH / 4 : 1)) * (TW + 2);
    constexpr int AI = (XMAXPIX * (KC / EL) + NTHREADS - 1) / NTHREADS;
    if (a.rl * a.cl * (KC / EL) > AI * NTHREADS) return SATCV_ERR_UNSUPPORTED;     // register-staged items per thread
  }
  if (a.ldy % (16 / (int)sizeof(T)) != 0 || ((uintptr_t)a.y % 16) != 0) return SATCV_ERR_UNSUPPORTED;
  if (a.mode_out == 0 && a.cout % (16 / (int)sizeof(T)) != 0 && a.cout < a.ldy) { /* tail handled by scalar stores */ }
  size_t lds_stage = (((size_t)KC * a.rl * a.pitch + (size_t)TAPS * KC * BN) * sizeof(T) + (size_t)(KC / EL) * 128) * (DB ? 2 : 1);      // + slot padding (< 128 B per plane)
  if (WDMA) lds_stage = 2 * ((size_t)KC * a.rl * a.pitch * sizeof(T) + (size_t)(KC / EL) * 128) + 3 * (size_t)TAPS * KC * BN * sizeof(T);      // two A stages + a three-slot weight ring
  if (M16) {      // slot planes padded to multiples of 256 bytes (exact: the TW = 32 form with a 1024-channel table fills the 160 KB to the byte)
    const size_t plane_b = (((size_t)a.rl * a.pitch * EL * sizeof(T) + 255) / 256) * 256;
    const size_t a_st = (size_t)(KC / EL) * plane_b, b_sl = (size_t)TAPS * KC * BN * sizeof(T);
    lds_stage = WDMA ? 2 * a_st + 3 * b_sl : 2 * (a_st + b_sl);
  }
  if (WDMA && (((uintptr_t)a.w % 16) != 0 || a.cout_pad % 64 != 0)) return SATCV_ERR_UNSUPPORTED;
  if (DB && ((TL && TAPS != 1) || a.mode_in != 0)) return SATCV_ERR_UNSUPPORTED;
  size_t lds_out = (size_t)BM * (BN + 16 / sizeof(T)) * sizeof(T) + (size_t)(WM + 1) * 2 * BN * sizeof(float);
  if (a.bst_y) {
    // fused BatchNorm-backward reduce: only the epilogue's interior-tile path does it, so EVERY tile must be one; a second staging
    // tile holds the layer's raw outputs
    // (the dilated-halo instantiation DYN compiles only the general epilogue, which does not form the fused sums: refuse it here, before
    //  the dry-run answer, so that satcv_conv2d_igemm_pipelined never promises sums the launch will not produce)
    if (sizeof(T) != 2 || (TAPS == 9 && a.dil != 1 && !TL) || a.imgs != 1 || a.h % TH != 0 || a.w_ % TW != 0 || a.cout % BN != 0 || a.cout % 8 != 0 || a.bst_ld % 8 != 0 ||
        ((uintptr_t)a.bst_y % 16) != 0 || (a.bst_y1 && (a.bst_split % 8 != 0 || a.bst_ld1 % 8 != 0 || ((uintptr_t)a.bst_y1 % 16) != 0)))
      return SATCV_ERR_UNSUPPORTED;
    lds_out += (size_t)BM * (BN + 16 / sizeof(T)) * sizeof(T);
  }
  const size_t lds_tab = a.in_scale ? (size_t)2 * cin * sizeof(float) : 0;         // scale / shift table behind the stage(s)
  const size_t lds = lds_stage + lds_tab > lds_out ? lds_stage + lds_tab : lds_out;
  if (lds > 160 * 1024) return SATCV_ERR_UNSUPPORTED;
  const bool dyn = TAPS == 9 && a.dil != 1;
  long long blocks = (long long)a.ngroups * a.tiles_y * a.tiles_x * a.n_tiles;
  if (blocks <= 0 || blocks > 0x7fffffffLL) return SATCV_ERR_UNSUPPORTED;
  // ---- split-K (SK instantiations): a launch that leaves most of the chip idle (small maps: the 8 x 8 centre of the U-Net, a single
  // DeepLab tile) with a long K loop is cut 2 - 4 ways over K; fp32 slabs + an ordered sum in a finish kernel (deterministic).  Plain NHWC
  // stores only (no pooling, depth-to-space, accumulation, fused BatchNorm-backward sums); SATCV_SPLITK=0 turns it off.
  a.ksplit = 1; a.kslab = nullptr;
  int ks = 1;
  if constexpr (SK) {
    // Plain (halo-tile / 1x1) launches: OPT-IN (SATCV_SPLITK=1).  Measured on the U-Net's 8 x 8 data gradient (1024 -> 512, 128 workgroups):
    // 87.8 -> 72 us alone, nothing in the step (the weight-gradient stream fills the idle CUs anyway), and a K order that depends on the
    // launch's workgroup count would break the bit-identity of inference across batch splits.  Tap-loop launches (dilated / strided
    // convolutions: ASPP, the ResNet backbone of the build-defined DeepLab): ON (SATCV_SPLITK_TL=0 turns it off) -- a single 512 x 512
    // tile puts 16 workgroups on the chip for 576 chunks of its stage-4 convolutions, 318 us per launch whatever the batch.
    const int splitk = g_opt_splitk;            // (satcv_set_option("splitk", 1): the build-defined DeepLab's inference plans set it around their launches)
    static const int splitk_tl = [] { const char* e = getenv("SATCV_SPLITK_TL"); return e ? atoi(e) : 1; }();
    if ((TL ? splitk_tl : splitk) && !dyn && sizeof(T) == 2 && a.mode_out == 0 && !a.pool_y && !a.accumulate && !a.bst_y && a.cout % 8 == 0 && a.ldy % 8 == 0 &&
        a.stride == 1 && (!a.stats || (a.cout <= 1024 && 256 % (a.cout / 8) == 0))) {
      // (1 x 1 launches of a single tile: a 64-channel chunk costs a workgroup ~1.2 us of load latency whatever its MFMA count, so even
      //  16 chunks are worth cutting four ways -- 4 chunks per split + a ~7 us finish launch against 16 chunks in a row)
      //  (measured on DeepLab: +3.5 % at batch 1, -2.8 % at batch 4 when launches of 64+ workgroups were cut too: those stay whole)
      constexpr bool DEEP1X1 = TAPS == 1 && !TL && KS == 4;
      //  option value 2 (what the DeepLab plans set): plain launches of fewer than 64 workgroups only
      const bool small_only = !TL && (DEEP1X1 || splitk == 2);
      while (ks < (TL ? 16 : 4) && blocks * ks * 2 <= 256 && a.nchunks / (ks * 2) >= (DEEP1X1 ? 4 : 8) && !(small_only && blocks >= 64)) ks *= 2;
    }
    if (ks == 1) return SATCV_ERR_UNSUPPORTED;          // (the caller continues with the single-pass instantiation)
  }
  if (dry) return SATCV_OK;
  if (!dyn) {            // the kernel hard-codes these for the undilated case: keep the two derivations in lock step
    constexpr int CLc = TW + (TAPS == 9 ? 2 : 0);
    constexpr int PITCHc = igemm_pitch(TW, CLc, M16);
    if (a.cl != CLc || a.pitch != PITCHc) { satcv_set_error("igemm_fast: internal pitch mismatch (%d/%d vs %d/%d)", a.cl, a.pitch, CLc, PITCHc); return SATCV_ERR_INVALID; }
  }
  auto kern = igemm_fast_kernel<T, TW, WM, WN, MT, NT, KS, TAPS, false, TL, DB, WPS, WDMA, SK, M16>;
  if constexpr (TAPS == 9 && !DB && WPS == 0 && !SK) { if (dyn) kern = igemm_fast_kernel<T, TW, WM, WN, MT, NT, KS, TAPS, true>; }
  if ((DB || WPS || SK) && dyn) return SATCV_ERR_UNSUPPORTED;
  { const int rc = satcv_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds); if (rc) return rc; }
  if constexpr (SK) {
    float* ws = igemm_splitk_workspace(st, (size_t)ks * a.n * a.h * a.w_ * a.cout * sizeof(float));
    if (!ws) return SATCV_ERR_UNSUPPORTED;
    a.ksplit = ks; a.kslab = ws; blocks *= ks;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NTHREADS), lds, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { satcv_set_error("igemm_fast launch: %s", hipGetErrorString(e)); return SATCV_ERR_HIP; }
  if (a.ksplit > 1) return igemm_splitk_finish<T>(a, st);
  return SATCV_OK;
}

template <typename T, int TW, int TAPS>
static int fast_tw(IgemmArgs& a, hipStream_t st, bool dry) {
  const int cin = a.c0 + a.c1;
  // transposed conv: N = f*f sub-pixel positions x cstat channels.  Tiles spanning several positions read the input tile once
  // instead of once per position and write whole lines (SATCV_CONVT_WIDE=0: one position per tile, the earlier behaviour)
  static const bool convt_wide = !(getenv("SATCV_CONVT_WIDE") && atoi(getenv("SATCV_CONVT_WIDE")) == 0);
  const int nspace = (a.mode_out && !convt_wide) ? a.cstat : a.cout;
  if constexpr (TAPS == 9 && std::is_same<T, bf16>::value) {
    // deep 3x3 layers: 256-pixel x 128-channel tile, 8 waves, double-buffered stages (SATCV_DB=0 keeps the 128 x 128 tile)
    const int db_mode = g_opt_igemm_db;
    a.dbg = g_opt_igemm_sched;            // (experiment bits; none wired at present)
    // round 5: the persistent 16x16x32 kernel (conv_igemm_m16p.hip) takes the mid layers -- 64 ... 256 input channels, Cout % 128 == 0, several
    // tiles per workgroup -- under the same option as the one-tile 16x16x32 kernels below (launches that write statistics, or every launch of a
    // training plan)
    // (the per-launch tile_policy of a training plan raises the option's default 1 to 2; SATCV_M16=0 still wins)
    const int m16 = (g_opt_igemm_m16 > 0 && a.tile_policy > g_opt_igemm_m16) ? a.tile_policy : g_opt_igemm_m16;
    if (TW == 32 && a.dil == 1 && (m16 >= 2 || (m16 == 1 && (a.stats || a.bst_y)))) {
      const int rc = igemm_m16p_launch(a, SATCV_BF16, st, dry);
      if (rc != SATCV_ERR_UNSUPPORTED) return rc;
    }
    // (K = 9 x 64 is four chunks: the double-buffered tile's longer set-up and epilogue are not amortised -- 64 -> 128 channels at 128 x 128
    //  258 vs 235 us, at 64 x 64 67.6 vs 63.3 us on the 128 x 128 tile; from 128 input channels on it wins, profiles/r03_db_vs_single.txt)
    if (db_mode && a.dil == 1 && a.mode_in == 0 && a.mode_out == 0 && !a.pool_y && nspace % 128 == 0 && (cin >= 128 || db_mode >= 2)) {
      const long long tiles256 = (long long)cdiv(a.n * a.h * a.w_, 256) * (nspace / 128);
      // (a 512-pixel x 128-channel tile -- wave tile 64 x 128, 128 accumulator registers -- needs ~300 bytes of scratch per lane at the
      //  256-register cap of two waves per SIMD: not kept)
      if (db_mode >= 2 || tiles256 >= 192) {
        // weights by LDS-DMA into a three-slot ring (SATCV_WDMA=0: register-staged weights; the ring does not fit beside the halo tiles
        // of the 8-pixel-wide maps, which stay on the register path)
        static const int wdma = [] { const char* e = getenv("SATCV_WDMA"); return e ? atoi(e) : 1; }();
        // round 5: the 16x16x32 tile of conv_igemm_m16.hip -- 32-channel chunks, weights in tap-row units.  Maps at least 16 pixels wide,
        // Cin % 64 == 0 (the 8-pixel-wide maps keep the 32x32x16 register path).  It sums K in another grouping than the 32x32x16 tiles, and
        // which tile serves a shape depends on the launch's tile count: option igemm_m16 = 1 (default) therefore limits it to launches that
        // write statistics -- training-mode forward convolutions and data gradients with fused BatchNorm-backward sums --, so that inference
        // stays bit-identical across batch splits; a training plan raises the option to 2 (every eligible launch)
        // around its steps (engine.Plan.run_forward / run_backward); SATCV_M16=0 turns the tile off
        if (m16 >= 2 || (m16 == 1 && (a.stats || a.bst_y))) {
          const int rc = igemm_m16_launch(a, SATCV_BF16, st, dry);
          if (rc != SATCV_ERR_UNSUPPORTED) return rc;
        }
        if (wdma) {
          const int rc = fast_cfg<T, TW, 4, 2, 2, 2, 1, TAPS, false, true, 0, true>(a, st, dry);
          if (rc != SATCV_ERR_UNSUPPORTED) return rc;
        }
        const int rc = fast_cfg<T, TW, 4, 2, 2, 2, 1, TAPS, false, true>(a, st, dry);
        if (rc != SATCV_ERR_UNSUPPORTED) return rc;
      }
    }
    // 64 output channels with deep K (64 + 64 -> 64 at 128 x 128): a 512-pixel x 64-channel tile, 8 waves of 64 x 64 -- the 9-tap weight slab
    // (18 KB per 16-channel chunk) is fetched once per 512 pixels instead of once per 128 (SATCV_DB64=0: the 128 x 64 tile)
    static const int db64 = [] { const char* e = getenv("SATCV_DB64"); return e ? atoi(e) : 1; }();
    if (db_mode && db64 && a.dil == 1 && a.mode_in == 0 && a.mode_out == 0 && !a.pool_y && nspace % 64 == 0 && (nspace == 64 || db64 >= 2) && (cin >= 128 || db64 >= 3) &&
        (long long)cdiv(a.n * a.h * a.w_, 512) * (nspace / 64) >= 192) {
      const int rc = fast_cfg<T, TW, 8, 1, 2, 2, 1, TAPS, false, true>(a, st, dry);
      if (rc != SATCV_ERR_UNSUPPORTED) return rc;
    }
    // (a 256-pixel x 64-channel form of the same loop for the 64-channel layers measured equal to the 128 x 64 tile: not kept)
  }
  // 32-channel chunks only for 1x1 taps (the 9-tap weight slab of a 32-channel chunk would not leave
  // room for 2-3 workgroups per CU)
  if constexpr (KTraits<T>::SUB == 2) {
    // scaled fp8: a chunk is 64 channels, so the 9-tap weight slab of a 128-wide N tile (74 KB) would leave one workgroup per CU
    // (a 256x64 tile was measured 10-30 % slower than 128x64 here; the double-buffered loop on a 256 x 64 tile of 8 waves needs ~300 bytes of
    //  scratch at its 256-register cap: the K = 64 fragments are 8 registers each)
    if (nspace >= 64 && nspace % 64 == 0) return fast_cfg<T, TW, 2, 2, 2, 1, 1, TAPS>(a, st, dry);
    return fast_cfg<T, TW, 4, 1, 2, 1, 1, TAPS>(a, st, dry);
  }
  if constexpr (TAPS == 1 && KTraits<T>::SUB == 1) {
    const bool ks2 = (cin % 32 == 0) && (!a.x1 || a.c0 % 32 == 0) && (a.mode_in != 1 || a.c0 % 32 == 0);
    if (a.taploop) {
      if (!ks2) {            // 16-channel chunks (the 7x7 stem on a 4-band tile stored as 16 channels)
        if (nspace >= 64 && nspace % 64 == 0) return fast_cfg<T, TW, 2, 2, 2, 1, 1, TAPS, true>(a, st, dry);
        return fast_cfg<T, TW, 4, 1, 2, 1, 1, TAPS, true>(a, st, dry);
      }
      if (nspace >= 128 && nspace % 128 == 0) {
        if constexpr (std::is_same<T, bf16>::value) {      // few workgroups, hundreds of chunks (a single DeepLab tile): split-K
          const int rc = fast_cfg<T, TW, 2, 2, 2, 2, 2, TAPS, true, false, 0, false, true>(a, st, dry);
          if (rc != SATCV_ERR_UNSUPPORTED) return rc;
          // the double-buffered 256-pixel x 128-channel tile with 64-channel chunks (one barrier per chunk, the next chunk's loads in
          // flight under the MFMAs; the gather table is re-derived when the chunk stream crosses into the next tap): the dilated 3 x 3
          // convolutions of a DeepLab batch / the ASPP ran at 230-460 TFLOP/s on the single-buffered 128 x 128 tile with 32-channel
          // chunks (8 MFMAs per wave between two barrier pairs).  SATCV_DB_TL=0: the single-buffered tile.  (Different chunk size: the
          // two forms sum K in a different grouping -- the choice depends on the shape AND the tile count, like split-K above)
          static const int db_tl = [] { const char* e = getenv("SATCV_DB_TL"); return e ? atoi(e) : 1; }();
          if (db_tl && g_opt_igemm_db != 0 && cin % 64 == 0 && (!a.x1 || a.c0 % 64 == 0) &&
              ((long long)cdiv(a.n * a.h * a.w_, 256) * (nspace / 128) >= (db_tl > 1 ? db_tl : 96))) {
            const int rc2 = fast_cfg<T, TW, 4, 2, 2, 2, 4, TAPS, true, true>(a, st, dry);
            if (rc2 != SATCV_ERR_UNSUPPORTED) return rc2;
          }
        }
        return fast_cfg<T, TW, 2, 2, 2, 2, 2, TAPS, true>(a, st, dry);
      }
      if (nspace >= 64 && nspace % 64 == 0) return fast_cfg<T, TW, 2, 2, 2, 1, 2, TAPS, true>(a, st, dry);
      // (the 256x32 tile with 32-channel chunks needs 64-92 bytes of scratch in its tap-loop form: the 16-channel form does not)
      return fast_cfg<T, TW, 4, 1, 2, 1, 1, TAPS, true>(a, st, dry);
    }
    if (ks2) {
      if constexpr (std::is_same<T, bf16>::value) {
        // deep 1x1 / transposed convolutions (K = Cin >= 512; at 256 it measured slower): 64-channel chunks -- with 32 a chunk is 8 MFMAs per wave between two
        // barrier pairs; SATCV_DB=0 keeps the 32-channel form
        const bool ks4 = g_opt_igemm_db != 0 && (cin % 64 == 0) && cin >= 512 && (!a.x1 || a.c0 % 64 == 0) && (a.mode_in != 1 || a.c0 % 64 == 0);
        // ... and on the double-buffered 256-pixel x 128-channel tile, one barrier per chunk (SATCV_DB1X1=0: the single-buffered tile)
        static const bool db1 = !(getenv("SATCV_DB1X1") && atoi(getenv("SATCV_DB1X1")) == 0);
        // (forward transposed convs: 48.5 -> 40.2 and 50.3 -> 42.8 us at 8 x 8 and 16 x 16; their space-to-depth data gradients measured slower on it)
        // ... or when even the 128-pixel tiles leave CUs idle (a single DeepLab tile: 16-128 workgroups): the launch then lasts one
        // workgroup's K loop, whose chunk costs the single-buffered tile a full load latency between two barriers (~1.5 us for 0.3 us of
        // MFMAs) -- SATCV_DB1X1_SMALL=0 turns this case off.  Same 64-channel chunks and k order: the two tiles give identical bits.
        static const bool db1s = !(getenv("SATCV_DB1X1_SMALL") && atoi(getenv("SATCV_DB1X1_SMALL")) == 0);
        const long long mt256 = cdiv(a.n * a.h * a.w_, 256), mt128 = cdiv(a.n * a.h * a.w_, 128);
        if (ks4 && g_opt_splitk && nspace >= 128 && nspace % 128 == 0) {       // opt-in split-K of under-filled deep 1x1 launches
          const int rc = fast_cfg<T, TW, 2, 2, 2, 2, 4, TAPS, false, false, 2, false, true>(a, st, dry);
          if (rc != SATCV_ERR_UNSUPPORTED) return rc;
        }
        if (ks4 && db1 && a.mode_in == 0 && nspace >= 128 && nspace % 128 == 0 &&
            (mt256 * (nspace / 128) >= 192 || (db1s && mt128 * (nspace / 128) <= 256))) {
          const int rc = fast_cfg<T, TW, 4, 2, 2, 2, 4, TAPS, false, true>(a, st, dry);
          if (rc != SATCV_ERR_UNSUPPORTED) return rc;
        }
        if (ks4 && nspace >= 128 && nspace % 128 == 0) {
          const int rc = fast_cfg<T, TW, 2, 2, 2, 2, 4, TAPS, false, false, 2>(a, st, dry);
          if (rc != SATCV_ERR_UNSUPPORTED) return rc;
        }
        if (nspace >= 128 && nspace % 128 == 0) {       // under-filled 1x1 launches with a long K loop (single DeepLab tiles): split-K
          const int rc = fast_cfg<T, TW, 2, 2, 2, 2, 2, TAPS, false, false, 0, false, true>(a, st, dry);
          if (rc != SATCV_ERR_UNSUPPORTED) return rc;
        }
      }
      if (nspace >= 128 && nspace % 128 == 0) return fast_cfg<T, TW, 2, 2, 2, 2, 2, TAPS>(a, st, dry);
      if (nspace >= 64 && nspace % 64 == 0) return fast_cfg<T, TW, 2, 2, 2, 1, 2, TAPS>(a, st, dry);
      return fast_cfg<T, TW, 4, 1, 2, 1, 2, TAPS>(a, st, dry);
    }
  }
  if (nspace >= 128 && nspace % 128 == 0) {      // (128x64 tiles on these layers: 5-10 % slower)
    // (a 256x128 tile -- 4x2 MFMA tiles per wave, 128 accumulator registers, one workgroup per CU -- was measured slower:
    //  15.9 vs 15.4 ms/step)
    if constexpr (std::is_same<T, bf16>::value) {      // under-filled launches with a long K loop: split-K
      const int rc = fast_cfg<T, TW, 2, 2, 2, 2, 1, TAPS, false, false, 0, false, true>(a, st, dry);
      if (rc != SATCV_ERR_UNSUPPORTED) return rc;
    }
    return fast_cfg<T, TW, 2, 2, 2, 2, 1, TAPS>(a, st, dry);
  }
  // (N tiles wider than 128 -- tried on synthetic 96 / 192-channel outputs -- leave one workgroup per CU and were slower)
  // (32-channel chunks for the thin 3x3 layers were measured SLOWER: 144 vs 113 us on enc1, 425 vs 351 us on
  //  dec0.conv1 -- fewer resident workgroups outweigh the halved barrier count)
  // (8-wave variants of the thin tiles -- 16 accumulator registers, 6 waves/SIMD -- were measured 8-20 % SLOWER: occupancy is
  //  no longer what limits the bytes in flight)
  if (nspace >= 64 && nspace % 64 == 0) return fast_cfg<T, TW, 2, 2, 2, 1, 1, TAPS>(a, st, dry);
  // (round 2, measured on the 32-output-channel layers at 256 x 256 and not kept: a 512-pixel x 32-channel tile with 128 x 32 per wave,
  //  8-15 % slower; a persistent form of this tile -- resident grid, tile loop around the body, set-up once -- needs 56-160 bytes of
  //  scratch per lane at 3-4 waves per SIMD and is 8-25 % slower: occupancy is what these layers need)
  return fast_cfg<T, TW, 4, 1, 2, 1, 1, TAPS>(a, st, dry);
}

template <typename T, int TAPS>
static int fast_t(IgemmArgs& a, hipStream_t st, bool dry) {
  switch (igemm_pick_tw(a.w_)) {
    case 32: return fast_tw<T, 32, TAPS>(a, st, dry);
    case 16: return fast_tw<T, 16, TAPS>(a, st, dry);
    default: return fast_tw<T, 8, TAPS>(a, st, dry);
  }
}

int igemm_fast_launch(IgemmArgs& a, int dtype, hipStream_t st, bool dry) {
  const int taps = a.kh * a.kw;
  const bool k3 = a.kh == 3 && a.kw == 3;
  const bool odd_sq = a.kh == a.kw && (a.kh & 1) && a.kh <= 7;
  a.taploop = 0;
  if (a.stride < 1) a.stride = 1;
  if (taps > 1 && odd_sq && (a.dil > 1 || a.stride > 1 || !k3) && dtype != SATCV_FP8X) {
    // dilated 3x3: the halo-tile form when its staged tile fits the register budget (small dilation on narrow tiles), else the tap loop;
    // strided and larger odd kernels (ResNet stem / stage transitions): always the tap loop
    int rc = SATCV_ERR_UNSUPPORTED;
    if (k3 && a.stride == 1) {
      if (dtype == SATCV_BF16) rc = fast_t<bf16, 9>(a, st, true);
      else if (dtype == SATCV_F32) rc = fast_t<float, 9>(a, st, true);
      else if (dtype == SATCV_FP8) rc = fast_t<fp8, 9>(a, st, true);
    }
    if (rc != SATCV_OK) {
      a.taploop = 1;
      if (dtype == SATCV_BF16) return fast_t<bf16, 1>(a, st, dry);
      if (dtype == SATCV_F32) return fast_t<float, 1>(a, st, dry);
      return fast_t<fp8, 1>(a, st, dry);
    }
  }
  if (!(taps == 1 || k3) || (a.stride != 1 && taps != 1)) return SATCV_ERR_UNSUPPORTED;
  if (dtype == SATCV_BF16) return taps == 1 ? fast_t<bf16, 1>(a, st, dry) : fast_t<bf16, 9>(a, st, dry);
  if (dtype == SATCV_F32) return taps == 1 ? fast_t<float, 1>(a, st, dry) : fast_t<float, 9>(a, st, dry);
  if (dtype == SATCV_FP8) return taps == 1 ? fast_t<fp8, 1>(a, st, dry) : fast_t<fp8, 9>(a, st, dry);
  if (dtype == SATCV_FP8X) return taps == 1 ? fast_t<fp8s, 1>(a, st, dry) : fast_t<fp8s, 9>(a, st, dry);
  return SATCV_ERR_UNSUPPORTED;
}
