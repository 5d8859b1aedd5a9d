// Round 5: the deep 3x3 convolution tile rebuilt around v_mfma_f32_16x16x32_bf16 (forward and data gradient of Conv2D at
// /root/reference/utils/model_tools.py:178, 312, 315 for Cin >= 128: the 64 x 64 ... 16 x 16 levels of the U-Net).
//
// Why a kernel of its own.  The 32x32x16 tile of conv_igemm_fast.hip sums 16 input channels per K step; the 16x16x32 instruction sums 32, so
// a K step wants FOUR 8-channel slot planes of one tap, i.e. 32-channel chunks -- whose 9-tap weight slab for 128 output channels is 72 KB:
// two of them (double buffering) plus two activation stages do not fit the 160 KB of LDS, and the three-slot ring of the 32x32x16 tile
// (221 KB) fits even less.  Here the weights move in TAP-ROW units (3 taps x 32 channels x 128 output channels = 24 KB) and the K loop runs
// in INTERVALS of two units between two barriers:
//      interval 3p    : chunk 2p   rows 0, 1          (activation stage 0)
//      interval 3p + 1: chunk 2p   row 2, chunk 2p + 1 row 0      (stages 0 and 1)
//      interval 3p + 2: chunk 2p + 1 rows 1, 2        (stage 1)
// Two weight buffers of two units (96 KB) alternate per interval: while interval k is multiplied out of buffer k & 1, every wave moves
// its six 1-KB pieces of interval k + 1 into the other buffer with global_load_lds_dwordx4 (no staging registers, no ds_write).  The
// activations (the operand that needs the producing layer's BatchNorm + ReLU and the zero padding) keep the register path, a 32-channel
// chunk at a time: loaded one interval before they are stored, stored into the stage whose last reader finished an interval earlier.
// One `s_waitcnt vmcnt(0)` + ONE barrier per interval = per 96 MFMAs of a wave (the 16-channel loop: one per 36 of twice the size --
// 25 % fewer barriers per FLOP, half the staging instructions per FLOP).  LDS: 2 x 22 KB + 96 KB + the scale / shift table = 148 KB.
//
// Fragment reads: lane l feeds row / column l & 15 and the 8 channels of slot l >> 4.  Slot planes are multiples of 256 bytes, so the
// four 16-lane groups of a ds_read_b128 ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS table) take 16 distinct 16-byte bank slots
// each; the halo tile's row pitch equals its width (a lane quarter reads 16 consecutive pixels of one row).  Staged items are dealt to
// lanes so that the 8 lanes of a ds_write_b128 group hold 8 CONSECUTIVE pixels (distinct banks although all four slot planes start on
// the same bank) while a wave-instruction's global loads still cover 16 pixels x 64 contiguous bytes.
// Accumulators: 4 x 4 blocks of 16 x 16 per wave in 64 registers (AccMap<true>); epilogue = the shared one (statistics, fused
// BatchNorm-backward sums, LDS-staged 16-byte stores).
#include "igemm_common.hpp"
#include <cstdlib>

// compile-time ablation switches for profiling builds (tools/scripts/build_m16_variants.sh; the product build has none set): 1 the fragment
// reads of steps 1-5 of an interval, 2 the staging (stores, loads, LDS-DMA), 4 the wait + barrier that ends an interval, 8 the MFMAs,
// 16 the epilogue, 32 the LDS-DMA pieces only, 64 the activation loads only, 128 the activation stores only -- timing only, wrong results
#ifndef SATCV_M16_ABL
#define SATCV_M16_ABL 0
#endif
#define MABL(bit) ((SATCV_M16_ABL & (bit)) != 0)

template <int TW>
struct M16Geom {
  static constexpr int BM = 256, BN = 128, TH = BM / TW, CL = TW + 2, RL = TH + 2;
  static constexpr int NPIX = ((RL * CL + 7) / 8) * 8;                       // halo pixels, padded to whole octets
  static constexpr int PLANE_E = (((NPIX * 16 + 255) / 256) * 256) / 2;      // elements per slot plane (a multiple of 256 bytes)
  static constexpr int A_STAGE_E = 4 * PLANE_E;
  static constexpr int UNIT_E = 3 * 4 * BN * 8;                              // a tap row of one 32-channel chunk: [3 taps][4 slots][128][8]
  static constexpr int WBUF_E = 2 * UNIT_E;
  static constexpr int A_ITEMS = NPIX * 4;
  static constexpr int AI = (A_ITEMS + 511) / 512;
  static constexpr size_t LDS_OPERANDS = (size_t)(2 * A_STAGE_E + 2 * WBUF_E) * 2;
};

template <int TW>
__global__ __launch_bounds__(512, 1) void igemm_m16sym_kernel(const IgemmArgs a) {
  using T = bf16;
  using G = M16Geom<TW>;
  constexpr int NTHREADS = 512, WM = 4, WN = 2, MT = 2, NT = 2, BN = G::BN, TH = G::TH, CL = G::CL, AI = G::AI;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* ldsA = reinterpret_cast<T*>(smem_raw);
  T* ldsW = ldsA + 2 * G::A_STAGE_E;
  float* ldsT = reinterpret_cast<float*>(ldsW + 2 * G::WBUF_E);              // [2][cin] scale, shift of the fused input BatchNorm
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN, g4 = lane >> 4, l16 = lane & 15;
  const int cin = a.c0 + a.c1, nch = cin / 32, nint = (3 * nch) / 2;

  // XCD-aware tile id (blocks b and b + 8 share an XCD): contiguous tile ranges per XCD, neighbouring halos side by side
  int bid;
  {
    const int G_ = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q = G_ >> 3, rem = G_ & 7;
    bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (orig >> 3);
  }
  const int nbase = (bid % a.n_tiles) * BN;
  int n0, y0, x0;
  {
    int mt = bid / a.n_tiles;
    const int tx = mt % a.tiles_x; mt /= a.tiles_x;
    const int ty = mt % a.tiles_y;
    n0 = mt / a.tiles_y; y0 = ty * TH; x0 = tx * TW;
  }

  // ---- staged activation items of this thread: item it = (pixel octet it >> 5, lane-in-octet it & 7, slot ((it >> 3) + it) & 3).  512 is a
  // multiple of 32, so the slot is a constant of the thread
  const int slot_t = ((tid >> 3) + tid) & 3;
  int a_l[AI], a_p[AI];
#pragma unroll
  for (int j = 0; j < AI; ++j) {
    const int it = tid + j * NTHREADS;
    const int pix = ((it >> 5) << 3) | (it & 7);
    const int L = pix / CL, c = pix - L * CL;
    const int y = y0 + L - 1, x = x0 + c - 1;
    a_l[j] = (it < G::A_ITEMS) ? slot_t * G::PLANE_E + pix * 8 : -1;
    a_p[j] = (it < G::A_ITEMS && L < G::RL && y >= 0 && y < a.h && x >= 0 && x < a.w_) ? (n0 * a.h + y) * a.w_ + x : -1;
  }
  if (a.in_scale) {
    for (int i = tid; i < cin; i += NTHREADS) { ldsT[i] = a.in_scale[i]; ldsT[cin + i] = a.in_shift[i]; }
  }
  // source of a 32-channel chunk (two-source input = the never materialised concat([skip, up]) of decoder_block)
  auto chunk_ptr = [&](int x, int& cs) -> const T* {
    const int cg0 = x * 32;
    if (cg0 < a.c0) { cs = a.c0; return reinterpret_cast<const T*>(a.x0) + cg0 + slot_t * 8; }
    cs = a.c1; return reinterpret_cast<const T*>(a.x1) + (cg0 - a.c0) + slot_t * 8;
  };
  Raw8<T> ra[AI];
  float4 rs[4];
  // (no vector-memory instruction inside a lane-dependent branch: items outside the image load pixel 0 and are zeroed when stored)
  auto load_a = [&](int x, int j) {
    int cs; const T* src = chunk_ptr(x, cs);
    const int p = a_p[j] < 0 ? 0 : a_p[j];
    ra[j] = gload8<T>(src + (size_t)p * cs);
  };
  auto load_rs = [&](int x) {
    if (a.in_scale) {
      const float4* tp = reinterpret_cast<const float4*>(ldsT + x * 32 + slot_t * 8);
      const float4* hp = reinterpret_cast<const float4*>(ldsT + cin + x * 32 + slot_t * 8);
      rs[0] = tp[0]; rs[1] = tp[1]; rs[2] = hp[0]; rs[3] = hp[1];
    }
  };
  auto store_a = [&](int stage, int j) {
    Raw8<T> v = ra[j];
    if (a.in_scale) v = affine8r<T>(v, rs, a.in_relu);
    v = select8<T>(a_p[j] >= 0, v);
    if (a_l[j] >= 0) lstore8<T>(ldsA + stage * G::A_STAGE_E + a_l[j], v);
  };
  // ---- weights: piece p = wave + 8 r (r = 0 .. 2) of unit uu = 3 chunk + tap row: half (p & 1) of the 2-KB row of (tap-in-row, slot) =
  // ((p >> 1) >> 2, (p >> 1) & 3); contiguous in the packed image [tap][cin / 8][cout_pad][8] and in LDS
  const T* wp = reinterpret_cast<const T*>(a.w);
  const unsigned lds_w = lds_addr_of(ldsW);
  auto dma_piece = [&](int uu, int r, int buf, int ub) {
    const int chunk = uu / 3, ky = uu - chunk * 3;
    const int p = wave + 8 * r, run = p >> 1;
    const int tap = ky * 3 + (run >> 2), slot = run & 3;
    const size_t off = ((size_t)(tap * (cin / 8) + chunk * 4 + slot) * a.cout_pad + nbase + (p & 1) * 64) * 8;
    lds_dma16(wp + off, (unsigned)lane * 16u, lds_w + (unsigned)((buf * G::WBUF_E + ub * G::UNIT_E) * 2) + (unsigned)p * 1024u);
  };

  // fragment addresses (elements): A = pixel (wm * 64 + 16 m + l16) of the tile in the slot plane of this lane quarter; B = column
  int a_off[2 * MT];
#pragma unroll
  for (int m = 0; m < 2 * MT; ++m) {
    const int q = wm * MT * 32 + m * 16 + l16;
    a_off[m] = g4 * G::PLANE_E + ((q / TW) * CL + q % TW) * 8;
  }
  const int b_lane = (g4 * BN + wn * NT * 32 + l16) * 8;

  // ---- prologue: interval 0's weights, chunk 0 staged, chunk 1 in registers
#pragma unroll
  for (int s = 0; s < 6; ++s) dma_piece(s / 3, s % 3, 0, s / 3);
#pragma unroll
  for (int j = 0; j < AI; ++j) load_a(0, j);
  __syncthreads();                                   // the scale / shift table is visible
  load_rs(0);
#pragma unroll
  for (int j = 0; j < AI; ++j) store_a(0, j);
#pragma unroll
  for (int j = 0; j < AI; ++j) load_a(1, j);
  dma_wait_all();
  __syncthreads();

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

  // one interval: six K = 32 steps (two units x three taps of a row).  P = k % 3 fixes the tap rows and stages at compile time, so every
  // tap offset of the fragment reads is an instruction immediate; c = the even chunk of the pair this period works on.
  // LATE: the two waves of a SIMD (w and w + 4) run the same stream and leave a barrier together -- with the staging in the same steps
  // both sit in their store / load / DMA issue at the same time and the matrix pipe idles (MI355X_MICROARCH.md, two waves per SIMD, item 9).
  // Waves 4-7 therefore do their staging in steps 3-5, waves 0-3 in steps 0-2: one wave's serial part runs under its partner's MFMAs.
  auto interval = [&](auto PC, auto LATEC, int k, int c) {
    constexpr int P = decltype(PC)::value;
    constexpr bool LATE = decltype(LATEC)::value;
    const int wb = (k & 1) * G::WBUF_E;
    const bool more = k + 1 < nint;
    FragT<T> af[2][2 * MT], bf[2][2 * NT];
    auto read_step = [&](int s, int buf) {
      const int ub = s / 3, kx = s % 3;
      const int ky = P == 0 ? ub : (P == 1 ? (ub == 0 ? 2 : 0) : ub + 1);
      const int stage = P == 0 ? 0 : (P == 1 ? ub : 1);
      const int aoff = stage * G::A_STAGE_E + (ky * CL + kx) * 8;
#pragma unroll
      for (int m = 0; m < 2 * MT; ++m) af[buf][m] = lds_frag<T>(ldsA + a_off[m] + aoff);
#pragma unroll
      for (int n = 0; n < 2 * NT; ++n) bf[buf][n] = lds_frag<T>(ldsW + wb + b_lane + ub * G::UNIT_E + (kx * 4 * BN + n * 16) * 8);
    };
    read_step(0, 0);
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      asm volatile("" ::: "memory");                  // IR-level fence: later steps' LDS reads stay behind this point
      if (s + 1 < 6 && !MABL(1)) read_step(s + 1, (s + 1) & 1);
      // this step's share of the staging (slot t = 0 .. 2).  Everything an interval consumes from registers was loaded in the interval
      // before and is complete (the vmcnt(0) at its end), so ALL stores come first, before this wave issues any vector-memory operation
      // of the interval: hipcc does not see the LDS-DMA pieces in its wait-count bookkeeping, and a store placed behind younger loads
      // and pieces made it wait for those.  Then per slot one activation load (the registers just freed) and two weight pieces
      const int t = MABL(2) ? -1 : (LATE ? s - 3 : s);
      if (t == 0) {
        if (P == 0) load_rs(c + 1);
        if (P == 2) load_rs(c + 2 < nch ? c + 2 : c);
#pragma unroll
        for (int j = 0; j < AI; ++j) {
          if (MABL(128)) { asm volatile("" :: "v"(ra[j].q[0].x), "v"(ra[j].q[0].y), "v"(ra[j].q[0].z), "v"(ra[j].q[0].w)); continue; }
          if (P == 0) store_a(1, j);
          if (P == 2) { if (c + 2 < nch) store_a(0, j); }
        }
      }
      if (t >= 0 && t < 3) {
        if (t < AI && !MABL(64)) {
          if (P == 1) load_a(c + 2 < nch ? c + 2 : nch - 1, t);
          if (P == 2) load_a(c + 3 < nch ? c + 3 : nch - 1, t);
        }
        if (more && !MABL(32)) {
          dma_piece(2 * (k + 1) + (2 * t) / 3, (2 * t) % 3, (k + 1) & 1, (2 * t) / 3);
          dma_piece(2 * (k + 1) + (2 * t + 1) / 3, (2 * t + 1) % 3, (k + 1) & 1, (2 * t + 1) / 3);
        }
      }
      __builtin_amdgcn_sched_barrier(0);              // keep the prefetch and the staging ahead of this step's MFMAs
      if (!MABL(8)) mma16_step<MT, NT>(acc, af[MABL(1) ? 0 : (s & 1)], bf[MABL(1) ? 0 : (s & 1)]);
      else { for (int m = 0; m < 2 * MT; ++m) asm volatile("" :: "v"(af[s & 1][m].v)); for (int n = 0; n < 2 * NT; ++n) asm volatile("" :: "v"(bf[s & 1][n].v)); }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) asm volatile("" : "+v"(acc[m][n]));      // the MFMAs stay in front of the barrier
    if (!MABL(4)) {
      dma_wait_all();                                 // next interval's weights (and the activation loads issued above) have landed
      __syncthreads();
    }
  };
  static_assert(AI <= 3, "one activation load per staging slot");
  auto k_loop = [&](auto LATEC) {
    for (int k = 0, c = 0; k < nint; k += 3, c += 2) {
      interval(std::integral_constant<int, 0>{}, LATEC, k, c);
      interval(std::integral_constant<int, 1>{}, LATEC, k + 1, c);
      interval(std::integral_constant<int, 2>{}, LATEC, k + 2, c);
    }
  };
#ifdef SATCV_M16_NOSTAGGER
  k_loop(std::false_type{});
#else
  if (wave >= 4) k_loop(std::true_type{});            // (wave-uniform; both copies pass the same barriers)
  else k_loop(std::false_type{});
#endif
  if (MABL(4)) { dma_wait_all(); __syncthreads(); }
  if (MABL(16)) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) asm volatile("" :: "v"(acc[m][n]));
    return;
  }
  igemm_epilogue<T, TW, WM, WN, MT, NT, false, true, true, true>(a, acc, n0, y0, x0, nbase, smem_raw);
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same tile with WAVE ROLES (the default; SATCV_M16_WS=0 runs the symmetric kernel above).  Ablation builds of the symmetric kernel
// (profiles/r05_ablation_m16.txt, r05_ablation_m16_staging.txt) put numbers on what keeps its matrix pipe ~50 % busy: 1024 -> 512 channels
// at 16 x 16 takes 143-148 us; without the staging (activation loads, BatchNorm affine + stores, LDS-DMA issue) 110, without the fragment
// reads 134, without the interval barrier 138, MFMAs alone 88.  The staging is in-order instruction issue in the SAME waves that issue the
// MFMAs: under v_mfma_f32_16x16x32 an MFMA holds the SIMD's vector issue for 8 of its 16 cycles, so ~100 vector instructions of affine /
// select / address work per chunk and wave are paid almost in full.  Here 12 waves: waves 0-7 issue nothing but fragment reads and MFMAs
// (a 64 x 64 output each, as before); waves 8-11 (one per SIMD) do ALL the staging -- per interval: store the activation chunk loaded an
// interval earlier (BatchNorm + ReLU in registers), reload the registers, issue the 48 weight pieces of the next interval, wait for them,
// barrier -- and end before the epilogue (a barrier counts only the waves still alive).  Three waves per SIMD: 168 registers each.
template <int TW>
__global__ __launch_bounds__(768, 1) void igemm_m16_kernel(const IgemmArgs a) {
  using T = bf16;
  using G = M16Geom<TW>;
  constexpr int WM = 4, WN = 2, MT = 2, NT = 2, BN = G::BN, TH = G::TH, CL = G::CL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* ldsA = reinterpret_cast<T*>(smem_raw);
  T* ldsW = ldsA + 2 * G::A_STAGE_E;
  float* ldsT = reinterpret_cast<float*>(ldsW + 2 * G::WBUF_E);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cin = a.c0 + a.c1, nch = cin / 32, nint = (3 * nch) / 2;
  int bid;
  {
    const int G_ = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q = G_ >> 3, rem = G_ & 7;
    bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (orig >> 3);
  }
  const int nbase = (bid % a.n_tiles) * BN;
  int n0, y0, x0;
  {
    int mt = bid / a.n_tiles;
    const int tx = mt % a.tiles_x; mt /= a.tiles_x;
    const int ty = mt % a.tiles_y;
    n0 = mt / a.tiles_y; y0 = ty * TH; x0 = tx * TW;
  }
  if (a.in_scale) {
    for (int i = tid; i < cin; i += 768) { ldsT[i] = a.in_scale[i]; ldsT[cin + i] = a.in_shift[i]; }
  }

  if (wave >= 8) {
    // ================================================================ staging waves (256 threads)
    constexpr int NS = 256, AI = (G::A_ITEMS + NS - 1) / NS;
    const int sid = tid - 512, ws = wave - 8;
#ifndef SATCV_M16_PRIO
#define SATCV_M16_PRIO 2
#endif
    // the staging wave is the youngest of its SIMD and would get the vector-issue slots the two matrix waves leave (priority, then age:
    // MI355X_MICROARCH.md, two waves per SIMD, item 2); with the BatchNorm affine it then trails the matrix waves.  Raised priority: its
    // ~250 vector instructions per interval go first, the matrix waves' MFMAs (8 issue cycles of 16) fill in behind them
    if (SATCV_M16_PRIO) __builtin_amdgcn_s_setprio(SATCV_M16_PRIO);
    const int slot_t = ((sid >> 3) + sid) & 3;            // (item -> (pixel, slot) as in the symmetric kernel; 256 is a multiple of 32)
    int a_l[AI], a_p[AI];
#pragma unroll
    for (int j = 0; j < AI; ++j) {
      const int it = sid + j * NS;
      const int pix = ((it >> 5) << 3) | (it & 7);
      const int L = pix / CL, c = pix - L * CL;
      const int y = y0 + L - 1, x = x0 + c - 1;
      a_l[j] = (it < G::A_ITEMS) ? slot_t * G::PLANE_E + pix * 8 : -1;
      a_p[j] = (it < G::A_ITEMS && L < G::RL && y >= 0 && y < a.h && x >= 0 && x < a.w_) ? (n0 * a.h + y) * a.w_ + x : -1;
    }
    auto chunk_ptr = [&](int x, int& cs) -> const T* {
      const int cg0 = x * 32;
      if (cg0 < a.c0) { cs = a.c0; return reinterpret_cast<const T*>(a.x0) + cg0 + slot_t * 8; }
      cs = a.c1; return reinterpret_cast<const T*>(a.x1) + (cg0 - a.c0) + slot_t * 8;
    };
    // two register sets, one per activation stage: a chunk is loaded THREE intervals before it is stored (set s holds the chunk that goes to
    // stage s next), so no load latency is ever waited for
    Raw8<T> ra[2][AI];
    float4 rs[4];
    auto load_a = [&](int x, int set) {
      int cs; const T* src = chunk_ptr(x, cs);
#pragma unroll
      for (int j = 0; j < AI; ++j) {
        const int p = a_p[j] < 0 ? 0 : a_p[j];          // (outside the image: pixel 0, zeroed when stored -- no load inside a lane-dependent branch)
        ra[set][j] = gload8<T>(src + (size_t)p * cs);
      }
    };
    // hipcc does not see the LDS-DMA pieces (inline asm) in its wait-count bookkeeping: a store of registers whose loads it still counts as
    // pending, placed behind freshly issued pieces, made it wait for the pieces too.  Touching the set first, while nothing younger is in
    // flight, retires those loads in its books
    auto touch = [&](int set) {
#pragma unroll
      for (int j = 0; j < AI; ++j) asm volatile("" : "+v"(ra[set][j].q[0].x), "+v"(ra[set][j].q[0].y), "+v"(ra[set][j].q[0].z), "+v"(ra[set][j].q[0].w));
    };
    auto store_a = [&](int x, int stage) {
      if (a.in_scale) {
        const float4* tp = reinterpret_cast<const float4*>(ldsT + x * 32 + slot_t * 8);
        const float4* hp = reinterpret_cast<const float4*>(ldsT + cin + x * 32 + slot_t * 8);
        rs[0] = tp[0]; rs[1] = tp[1]; rs[2] = hp[0]; rs[3] = hp[1];
      }
#pragma unroll
      for (int j = 0; j < AI; ++j) {
        Raw8<T> v = ra[stage][j];
        if (a.in_scale) v = affine8r<T>(v, rs, a.in_relu);
        v = select8<T>(a_p[j] >= 0, v);
        if (a_l[j] >= 0) lstore8<T>(ldsA + stage * G::A_STAGE_E + a_l[j], v);
      }
    };
    const T* wp = reinterpret_cast<const T*>(a.w);
    const unsigned lds_w = lds_addr_of(ldsW);
    // the 48 pieces of interval kk (units 2 kk, 2 kk + 1) into buffer kk & 1: this wave's pieces p = ws + 4 r of each unit
    auto dma_interval = [&](int kk) {
#pragma unroll
      for (int ub = 0; ub < 2; ++ub) {
        const int uu = 2 * kk + ub, chunk = uu / 3, ky = uu - chunk * 3;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          const int p = ws + 4 * r, run = p >> 1;
          const int tap = ky * 3 + (run >> 2), slot = run & 3;
          const size_t off = ((size_t)(tap * (cin / 8) + chunk * 4 + slot) * a.cout_pad + nbase + (p & 1) * 64) * 8;
          lds_dma16(wp + off, (unsigned)lane * 16u, lds_w + (unsigned)(((kk & 1) * G::WBUF_E + ub * G::UNIT_E) * 2) + (unsigned)p * 1024u);
        }
      }
    };
    auto clampc = [&](int x) { return x < nch ? x : nch - 1; };
    dma_interval(0);
    load_a(0, 0);
    __syncthreads();                                   // (1) the scale / shift table is visible
    store_a(0, 0);
    load_a(1, 1);
    load_a(clampc(2), 0);
    dma_wait_all();
    __syncthreads();                                   // (2) interval 0 may start
    // per interval: the weight pieces of the next interval FIRST (they land while this wave does its vector work), then the store of the set
    // whose stage fell free at the last barrier, the reload of that set, and a counted wait that leaves only the reload in flight
    for (int k = 0, c = 0; k < nint; k += 3, c += 2) {
      // k % 3 == 0: chunk c + 1 (set 1) -> stage 1 (last read two intervals ago); set 1 <- chunk c + 3
      touch(1);
      dma_interval(k + 1);
      store_a(c + 1, 1);
      load_a(clampc(c + 3), 1);
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(AI) : "memory");
      __syncthreads();
      // k % 3 == 1: weights only (the counter runs in order: the reload above retires in front of the pieces)
      dma_interval(k + 2);
      dma_wait_all();
      __syncthreads();
      // k % 3 == 2: chunk c + 2 (set 0) -> stage 0 (its last reader was the interval that just ended); set 0 <- chunk c + 4
      touch(0);
      if (k + 3 < nint) dma_interval(k + 3);
      if (c + 2 < nch) store_a(c + 2, 0);
      load_a(clampc(c + 4), 0);
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(AI) : "memory");
      __syncthreads();
    }
    return;                                            // (the epilogue's barriers count the surviving waves only)
  }

  // ================================================================ matrix waves (512 threads)
  const int wm = wave / WN, wn = wave % WN, g4 = lane >> 4, l16 = lane & 15;
  int a_off[2 * MT];
#pragma unroll
  for (int m = 0; m < 2 * MT; ++m) {
    const int q = wm * MT * 32 + m * 16 + l16;
    a_off[m] = g4 * G::PLANE_E + ((q / TW) * CL + q % TW) * 8;
  }
  const int b_lane = (g4 * BN + wn * NT * 32 + l16) * 8;
  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
  __syncthreads();                                     // (1)
  __syncthreads();                                     // (2)
  auto interval = [&](auto PC, int k) {
    constexpr int P = decltype(PC)::value;
    const int wb = (k & 1) * G::WBUF_E;
    FragT<T> af[2][2 * MT], bf[2][2 * NT];
    auto read_step = [&](int s, int buf) {
      const int ub = s / 3, kx = s % 3;
      const int ky = P == 0 ? ub : (P == 1 ? (ub == 0 ? 2 : 0) : ub + 1);
      const int stage = P == 0 ? 0 : (P == 1 ? ub : 1);
      const int aoff = stage * G::A_STAGE_E + (ky * CL + kx) * 8;
#pragma unroll
      for (int m = 0; m < 2 * MT; ++m) af[buf][m] = lds_frag<T>(ldsA + a_off[m] + aoff);
#pragma unroll
      for (int n = 0; n < 2 * NT; ++n) bf[buf][n] = lds_frag<T>(ldsW + wb + b_lane + ub * G::UNIT_E + (kx * 4 * BN + n * 16) * 8);
    };
    read_step(0, 0);
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      asm volatile("" ::: "memory");
      if (s + 1 < 6) read_step(s + 1, (s + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      mma16_step<MT, NT>(acc, af[s & 1], bf[s & 1]);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) asm volatile("" : "+v"(acc[m][n]));
    __syncthreads();
  };
  for (int k = 0; k < nint; k += 3) {
    interval(std::integral_constant<int, 0>{}, k);
    interval(std::integral_constant<int, 1>{}, k + 1);
    interval(std::integral_constant<int, 2>{}, k + 2);
  }
  igemm_epilogue<T, TW, WM, WN, MT, NT, false, true, true, true>(a, acc, n0, y0, x0, nbase, smem_raw);
}

template <int TW>
static int m16_cfg(IgemmArgs& a, hipStream_t st, bool dry) {
  using G = M16Geom<TW>;
  const int cin = a.c0 + a.c1;
  if (a.h < G::TH) return SATCV_ERR_UNSUPPORTED;                               // (several small images per tile: the 32x32x16 tile)
  if (cin % 64 != 0 || (a.x1 && a.c0 % 32 != 0) || a.cout % 128 != 0 || a.cout_pad % 64 != 0 || a.cout_pad < a.cout) return SATCV_ERR_UNSUPPORTED;
  if (((uintptr_t)a.w % 16) != 0 || a.ldy % 8 != 0 || ((uintptr_t)a.y % 16) != 0) return SATCV_ERR_UNSUPPORTED;
  a.halh = a.halw = 1;
  a.tiles_x = cdiv(a.w_, TW); a.tiles_y = cdiv(a.h, G::TH);
  a.rpi = G::TH; a.imgs = 1; a.ngroups = a.n;
  a.seg = G::RL; a.rl = G::RL; a.cl = G::CL; a.pitch = G::CL;
  a.n_tiles = a.cout / 128;
  a.cpt = cin / 32; a.nchunks = a.cpt; a.taploop = 0; a.halh_tl = a.halw_tl = 1;
  a.ksplit = 1; a.kslab = nullptr;
  size_t lds_out = (size_t)G::BM * (G::BN + 8) * 2 + (size_t)(4 + 1) * 2 * G::BN * sizeof(float);
  if (a.bst_y) {      // fused BatchNorm-backward sums: interior tiles only (igemm_epilogue's fast path), a second staging tile
    if (a.h % G::TH != 0 || a.w_ % TW != 0 || a.cout % 8 != 0 || a.bst_ld % 8 != 0 || ((uintptr_t)a.bst_y % 16) != 0 ||
        (a.bst_y1 && (a.bst_split % 8 != 0 || a.bst_ld1 % 8 != 0 || ((uintptr_t)a.bst_y1 % 16) != 0)))
      return SATCV_ERR_UNSUPPORTED;
    lds_out += (size_t)G::BM * (G::BN + 8) * 2;
  }
  const size_t lds_op = G::LDS_OPERANDS + (a.in_scale ? (size_t)2 * cin * sizeof(float) : 0);
  const size_t lds = lds_op > lds_out ? lds_op : lds_out;
  if (lds > 160 * 1024) return SATCV_ERR_UNSUPPORTED;
  const long long blocks = (long long)a.n * a.tiles_y * a.tiles_x * a.n_tiles;
  if (blocks <= 0 || blocks > 0x7fffffffLL) return SATCV_ERR_UNSUPPORTED;
  if (dry) return SATCV_OK;
  // wave roles from 256 input channels on (SATCV_M16_WS=2: always, =0: never): with four chunks the longer prologue of the role kernel (three
  // chunks loaded before the first MFMA) is not amortised -- 128 -> 128 at 64 x 64: 97-99 us symmetric, 101-105 with roles, 108 on the 32x32x16 tile
  static const int roles_opt = [] { const char* e = getenv("SATCV_M16_WS"); return e ? atoi(e) : 1; }();
  const bool roles = roles_opt >= 2 || (roles_opt == 1 && cin >= 256);
  auto kern = roles ? igemm_m16_kernel<TW> : igemm_m16sym_kernel<TW>;
  { const int rc = satcv_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds); if (rc) return rc; }
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(roles ? 768 : 512), lds, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { satcv_set_error("igemm_m16 launch: %s", hipGetErrorString(e)); return SATCV_ERR_HIP; }
  return SATCV_OK;
}

// bf16 3x3, dilation 1, plain NHWC in and out, Cout % 128 == 0, Cin % 64 == 0, maps at least 16 pixels wide and a tile high;
// SATCV_ERR_UNSUPPORTED otherwise (the caller continues with the 32x32x16 tiles)
int igemm_m16_launch(IgemmArgs& a, int dtype, hipStream_t st, bool dry) {
  if (dtype != SATCV_BF16) return SATCV_ERR_UNSUPPORTED;
  if (!(a.kh == 3 && a.kw == 3 && a.dil == 1 && a.stride == 1 && a.mode_in == 0 && a.mode_out == 0 && !a.pool_y && !a.out_scale)) return SATCV_ERR_UNSUPPORTED;
  switch (igemm_pick_tw(a.w_)) {
    case 32: return m16_cfg<32>(a, st, dry);
    case 16: return m16_cfg<16>(a, st, dry);
    default: return SATCV_ERR_UNSUPPORTED;
  }
}
