// Round 5: the thin 3x3 convolutions (16 / 32 / 64 -> 32 / 64 channels at full and half resolution: conv_batch_act of enc0, enc1, dec1.conv2,
// dec0.*, utils/model_tools.py:174-186; forward, data gradient and the folded inference convs) with WAVE ROLES.
//
// These layers are HBM-bound on paper (50-100 us at batch 64) and ran at 0.45-0.55 of that on the persistent weights-stationary kernel
// (conv_igemm_ws.hip), whose waves each did everything in turn: BatchNorm affine + LDS stores of the next tile, MFMAs, epilogue -- ~450
// vector instructions per tile and wave around 36 MFMAs, the phases of a workgroup adding up instead of overlapping.
// The 16x16x32 tile and the fused transposed-conv backward showed what fixes that: the vector work must sit in OTHER waves than the MFMAs.
//   * waves 8-11 STAGE and STORE: two tiles of input loads in flight (register sets by tile parity); per iteration they write out the
//     PREVIOUS tile (staging tile -> 16-byte row stores, BatchNorm sum / sum of squares of the stored values, the fused max-pool of the folded
//     inference graph), stage the NEXT tile's halo image (the producing layer's BatchNorm + ReLU in registers, zero padding) into the
//     A buffer the matrix waves are not reading, and refill the register set;
//   * waves 0-7 MULTIPLY: the whole 9-tap weight tensor is resident in LDS, a tile is 9 x Cin / 32 K steps of v_mfma_f32_16x16x32_bf16 per
//     16 x 16 block; bias / folded multiplier / ReLU on the accumulators, bf16 into the other staging tile.  ONE barrier per tile.
// Tile = TH x 32 pixels (TH = 8, or 4 where two A buffers of a 64-channel halo tile would not fit); slot planes are multiples of 256 bytes
// (conflict-free ds_read_b128 of the 16x16x32 operands), staged items dealt so that 8 consecutive lanes write 8 consecutive pixels.
// Whole tiles only; the kernel is chosen by shape alone (bit-identical inference across batch splits).
#include "igemm_common.hpp"
#include <cstdlib>
extern int g_opt_igemm_thin, g_opt_thin_roles;      // api.hip: satcv_set_option
extern int g_ws_launches, g_tr_launches;                  // conv_igemm_ws.hip: launches of the persistent thin-layer kernels (tests assert the path taken)

template <int CINS, int COUT, int TH>      // CINS: stored input channels (16 / 32 / 64)
struct TrGeom {
  static constexpr int CINP = CINS < 32 ? 32 : CINS;                       // channels the K steps cover (a 16-channel input: two zero planes)
  static constexpr int SX = CINP / 8, SXR = CINS / 8, PX = TH * 32, RL = TH + 2, CL = 34, NPIX = RL * CL, NPIXP = (NPIX + 7) / 8 * 8;
  static constexpr int PLANE_E = ((NPIXP * 16 + 255) / 256 * 256) / 2;
  static constexpr int ABUF_E = SX * PLANE_E;
  static constexpr int W_E = 9 * CINP * COUT;                              // [tap][CINP / 8][COUT][8]
  static constexpr int OPITCH = COUT + 8, O_E = PX * OPITCH;
  static constexpr size_t LDS = (size_t)(2 * ABUF_E + W_E + 2 * O_E) * 2;
  static constexpr int A_ITEMS = NPIXP * SXR, AI = (A_ITEMS + 255) / 256;
  static_assert(LDS >= 16 * 256 * sizeof(float), "end-of-kernel reduction of the statistics");
};

template <int CINS, int COUT, int TH>
__global__ __launch_bounds__(768, 1) void igemm_tr_kernel(const IgemmArgs a, const int total_tiles) {
  typedef bf16 T;
  using G = TrGeom<CINS, COUT, TH>;
  constexpr int CINP = G::CINP, SX = G::SX, SXR = G::SXR, PX = G::PX, CL = G::CL, OPITCH = G::OPITCH, AI = G::AI;
  constexpr int NS = 256, VQ = COUT / 8, OI = PX * VQ / NS, OQS = NS / VQ;                 // output vectors per staging thread; pixels between two of them
  static_assert((PX * VQ) % NS == 0 && NS % VQ == 0 && NS % (8 * SXR) == 0, "item -> thread mapping");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* ldsA = reinterpret_cast<T*>(smem_raw);                                // two halo images [SX][PLANE_E]
  T* ldsW = ldsA + 2 * G::ABUF_E;
  T* ldsO = ldsW + G::W_E;                                                 // two output staging tiles [PX][OPITCH]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware contiguous tile ranges (conv_igemm_ws.hip)
  const int Gd = gridDim.x;
  const int xcd = blockIdx.x & 7, nx = Gd >> 3, remx = Gd & 7;
  const int bid = (xcd < remx ? xcd * (nx + 1) : remx * (nx + 1) + (xcd - remx) * nx) + (blockIdx.x >> 3);
  const int per = total_tiles / Gd, extra = total_tiles % Gd;
  const int t_lo = bid * per + (bid < extra ? bid : extra), t_hi = t_lo + per + (bid < extra ? 1 : 0);
  // ---- once per workgroup: the weight tensor [tap][k slot][COUT][8] (padded k slots zero), zero planes of a 16-channel input
  {
    const T* wp = reinterpret_cast<const T*>(a.w);
    for (int it = tid; it < 9 * SX * COUT; it += 768) {
      const int co = it % COUT, slot = (it / COUT) % SX, tap = it / (COUT * SX);
      Raw8<T> v = zero8<T>();
      if (slot < SXR) v = gload8<T>(wp + ((size_t)(tap * SXR + slot) * COUT + co) * 8);
      lstore8<T>(ldsW + (size_t)it * 8, v);
    }
    if constexpr (SXR < SX) {
      for (int it = tid; it < 2 * (SX - SXR) * G::PLANE_E / 8; it += 768) {
        const int b = it / ((SX - SXR) * G::PLANE_E / 8), rest = it % ((SX - SXR) * G::PLANE_E / 8);
        lstore8<T>(ldsA + b * G::ABUF_E + SXR * G::PLANE_E + rest * 8, zero8<T>());
      }
    }
  }
  auto tile_origin = [&](int v, int& n0, int& y0, int& x0) __attribute__((always_inline)) {
    const int tx = v % a.tiles_x; v /= a.tiles_x;
    const int ty = v % a.tiles_y;
    n0 = v / a.tiles_y; y0 = ty * TH; x0 = tx * 32;
  };

  if (wave >= 8) {
    // ================================================================ staging / storing waves
    const int sid = tid - 512;
    // input items: group of 8 SXR items = 8 pixels x SXR slots; lane i of block b takes pixel i, slot (b + i) % SXR -- 8 consecutive lanes
    // write 8 consecutive pixels of one plane (distinct banks), a wave-instruction still covers whole pixels
    const int slot_t = ((sid >> 3) + sid) & (SXR - 1);
    const int ch0 = slot_t * 8;
    const bool xsecond = a.x1 != nullptr && ch0 >= a.c0;
    const T* xsrc = xsecond ? reinterpret_cast<const T*>(a.x1) + (ch0 - a.c0) : reinterpret_cast<const T*>(a.x0) + ch0;
    const int xcs = xsecond ? a.c1 : a.c0;
    int it_l[AI], it_c[AI];                                                // halo row / column of item j (or -1)
#pragma unroll
    for (int j = 0; j < AI; ++j) {
      const int it = sid + j * NS;
      const int pix = (it / (8 * SXR)) * 8 + (it & 7);
      it_l[j] = (it < G::A_ITEMS && pix < G::NPIX) ? pix / CL : -1;
      it_c[j] = pix % CL;
    }
    float xsc[8], xsh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { xsc[e] = a.in_scale ? a.in_scale[ch0 + e] : 1.f; xsh[e] = a.in_scale ? a.in_shift[ch0 + e] : 0.f; }
    const bool xaff = a.in_scale != nullptr;
    const unsigned xrelu_lim = a.in_relu != 0 ? 0u : 0x80008000u;
    const int rowstride = a.w_ * xcs;
    Raw8<T> rx[2][AI];
    unsigned vmask[2] = {0u, 0u};
    auto issue = [&](int v, auto SET) __attribute__((always_inline)) {
      constexpr int S = decltype(SET)::value;
      int n0, y0, x0; tile_origin(v, n0, y0, x0);
      const int ylo = 1 - y0, yhi = a.h - y0 + 1, xlo = 1 - x0, xhi = a.w_ - x0 + 1;      // halo rows / columns inside the image
      const long long pb = (long long)(n0 * a.h + y0 - 1) * a.w_ + (x0 - 1);               // halo origin (may lie before the tensor: never dereferenced)
      const T* xb = xsrc + pb * xcs;
      const unsigned centre = (unsigned)(rowstride + xcs);                                 // the tile's first pixel: always inside
      unsigned vm = 0;
#pragma unroll
      for (int j = 0; j < AI; ++j) {
        const int L = it_l[j], c = it_c[j];
        const bool ok = L >= ylo && L < yhi && c >= xlo && c < xhi && L >= 0;
        vm |= (ok ? 1u : 0u) << j;
        unsigned off = ok ? (unsigned)(__mul24(L, rowstride) + __mul24(c, xcs)) : centre;
        asm volatile("" : "+v"(off));
        rx[S][j] = gload8<T>(xb + off);
      }
      vmask[S] = vm;
    };
    auto store_tile = [&](int b, auto SET) __attribute__((always_inline)) {
      constexpr int S = decltype(SET)::value;
      T* d = ldsA + b * G::ABUF_E + slot_t * G::PLANE_E;
#pragma unroll
      for (int j = 0; j < AI; ++j) {
        Raw8<T> v = rx[S][j];
        if (xaff) v = affine8_lim(v, xsc, xsh, xrelu_lim);
        v = select8<T>((vmask[S] >> j) & 1u, v);
        const int it = sid + j * NS;
        const int pix = (it / (8 * SXR)) * 8 + (it & 7);
        if (it < G::A_ITEMS) lstore8<T>(d + pix * 8, v);
      }
    };
    // output vectors of this thread: channel group vq, pixels oq0 + j * OQS
    const int vq = sid % VQ, oq0 = sid / VQ;
    float st1[8], st2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { st1[e] = 0.f; st2[e] = 0.f; }
    const unsigned o_rowstride = (unsigned)a.w_ * (unsigned)a.ldy;
    auto epilogue = [&](int v, int b) __attribute__((always_inline)) {
      int n0, y0, x0; tile_origin(v, n0, y0, x0);
      T* yp = reinterpret_cast<T*>(a.y) + ((size_t)(n0 * a.h + y0) * a.w_ + x0) * a.ldy + vq * 8;
      const T* so = ldsO + b * G::O_E + vq * 8;
#pragma unroll
      for (int j = 0; j < OI; ++j) {
        const int q = oq0 + j * OQS;
        const uint4 dv = *reinterpret_cast<const uint4*>(so + q * OPITCH);
        unsigned off = (unsigned)(q >> 5) * o_rowstride + (unsigned)(q & 31) * (unsigned)a.ldy;
        asm volatile("" : "+v"(off));
        *reinterpret_cast<uint4*>(yp + off) = dv;
        if (a.stats) {
          const bf16x8 d8 = __builtin_bit_cast(bf16x8, dv);
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float f = (float)d8[e]; st1[e] += f; st2[e] += f * f; }
        }
      }
      if (a.pool_y) {
        // fused max-pool of the tile just stored (folded inference encoder blocks): window == stride == pool_f, whole windows inside the tile
        const int f = a.pool_f, pw = 32 / f, ph = TH / f, hp = a.h / f, wp = a.w_ / f;
        T* pp = reinterpret_cast<T*>(a.pool_y) + vq * 8;
        for (int pq = oq0; pq < ph * pw; pq += OQS) {
          const int t0 = (pq / pw) * f, c0 = (pq % pw) * f;
          float mx[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) mx[e] = -INFINITY;
          for (int i = 0; i < f; ++i)
            for (int jj = 0; jj < f; ++jj) {
              const bf16x8 s8 = *reinterpret_cast<const bf16x8*>(so + ((t0 + i) * 32 + c0 + jj) * OPITCH);
#pragma unroll
              for (int e = 0; e < 8; ++e) mx[e] = fmaxf(mx[e], (float)s8[e]);
            }
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16)mx[e];
          *reinterpret_cast<bf16x8*>(pp + ((size_t)(n0 * hp + (y0 + t0) / f) * wp + (x0 + c0) / f) * a.pool_ld) = o;
        }
      }
    };
    const std::integral_constant<int, 0> S0{};
    const std::integral_constant<int, 1> S1{};
    if (t_lo < t_hi) issue(t_lo, S0);
    if (t_lo + 1 < t_hi) issue(t_lo + 1, S1);
    __syncthreads();                                                        // (1) weights in LDS
    if (t_lo < t_hi) {
      store_tile(0, S0);
      if (t_lo + 2 < t_hi) issue(t_lo + 2, S0);
    }
    __syncthreads();                                                        // (2) tile t_lo staged
    auto iteration = [&](int t, auto KC) __attribute__((always_inline)) {
      constexpr int K = decltype(KC)::value;
      if (t > t_lo) epilogue(t - 1, K ^ 1);
      if (t + 1 < t_hi) {
        store_tile(K ^ 1, std::integral_constant<int, K ^ 1>{});
        if (t + 3 < t_hi) issue(t + 3, std::integral_constant<int, K ^ 1>{});
      }
      __syncthreads();
    };
    for (int t = t_lo; t < t_hi; t += 2) {
      iteration(t, S0);
      if (t + 1 < t_hi) iteration(t + 1, S1);
    }
    if (t_lo < t_hi) epilogue(t_hi - 1, (t_hi - 1 - t_lo) & 1);
    __syncthreads();                                                        // (3) every staging read is done; the matrix waves have ended
    if (a.stats && t_lo < t_hi) {
      // statistics of the stored values: the threads of a channel group summed through LDS in a fixed order, one pair of atomics per channel
      float* r2 = reinterpret_cast<float*>(smem_raw);                       // [16][256]
#pragma unroll
      for (int e = 0; e < 8; ++e) { r2[e * NS + sid] = st1[e]; r2[(8 + e) * NS + sid] = st2[e]; }
      __syncthreads();
      if (sid < COUT) {
        const int gq = sid >> 3, e = sid & 7;
        double t1 = 0.0, t2 = 0.0;
        for (int k = 0; k < NS / VQ; ++k) { t1 += (double)r2[e * NS + k * VQ + gq]; t2 += (double)r2[(8 + e) * NS + k * VQ + gq]; }
        satcv_stat_t* rowp = a.stats + (size_t)(blockIdx.x % SATCV_STAT_ROWS) * 2 * a.stats_ld;
        atomicAdd(rowp + sid, (satcv_stat_t)t1);
        atomicAdd(rowp + a.stats_ld + sid, (satcv_stat_t)t2);
      }
    }
    return;
  }

  // ================================================================ matrix waves
  const int g4 = lane >> 4, l16 = lane & 15;
  constexpr int NPB = PX / 16, PBW = NPB / 8, NB = COUT / 16, KS = CINP / 32;
  static_assert(PBW * 8 == NPB, "pixel blocks per wave");
  int a_off[PBW];
#pragma unroll
  for (int p = 0; p < PBW; ++p) {
    const int q = (wave * PBW + p) * 16 + l16;
    a_off[p] = g4 * G::PLANE_E + ((q >> 5) * CL + (q & 31)) * 8;
  }
  const int b_off = (g4 * COUT + l16) * 8;
  float bia[NB], osc[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n) { bia[n] = a.bias ? a.bias[n * 16 + l16] : 0.f; osc[n] = a.out_scale ? a.out_scale[n * 16 + l16] : 1.f; }
  const bool orelu = a.out_relu != 0;
  __syncthreads();                                                          // (1)
  __syncthreads();                                                          // (2)
  for (int t = t_lo; t < t_hi; ++t) {
    const int b = (t - t_lo) & 1;
    const T* A = ldsA + b * G::ABUF_E;
    f32x4 acc[PBW][NB];
#pragma unroll
    for (int p = 0; p < PBW; ++p)
#pragma unroll
      for (int n = 0; n < NB; ++n) acc[p][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      constexpr int STEPS = 9 * KS;
      FragT<T> af[2][PBW], bf[2][NB];
      auto read_step = [&](int st, int buf) __attribute__((always_inline)) {
        const int tap = st / KS, ks = st % KS;
        const int toff = ((tap / 3) * CL + (tap % 3)) * 8 + ks * 4 * G::PLANE_E;
#pragma unroll
        for (int p = 0; p < PBW; ++p) af[buf][p] = lds_frag<T>(A + a_off[p] + toff);
#pragma unroll
        for (int n = 0; n < NB; ++n) bf[buf][n] = lds_frag<T>(ldsW + ((tap * SX + ks * 4) * COUT + n * 16) * 8 + b_off);
      };
      read_step(0, 0);
      // (three compile-time groups of taps: hipcc does not fully unroll 72 MFMAs in one loop, and a run-time step index sends the fragment
      //  arrays to scratch)
      auto tap_row = [&](auto GC) __attribute__((always_inline)) {
        constexpr int S0_ = decltype(GC)::value * 3 * KS;
#pragma unroll
        for (int s_ = 0; s_ < 3 * KS; ++s_) {
          const int st = S0_ + s_;
          asm volatile("" ::: "memory");
          if (st + 1 < STEPS) read_step(st + 1, (st + 1) & 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int p = 0; p < PBW; ++p)
#pragma unroll
            for (int n = 0; n < NB; ++n) acc[p][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[st & 1][p].v, bf[st & 1][n].v, acc[p][n], 0, 0, 0);
        }
      };
      tap_row(std::integral_constant<int, 0>{});
      tap_row(std::integral_constant<int, 1>{});
      tap_row(std::integral_constant<int, 2>{});
    }
    // accumulators -> multiplier / bias / ReLU -> bf16 -> staging tile b: block rows 4 g4 + j, column l16
#pragma unroll
    for (int p = 0; p < PBW; ++p) {
      T* op = ldsO + b * G::O_E + ((wave * PBW + p) * 16 + 4 * g4) * OPITCH + l16;
#pragma unroll
      for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = acc[p][n][j] * osc[n] + bia[n];
          if (orelu) v = fmaxf(v, 0.f);
          op[j * OPITCH + n * 16] = (T)v;
        }
    }
    __syncthreads();
  }
  __syncthreads();                                                          // (3)
}

// ------------------------------------------------------------------ host side
template <int CINS, int COUT, int TH>
static int tr_cfg(IgemmArgs& a, hipStream_t st, bool dry) {
  using G = TrGeom<CINS, COUT, TH>;
  static_assert(G::LDS <= 160 * 1024, "buffers + weights exceed the LDS");
  if (a.h % TH != 0) return SATCV_ERR_UNSUPPORTED;
  if (a.pool_y && (TH % a.pool_f != 0 || 32 % a.pool_f != 0)) return SATCV_ERR_UNSUPPORTED;
  a.halh = a.halw = 1;
  a.tiles_x = a.w_ / 32; a.tiles_y = a.h / TH;
  a.rpi = TH; a.imgs = 1; a.ngroups = a.n; a.seg = TH + 2; a.rl = TH + 2; a.cl = 34; a.pitch = 34; a.n_tiles = 1;
  const long long total = (long long)a.n * a.tiles_y * a.tiles_x;
  if (total <= 0 || total > 0x7fffffffLL) return SATCV_ERR_UNSUPPORTED;
  if ((long long)(TH + 2) * a.w_ * (a.c0 > a.c1 ? a.c0 : a.c1) >= (1LL << 23)) return SATCV_ERR_UNSUPPORTED;      // 24-bit multiplies of the halo offsets
  if (dry) return SATCV_OK;
  auto kern = igemm_tr_kernel<CINS, COUT, TH>;
  { const int rc = satcv_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), G::LDS); if (rc) return rc; }
  static int ncu = 0;
  if (!ncu) {
    int dev = 0; hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) { satcv_set_error("igemm_tr: device query failed"); return SATCV_ERR_HIP; }
    ncu = p.multiProcessorCount;
  }
  long long grid = ncu;
  if (grid > total) grid = total;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(768), G::LDS, st, a, (int)total);
  ++g_ws_launches; ++g_tr_launches;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { satcv_set_error("igemm_tr launch: %s", hipGetErrorString(e)); return SATCV_ERR_HIP; }
  return SATCV_OK;
}

// bf16 3x3, dilation 1, whole TH x 32 tiles, 16 / 32 / 64 stored input channels -> 32 / 64 output channels (64 -> 64 stays on the
// weights-stationary kernel: its 74 KB of weights leave no room for two halo images and two staging tiles).  SATCV_ERR_UNSUPPORTED otherwise.
int igemm_tr_launch(IgemmArgs& a, int dtype, hipStream_t st, bool dry) {
  const int on = g_opt_thin_roles;
  if (!on || !g_opt_igemm_thin || dtype != SATCV_BF16) return SATCV_ERR_UNSUPPORTED;
  const int cin = a.c0 + a.c1;
  if (a.kh != 3 || a.kw != 3 || a.dil != 1 || a.stride != 1 || a.mode_in || a.mode_out || a.accumulate || a.bst_y) return SATCV_ERR_UNSUPPORTED;
  if (!(cin == 16 || cin == 32 || cin == 64) || !(a.cout == 32 || a.cout == 64) || a.cout_pad != a.cout || a.cstat != a.cout) return SATCV_ERR_UNSUPPORTED;
  if (a.x1 && (a.c0 % 8 != 0)) return SATCV_ERR_UNSUPPORTED;
  if (a.w_ % 32 != 0 || a.ldy % 8 != 0 || ((uintptr_t)a.y % 16) != 0 || ((uintptr_t)a.w % 16) != 0) return SATCV_ERR_UNSUPPORTED;
  if (a.pool_y && (a.pool_ld % 8 != 0 || ((uintptr_t)a.pool_y % 16) != 0)) return SATCV_ERR_UNSUPPORTED;
  // MEASURED against the weights-stationary kernel (profiles/r05_ab_thin_roles_probe.txt, batch 64): 32 -> 32 at 256 x 256 179 vs 189 us, 32 -> 64
  // 236 vs 248 (128 x 128: 61 vs 66) -- kept; 16 -> 32 140 vs 112 (two 11-KB tiles in flight per CU are too few bytes), 64 -> 32 with 4-row
  // tiles 311 vs 259 (1.6x halo re-reads) -- those stay on the weights-stationary kernel, whose three workgroups per CU put twelve waves on
  // the vector work of these layers where this kernel has four (option thin_roles = 2 / SATCV_THIN_ROLES=2 routes every served shape here)
  if (on < 2 && cin != 32) return SATCV_ERR_UNSUPPORTED;
  if (a.cout == 32) {
    if (cin == 16) return tr_cfg<16, 32, 8>(a, st, dry);
    if (cin == 32) return tr_cfg<32, 32, 8>(a, st, dry);
    return tr_cfg<64, 32, 4>(a, st, dry);
  }
  if (cin == 16) return tr_cfg<16, 64, 8>(a, st, dry);
  if (cin == 32) return tr_cfg<32, 64, 8>(a, st, dry);
  return SATCV_ERR_UNSUPPORTED;
}
