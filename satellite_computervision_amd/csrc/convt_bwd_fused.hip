// Round 5: fused backward of decoder_block's up-sampling path (utils/model_tools.py:306-309: Conv2DTranspose(k = s = 2) -> concatenate([skip, up])
// -> BatchNormalization -> ReLU, differentiated): ONE launch replaces, for the `up` half of the concatenation,
//     satcv_bn_bwd_apply (write dup)  +  the space-to-depth data gradient (read dup, write dx)  +  the weight gradient (read dup and x).
// dup = scale (g [scale y + shift > 0] - c1 - xhat c2) is formed in registers from g (the gradient of the activated concatenation, its `up`
// channels) and y (the transposed convolution's stored output), never written to HBM, and used twice from LDS:
//     dx[p][ci]        = sum_{ij, co} dup[up(p, ij)][co] K[ij][co][ci]                    (A = dup rows, B = the data-gradient operand image)
//     dK[ij][co][ci]  += sum_p        dup[up(p, ij)][co] x[p][ci]                         (both operands by transposing reads)
// A transposed convolution with kernel = stride is a 1 x 1 convolution from Cin to 4 Cout channels on the INPUT grid, stored depth-to-space: the
// backward is that of a 1 x 1 convolution whose gradient is gathered space-to-depth -- no halo.  HBM bytes per input pixel: g and y (4 Cout
// each), x and dx (Cin each) = 12 Cout x 2 B instead of 24 (SURVEY section 8d counts the unfused passes).
//
// Structure (what the 16x16x32 tile of conv_igemm_m16.hip taught: the staging must not sit in the waves that issue the MFMAs).  12 waves:
//   * waves 8-11 STAGE: per tile of PX input pixels (a segment of one image row) they hold g, y, x of the NEXT tile in registers (one tile =
//     ~80-100 KB in flight per CU), form dup / the activated x, write them as [8-channel plane][pixel][8] images (plane stride 64 B modulo
//     256 B: conflict-free transposing reads) into the buffer the matrix waves are not reading, and write out the PREVIOUS tile's dx from its
//     staging tile with 16-byte row stores, forming the BatchNorm-backward sums of the layer below (sum dx [a > 0], sum dx a) on the way;
//   * waves 0-7 MULTIPLY: weight gradient (each wave owns whole 32 x 32 (ci, k) tiles of dK for the whole launch: one fp32 slab per
//     workgroup at the end, summed in fixed order by the slab-sum kernel) and data gradient (16 x 16 blocks on v_mfma_f32_16x16x32_bf16), whose
//     result goes to a double-buffered staging tile.  ONE barrier per tile.
// A workgroup owns a block of 64 input channels (its slice of the operand image stays in LDS) and a contiguous range of tiles; persistent.
#include "igemm_common.hpp"
#include <cstdlib>
#include <cstring>

void reduce_job_fill(satcv_reduce_job* job, const float* ws, float* dw, int nslab, int taps, int kpad, int npad, int cin, int nvalid, int transposed, int accumulate);
int wgrad_reduce_slabs_t(const float* ws, float* dw, int nslab, int kpad, int npad, int cin, int nvalid, int accumulate, hipStream_t st);      // conv_wgrad.hip
void satcv_prof_begin(int kind, double flops, hipStream_t st);
void satcv_prof_end(int kind, hipStream_t st);

struct CtbfArgs {
  const void* g; int ldg;              // gradient of the activated concatenation, offset to the first `up` channel; (n, 2h, 2w, ldg)
  const void* yup; int ldy;            // raw output of the transposed convolution (n, 2h, 2w, ldy)
  const float* bn_scale; const float* bn_shift; const float* bn_mean; const float* bn_rstd;      // of the `up` channels
  const float* bn_c1; const float* bn_c2; int linear;
  const void* x; int ldx;              // the transposed convolution's input (n, h, w, ldx), raw with a pending BatchNorm + ReLU (or activated)
  const float* in_scale; const float* in_shift; int in_relu;
  const void* w; int npad;             // data-gradient operand image [4 Cout / 8][npad][8]
  void* dx; int lddx;
  float* ws;                           // [slabs][cin][4 Cout] partial weight gradients
  int n, h, w_, cin;
  satcv_stat_t* bst_sums; int bst_ld; const float* bst_mean; const float* bst_rstd;
  int tiles, nblk, slabs;              // tiles of PX pixels; cin / 64 channel blocks; workgroups per channel block
};

__device__ __forceinline__ bf16x4 ct_tr_read4(const bf16* p) {
  short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(p));
  return __builtin_bit_cast(bf16x4, v);
}

template <int COUT4, int PX, int CBLK>
struct CtbfGeom {
  static constexpr int SD = COUT4 / 8, C8 = SD / 4, SX = CBLK / 8;
  static constexpr int STR = PX * 8 + 32;                                  // plane stride in elements: PX * 16 B + 64 B
  static constexpr int BUF_E = (SD + SX) * STR;
  static constexpr int W_E = COUT4 * CBLK;                                 // [COUT4 / 8][CBLK][8]
  static constexpr int OPITCH = CBLK + 8;
  static constexpr int O_E = PX * OPITCH;
  static constexpr int TAB_F = C8 * 32 + SX * 16;                          // floats: [C8][4][8] dup coefficients, [SX][2][8] input scale / shift
  static constexpr size_t LDS = (size_t)(2 * BUF_E + W_E + 2 * O_E) * 2 + (size_t)TAB_F * sizeof(float);
  static_assert(LDS >= 16 * 256 * sizeof(float), "end-of-kernel reduction of the fused sums");
  static_assert(((STR * 2) % 256) == 64, "plane stride must be 64 B modulo 256 B");
};

template <int COUT4, int PX, int CBLK>
__global__ __launch_bounds__(768, 1) void convt_bwd_fused_kernel(const CtbfArgs a) {
  typedef bf16 T;
  using G = CtbfGeom<COUT4, PX, CBLK>;
  constexpr int SD = G::SD, C8 = G::C8, SX = G::SX, STR = G::STR, OPITCH = G::OPITCH, COUT = COUT4 / 4;
  constexpr int NS = 256, DI = PX * SD / NS, XI = PX * SX / NS, XQS = NS / SX;      // XQS: pixels between two x / dx items of a thread
  static_assert(PX * SD % NS == 0 && NS % SD == 0 && (PX * SX) % NS == 0 && NS % SX == 0, "item -> thread mapping");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* lds0 = reinterpret_cast<T*>(smem_raw);                                // buffer b: dup planes [SD], then x planes [8]
  T* ldsW = lds0 + 2 * G::BUF_E;
  T* ldsO = ldsW + G::W_E;                                                 // two dx staging tiles [PX][OPITCH]
  float* tabD = reinterpret_cast<float*>(ldsO + 2 * G::O_E);               // [C8][4][8]: sc, sh, B, C with dup = sc gm + B y + C, act = sc y + sh
  float* tabX = tabD + C8 * 32;                                            // [8][2][8]: scale, shift of the input's BatchNorm (this channel block)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroup -> (channel block, slab): the workgroups of one pixel range and different channel blocks have consecutive ids (one XCD: they
  // read the same g / y tiles, three times out of four from its L2)
  const int blk = blockIdx.x % a.nblk, slab = blockIdx.x / a.nblk;
  const int ci0 = blk * CBLK;
  const int per = a.tiles / a.slabs, extra = a.tiles % a.slabs;
  const int t_lo = slab * per + (slab < extra ? slab : extra), t_hi = t_lo + per + (slab < extra ? 1 : 0);
  const int segs = a.w_ / PX;
  // operand image slice of this channel block: [k / 8][64][8]
  {
    const T* wp = reinterpret_cast<const T*>(a.w);
    for (int it = tid; it < SD * CBLK; it += 768) lstore8<T>(ldsW + (size_t)it * 8, gload8<T>(wp + ((size_t)(it / CBLK) * a.npad + ci0 + it % CBLK) * 8));
    for (int ch = tid; ch < COUT; ch += 768) {
      const float sc = a.bn_scale[ch], sh = a.bn_shift[ch], mu = a.bn_mean[ch], rs = a.bn_rstd[ch], c1 = a.bn_c1[ch], c2 = a.bn_c2[ch];
      float* t = tabD + (ch >> 3) * 32 + (ch & 7);
      t[0] = sc; t[8] = sh; t[16] = -sc * c2 * rs; t[24] = sc * (c2 * rs * mu - c1);
    }
    for (int ch = tid; ch < CBLK; ch += 768) {
      float* t = tabX + (ch >> 3) * 16 + (ch & 7);
      t[0] = a.in_scale ? a.in_scale[ci0 + ch] : 1.f; t[8] = a.in_scale ? a.in_shift[ci0 + ch] : 0.f;
    }
  }

  if (wave >= 8) {
    // ================================================================ staging waves
    const int sid = tid - 512;
    const int plane = sid % SD, c8 = plane % C8, ij = plane / C8;          // (thread-fixed: NS % SD == 0)
    const int q0 = sid / SD;                                               // pixel of item j: q0 + j * (NS / SD)
    const int vq = sid % SX, xq0 = sid / SX;                               // x / dx items: channel group vq, pixel xq0 + j * XQS
    // (the per-channel constants of a thread's 8 channels are read from the LDS tables inside the phases that use them: held in registers for the
    //  whole launch beside one tile of loads they spilled)
    const bool xaff = a.in_scale != nullptr;
    const unsigned xrelu_lim = a.in_relu != 0 ? 0u : 0x80008000u;
    const float lin_lo = a.linear ? -INFINITY : 0.f;
    // source offsets relative to the tile's first output pixel (elements): item j is pixel q0 + j * (NS / SD)
    const int wo = 2 * a.w_;
    const unsigned pix0 = (unsigned)((ij >> 1) * wo + 2 * q0 + (ij & 1));
    const unsigned g_off0 = pix0 * (unsigned)a.ldg + (unsigned)c8 * 8u, g_step = (unsigned)(2 * (NS / SD)) * (unsigned)a.ldg;
    const unsigned y_off0 = pix0 * (unsigned)a.ldy + (unsigned)c8 * 8u, y_step = (unsigned)(2 * (NS / SD)) * (unsigned)a.ldy;
    // TWO tiles of loads in flight (register sets 0 / 1 by tile parity): with one, every tile waited for a full memory latency -- 40 KB in flight
    // per CU sustain ~14 GB/s per CU at ~3 us under load, 3.5 of the ~6 TB/s this pattern can reach
    Raw8<T> rg[2][DI], ry[2][DI], rx[2][XI];
    float bs1[8], bs2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { bs1[e] = 0.f; bs2[e] = 0.f; }
    auto origin = [&](int v, size_t& opix, size_t& ipix) {                 // first output / input pixel of tile v
      const int row = v / segs, x0 = (v - row * segs) * PX;
      const int n = row / a.h, y = row - n * a.h;
      opix = ((size_t)(n * 2 * a.h + 2 * y)) * wo + 2 * x0;
      ipix = ((size_t)(n * a.h + y)) * a.w_ + x0;
    };
    // every global access is (wave-uniform pointer of the tile) + (32-bit lane offset made opaque at its use): as 64-bit lane pointers hipcc
    // hoisted the ~20 per-item addresses out of the tile loop and spilled them (a scratch reload queues behind the prefetched tile on the
    // in-order vmcnt counter: conv_bwd_fused.hip)
    const unsigned x_off0 = (unsigned)xq0 * (unsigned)a.ldx + (unsigned)(ci0 + vq * 8), x_step = (unsigned)XQS * (unsigned)a.ldx;
    const unsigned dx_off0 = (unsigned)xq0 * (unsigned)a.lddx + (unsigned)(ci0 + vq * 8), dx_step = (unsigned)XQS * (unsigned)a.lddx;
    auto issue = [&](int v, auto SET) {
      constexpr int S = decltype(SET)::value;
      size_t opix, ipix; origin(v, opix, ipix);
      const T* gb = reinterpret_cast<const T*>(a.g) + opix * a.ldg;
      const T* yb = reinterpret_cast<const T*>(a.yup) + opix * a.ldy;
#pragma unroll
      for (int j = 0; j < DI; ++j) {
        unsigned go = g_off0 + j * g_step, yo = y_off0 + j * y_step;
        asm volatile("" : "+v"(go), "+v"(yo));
        rg[S][j] = gload8<T>(gb + go); ry[S][j] = gload8<T>(yb + yo);
      }
      const T* xb = reinterpret_cast<const T*>(a.x) + ipix * a.ldx;
#pragma unroll
      for (int j = 0; j < XI; ++j) {
        unsigned xo = x_off0 + j * x_step;
        asm volatile("" : "+v"(xo));
        rx[S][j] = gload8<T>(xb + xo);
      }
    };
    auto store_tile = [&](int b, auto SET) {                                // register set -> buffer b (x first: its registers and constants are gone before the dup phase)
      constexpr int S = decltype(SET)::value;
      {
        float xsc[8], xsh[8];
        int toff = vq * 16;
        asm volatile("" : "+v"(toff));
        const float4* tp = reinterpret_cast<const float4*>(tabX + toff);
        const float4 s0 = tp[0], s1 = tp[1], h0 = tp[2], h1 = tp[3];
        xsc[0] = s0.x; xsc[1] = s0.y; xsc[2] = s0.z; xsc[3] = s0.w; xsc[4] = s1.x; xsc[5] = s1.y; xsc[6] = s1.z; xsc[7] = s1.w;
        xsh[0] = h0.x; xsh[1] = h0.y; xsh[2] = h0.z; xsh[3] = h0.w; xsh[4] = h1.x; xsh[5] = h1.y; xsh[6] = h1.z; xsh[7] = h1.w;
        T* xd = lds0 + b * G::BUF_E + (SD + vq) * STR;
#pragma unroll
        for (int j = 0; j < XI; ++j) {
          Raw8<T> v = rx[S][j];
          if (xaff) v = affine8_lim(v, xsc, xsh, xrelu_lim);
          lstore8<T>(xd + (xq0 + j * XQS) * 8, v);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      int toff = c8 * 32;
      asm volatile("" : "+v"(toff));
      const float4* tp = reinterpret_cast<const float4*>(tabD + toff);
      float prm[32];
#pragma unroll
      for (int k = 0; k < 8; ++k) { const float4 v4 = tp[k]; prm[4 * k] = v4.x; prm[4 * k + 1] = v4.y; prm[4 * k + 2] = v4.z; prm[4 * k + 3] = v4.w; }
      T* d = lds0 + b * G::BUF_E + plane * STR;
#pragma unroll
      for (int j = 0; j < DI; ++j) {
        float gv[8], yv[8];
        unpack8<T>(rg[S][j], gv); unpack8<T>(ry[S][j], yv);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float act = yv[e] * prm[e] + prm[8 + e];
          const float gm = act > lin_lo ? gv[e] : 0.f;
          o[e] = (bf16)(prm[e] * gm + (prm[16 + e] * yv[e] + prm[24 + e]));
        }
        Raw8<T> v; v.q[0] = __builtin_bit_cast(uint4, o);
        lstore8<T>(d + (q0 + j * (NS / SD)) * 8, v);
        __builtin_amdgcn_sched_barrier(0);                                  // (one item's unpacked values at a time)
      }
    };
    auto epilogue = [&](int v, int b) {                                     // dx of tile v: staging tile b -> global, + the sums of the layer below
      size_t opix, ipix; origin(v, opix, ipix);
      T* yp = reinterpret_cast<T*>(a.dx) + ipix * a.lddx;
      const T* so = ldsO + b * G::O_E + vq * 8;
      const T* xa = lds0 + b * G::BUF_E + (SD + vq) * STR;
#pragma unroll
      for (int j = 0; j < XI; ++j) {
        const int q = xq0 + j * XQS;
        const uint4 dv = *reinterpret_cast<const uint4*>(so + q * OPITCH);
        unsigned dxo = dx_off0 + j * dx_step;
        asm volatile("" : "+v"(dxo));
        *reinterpret_cast<uint4*>(yp + dxo) = dv;
        if (a.bst_sums) {
          const bf16x8 d8 = __builtin_bit_cast(bf16x8, dv);
          const bf16x8 a8 = *reinterpret_cast<const bf16x8*>(xa + q * 8);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float df = (float)d8[e], af = (float)a8[e];
            bs1[e] += af > 0.f ? df : 0.f;
            bs2[e] += df * af;
          }
        }
      }
    };
    const std::integral_constant<int, 0> S0{};
    const std::integral_constant<int, 1> S1{};
    if (t_lo < t_hi) issue(t_lo, S0);
    if (t_lo + 1 < t_hi) issue(t_lo + 1, S1);
    __syncthreads();                                                        // (1) operand image and tables in LDS
    if (t_lo < t_hi) {
      store_tile(0, S0);
      if (t_lo + 2 < t_hi) issue(t_lo + 2, S0);
    }
    __syncthreads();                                                        // (2) tile t_lo staged
    // iteration t (tile parity k = buffer the matrix waves work on): write out tile t - 1 (staging tile and x planes of buffer k ^ 1 -- every
    // thread reads exactly the x items it overwrites below, so no barrier is needed between the two), stage tile t + 1 from register set k ^ 1
    // into buffer k ^ 1, refill that set with tile t + 3
    auto iteration = [&](int t, auto KC) {
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
    // ---- fused sums: the 32 threads of a channel group are summed through LDS in a fixed order; one pair of atomics per channel and workgroup
    __syncthreads();                                                        // (3) the matrix waves have written their slabs
    if (a.bst_sums && t_lo < t_hi) {
      // every LDS region is free now (the last epilogue ended before barrier 3): [16][256] floats over the buffers
      float* r2 = reinterpret_cast<float*>(smem_raw);
#pragma unroll
      for (int e = 0; e < 8; ++e) { r2[e * NS + sid] = bs1[e]; r2[(8 + e) * NS + sid] = bs2[e]; }
      // (staging waves only from here on: an LDS-visible hand-off between them needs no workgroup barrier once every wave has WRITTEN -- they
      //  all pass the same s_barrier below together with nobody else, because the matrix waves have ended)
      __syncthreads();
      if (sid < CBLK) {
        const int gq = sid >> 3, e = sid & 7;                               // channel ci0 + sid; contributors: threads s with s % SX == gq
        double t1 = 0.0, t2 = 0.0;
        for (int k = 0; k < NS / SX; ++k) { t1 += (double)r2[e * NS + k * SX + gq]; t2 += (double)r2[(8 + e) * NS + k * SX + gq]; }
        const int ch = ci0 + sid;
        double s2 = t2;
        if (a.in_scale) {
          const double sc = (double)a.in_scale[ch], sh = (double)a.in_shift[ch], mu = (double)a.bst_mean[ch], rs = (double)a.bst_rstd[ch];
          s2 = sc != 0.0 ? (t2 - (sh + sc * mu) * t1) * (rs / sc) : 0.0;
        }
        satcv_stat_t* rowp = a.bst_sums + (size_t)(blockIdx.x % SATCV_STAT_ROWS) * 2 * a.bst_ld;
        atomicAdd(rowp + ch, (satcv_stat_t)t1);
        atomicAdd(rowp + a.bst_ld + ch, (satcv_stat_t)s2);
      }
    }
    return;
  }

  // ================================================================ matrix waves
  const int r = lane & 31, hh = lane >> 5, g4 = lane >> 4, l16 = lane & 15;
  // weight gradient: tiles id = wave + 8 t -> (ci tile id & 1, k tile id >> 1); transposing reads as in conv_bwd_fused.hip
  constexpr int NCT = CBLK / 32, WT = (NCT * COUT4 / 32) / 8;
  f32x16 wacc[WT];
#pragma unroll
  for (int t = 0; t < WT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) wacc[t][i] = 0.f;
  const int gi = lane >> 4, i16 = lane & 15;
  const int chb = 16 * (gi & 1) + 4 * (i16 & 3);
  const int pk = 8 * (gi >> 1) + (i16 >> 2);
  const int tr_o = (chb >> 3) * STR + pk * 8 + (chb & 7);
  // data gradient: 16 x 16 blocks; this wave's pixel block pb and its NBW channel blocks from nb0
  constexpr int NPB = PX / 16, WPP = 8 / NPB, NBT = CBLK / 16, NBW = NBT / WPP;
  static_assert(NPB * WPP == 8 && NBW * WPP == NBT && WT * 8 == NCT * COUT4 / 32, "block distribution");
  const int pb = wave / WPP, nb0 = (wave % WPP) * NBW;
  const int da_off = g4 * STR + (pb * 16 + l16) * 8;                       // A: plane (4 ks + g4), pixel pb * 16 + l16
  const int db_off = (g4 * CBLK + nb0 * 16 + l16) * 8;                     // B: [slot][CBLK][8]
  __syncthreads();                                                          // (1)
  __syncthreads();                                                          // (2)
  for (int t = t_lo; t < t_hi; ++t) {
    const int b = (t - t_lo) & 1;
    const T* ldsD = lds0 + b * G::BUF_E;
    const T* ldsX = ldsD + SD * STR;
    // ---- weight gradient: K = the PX pixels of the tile, 16 per step
#pragma unroll
    for (int tt = 0; tt < WT; ++tt) {
      const int id = wave + tt * 8, ct = id % NCT, kt = id / NCT;
      const T* xa = ldsX + ct * 4 * STR + tr_o;
      const T* da = ldsD + kt * 4 * STR + tr_o;
      bf16x4 fa[2][2], fb[2][2];
      auto read_k = [&](int u, int buf) {
        fa[buf][0] = ct_tr_read4(xa + u * 128); fa[buf][1] = ct_tr_read4(xa + u * 128 + 32);
        fb[buf][0] = ct_tr_read4(da + u * 128); fb[buf][1] = ct_tr_read4(da + u * 128 + 32);
      };
      read_k(0, 0);
#pragma unroll
      for (int u = 0; u < PX / 16; ++u) {
        asm volatile("" ::: "memory");
        if (u + 1 < PX / 16) read_k(u + 1, (u + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 afr = __builtin_shufflevector(fa[u & 1][0], fa[u & 1][1], 0, 1, 2, 3, 4, 5, 6, 7);
        const bf16x8 bfr = __builtin_shufflevector(fb[u & 1][0], fb[u & 1][1], 0, 1, 2, 3, 4, 5, 6, 7);
        wacc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr, wacc[tt], 0, 0, 0);
      }
    }
    // ---- data gradient: K = 4 Cout, 32 per step
    f32x4 acc[NBW];
#pragma unroll
    for (int n = 0; n < NBW; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      constexpr int KS = COUT4 / 32;
      FragT<T> af[2], bf[2][NBW];
      auto read_step = [&](int ks, int buf) {
        af[buf] = lds_frag<T>(ldsD + ks * 4 * STR + da_off);
#pragma unroll
        for (int n = 0; n < NBW; ++n) bf[buf][n] = lds_frag<T>(ldsW + ks * 4 * CBLK * 8 + db_off + n * 16 * 8);
      };
      read_step(0, 0);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        asm volatile("" ::: "memory");
        if (ks + 1 < KS) read_step(ks + 1, (ks + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NBW; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks & 1].v, bf[ks & 1][n].v, acc[n], 0, 0, 0);
      }
    }
    // accumulators -> bf16 -> staging tile b: block (pb, nb): rows 4 g4 + j, column l16
    {
      T* op = ldsO + b * G::O_E + (pb * 16 + 4 * g4) * OPITCH + nb0 * 16 + l16;
#pragma unroll
      for (int n = 0; n < NBW; ++n)
#pragma unroll
        for (int j = 0; j < 4; ++j) op[j * OPITCH + n * 16] = (T)acc[n][j];
    }
    __syncthreads();
  }
  // ---- this workgroup's partial weight gradient: ws[slab][ci][k]
#pragma unroll
  for (int tt = 0; tt < WT; ++tt) {
    const int id = wave + tt * 8, ct = id % NCT, kt = id / NCT;
    float* dst = a.ws + ((size_t)slab * a.cin + ci0 + ct * 32) * COUT4 + kt * 32 + r;
#pragma unroll
    for (int i = 0; i < 16; ++i) dst[(size_t)((i & 3) + 8 * (i >> 2) + 4 * hh) * COUT4] = wacc[tt][i];
  }
  __syncthreads();                                                          // (3)
}

// ------------------------------------------------------------------ host side
static int g_ct_ncu = 0;
template <int COUT4, int PX, int CBLK>
static int ctbf_launch(const satcv_ctbf_desc* d, hipStream_t st, bool query, int64_t* ws_bytes, satcv_reduce_job* job) {
  using G = CtbfGeom<COUT4, PX, CBLK>;
  static_assert(G::LDS <= 160 * 1024, "buffers + operand image exceed the LDS");
  if (d->w_ % PX != 0 || d->cin % CBLK != 0) return SATCV_ERR_UNSUPPORTED;
  if (!g_ct_ncu) {
    int dev = 0; hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) { (void)hipGetLastError(); g_ct_ncu = 256; }
    else g_ct_ncu = p.multiProcessorCount;
  }
  const int nblk = d->cin / CBLK;
  const long long tiles = (long long)d->n * d->h * (d->w_ / PX);
  long long slabs = g_ct_ncu / nblk; if (slabs < 1) slabs = 1; if (slabs > tiles) slabs = tiles;
  const size_t need = (size_t)slabs * d->cin * COUT4 * sizeof(float);
  if (job) { reduce_job_fill(job, d->workspace, d->dw, (int)slabs, 1, d->cin, COUT4, d->cin, COUT4, 1, d->accumulate); return SATCV_OK; }
  if (query) { *ws_bytes = (int64_t)need; return SATCV_OK; }
  SATCV_CHECK((size_t)d->workspace_bytes >= need, "convt_bwd_fused: workspace %lld < %zu", (long long)d->workspace_bytes, need);
  CtbfArgs a;
  memset(&a, 0, sizeof(a));
  a.g = d->g; a.ldg = d->ldg; a.yup = d->yup; a.ldy = d->ldy;
  a.bn_scale = d->bn_scale; a.bn_shift = d->bn_shift; a.bn_mean = d->bn_mean; a.bn_rstd = d->bn_rstd; a.bn_c1 = d->bn_c1; a.bn_c2 = d->bn_c2; a.linear = d->linear;
  a.x = d->x; a.ldx = d->ldx; a.in_scale = d->in_scale; a.in_shift = d->in_shift; a.in_relu = d->in_relu;
  a.w = d->w_dgrad; a.npad = d->w_npad; a.dx = d->dx; a.lddx = d->lddx; a.ws = d->workspace;
  a.n = d->n; a.h = d->h; a.w_ = d->w_; a.cin = d->cin;
  a.bst_sums = d->bst_sums; a.bst_ld = d->bst_sums_ld; a.bst_mean = d->bst_mean; a.bst_rstd = d->bst_rstd;
  a.tiles = (int)tiles; a.nblk = nblk; a.slabs = (int)slabs;
  auto kern = convt_bwd_fused_kernel<COUT4, PX, CBLK>;
  { const int rc = satcv_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), G::LDS); if (rc) return rc; }
  const double flops = 4.0 * d->n * d->h * d->w_ * (double)COUT4 * d->cin;
  satcv_prof_begin(1, flops, st);
  hipLaunchKernelGGL(kern, dim3((unsigned)(slabs * nblk)), dim3(768), G::LDS, st, a);
  satcv_prof_end(1, st);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { satcv_set_error("convt_bwd_fused launch: %s", hipGetErrorString(e)); return SATCV_ERR_HIP; }
  if (d->defer_reduce) return SATCV_OK;
  return wgrad_reduce_slabs_t(d->workspace, d->dw, (int)slabs, d->cin, COUT4, d->cin, COUT4, d->accumulate, st);
}

static int ctbf_dispatch(const satcv_ctbf_desc* d, hipStream_t st, bool query, int64_t* ws_bytes, satcv_reduce_job* job = nullptr) {
  if (!d || d->dtype != SATCV_BF16 || d->f != 2 || d->cin <= 0 || d->cin % 64 != 0 || d->n <= 0 || d->h <= 0 || d->w_ <= 0) return SATCV_ERR_UNSUPPORTED;
  if (!satcv_pixels_ok(d->n, d->h, d->w_, 2) || d->cin > (1 << 16)) return SATCV_ERR_UNSUPPORTED;
  if (d->ldg % 8 || d->ldy % 8 || d->ldx % 8 || d->lddx % 8 || d->w_npad < d->cin || d->w_npad % 8) return SATCV_ERR_UNSUPPORTED;
  if (((uintptr_t)d->g | (uintptr_t)d->yup | (uintptr_t)d->x | (uintptr_t)d->dx | (uintptr_t)d->w_dgrad) % 16) return SATCV_ERR_UNSUPPORTED;
  if ((long long)(2 * d->w_ + 2 * 128 + 2) * (d->ldg > d->ldy ? d->ldg : d->ldy) >= (1LL << 31)) return SATCV_ERR_UNSUPPORTED;
  if (d->bst_sums && !(d->bst_sums_ld >= d->cin && (d->in_scale == nullptr || (d->in_shift && d->in_relu && d->bst_mean && d->bst_rstd)))) return SATCV_ERR_UNSUPPORTED;
  // tile and channel-block shapes per level (two tiles of loads in flight per workgroup): Cout 32 -> 64 pixels x 64 input channels; Cout 64 ->
  // 32 pixels x 128 input channels (ONE channel block for the U-Net's 128-channel input: the g / y tile is formed once, not once per block).
  // Cout 128 (512 gradient channels: 64 KB of operand image per 64 input channels, four blocks each re-forming the tile) measured no
  // faster than the three launches it would replace (141 us against 152 at batch 64) and is not instantiated
  switch (d->cout) {
    case 32: return ctbf_launch<128, 64, 64>(d, st, query, ws_bytes, job);
    case 64: return d->cin % 128 == 0 ? ctbf_launch<256, 32, 128>(d, st, query, ws_bytes, job) : ctbf_launch<256, 32, 64>(d, st, query, ws_bytes, job);
    default: return SATCV_ERR_UNSUPPORTED;
  }
}

extern "C" int64_t satcv_convt_bwd_fused_workspace(const satcv_ctbf_desc* d) {
  int64_t nb = -1;
  if (ctbf_dispatch(d, nullptr, true, &nb) != SATCV_OK) return -1;
  return nb;
}
extern "C" int satcv_convt_bwd_fused_reduce_job(const satcv_ctbf_desc* d, satcv_reduce_job* job) {
  SATCV_CHECK(d && job && d->dw && d->workspace, "convt_bwd_fused_reduce_job: null pointer");
  return ctbf_dispatch(d, nullptr, false, nullptr, job);
}
extern "C" int satcv_convt_bwd_fused(const satcv_ctbf_desc* d, void* stream) {
  SATCV_CHECK(d && d->g && d->yup && d->x && d->dx && d->dw && d->w_dgrad && d->workspace, "convt_bwd_fused: null pointer");
  SATCV_CHECK(d->bn_scale && d->bn_shift && d->bn_mean && d->bn_rstd && d->bn_c1 && d->bn_c2, "convt_bwd_fused: BatchNorm coefficients missing");
  const int rc = ctbf_dispatch(d, reinterpret_cast<hipStream_t>(stream), false, nullptr);
  if (rc == SATCV_ERR_UNSUPPORTED) satcv_set_error("convt_bwd_fused: shape outside the kernel's limits (bf16, f = 2, Cout 32 / 64, Cin %% 64 == 0, row length a multiple of the tile)");
  return rc;
}
