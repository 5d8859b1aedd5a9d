// Thin transposed convolutions (Conv2DTranspose, kernel == stride == 2, of the full- and half-resolution decoder levels:
// utils/model_tools.py:306) as a streaming kernel.
//
// These layers are HBM-bound by a wide margin (64 -> 4 x 32 channels at 128 x 128: 402 MB of tensors, 17 GFLOP) and ran at 0.3-0.36 of
// their HBM roofline on the tiled implicit-GEMM kernel, whose tile pays a gather-table set-up, two barriers per chunk and an LDS-staged
// epilogue for 4-8 MFMAs per wave.  Here a WAVE is the unit of work and nothing in the loop synchronises:
//   * a wave owns strips of 32 consecutive input pixels (one row segment).  Its activation fragments come straight from global memory
//     in MFMA operand layout (lane = pixel, 8 consecutive channels: one 16-byte load per k-step), get the producing layer's
//     BatchNorm + ReLU in registers and are never staged in LDS;
//   * the whole weight tensor (16-64 KB) is resident in LDS in fragment layout;
//   * every product is formed in BOTH orientations from the same two fragments: W^T x X^T leaves the output with the pixel on the
//     lane and the channels in the registers -- after the bias and bf16 packing, v_permlane32_swap pairs give each lane 8 consecutive
//     channels, i.e. 16-byte stores straight to the depth-to-space position, no LDS transpose --, and X x W leaves the channel on the
//     lane, which makes the BatchNorm sum / sum-of-squares of the stored values an in-lane accumulation (two registers per channel
//     tile instead of a cross-lane reduction per strip).  The second product costs MFMA time the layer does not use anyway (the
//     matrix pipe is < 25 % busy at the HBM rate) and no bytes;
//   * the next strip's loads are in flight while the current one is multiplied and stored.
#include "igemm_common.hpp"
#include <cstdlib>

struct ConvtArgs {
  const void* x; const float* in_scale; const float* in_shift; int in_relu;
  const void* w; const float* bias;
  const float* out_scale; int out_relu;      // folded inference graph: y = relu(acc * out_scale + bias)
  void* y; int ldy;
  satcv_stat_t* stats; int stats_ld;
  int h, w_;                  // input map
  int total_strips;           // n * h * w_ / 32
};

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v;
  v[0] = (bf16)a; v[1] = (bf16)b;
  return __builtin_bit_cast(unsigned, v);
}

// NSPLIT > 1: the 4 COUT output columns are cut into NSPLIT blocks, one per workgroup of a group of NSPLIT (256 -> 4 x 128 channels: 256 KB of
// weights, 64 KB per workgroup = one position; the input strip is read by the group's four workgroups, three times out of four from L2)
template <int CIN, int COUT, int NW, int WPS, int NSPLIT = 1>
__global__ __launch_bounds__(NW * 64, WPS) void convt_thin_kernel(const ConvtArgs a) {
  typedef bf16 T;
  constexpr int KS = CIN / 16, NCOL = 4 * COUT, NTW = NCOL / 32 / NSPLIT, NT = NTW, NCOLW = NTW * 32, NTHREADS = NW * 64;
  static_assert(NTW % 2 == 0 && (NTW * 32) % 64 == 0 && (NSPLIT == 1 || (NTW * 32) % COUT == 0 || COUT % (NTW * 32) == 0), "column blocks of whole 64-channel pairs");
  constexpr size_t W_BYTES = (size_t)CIN * NCOLW * sizeof(T);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* ldsW = reinterpret_cast<T*>(smem_raw);                                   // [CIN / 8][NCOLW][8]: this workgroup's columns of the packed forward image
  float* tab = reinterpret_cast<float*>(smem_raw + W_BYTES);                  // scale[CIN], shift[CIN], bias[COUT]; later the statistics
  // wave-private output staging: 32 input pixels x 128 bytes (a pair of channel tiles), rows padded to 144 bytes so that the 16-byte
  // stores of 8 neighbouring pixels fall on distinct banks.  No barrier: only this wave touches its region, and a wave's LDS operations
  // execute in order.
  constexpr int OPITCH = 144, TAB_FLOATS = (2 * CIN + 2 * COUT > 2 * NW * COUT ? 2 * CIN + 2 * COUT : 2 * NW * COUT);
  // (the loads stay fragment-shaped -- 32-byte pieces of 32 lines per instruction: whole-line loads redistributed through these rows
  //  measured the same, 89.4 vs 90.6 and 48.5 vs 48.1 us)
  unsigned char* ldsO = smem_raw + W_BYTES + (size_t)TAB_FLOATS * sizeof(float) + (size_t)(threadIdx.x >> 6) * (32 * OPITCH);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;

  {
    const T* wp = reinterpret_cast<const T*>(a.w);
    const int col0 = (int)((blockIdx.x >> 3) % NSPLIT) * NCOLW;
    for (int it = tid; it < (CIN / 8) * NCOLW; it += NTHREADS) lstore8<T>(ldsW + (size_t)it * 8, gload8<T>(wp + ((size_t)(it / NCOLW) * NCOL + col0 + it % NCOLW) * 8));
    for (int ch = tid; ch < CIN; ch += NTHREADS) {
      tab[ch] = a.in_scale ? a.in_scale[ch] : 1.f;
      tab[CIN + ch] = a.in_scale ? a.in_shift[ch] : 0.f;
    }
    for (int ch = tid; ch < COUT; ch += NTHREADS) { tab[2 * CIN + ch] = a.bias ? a.bias[ch] : 0.f; tab[2 * CIN + COUT + ch] = a.out_scale ? a.out_scale[ch] : 1.f; }
  }
  __syncthreads();
  const bool xaff = a.in_scale != nullptr;
  const unsigned relu_lim = a.in_relu != 0 ? 0u : 0x80008000u;
  const unsigned orelu_lim = a.out_relu != 0 ? 0u : 0x80008000u;
  const bool want_stats = a.stats != nullptr;

  // ---- this wave's strips: XCD-aware contiguous ranges (blocks b and b + 8 share an XCD)
  // (NSPLIT > 1: the workgroups of a group are neighbours in the launch order -- they read the same strips at about the same time)
  // (NSPLIT > 1: the launch has a multiple of 8 NSPLIT workgroups; a group's members are blocks b, b + 8, ... -- one XCD, one L2)
  const int G = gridDim.x / NSPLIT;
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, by = jx % NSPLIT, gj = jx / NSPLIT;
  const int gnt0 = __builtin_amdgcn_readfirstlane(by * NTW);                  // first global channel tile of this workgroup
  const int nx = G >> 3, remx = G & 7;
  const int bid = (xcd < remx ? xcd * (nx + 1) : remx * (nx + 1) + (xcd - remx) * nx) + gj;
  const int gw = __builtin_amdgcn_readfirstlane(bid * NW + wave), GW = G * NW;
  const int per = a.total_strips / GW, extra = a.total_strips % GW;
  const int t_lo = gw * per + (gw < extra ? gw : extra), t_hi = t_lo + per + (gw < extra ? 1 : 0);

  const T* xlane = reinterpret_cast<const T*>(a.x) + (size_t)r * CIN + hh * 8;      // + strip * 32 * CIN + ks * 16
  const T* wlane = ldsW + (size_t)(hh * NCOLW + r) * 8;                              // + (ks * 2 * NCOLW + nt * 32) * 8
  const int wo = 2 * a.w_;
  float st1[NT], st2[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) { st1[n] = 0.f; st2[n] = 0.f; }

  constexpr int XSTEP = 16;
  Raw8<T> xr[KS];
  if (t_lo < t_hi) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xr[ks] = gload8<T>(xlane + (size_t)t_lo * 32 * CIN + ks * XSTEP);
  }
  for (int t = t_lo; t < t_hi; ++t) {
    // ---- the strip's fragments: BatchNorm + ReLU of the producing layer on the loaded registers
    bf16x8 xf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      Raw8<T> v = xr[ks];
      if (xaff) {
        int toff = ks * 16 + hh * 8;
        asm volatile("" : "+v"(toff));                                        // (read per strip: hoisted, the 16 KS values would stay live)
        const float4* sp = reinterpret_cast<const float4*>(tab + toff);
        const float4* hp = reinterpret_cast<const float4*>(tab + CIN + toff);
        const float4 s0 = sp[0], s1 = sp[1], h0 = hp[0], h1 = hp[1];
        const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
        v = affine8_lim(v, sc, sh, relu_lim);
      }
      xf[ks] = __builtin_bit_cast(bf16x8, v.q[0]);
    }
    // ---- the next strip's loads
    if (t + 1 < t_hi) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) xr[ks] = gload8<T>(xlane + (size_t)(t + 1) * 32 * CIN + ks * XSTEP);
    }
    // strip origin: pixel 32 t = (n, y, x0) of the input map; output rows 2 y, 2 y + 1
    const int p0 = t * 32;
    const int row = p0 / a.w_, x0 = p0 - row * a.w_;                          // row = n * h + y (the output has 2 h rows per image: 2 row is right)
    // channel tiles in pairs: the pair's four 32-byte pieces complete whole 128-byte lines of the output (two neighbouring pixels of
    // 32 channels, or one pixel of 64), and all four stores are issued together -- issued a tile apart, the pieces reached the L2
    // microseconds apart under load and part of the lines were evicted half written (3.1 instead of 3.5-4.2 TB/s)
#pragma unroll
    for (int np = 0; np < NT / 2; ++np) {
      f32x16 accT[2], accD[2];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) { accT[u][i] = 0.f; accD[u][i] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wlane + (size_t)(ks * 2 * NCOLW + (2 * np + u) * 32) * 8);
          accT[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf[ks], accT[u], 0, 0, 0);       // [channel][pixel]: stores
          if (want_stats) accD[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[ks], wf, accD[u], 0, 0, 0);      // [pixel][channel]: statistics
        }
      }
      // ---- stores: rows of accT are channels cb + 8 g + 4 hh + e (register 4 g + e), the column is this lane's pixel.  After the bias
      // and the bf16 packing a half exchange (v_permlane32_swap) gives every lane 8 consecutive channels of its pixel; the pair's 128 bytes
      // per pixel go through the wave's staging rows and leave as whole lines: a store instruction writes 8 pixels x 128 contiguous bytes
      // (32-byte pieces straight from the registers ran the L2 write path at a quarter of its width: 3.4 TB/s)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int cb = ((gnt0 + 2 * np + u) * 32) % COUT;
        uint2 o2[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 b4 = *reinterpret_cast<const float4*>(tab + 2 * CIN + cb + 8 * g + 4 * hh);
          const float4 m4 = *reinterpret_cast<const float4*>(tab + 2 * CIN + COUT + cb + 8 * g + 4 * hh);      // (1 without out_scale: acc * 1 + b == acc + b)
          o2[g].x = relu_pk_bf16(pk_bf16(fmaf(accT[u][4 * g], m4.x, b4.x), fmaf(accT[u][4 * g + 1], m4.y, b4.y)), orelu_lim);
          o2[g].y = relu_pk_bf16(pk_bf16(fmaf(accT[u][4 * g + 2], m4.z, b4.z), fmaf(accT[u][4 * g + 3], m4.w, b4.w)), orelu_lim);
        }
#pragma unroll
        for (int k = 0; k < 4; k += 2) {
          // afterwards lanes 0-31 hold channels 8 k ... 8 k + 7 of their pixel, lanes 32-63 channels 8 k + 8 ... 8 k + 15
          const uint2 lo = o2[k], hi = o2[k + 1];
          auto sx = __builtin_amdgcn_permlane32_swap(lo.x, hi.x, false, false);
          auto sy = __builtin_amdgcn_permlane32_swap(lo.y, hi.y, false, false);
          *reinterpret_cast<uint4*>(ldsO + r * OPITCH + u * 64 + k * 16 + hh * 16) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
        }
      }
      {
        // the pair's position: channel tiles 2 np, 2 np + 1 are the two horizontal positions (j = 0, 1) of one output row for COUT = 32,
        // the two channel halves of one position for COUT = 64
        const int ij = ((gnt0 + 2 * np) * 32) / COUT, cbp = ((gnt0 + 2 * np) * 32) % COUT;      // (cbp: 0 unless a position has more than 64 channels)
        T* yrow = reinterpret_cast<T*>(a.y) + ((size_t)(2 * row + (ij >> 1)) * wo + 2 * x0 + (COUT == 32 ? 0 : (ij & 1))) * a.ldy + (COUT == 32 ? 0 : cbp);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int c = lane + 64 * m, q = c >> 3, piece = c & 7;              // 16-byte piece of input pixel q
          const uint4 v = *reinterpret_cast<const uint4*>(ldsO + q * OPITCH + piece * 16);
          const int eoff = COUT == 32 ? (2 * q + (piece >> 2)) * a.ldy + (piece & 3) * 8 : 2 * q * a.ldy + piece * 8;
          *reinterpret_cast<uint4*>(yrow + eoff) = v;
        }
      }
      // ---- statistics of the stored values: column of accD = channel cb + r, its 16 registers are pixels
      if (want_stats) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int cb = ((gnt0 + 2 * np + u) * 32) % COUT;
          const float bv = tab[2 * CIN + cb + r];
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float fv = (float)(T)(accD[u][i] + bv);
            s1 += fv; s2 += fv * fv;
          }
          st1[2 * np + u] += s1; st2[2 * np + u] += s2;
        }
      }
    }
  }
  // ---- statistics: lane halves, channel tiles of one channel and the waves of the workgroup summed in a fixed order; one pair of
  //      atomics per channel and workgroup
  if (want_stats) {
    float c1[COUT / 32], c2[COUT / 32];
#pragma unroll
    for (int u = 0; u < COUT / 32; ++u) { c1[u] = 0.f; c2[u] = 0.f; }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int grp = (gnt0 + nt) % (COUT / 32);
#pragma unroll
      for (int u = 0; u < COUT / 32; ++u) { c1[u] += u == grp ? st1[nt] : 0.f; c2[u] += u == grp ? st2[nt] : 0.f; }
    }
    __syncthreads();                                                          // (the tables are dead: every wave is past its last strip)
    float* red = tab;                                                         // [NW][2][COUT]
#pragma unroll
    for (int u = 0; u < COUT / 32; ++u) {
      const float s1 = c1[u] + __shfl_xor(c1[u], 32, 64), s2 = c2[u] + __shfl_xor(c2[u], 32, 64);
      if (hh == 0) { red[(wave * 2 + 0) * COUT + u * 32 + r] = s1; red[(wave * 2 + 1) * COUT + u * 32 + r] = s2; }
    }
    __syncthreads();
    if (tid < COUT) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) { t1 += red[(w * 2 + 0) * COUT + tid]; t2 += red[(w * 2 + 1) * COUT + tid]; }
      satcv_stat_t* rowp = a.stats + (size_t)(blockIdx.x % SATCV_STAT_ROWS) * 2 * a.stats_ld;
      atomicAdd(rowp + tid, (satcv_stat_t)t1);
      atomicAdd(rowp + a.stats_ld + tid, (satcv_stat_t)t2);
    }
  }
}

template <int CIN, int COUT, int NW, int WPS, int NSPLIT = 1>
static int convt_thin_cfg(const ConvtArgs& ca, hipStream_t st) {
  constexpr size_t lds = (size_t)CIN * 4 * COUT / NSPLIT * sizeof(bf16) + (size_t)(2 * CIN + 2 * COUT > 2 * NW * COUT ? 2 * CIN + 2 * COUT : 2 * NW * COUT) * sizeof(float) + (size_t)NW * 32 * 144;
  auto kern = convt_thin_kernel<CIN, COUT, NW, WPS, NSPLIT>;
  { const int rc = satcv_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds); if (rc) return rc; }
  static const int cus = [] {
    int dev = 0, v = 256;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) v = p.multiProcessorCount;
    return v;
  }();
  int per_cu = WPS * 4 / NW;                                                  // resident workgroups: WPS waves per SIMD
  while (per_cu > 1 && (size_t)per_cu * lds > 150 * 1024) --per_cu;
  if (per_cu < 1) per_cu = 1;
  long long grid = (long long)cus * per_cu;
  const long long need = (ca.total_strips + NW - 1) / NW * NSPLIT;             // a wave should see at least one strip
  if (grid > need) grid = need;
  if (NSPLIT > 1) { grid -= grid % (8 * NSPLIT); if (grid < 8 * NSPLIT) grid = 8 * NSPLIT; }      // whole groups on every XCD
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), lds, st, ca);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { satcv_set_error("convt_thin launch: %s", hipGetErrorString(e)); return SATCV_ERR_HIP; }
  return SATCV_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Data gradient of the same layers: dx[q][ci] = sum over the 2 x 2 positions and co of dy[2 q + (i, j)][co] w[i][j][ci][co], K = 4 COUT.
// Same structure: a wave owns strips of 32 input-resolution pixels, the dy fragments come straight from global memory (lane = pixel,
// 8 consecutive channels of one position), the data-gradient weight image is resident in LDS, the product is formed with the pixel on
// the lane, packed, exchanged between the lane halves and sent through the wave's staging rows so that every store instruction
// writes whole lines of dx.  The fused BatchNorm-backward sums of the layer below (bst_*, satcv.h) are formed from the staged row
// pieces: a lane reads back 8 consecutive channels of a pixel -- always the same 8 channels --, loads that layer's raw output at the same
// address and accumulates sum g [a > 0] and sum g xhat in registers; lanes, waves and workgroups are combined once at the end.
struct ConvtDgradArgs {
  const void* dy; int lddy;            // [n][2 h][2 w_][lddy], COUT channels used
  const void* w;                       // data-gradient image [4 COUT / 8][CIN][8]
  void* dx; int lddx;
  satcv_stat_t* stats; int stats_ld;
  const void* bst_y; int bst_ld;
  const float* bst_scale; const float* bst_shift; const float* bst_mean; const float* bst_rstd; int bst_relu;
  int h, w_, total_strips;
};

template <int CIN, int COUT, int NW, int WPS>
__global__ __launch_bounds__(NW * 64, WPS) void convt_thin_dgrad_kernel(const ConvtDgradArgs a) {
  typedef bf16 T;
  constexpr int K = 4 * COUT, KS = K / 16, NT = CIN / 32, NP = NT / 2, NTHREADS = NW * 64, KPP = COUT / 16;      // KPP: k-steps per position
  constexpr size_t W_BYTES = (size_t)K * CIN * sizeof(T);
  constexpr int OPITCH = 144, TAB_FLOATS = (4 * CIN > 2 * NW * CIN ? 4 * CIN : 2 * NW * CIN);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* ldsW = reinterpret_cast<T*>(smem_raw);                                   // [K / 8][CIN][8]
  float* tab = reinterpret_cast<float*>(smem_raw + W_BYTES);                  // scale, shift of the layer below [2][CIN]; later the sums
  unsigned char* ldsO = smem_raw + W_BYTES + (size_t)TAB_FLOATS * sizeof(float) + (size_t)(threadIdx.x >> 6) * (32 * OPITCH);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const bool bst = a.bst_y != nullptr;
  {
    const T* wp = reinterpret_cast<const T*>(a.w);
    for (int it = tid; it < (K / 8) * CIN; it += NTHREADS) lstore8<T>(ldsW + (size_t)it * 8, gload8<T>(wp + (size_t)it * 8));
    if (bst) {
      for (int ch = tid; ch < CIN; ch += NTHREADS) {
        tab[ch] = a.bst_scale[ch]; tab[CIN + ch] = a.bst_shift[ch];
      }
    }
  }
  __syncthreads();
  const float lin_lo = a.bst_relu == 0 ? -INFINITY : 0.f;

  const int G = gridDim.x;
  const int xcd = blockIdx.x & 7, nx = G >> 3, remx = G & 7;
  const int bid = (xcd < remx ? xcd * (nx + 1) : remx * (nx + 1) + (xcd - remx) * nx) + (blockIdx.x >> 3);
  const int gw = __builtin_amdgcn_readfirstlane(bid * NW + wave), GW = G * NW;
  const int per = a.total_strips / GW, extra = a.total_strips % GW;
  const int t_lo = gw * per + (gw < extra ? gw : extra), t_hi = t_lo + per + (gw < extra ? 1 : 0);

  const int wo = 2 * a.w_;
  const T* wlane = ldsW + (size_t)(hh * CIN + r) * 8;                         // + (ks * 2 * CIN + nt * 32) * 8
  // this lane's dy item of k-step ks: position ij = ks / KPP, channels (ks % KPP) * 16 + hh * 8 of output pixel (2 y + i, 2 (x0 + r) + j)
  const unsigned dy_lane = (unsigned)(2 * r) * a.lddy + hh * 8;
  auto issue = [&](Raw8<T> (&dst)[KS], int t) {
    const int p0 = t * 32;
    const int row = p0 / a.w_, x0 = p0 - row * a.w_;
    const T* base = reinterpret_cast<const T*>(a.dy) + ((size_t)(2 * row) * wo + 2 * x0) * a.lddy + dy_lane;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int ij = ks / KPP, cq = (ks % KPP) * 16;
      dst[ks] = gload8<T>(base + ((size_t)(ij >> 1) * wo + (ij & 1)) * a.lddy + cq);
    }
  };
  float bs1[NP][8], bs2[NP][8];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int e = 0; e < 8; ++e) { bs1[p][e] = 0.f; bs2[p][e] = 0.f; }

  Raw8<T> dr[KS];
  if (t_lo < t_hi) issue(dr, t_lo);
  for (int t = t_lo; t < t_hi; ++t) {
    // one register set for dy: a k-step's item is re-requested for the next strip as soon as its MFMAs have read it
    const bool more = t + 1 < t_hi;
    const int p1 = (more ? t + 1 : t) * 32;
    const int row1 = p1 / a.w_, x1 = p1 - row1 * a.w_;
    const T* base1 = reinterpret_cast<const T*>(a.dy) + ((size_t)(2 * row1) * wo + 2 * x1) * a.lddy + dy_lane;
    const size_t pix0 = (size_t)t * 32;
    const int piece = lane & 7;
    // the raw outputs of the layer below at the pieces this lane will store: requested BEFORE the next strip's dy (the in-order load
    // counter would otherwise make their use wait for that whole strip)
    uint4 vv[NP][4];
    if (bst) {
      const T* vrow = reinterpret_cast<const T*>(a.bst_y) + pix0 * a.bst_ld + piece * 8;
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int m = 0; m < 4; ++m) vv[p][m] = *reinterpret_cast<const uint4*>(vrow + (size_t)((lane >> 3) + 8 * m) * a.bst_ld + p * 64);
    }
    f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 df = __builtin_bit_cast(bf16x8, dr[ks].q[0]);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wlane + (size_t)(ks * 2 * CIN + n * 32) * 8);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, df, acc[n], 0, 0, 0);       // [ci][pixel]
      }
      if (more) {
        const int ij = ks / KPP, cq = (ks % KPP) * 16;
        dr[ks] = gload8<T>(base1 + ((size_t)(ij >> 1) * wo + (ij & 1)) * a.lddy + cq);
      }
    }
#pragma unroll
    for (int np = 0; np < NP; ++np) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        uint2 o2[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          o2[g].x = pk_bf16(acc[2 * np + u][4 * g], acc[2 * np + u][4 * g + 1]);
          o2[g].y = pk_bf16(acc[2 * np + u][4 * g + 2], acc[2 * np + u][4 * g + 3]);
        }
#pragma unroll
        for (int k = 0; k < 4; k += 2) {
          const uint2 lo = o2[k], hi = o2[k + 1];
          auto sx = __builtin_amdgcn_permlane32_swap(lo.x, hi.x, false, false);
          auto sy = __builtin_amdgcn_permlane32_swap(lo.y, hi.y, false, false);
          *reinterpret_cast<uint4*>(ldsO + r * OPITCH + u * 64 + k * 16 + hh * 16) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
        }
      }
      // the pair's 64 channels of the strip: 8 pixels x 128 bytes per store instruction
      T* xrow = reinterpret_cast<T*>(a.dx) + pix0 * a.lddx + np * 64;
      float sc[8], sh[8];
      if (bst) {
        int toff = np * 64 + piece * 8;
        asm volatile("" : "+v"(toff));
        const float4* tp = reinterpret_cast<const float4*>(tab + toff);
        const float4 a0 = tp[0], a1 = tp[1], b0 = tp[CIN / 4], b1 = tp[CIN / 4 + 1];
        sc[0] = a0.x; sc[1] = a0.y; sc[2] = a0.z; sc[3] = a0.w; sc[4] = a1.x; sc[5] = a1.y; sc[6] = a1.z; sc[7] = a1.w;
        sh[0] = b0.x; sh[1] = b0.y; sh[2] = b0.z; sh[3] = b0.w; sh[4] = b1.x; sh[5] = b1.y; sh[6] = b1.z; sh[7] = b1.w;
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int q = (lane >> 3) + 8 * m;                                    // 16-byte piece `piece` of pixel q
        const uint4 v = *reinterpret_cast<const uint4*>(ldsO + q * OPITCH + piece * 16);
        *reinterpret_cast<uint4*>(xrow + (size_t)q * a.lddx + piece * 8) = v;
        if (bst) {
          const bf16x8 g8 = __builtin_bit_cast(bf16x8, v), y8 = __builtin_bit_cast(bf16x8, vv[np][m]);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float fv = (float)g8[e], yv = (float)y8[e];
            const float gg = yv * sc[e] + sh[e] > lin_lo ? fv : 0.f;
            bs1[np][e] += gg; bs2[np][e] += gg * yv;            // sum g y: turned into sum g xhat = rstd (sum g y - mean sum g) at the end
          }
        }
      }
    }
  }
  // ---- fused sums: the lanes holding one channel group (lane % 8), then the waves, in a fixed order; one pair of atomics per channel
  if (bst) {
    __syncthreads();
    float* red = tab;                                                         // [NW][2][CIN]
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float s1 = bs1[p][e], s2 = bs2[p][e];
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        if (lane < 8) { red[(wave * 2 + 0) * CIN + p * 64 + lane * 8 + e] = s1; red[(wave * 2 + 1) * CIN + p * 64 + lane * 8 + e] = s2; }
      }
    __syncthreads();
    if (tid < CIN) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) { t1 += red[(w * 2 + 0) * CIN + tid]; t2 += red[(w * 2 + 1) * CIN + tid]; }
      satcv_stat_t* rowp = a.stats + (size_t)(blockIdx.x % SATCV_STAT_ROWS) * 2 * a.stats_ld;
      atomicAdd(rowp + tid, (satcv_stat_t)t1);
      atomicAdd(rowp + a.stats_ld + tid, (satcv_stat_t)(((double)t2 - (double)a.bst_mean[tid] * (double)t1) * (double)a.bst_rstd[tid]));
    }
  }
}

template <int CIN, int COUT, int NW, int WPS>
static int convt_thin_dgrad_cfg(const ConvtDgradArgs& ca, hipStream_t st) {
  constexpr size_t lds = (size_t)4 * COUT * CIN * sizeof(bf16) + (size_t)(4 * CIN > 2 * NW * CIN ? 4 * CIN : 2 * NW * CIN) * sizeof(float) + (size_t)NW * 32 * 144;
  auto kern = convt_thin_dgrad_kernel<CIN, COUT, NW, WPS>;
  { const int rc = satcv_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds); if (rc) return rc; }
  static const int cus = [] {
    int dev = 0, v = 256;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) v = p.multiProcessorCount;
    return v;
  }();
  int per_cu = WPS * 4 / NW;
  while (per_cu > 1 && (size_t)per_cu * lds > 150 * 1024) --per_cu;
  if (per_cu < 1) per_cu = 1;
  long long grid = (long long)cus * per_cu;
  const long long need = (ca.total_strips + NW - 1) / NW;
  if (grid > need) grid = need;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), lds, st, ca);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { satcv_set_error("convt_thin_dgrad launch: %s", hipGetErrorString(e)); return SATCV_ERR_HIP; }
  return SATCV_OK;
}

// the space-to-depth launch of satcv_conv2d_igemm (mode_in == 1): x0 = dy with c0 = COUT channels, cout = CIN
int convt_thin_dgrad_launch(const IgemmArgs& a, int dtype, hipStream_t st) {
  static const bool on = [] { const char* e = getenv("SATCV_CONVT_THIN"); return !e || atoi(e) != 0; }();
  if (!on || dtype != SATCV_BF16) return SATCV_ERR_UNSUPPORTED;
  if (a.kh != 1 || a.kw != 1 || a.mode_in != 1 || a.mode_out != 0 || a.f != 2 || a.x1 || a.stride != 1) return SATCV_ERR_UNSUPPORTED;
  if (a.out_scale || a.pool_y || a.accumulate || a.out_relu || a.in_scale || a.bias || a.bst_y1) return SATCV_ERR_UNSUPPORTED;
  if ((a.stats != nullptr) != (a.bst_y != nullptr)) return SATCV_ERR_UNSUPPORTED;      // (plain output statistics are not formed here)
  const int cout_t = a.c0, cin = a.cout;
  if (a.cout_pad != cin || a.w_ % 32 != 0 || a.ldy % 8 != 0 || a.ldy < cin) return SATCV_ERR_UNSUPPORTED;
  if (((uintptr_t)a.y % 16) != 0 || ((uintptr_t)a.x0 % 16) != 0 || ((uintptr_t)a.w % 16) != 0) return SATCV_ERR_UNSUPPORTED;
  if (a.bst_y && (a.bst_ld % 8 != 0 || a.bst_ld < cin || ((uintptr_t)a.bst_y % 16) != 0 || a.stats_ld < cin || a.cstat != cin)) return SATCV_ERR_UNSUPPORTED;
  const long long strips = (long long)a.n * a.h * a.w_ / 32;
  if (strips < 8 || strips > 0x3ffffff) return SATCV_ERR_UNSUPPORTED;      // (pixel indices of a strip stay in 31 bits)
  ConvtDgradArgs ca;
  ca.dy = a.x0; ca.lddy = a.c0; ca.w = a.w; ca.dx = a.y; ca.lddx = a.ldy; ca.stats = a.stats; ca.stats_ld = a.stats_ld;
  ca.bst_y = a.bst_y; ca.bst_ld = a.bst_ld; ca.bst_scale = a.bst_scale; ca.bst_shift = a.bst_shift; ca.bst_mean = a.bst_mean; ca.bst_rstd = a.bst_rstd;
  ca.bst_relu = a.bst_relu;
  ca.h = a.h; ca.w_ = a.w_; ca.total_strips = (int)strips;
  // (64 <- 4 x 32 channels at 128 x 128 moves 536 MB -- dy, the raw outputs for the fused sums, dx -- and the tiled kernel already does it at
  //  5.0 TB/s, 107 us; this kernel measured 113-120 us there.  SATCV_CONVT_THIN=2 runs it anyway.)
  static const bool all = [] { const char* e = getenv("SATCV_CONVT_THIN"); return e && atoi(e) >= 2; }();
  if (cin == 64 && cout_t == 32 && all) return convt_thin_dgrad_cfg<64, 32, 4, 3>(ca, st);
  if (cin == 128 && cout_t == 64) return convt_thin_dgrad_cfg<128, 64, 8, 2>(ca, st);
  return SATCV_ERR_UNSUPPORTED;
}

// SATCV_ERR_UNSUPPORTED outside the kernel's limits (the caller falls back to the tiled kernels)
int convt_thin_launch(const IgemmArgs& a, int dtype, hipStream_t st) {
  static const bool on = [] { const char* e = getenv("SATCV_CONVT_THIN"); return !e || atoi(e) != 0; }();
  if (!on || dtype != SATCV_BF16) return SATCV_ERR_UNSUPPORTED;
  if (a.kh != 1 || a.kw != 1 || a.mode_out != 1 || a.mode_in != 0 || a.f != 2 || a.x1 || a.stride != 1) return SATCV_ERR_UNSUPPORTED;
  if (a.pool_y || a.accumulate || a.bst_y || ((a.out_scale || a.out_relu) && a.stats)) return SATCV_ERR_UNSUPPORTED;      // (statistics are of the plain training output)
  const int cin = a.c0, cout_t = a.cstat;
  if (a.cout != 4 * cout_t || a.cout_pad != 4 * cout_t) return SATCV_ERR_UNSUPPORTED;
  if (a.w_ % 32 != 0 || a.ldy % 8 != 0 || a.ldy < cout_t || ((uintptr_t)a.y % 16) != 0 || ((uintptr_t)a.x0 % 16) != 0 || ((uintptr_t)a.w % 16) != 0) return SATCV_ERR_UNSUPPORTED;
  const long long strips = (long long)a.n * a.h * a.w_ / 32;
  if (strips < 8 || strips > 0x3ffffff) return SATCV_ERR_UNSUPPORTED;      // (pixel indices of a strip stay in 31 bits)
  if (a.stats && a.stats_ld < cout_t) return SATCV_ERR_UNSUPPORTED;
  ConvtArgs ca;
  ca.x = a.x0; ca.in_scale = a.in_scale; ca.in_shift = a.in_shift; ca.in_relu = a.in_relu;
  ca.w = a.w; ca.bias = a.bias; ca.out_scale = a.out_scale; ca.out_relu = a.out_relu; ca.y = a.y; ca.ldy = a.ldy; ca.stats = a.stats; ca.stats_ld = a.stats_ld;
  ca.h = a.h; ca.w_ = a.w_; ca.total_strips = (int)strips;
  static const int wps = [] { const char* e = getenv("SATCV_CONVT_WPS"); return e ? atoi(e) : 3; }();
  if (cin == 64 && cout_t == 32) {
    if (wps == 2) return convt_thin_cfg<64, 32, 4, 2>(ca, st);
    return convt_thin_cfg<64, 32, 4, 3>(ca, st);
  }
  if (cin == 128 && cout_t == 64) return convt_thin_cfg<128, 64, 8, 2>(ca, st);
  static const bool mid = [] { const char* e = getenv("SATCV_CONVT_MID"); return !e || atoi(e) != 0; }();
  if (cin == 256 && cout_t == 128 && mid) return convt_thin_cfg<256, 128, 8, 2, 4>(ca, st);      // one position (64 KB of weights) per workgroup
  return SATCV_ERR_UNSUPPORTED;
}
