// Thin 3x3 convolutions (Cin 16 / 32 / 64 -> Cout 32 / 64 on the full- and half-resolution levels: utils/model_tools.py:178, forward with the
// producing layer's BatchNorm + ReLU on the input, optional second source = decoder concatenation) as a STREAMING kernel: a wave walks down a
// column of 32-pixel row strips and nothing in its loop synchronises with another wave.
//
// The tiled persistent kernel of these layers (conv_igemm_ws.hip) runs at 3.3-3.9 TB/s: its workgroup stages an 8 x 32 tile + halo, meets at
// four barriers per tile and its phases (stage / MFMA / epilogue) do not overlap.  conv_transpose_thin.hip showed what the same bytes do
// without the barriers (4.2-4.6 TB/s).  A 3x3 conv needs the rows above and below, so here:
//   * a wave owns a contiguous run of row strips (32 pixels of one image row) in (image, column, row) order and keeps a ROLLING WINDOW of
//     activated input rows in its own LDS region: rows y - 1, y, y + 1 (34 pixels each: one halo pixel per side, zero outside the image)
//     while it computes output row y; row y + 2 arrives in registers during that time, gets the input BatchNorm + ReLU there and is
//     written over the slot of row y - 1 once that row's last reader is done.  Every input row is read from HBM once per column (34 / 32 of the bytes) plus two rows where a run
//     starts; no barrier, because only the owning wave touches its region and a wave's LDS operations execute in order;
//   * the 9-tap weight tensor (9-72 KB) is resident in LDS, shared by the workgroup's waves, read-only after the one barrier at the start;
//   * the product is formed with the pixel on the lane (A = W^T fragment, B = X^T fragment: the B fragment of tap (ky, kx) is a 16-byte
//     row read at pixel r + kx of window row ky); bias, bf16 packing and a half exchange (v_permlane32_swap) give every lane 8 consecutive
//     channels of its pixel, which go through wave-private staging rows and leave as whole lines (a row strip of 32 channels is 2 KB
//     contiguous);
//   * the BatchNorm sum / sum-of-squares of the stored values are taken from the staged pieces on their way out: a lane always reads the
//     same 8 channels (piece index = lane mod pieces per pixel), so the sums are 16 per-lane registers, combined across lanes, waves and
//     workgroups once at the end.
#include "igemm_common.hpp"
#include <cstdlib>

struct Conv3sArgs {
  const void* x0; const void* x1; int c0, c1;
  const float* in_scale; const float* in_shift; int in_relu;
  const void* w; const float* bias;
  void* y; int ldy;
  satcv_stat_t* stats; int stats_ld;
  int n, h, w_;
  int total;                  // row strips: n * (w_ / 32) * h
};

__device__ __forceinline__ unsigned pk_bf16_s(float a, float b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v;
  v[0] = (bf16)a; v[1] = (bf16)b;
  return __builtin_bit_cast(unsigned, v);
}

template <int CIN, int COUT, int NW, int WPS>
__global__ __launch_bounds__(NW * 64, WPS) void conv3_stream_kernel(const Conv3sArgs a) {
  typedef bf16 T;
  constexpr int SX = CIN / 8, KS = CIN / 16, NT = COUT / 32, PPR = COUT / 8, NTHREADS = NW * 64;
  constexpr int PXP = 36;                                 // pixels per plane row: 34 + 2 of padding
  constexpr int PLANEB = PXP * 16, ROWB = SX * PLANEB;    // bytes of a slot plane / of a window row
  constexpr int NRING = 3;                                // window rows y - 1, y, y + 1: row y + 2 is written over row y - 1 after the MFMAs of row y
  constexpr int OPITCH = COUT * 2 + 16;                   // staging row pitch (bytes): 8 neighbouring pixels' 16-byte stores on distinct banks
  constexpr int RJ = (34 * SX + 63) / 64;                 // 16-byte items of a window row per lane
  constexpr int OM = 32 * PPR / 64;                       // staged pieces per lane and strip
  constexpr size_t W_BYTES = (size_t)9 * CIN * COUT * sizeof(T);
  constexpr int TAB_FLOATS = (2 * CIN + COUT > 2 * NW * COUT ? 2 * CIN + COUT : 2 * NW * COUT);
  constexpr size_t WAVE_BYTES = (size_t)NRING * ROWB + 32 * OPITCH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* ldsW = reinterpret_cast<T*>(smem_raw);                                   // [9][SX][COUT][8]: the packed forward image as it is
  float* tab = reinterpret_cast<float*>(smem_raw + W_BYTES);                  // scale[CIN], shift[CIN], bias[COUT]; later the statistics
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  unsigned char* ring = smem_raw + W_BYTES + (size_t)TAB_FLOATS * sizeof(float) + (size_t)wave * WAVE_BYTES;
  unsigned char* ldsO = ring + NRING * ROWB;
  {
    const T* wp = reinterpret_cast<const T*>(a.w);
    for (int it = tid; it < 9 * SX * COUT; it += NTHREADS) lstore8<T>(ldsW + (size_t)it * 8, gload8<T>(wp + (size_t)it * 8));
    for (int ch = tid; ch < CIN; ch += NTHREADS) {
      tab[ch] = a.in_scale ? a.in_scale[ch] : 1.f;
      tab[CIN + ch] = a.in_scale ? a.in_shift[ch] : 0.f;
    }
    for (int ch = tid; ch < COUT; ch += NTHREADS) tab[2 * CIN + ch] = a.bias ? a.bias[ch] : 0.f;
  }
  __syncthreads();
  const bool xaff = a.in_scale != nullptr;
  const unsigned relu_lim = a.in_relu != 0 ? 0u : 0x80008000u;
  const bool want_stats = a.stats != nullptr;

  // ---- this wave's run of row strips (XCD-aware contiguous ranges: blocks b and b + 8 share an XCD)
  const int G = gridDim.x;
  const int xcd = blockIdx.x & 7, nx = G >> 3, remx = G & 7;
  const int bid = (xcd < remx ? xcd * (nx + 1) : remx * (nx + 1) + (xcd - remx) * nx) + (blockIdx.x >> 3);
  const int gw = __builtin_amdgcn_readfirstlane(bid * NW + wave), GW = G * NW;
  const int per = a.total / GW, extra = a.total % GW;
  const int t_lo = gw * per + (gw < extra ? gw : extra), t_hi = t_lo + per + (gw < extra ? 1 : 0);
  const int CX = a.w_ >> 5;

  // ---- this lane's items of a window row: pixel (0 ... 33, image column x0 - 1 + pixel) and 8-channel slot
  int it_src[RJ];                // element offset from the row's pixel x0 in its source tensor (x0 or x1), -1: no such item
  int it_lds[RJ];                // byte offset inside a window row
  int it_px[RJ];
  bool it_second[RJ];
#pragma unroll
  for (int j = 0; j < RJ; ++j) {
    const int it = lane + 64 * j, px = it / SX, slot = it % SX;
    const bool ok = it < 34 * SX;
    const int ch = slot * 8;
    it_second[j] = ok && a.x1 != nullptr && ch >= a.c0;
    it_src[j] = ok ? (it_second[j] ? (px - 1) * a.c1 + (ch - a.c0) : (px - 1) * a.c0 + ch) : -1;
    it_lds[j] = slot * PLANEB + px * 16;
    it_px[j] = px;
  }
  const T* xp0 = reinterpret_cast<const T*>(a.x0);
  const T* xp1 = reinterpret_cast<const T*>(a.x1);
  // load one window row (image row yy of image n at column origin x0) into registers; rows outside the image are not loaded
  auto load_row = [&](Raw8<T> (&dst)[RJ], int n, int yy, int x0) {
    if (yy < 0 || yy >= a.h) return;
    const size_t pix = ((size_t)n * a.h + yy) * a.w_ + x0;
#pragma unroll
    for (int j = 0; j < RJ; ++j) {
      const int x = x0 - 1 + it_px[j];
      const bool ok = it_src[j] != -1 && x >= 0 && x < a.w_;
      const T* p = it_second[j] ? xp1 + pix * a.c1 : xp0 + pix * a.c0;
      dst[j] = gload8<T>(p + (ok ? it_src[j] : 0));       // (outside: the strip's first pixel, zeroed when the row is written)
    }
  };
  auto write_row = [&](const Raw8<T> (&src)[RJ], int yy, int x0) {
    unsigned char* rowp = ring + ((yy + NRING) % NRING) * ROWB;
    const bool rowok = yy >= 0 && yy < a.h;
#pragma unroll
    for (int j = 0; j < RJ; ++j) {
      if (lane + 64 * j >= 34 * SX) continue;
      const int x = x0 - 1 + it_px[j];
      const bool ok = rowok && x >= 0 && x < a.w_;
      Raw8<T> v = src[j];
      if (xaff) {
        int toff = ((lane + 64 * j) % SX) * 8;
        asm volatile("" : "+v"(toff));
        const float4* sp = reinterpret_cast<const float4*>(tab + toff);
        const float4* hp = reinterpret_cast<const float4*>(tab + CIN + toff);
        const float4 s0 = sp[0], s1 = sp[1], h0 = hp[0], h1 = hp[1];
        const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
        v = affine8_lim(v, sc, sh, relu_lim);
      }
      v = select8<T>(ok, v);
      *reinterpret_cast<uint4*>(rowp + it_lds[j]) = v.q[0];
    }
  };

  float bs1[8], bs2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { bs1[e] = 0.f; bs2[e] = 0.f; }
  const T* wlane = ldsW + (size_t)(hh * COUT + r) * 8;                       // + ((tap * SX + ks * 2) * COUT + nt * 32) * 8
  const int xfrag = hh * PLANEB + r * 16;                                    // + row slot + ks * 2 * PLANEB + kx * 16

  int cur = t_lo;
  while (cur < t_hi) {
    // ---- a run inside one column: rows y ... y_end - 1 of image n, pixels x0 ... x0 + 31
    const int col = cur / a.h;
    int y = cur - col * a.h;
    const int n = col / CX, x0 = (col - n * CX) * 32;
    const int y_end = min(a.h, y + (t_hi - cur));
    cur += y_end - y;
    Raw8<T> rr[RJ];
    {
      // the window of the first row: y - 1, y, y + 1 -- requested together, then written
      Raw8<T> ra[RJ], rb[RJ];
#pragma unroll
      for (int j = 0; j < RJ; ++j) { ra[j] = zero8<T>(); rb[j] = zero8<T>(); rr[j] = zero8<T>(); }
      load_row(ra, n, y - 1, x0);
      load_row(rb, n, y, x0);
      load_row(rr, n, y + 1, x0);
      write_row(ra, y - 1, x0);
      write_row(rb, y, x0);
      write_row(rr, y + 1, x0);
      load_row(rr, n, y + 2, x0);
    }
    for (; y < y_end; ++y) {
      f32x16 acc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
      // (fragment reads two taps ahead of their MFMAs -- three register sets, 149 instead of 110 registers -- measured SLOWER: 156 vs 134 us on
      //  32 -> 32 at 256 x 256)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const unsigned char* rowp = ring + ((y - 1 + ky + NRING) % NRING) * ROWB + xfrag;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 xf = *reinterpret_cast<const bf16x8*>(rowp + ks * 2 * PLANEB + kx * 16);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wlane + (size_t)(((ky * 3 + kx) * SX + ks * 2) * COUT + t * 32) * 8);
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc[t], 0, 0, 0);       // [channel][pixel]
            }
          }
        }
      }
      // row y + 2 (in registers since the previous step) over the slot of row y - 1, which the MFMAs above were the last to read; then
      // request row y + 3
      write_row(rr, y + 2, x0);
      load_row(rr, n, y + 3, x0);
      // ---- bias, bf16, half exchange, staging rows
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        uint2 o2[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 b4 = *reinterpret_cast<const float4*>(tab + 2 * CIN + t * 32 + 8 * g + 4 * hh);
          o2[g].x = pk_bf16_s(acc[t][4 * g] + b4.x, acc[t][4 * g + 1] + b4.y);
          o2[g].y = pk_bf16_s(acc[t][4 * g + 2] + b4.z, acc[t][4 * g + 3] + b4.w);
        }
#pragma unroll
        for (int k = 0; k < 4; k += 2) {
          const uint2 lo = o2[k], hi = o2[k + 1];
          auto sx = __builtin_amdgcn_permlane32_swap(lo.x, hi.x, false, false);
          auto sy = __builtin_amdgcn_permlane32_swap(lo.y, hi.y, false, false);
          *reinterpret_cast<uint4*>(ldsO + r * OPITCH + t * 64 + k * 16 + hh * 16) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
        }
      }
      // ---- whole-line stores of the strip (+ the statistics of the stored values: this lane's 8 channels)
      T* yrow = reinterpret_cast<T*>(a.y) + (((size_t)n * a.h + y) * a.w_ + x0) * a.ldy;
#pragma unroll
      for (int m = 0; m < OM; ++m) {
        const int c = lane + 64 * m, q = c / PPR, piece = c % PPR;
        const uint4 v = *reinterpret_cast<const uint4*>(ldsO + q * OPITCH + piece * 16);
        *reinterpret_cast<uint4*>(yrow + (size_t)q * a.ldy + piece * 8) = v;
        if (want_stats) {
          const bf16x8 v8 = __builtin_bit_cast(bf16x8, v);
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float fv = (float)v8[e]; bs1[e] += fv; bs2[e] += fv * fv; }
        }
      }
    }
  }
  // ---- statistics: lanes with the same piece index, then the waves, in a fixed order; one pair of atomics per channel and workgroup
  if (want_stats) {
    __syncthreads();                                                          // (the tables are dead: every wave is past its last strip)
    float* red = tab;                                                         // [NW][2][COUT]
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float s1 = bs1[e], s2 = bs2[e];
#pragma unroll
      for (int o = PPR; o < 64; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
      if (lane < PPR) { red[(wave * 2 + 0) * COUT + lane * 8 + e] = s1; red[(wave * 2 + 1) * COUT + lane * 8 + e] = s2; }
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

extern int g_opt_igemm_thin;        // api.hip: 2 = tests force the tiled persistent kernel
int g_conv3s_launches = 0;          // launches served here (satcv_get_option("conv3_stream_launches"): tests check the path taken)

template <int CIN, int COUT, int NW, int WPS>
static int conv3s_cfg(const Conv3sArgs& ca, hipStream_t st) {
  constexpr int SX = CIN / 8;
  constexpr size_t lds = (size_t)9 * CIN * COUT * 2 + (size_t)(2 * CIN + COUT > 2 * NW * COUT ? 2 * CIN + COUT : 2 * NW * COUT) * 4 +
                         (size_t)NW * (3 * SX * 36 * 16 + 32 * (COUT * 2 + 16));
  static_assert(lds <= 160 * 1024, "weights + windows exceed the LDS");
  auto kern = conv3_stream_kernel<CIN, COUT, NW, WPS>;
  { const int rc = satcv_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds); if (rc) return rc; }
  static const int cus = [] {
    int dev = 0, v = 256;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) v = p.multiProcessorCount;
    return v;
  }();
  int per_cu = WPS * 4 / NW;
  while (per_cu > 1 && (size_t)per_cu * lds > 160 * 1024) --per_cu;
  if (per_cu < 1) per_cu = 1;
  long long grid = (long long)cus * per_cu;
  const long long need = (ca.total / 8 + NW - 1) / NW;                          // at least 8 rows per wave: a run starts with two extra rows
  if (grid > need) grid = need;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), lds, st, ca);
  ++g_conv3s_launches;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { satcv_set_error("conv3_stream launch: %s", hipGetErrorString(e)); return SATCV_ERR_HIP; }
  return SATCV_OK;
}

// SATCV_ERR_UNSUPPORTED outside the kernel's limits (the caller goes on to the persistent tiled kernel)
// OFF by default (satcv_set_option("conv3_stream", 1) / SATCV_CONV3_STREAM=1): against the tiled persistent kernel it measured 90.0 vs 92.4 us
// (16 -> 32 at 256 x 256), 51.8 vs 56.6 (32 -> 64 at 128 x 128), 134.3 vs 132.6 (32 -> 32), 248 vs 228 (32 + 32 -> 32), 112.6 vs 91.6 (64 -> 64):
// a 3x3 strip carries 9x the MFMAs and LDS fragment reads of a transposed-conv strip per byte, and the 64-channel windows leave room for
// 4-6 waves per CU; the structure that took the transposed convolutions from 2.9 to 4.5 TB/s does not carry over.  Kept as a tested option.
extern int g_opt_conv3_stream;
int conv3_stream_launch(const IgemmArgs& a, int dtype, hipStream_t st) {
  if (!g_opt_conv3_stream || g_opt_igemm_thin >= 2 || dtype != SATCV_BF16) return SATCV_ERR_UNSUPPORTED;
  const int cin = a.c0 + a.c1;
  if (a.kh != 3 || a.kw != 3 || a.dil != 1 || a.stride != 1 || a.mode_in || a.mode_out || a.accumulate || a.bst_y) return SATCV_ERR_UNSUPPORTED;
  if (a.out_scale || a.out_relu || a.pool_y) return SATCV_ERR_UNSUPPORTED;                      // (the folded inference graph stays on the tiled kernel)
  if (!(cin == 16 || cin == 32 || cin == 64) || !(a.cout == 32 || a.cout == 64) || a.cout_pad != a.cout || a.cstat != a.cout) return SATCV_ERR_UNSUPPORTED;
  if (a.x1 && (a.c0 % 8 != 0)) return SATCV_ERR_UNSUPPORTED;
  if (a.w_ % 32 != 0 || a.ldy % 8 != 0 || a.ldy < a.cout || ((uintptr_t)a.y % 16) != 0 || ((uintptr_t)a.x0 % 16) != 0 || (a.x1 && ((uintptr_t)a.x1 % 16) != 0)) return SATCV_ERR_UNSUPPORTED;
  if (a.stats && a.stats_ld < a.cout) return SATCV_ERR_UNSUPPORTED;
  const long long total = (long long)a.n * (a.w_ / 32) * a.h;
  if (total < 8 || total > 0x3fffffff || (long long)a.n * a.h * a.w_ * (a.c0 > a.c1 ? a.c0 : a.c1) > 0x7fffffffLL * 8) return SATCV_ERR_UNSUPPORTED;
  Conv3sArgs ca;
  ca.x0 = a.x0; ca.x1 = a.x1; ca.c0 = a.c0; ca.c1 = a.c1; ca.in_scale = a.in_scale; ca.in_shift = a.in_shift; ca.in_relu = a.in_relu;
  ca.w = a.w; ca.bias = a.bias; ca.y = a.y; ca.ldy = a.ldy; ca.stats = a.stats; ca.stats_ld = a.stats_ld;
  ca.n = a.n; ca.h = a.h; ca.w_ = a.w_; ca.total = (int)total;
  // waves per workgroup x workgroups per CU: as many waves as the LDS holds beside the resident weights (a wave's window + staging rows:
  // 6-18 KB)
  if (a.cout == 32) {
    if (cin == 16) return conv3s_cfg<16, 32, 4, 3>(ca, st);
    if (cin == 32) return conv3s_cfg<32, 32, 6, 3>(ca, st);
    return conv3s_cfg<64, 32, 6, 2>(ca, st);
  }
  if (cin == 16) return conv3s_cfg<16, 64, 4, 3>(ca, st);
  if (cin == 32) return conv3s_cfg<32, 64, 8, 2>(ca, st);
  return conv3s_cfg<64, 64, 4, 1>(ca, st);
}
