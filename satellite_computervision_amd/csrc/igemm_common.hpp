// Shared argument block and MFMA fragment helpers of the implicit-GEMM convolution kernels.
#pragma once
#include "common.hpp"

struct IgemmArgs {
  const void* x0; const void* x1;
  int c0, c1;
  const float* in_scale; const float* in_shift; int in_relu;
  const void* w; const float* bias; const float* out_scale;
  void* pool_y; int pool_ld, pool_f;
  void* y; int ldy;
  satcv_stat_t* stats; int stats_ld;
  int n, h, w_;            // GEMM pixel grid
  int hs, ws;              // source spatial dims
  int cout, cout_pad;
  int kh, kw, dil;
  int mode_in, mode_out, f;
  int cstat, out_relu, accumulate, stride;
  // derived tiling
  int tiles_x, tiles_y, ngroups;   // M tiles = ngroups * tiles_y * tiles_x
  int rpi, imgs, seg, rl, cl, pitch, halh, halw;
  int n_tiles;                     // N tiles
  int nchunks;
  int halh_tl, halw_tl;            // tap-loop form: offset of tap (0, 0) in source pixels
  int taploop, cpt;                // tap-loop form of a dilated 3x3 conv: K = 9 taps x cpt channel chunks, one shifted tile per tap
  // BatchNorm-backward reduce pass of the layer whose activation gradient this launch writes (satcv.h: bst_*)
  const void* bst_y; const void* bst_y1; int bst_ld, bst_ld1, bst_split;
  const float* bst_scale; const float* bst_shift; const float* bst_mean; const float* bst_rstd; int bst_relu;
  int dbg;                         // ablation bits (env SATCV_DBG): 1 skip stores, 2 skip MFMA, 4 skip A loads, 8 skip B loads
  // split-K (conv_igemm_fast.hip): ksplit workgroups share an output tile, each sums a contiguous range of the K chunks and writes its
  // fp32 partial tile to kslab[split][pixel][cout]; satcv's finish kernel adds the slabs in order, applies the epilogue and the statistics
  int ksplit; float* kslab;
  int tile_policy;                 // satcv_conv_desc::tile_policy (0: option igemm_m16 decides, 2: every eligible launch on the 16x16x32 tiles)
};

template <typename T>
struct FragT;
template <>
struct FragT<bf16> { bf16x8 v; };
template <>
struct FragT<float> { float4 lo, hi; };
template <>
struct FragT<fp8> { long v; };
template <>
struct FragT<fp8s> { i32x4 lo, hi; };

// p: the lane's item of sub-slot 0; sub_stride: element distance to its item of sub-slot 1 (fp8s only)
template <typename T>
__device__ __forceinline__ FragT<T> lds_frag(const T* p, int sub_stride = 0) {
  FragT<T> f;
  if constexpr (std::is_same<T, fp8s>::value) {
    f.lo = *reinterpret_cast<const i32x4*>(p);
    f.hi = *reinterpret_cast<const i32x4*>(p + sub_stride);
  } else if constexpr (std::is_same<T, bf16>::value) {
    f.v = *reinterpret_cast<const bf16x8*>(p);
  } else if constexpr (std::is_same<T, fp8>::value) {
    f.v = *reinterpret_cast<const long*>(p);
  } else {
    f.lo = reinterpret_cast<const float4*>(p)[0];
    f.hi = reinterpret_cast<const float4*>(p)[1];
  }
  return f;
}

template <typename T>
__device__ __forceinline__ void mma32(f32x16& acc, const FragT<T>& a, const FragT<T>& b) {
  if constexpr (std::is_same<T, bf16>::value) {
#ifdef SATCV_EXP_M16
    // TIMING EXPERIMENT ONLY (wrong results): the same fragments through two v_mfma_f32_16x16x32_bf16 -- equal matrix cycles and FLOPs;
    // the MFMA shape changes the clock the chip holds under load (MI355X_MICROARCH.md, DVFS item 7)
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    f32x4_ c0 = {acc[0], acc[1], acc[2], acc[3]}, c1 = {acc[8], acc[9], acc[10], acc[11]};
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c1, 0, 0, 0);
    acc[0] = c0[0]; acc[1] = c0[1]; acc[2] = c0[2]; acc[3] = c0[3];
    acc[8] = c1[0]; acc[9] = c1[1]; acc[10] = c1[2]; acc[11] = c1[3];
#else
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
#endif
  } else if constexpr (std::is_same<T, fp8>::value) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a.v, b.v, acc, 0, 0, 0);      // bf16 rate, half the operand bytes
  } else if constexpr (std::is_same<T, fp8s>::value) {
    // K = 64 per instruction, unit block scales (e8m0 127): twice the bf16 rate.  Any k order is fine as long as A and B agree
    // (tools/probes/mfma_scale_layout.hip): a lane half holds the 32 consecutive channels of its two 16-byte items.
    const i32x8 av = __builtin_shufflevector(a.lo, a.hi, 0, 1, 2, 3, 4, 5, 6, 7);
    const i32x8 bv = __builtin_shufflevector(b.lo, b.hi, 0, 1, 2, 3, 4, 5, 6, 7);
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  } else {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo.x, b.lo.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo.y, b.lo.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo.z, b.lo.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo.w, b.lo.w, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi.x, b.hi.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi.y, b.hi.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi.z, b.hi.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi.w, b.hi.w, acc, 0, 0, 0);
  }
}


// ---- accumulator maps.  acc[m][n] is a 32 x 32 output tile in 16 registers per lane.  With v_mfma_f32_32x32x16 (M16 = false) register i
// of lane l is (row (i & 3) + 8 (i >> 2) + 4 (l >> 5), column l & 31).  With v_mfma_f32_16x16x32 (M16 = true) the tile is 2 x 2 blocks of
// 16 x 16, block (mb, nb) in registers 4 (2 mb + nb) .. + 3: register i is (row 16 (i >> 3) + 4 (l >> 4) + (i & 3), column
// 16 ((i >> 2) & 1) + (l & 15)) -- a lane owns TWO columns of the tile, one per column block (`sub`).
template <bool M16>
struct AccMap {
  static constexpr int NSUB = M16 ? 2 : 1;            // column blocks per 32-wide tile (statistics slots per tile and lane)
  static constexpr int RPS = 16 / NSUB;               // registers per column block
  __device__ static __forceinline__ int reg(int sub, int k) { return M16 ? ((k >> 2) * 8 + sub * 4 + (k & 3)) : k; }      // k-th register of column block sub
  __device__ static __forceinline__ int row(int i, int lane) { return M16 ? (16 * (i >> 3) + 4 * (lane >> 4) + (i & 3)) : ((i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)); }
  __device__ static __forceinline__ int col(int sub, int lane) { return M16 ? (16 * sub + (lane & 15)) : (lane & 31); }
  __device__ static __forceinline__ int sub_of(int i) { return M16 ? ((i >> 2) & 1) : 0; }
  // sum over the lanes that share a column; true in the lane that keeps the result
  __device__ static __forceinline__ float colsum(float v) {
    if (M16) v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
  }
  __device__ static __forceinline__ bool owner(int lane) { return M16 ? lane < 16 : lane < 32; }
};

// one K = 32 step of a wave's MT x NT tiles on v_mfma_f32_16x16x32_bf16: a[2 m + mb] = the A fragment of 16-row block mb of tile m (lane l:
// row l & 15, k = 8 (l >> 4) .. + 7), b[2 n + nb] likewise for the 16-column blocks.  Same FLOPs per matrix cycle as 32x32x16; the chip holds a
// higher clock under this shape (MI355X_MICROARCH.md, DVFS give-back item 7; measured here: profiles/r04_exp_mfma_16x16x32_timing.txt)
template <int MT, int NT>
__device__ __forceinline__ void mma16_step(f32x16 (&acc)[MT][NT], const FragT<bf16> (&a)[2 * MT], const FragT<bf16> (&b)[2 * NT]) {
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const int o = (mb * 2 + nb) * 4;
          f32x4 c = {acc[m][n][o], acc[m][n][o + 1], acc[m][n][o + 2], acc[m][n][o + 3]};
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2 * m + mb].v, b[2 * n + nb].v, c, 0, 0, 0);
          acc[m][n][o] = c[0]; acc[m][n][o + 1] = c[1]; acc[m][n][o + 2] = c[2]; acc[m][n][o + 3] = c[3];
        }
}

// ------------------------------------------------------------------ shared epilogue of the implicit-GEMM kernels
// acc[MT][NT] (32x32 MFMA tiles of this wave) -> y: per-channel multiplier / bias / ReLU, rounding to T, BN sum / sum-of-squares of
// the STORED values, tile staged in LDS [BM][BN+pad] and written with 16-byte coalesced stores (NHWC rows; depth-to-space rows
// for the transposed conv).  Call with every wave past its last LDS fragment read (the staging aliases the operand images).
// carry: persistent kernels pass two per-thread doubles (threads < BN own one output channel each); the tile's statistics are added
// there instead of going to the replica rows with atomics, and the caller flushes them once per workgroup (stats_flush).
// GENERAL = false: the caller guarantees interior tiles (whole tiles inside the image, every column valid, plain store): only the fast
// path is compiled -- the persistent thin-layer kernel sits at its register cap and the masked path's live state spilled.
template <typename T, int TW, int WM, int WN, int MT, int NT, bool SKIP_STORES = false, bool FAST = true, bool GENERAL = true, bool M16 = false>
__device__ __forceinline__ void igemm_epilogue(const IgemmArgs& a, f32x16 (&acc)[MT][NT], int n0, int y0, int x0, int nbase, unsigned char* smem_raw,
                                               double* carry = nullptr) {
  constexpr int NTHREADS = WM * WN * 64, BM = WM * MT * 32, BN = WN * NT * 32;
  constexpr int OPITCH = BN + 16 / (int)sizeof(T);                   // output staging pitch (elements)
  T* ldsO = reinterpret_cast<T*>(smem_raw);
  float* ldsS = reinterpret_cast<float*>(smem_raw + (size_t)BM * OPITCH * sizeof(T));   // [WM][2][BN] partial BN sums
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  using AM = AccMap<M16>;
  constexpr int NSL = NT * AM::NSUB;                 // statistics slots (columns of this lane) per wave
  // ---- fast path (wave-uniform test): one whole image tile inside the image, every column valid, plain NHWC rows, no ReLU / pool /
  // accumulation -- no validity masks, no per-element selects, no divisions in the store loop.  The general path below cost ~4,700
  // cycles per 256 x 32 tile (19 % of a thin-layer tile) of which most was mask and address bookkeeping.
  if (FAST && (!GENERAL || (a.imgs == 1 && !a.accumulate && !a.pool_y && !a.out_relu && n0 < a.n && y0 + BM / TW <= a.h && x0 + TW <= a.w_ &&
      nbase + BN <= a.cout && (a.mode_out == 0 || a.cstat % 8 == 0))) && !SKIP_STORES && (sizeof(T) == 2 || (!GENERAL && sizeof(T) == 1))) {
    constexpr int EPV = 16 / (int)sizeof(T);      // elements per 16-byte vector (8, or 16 for the fp8 form of the thin-layer kernel)
    constexpr int VPR = BN / EPV;                 // 16-byte vectors per tile row
    static_assert(NTHREADS % VPR == 0, "column group of a thread must be loop-invariant");
    const int vq = tid % VPR;                     // this thread's 16-byte column group (fixed: NTHREADS is a multiple of VPR)
    // ---- fused BatchNorm-backward reduce (bst_*, satcv.h): this launch writes dL/d act of a conv -> BN -> ReLU layer; the sums that
    // layer's BN backward needs (sum g, sum g * xhat over the batch) are formed here from the tile in the accumulators and the
    // layer's raw outputs v at the same pixels, instead of by a separate pass that re-reads both tensors from HBM.  v arrives by
    // 16-byte row loads (issued first, hidden behind the accumulator -> LDS pass) in a second staging tile.
    const bool bst = GENERAL && sizeof(T) == 2 && a.bst_y != nullptr;      // (the fast-only instantiations never carry the fused sums)
    // (the 128 x 128 tile of 4 waves would pay for 8 held row vectors with its third workgroup per CU: it parks them in LDS at once)
    constexpr bool YHOLD = !(MT * NT >= 4 && NTHREADS <= 256);
    constexpr int YIT = (BM * VPR + NTHREADS - 1) / NTHREADS;
    uint4 yv[YIT];
#pragma unroll
    for (int j = 0; j < YIT; ++j) yv[j] = make_uint4(0u, 0u, 0u, 0u);
    T* ldsY = reinterpret_cast<T*>(smem_raw + (size_t)BM * OPITCH * sizeof(T) + (size_t)(WM + 1) * 2 * BN * sizeof(float));
    if (bst) {
      const int cg = nbase + vq * 8;
      const bool sec = a.bst_y1 != nullptr && cg >= a.bst_split;
      const size_t yld = sec ? (size_t)a.bst_ld1 : (size_t)a.bst_ld;
      const T* ys = (sec ? reinterpret_cast<const T*>(a.bst_y1) + (cg - a.bst_split) : reinterpret_cast<const T*>(a.bst_y) + cg) +
                    ((size_t)(n0 * a.h + y0) * a.w_ + x0) * yld;
#pragma unroll
      for (int j = 0; j < YIT; ++j) {
        const int it = tid + j * NTHREADS;
        const int q = it < BM * VPR ? it / VPR : 0;
        yv[j] = *reinterpret_cast<const uint4*>(ys + ((size_t)(q / TW) * a.w_ + q % TW) * yld);
      }
      if constexpr (!YHOLD) {
#pragma unroll
        for (int j = 0; j < YIT; ++j) {
          const int it = tid + j * NTHREADS;
          if (it < BM * VPR) *reinterpret_cast<uint4*>(ldsY + (it / VPR) * OPITCH + vq * 8) = yv[j];
        }
      }
    }
    float st1[NSL], st2[NSL];
    // (RELU: the folded inference graph clamps in the epilogue; only the interior-tile-only instantiations -- the persistent
    //  thin-layer kernel -- take such launches here, everything else sends them down the general path)
    auto first_pass = [&](auto RELU) {
#pragma unroll
      for (int ns = 0; ns < NSL; ++ns) {
        const int n = ns / AM::NSUB, sub = ns % AM::NSUB;
        const int cl_ = (wn * NT + n) * 32 + AM::col(sub, lane);
        const int cch = (nbase + cl_) % a.cstat;
        const float bv = a.bias ? a.bias[cch] : 0.f;
        const float osc = a.out_scale ? a.out_scale[cch] : 1.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
          for (int k = 0; k < AM::RPS; ++k) {
            const int i = AM::reg(sub, k);
            const int q = (wm * MT + m) * 32 + AM::row(i, lane);
            float vv = acc[m][n][i] * osc + bv;
            if (decltype(RELU)::value) vv = fmaxf(vv, 0.f);
            const T tv = (T)vv;
            ldsO[q * OPITCH + cl_] = tv;
            const float fv = (float)tv;
            s1 += fv; s2 += fv * fv;
          }
        }
        st1[ns] = s1; st2[ns] = s2;
      }
    };
    if (!GENERAL && a.out_relu) first_pass(std::true_type{});
    else first_pass(std::false_type{});
    if (bst) {
      if constexpr (YHOLD) {
#pragma unroll
        for (int j = 0; j < YIT; ++j) {
          const int it = tid + j * NTHREADS;
          if (it < BM * VPR) *reinterpret_cast<uint4*>(ldsY + (it / VPR) * OPITCH + vq * 8) = yv[j];
        }
      }
      __syncthreads();
#pragma unroll
      for (int ns = 0; ns < NSL; ++ns) {
        const int n = ns / AM::NSUB, sub = ns % AM::NSUB;
        const int cl_ = (wn * NT + n) * 32 + AM::col(sub, lane);
        const int cch = nbase + cl_;
        const float bv = a.bias ? a.bias[cch] : 0.f;
        const float osc = a.out_scale ? a.out_scale[cch] : 1.f;
        const float sc = a.bst_scale[cch], sh = a.bst_shift[cch], mu = a.bst_mean[cch], rs = a.bst_rstd[cch];
        const bool lin = a.bst_relu == 0;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
          for (int k = 0; k < AM::RPS; ++k) {
            const int i = AM::reg(sub, k);
            const int q = (wm * MT + m) * 32 + AM::row(i, lane);
            const float fv = (float)(T)(acc[m][n][i] * osc + bv);       // the value as stored (what the apply pass will read)
            const float v = (float)ldsY[q * OPITCH + cl_];
            const float gg = (v * sc + sh > 0.f || lin) ? fv : 0.f;
            s1 += gg; s2 += gg * ((v - mu) * rs);
          }
        }
        st1[ns] = s1; st2[ns] = s2;
      }
    }
    if (a.stats) {
#pragma unroll
      for (int ns = 0; ns < NSL; ++ns) {
        const int cl_ = (wn * NT + ns / AM::NSUB) * 32 + AM::col(ns % AM::NSUB, lane);
        const float s1 = AM::colsum(st1[ns]);
        const float s2 = AM::colsum(st2[ns]);
        if (AM::owner(lane)) { ldsS[(wm * 2 + 0) * BN + cl_] = s1; ldsS[(wm * 2 + 1) * BN + cl_] = s2; }
      }
    }
    __syncthreads();
    if (a.stats && tid < BN) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) { t1 += ldsS[(w * 2 + 0) * BN + tid]; t2 += ldsS[(w * 2 + 1) * BN + tid]; }
      const int cch = (nbase + tid) % a.cstat;
      if (carry) { carry[0] += (double)t1; carry[1] += (double)t2; }
      else {
        satcv_stat_t* rowp = a.stats + (size_t)(blockIdx.x % SATCV_STAT_ROWS) * 2 * a.stats_ld;
        atomicAdd(rowp + cch, (satcv_stat_t)t1);
        atomicAdd(rowp + a.stats_ld + cch, (satcv_stat_t)t2);
      }
    }
    T* yp; size_t row_pitch, col_pitch;
    if (a.mode_out == 1) {
      // depth-to-space (transposed conv): the column group belongs to ONE sub-pixel position (iy, ix) of cstat channels
      const int cn0 = nbase + vq * EPV, ij = cn0 / a.cstat, cb = cn0 - ij * a.cstat;
      const int f = a.f, wo = a.w_ * f;
      yp = reinterpret_cast<T*>(a.y) + ((size_t)(n0 * a.h * f + y0 * f + ij / f) * wo + (size_t)x0 * f + ij % f) * a.ldy + cb;
      row_pitch = (size_t)f * wo * a.ldy; col_pitch = (size_t)f * a.ldy;
    } else {
      yp = reinterpret_cast<T*>(a.y) + ((size_t)(n0 * a.h + y0) * a.w_ + x0) * a.ldy + nbase + vq * EPV;
      row_pitch = (size_t)a.w_ * a.ldy; col_pitch = (size_t)a.ldy;
    }
    // (the interior-only instantiations also take tiles wider than the layer: a 32-column tile for 16 output channels -- thin dilated layers of
    //  the atrous CNNs; its upper column groups hold zeros and are not stored)
    const bool col_ok = GENERAL || nbase + vq * EPV < a.cout;
#pragma unroll
    for (int it = tid; it < BM * VPR; it += NTHREADS) {
      const int q = it / VPR;                     // (compile-time divisors)
      const int t = q / TW, cx = q % TW;
      if (col_ok) {
        T* dst = yp + (size_t)t * row_pitch + (size_t)cx * col_pitch;
        uint4 nv = *reinterpret_cast<const uint4*>(ldsO + q * OPITCH + vq * EPV);
        if constexpr (!GENERAL && sizeof(T) == 2) {
          // accumulate == 1 on the interior-only instantiations: y += result (a data gradient joining the gradient another consumer of the same
          // tensor wrote -- residual sums of the atrous CNNs); same arithmetic as the general path below
          if (a.accumulate == 1) {
            uint4 ov = *reinterpret_cast<const uint4*>(dst);
            T* oe = reinterpret_cast<T*>(&ov);
            const T* nn = reinterpret_cast<const T*>(&nv);
#pragma unroll
            for (int e = 0; e < EPV; ++e) oe[e] = (T)((float)oe[e] + (float)nn[e]);
            nv = ov;
          }
        }
        *reinterpret_cast<uint4*>(dst) = nv;
      }
    }
    if constexpr (!GENERAL) {
      // fused max-pool of the tile just stored (folded inference encoder blocks; the interior-tile-only instantiations take these
      // launches here, see the general form at the end of this function): every window of the tile is whole and inside the image
      if (a.pool_y) {
        const int f = a.pool_f, pw = TW / f, ph = (BM / TW) / f;
        const int hp = a.h / f, wp = a.w_ / f;
        T* pp = reinterpret_cast<T*>(a.pool_y) + nbase + vq * EPV;
        for (int it = tid; it < ph * pw * VPR; it += NTHREADS) {
          const int pq = it / VPR;
          const int t0 = (pq / pw) * f, c0 = (pq % pw) * f;
          float mx[EPV];
#pragma unroll
          for (int e = 0; e < EPV; ++e) mx[e] = -INFINITY;
          for (int i = 0; i < f; ++i)
            for (int j = 0; j < f; ++j) {
              const T* sp = ldsO + ((t0 + i) * TW + c0 + j) * OPITCH + vq * EPV;
#pragma unroll
              for (int e = 0; e < EPV; ++e) mx[e] = fmaxf(mx[e], (float)sp[e]);
            }
          T o[EPV];
#pragma unroll
          for (int e = 0; e < EPV; ++e) o[e] = (T)mx[e];
          *reinterpret_cast<uint4*>(pp + ((size_t)(n0 * hp + (y0 + t0) / f) * wp + (x0 + c0) / f) * a.pool_ld) = *reinterpret_cast<const uint4*>(o);
        }
      }
    }
    return;
  }
  if constexpr (!GENERAL) return;
  // validity of the 16 accumulator rows of each MFMA tile (pixels outside the image must not enter the statistics): rows of one
  // MFMA tile span 32/TW tile rows; the column test needs no division
  unsigned pvmask[MT];
  {
    const int xlim = a.w_ - x0;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      pvmask[m] = 0;
      if (a.stats) {
        unsigned rowok = 0;
#pragma unroll
        for (int u = 0; u < 32 / TW; ++u) {
          const int t = (wm * MT + m) * (32 / TW) + u;
          const int k = (a.imgs == 1) ? 0 : t / a.rpi;
          const bool ok = (k < a.imgs) && (n0 + k < a.n) && (y0 + (t - k * a.rpi) < a.h);
          rowok |= (ok ? 1u : 0u) << u;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = AM::row(i, lane);
          const bool pv = ((rowok >> (row / TW)) & 1u) && ((row % TW) < xlim);
          pvmask[m] |= (pv ? 1u : 0u) << i;
        }
      }
    }
  }
  float st1[NSL], st2[NSL];
#pragma unroll
  for (int ns = 0; ns < NSL; ++ns) {
    const int n = ns / AM::NSUB, sub = ns % AM::NSUB;
    const int cl_ = (wn * NT + n) * 32 + AM::col(sub, lane);          // column inside the tile
    const int cn = nbase + cl_;
    const bool cvalid = cn < a.cout;
    const int cch = cvalid ? cn % a.cstat : 0;
    const float bv = (a.bias && cvalid) ? a.bias[cch] : 0.f;
    const float osc = (a.out_scale && cvalid) ? a.out_scale[cch] : 1.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int k = 0; k < AM::RPS; ++k) {
        const int i = AM::reg(sub, k);
        const int q = (wm * MT + m) * 32 + AM::row(i, lane);
        float vv = acc[m][n][i] * osc + bv;
        if (a.out_relu) vv = fmaxf(vv, 0.f);
        const T tv = (T)vv;
        ldsO[q * OPITCH + cl_] = tv;
        const float fv = ((pvmask[m] >> i) & 1u) ? (float)tv : 0.f;
        s1 += fv; s2 += fv * fv;
      }
    }
    st1[ns] = s1; st2[ns] = s2;
  }
  if (a.stats) {
#pragma unroll
    for (int ns = 0; ns < NSL; ++ns) {
      const int cl_ = (wn * NT + ns / AM::NSUB) * 32 + AM::col(ns % AM::NSUB, lane);
      const float s1 = AM::colsum(st1[ns]);
      const float s2 = AM::colsum(st2[ns]);
      if (AM::owner(lane)) { ldsS[(wm * 2 + 0) * BN + cl_] = s1; ldsS[(wm * 2 + 1) * BN + cl_] = s2; }
    }
  }
  __syncthreads();
  if (a.stats && tid < BN) {
    // one pair of atomics per output channel per workgroup (waves summed in fixed order)
    const int cn = nbase + tid;
    if (cn < a.cout) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) { t1 += ldsS[(w * 2 + 0) * BN + tid]; t2 += ldsS[(w * 2 + 1) * BN + tid]; }
      const int cch = cn % a.cstat;
      if (carry) { carry[0] += (double)t1; carry[1] += (double)t2; }
      else {
        satcv_stat_t* rowp = a.stats + (size_t)(blockIdx.x % SATCV_STAT_ROWS) * 2 * a.stats_ld;
        atomicAdd(rowp + cch, (satcv_stat_t)t1);
        atomicAdd(rowp + a.stats_ld + cch, (satcv_stat_t)t2);
      }
    }
  }
  // coalesced 16-byte stores of whole channel rows
  constexpr int EPV = 16 / (int)sizeof(T);        // elements per 16-byte vector
  constexpr int VPR = BN / EPV;                   // vectors per tile row
  T* yp = reinterpret_cast<T*>(a.y);
  const int ho = a.mode_out ? a.h * a.f : a.h;
  const int wo = a.mode_out ? a.w_ * a.f : a.w_;
  const int ncols = min(BN, a.cout - nbase);      // valid columns of this tile
  int cbase = nbase;
  // depth-to-space (transposed conv): a tile may span several (i, j) sub-pixel positions of cstat channels each -- with all f*f of
  // them in one workgroup the (j, channel) runs of an output row are written as whole contiguous lines.  NTHREADS is a multiple
  // of VPR, so the 16-byte column group of a thread (and with it its sub-pixel position) is fixed.
  static_assert(NTHREADS % VPR == 0, "column group of a thread must be loop-invariant");
  int ij = 0, cvec = (tid % VPR) * EPV;       // channel offset of this thread's vectors inside (ij, cbase)
  if (a.mode_out == 1) {
    const int cn0 = nbase + cvec;
    ij = cn0 / a.cstat;
    cbase = cn0 - ij * a.cstat; cvec = 0;
  }
  for (int it = tid; it < BM * VPR; it += NTHREADS) {
    const int q = it / VPR, vq = it % VPR;
    if (vq * EPV >= ncols) continue;
    const int t = q / TW, cx = q % TW;
    const int k = (a.imgs == 1) ? 0 : t / a.rpi;
    const int nimg = n0 + k, y = y0 + (t - k * a.rpi), x = x0 + cx;
    if (!((k < a.imgs) && (nimg < a.n) && (y < a.h) && (x < a.w_))) continue;
    size_t off;
    if (a.mode_out == 1) off = ((size_t)(nimg * ho + y * a.f + ij / a.f) * wo + x * a.f + ij % a.f) * a.ldy + cbase + cvec;
    else off = ((size_t)(nimg * ho + y) * wo + x) * a.ldy + cbase + vq * EPV;
    const T* sp = ldsO + q * OPITCH + vq * EPV;
    if (SKIP_STORES) continue;
    if (a.accumulate) {
      // accumulate == 2: y = ReLU(y + result) -- the residual join of a ResNet bottleneck written in place over the shortcut
      // (inference; engine.py fuses `add_relu` into the block's last convolution when nothing else reads the shortcut afterwards)
      const int ne = min(EPV, ncols - vq * EPV);
      if (ne == EPV && sizeof(T) == 2) {
        uint4 ov = *reinterpret_cast<const uint4*>(yp + off);
        const uint4 nv = *reinterpret_cast<const uint4*>(sp);
        T* oe = reinterpret_cast<T*>(&ov);
        const T* nn = reinterpret_cast<const T*>(&nv);
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
          float vv = (float)oe[e] + (float)nn[e];
          if (a.accumulate == 2) vv = fmaxf(vv, 0.f);
          oe[e] = (T)vv;
        }
        *reinterpret_cast<uint4*>(yp + off) = ov;
      } else {
        for (int e = 0; e < ne; ++e) {
          float vv = (float)yp[off + e] + (float)sp[e];
          if (a.accumulate == 2) vv = fmaxf(vv, 0.f);
          yp[off + e] = (T)vv;
        }
      }
    } else if (ncols - vq * EPV >= EPV) {
      *reinterpret_cast<uint4*>(yp + off) = *reinterpret_cast<const uint4*>(sp);
    } else {
      for (int e = 0; e < ncols - vq * EPV; ++e) yp[off + e] = sp[e];
    }
  }
  // fused max-pool of the tile just stored (inference encoder blocks): window == stride == pool_f, full windows only
  if (a.pool_y) {
    const int f = a.pool_f, pw = TW / f, ph = (BM / TW) / f;
    const int hp = a.h / f, wp = a.w_ / f;
    T* pp = reinterpret_cast<T*>(a.pool_y);
    for (int it = tid; it < ph * pw * VPR; it += NTHREADS) {
      const int vq = it % VPR, pq = it / VPR;
      if (vq * EPV + EPV > ncols) continue;
      const int t0 = (pq / pw) * f, c0 = (pq % pw) * f;
      const int k = (a.imgs == 1) ? 0 : t0 / a.rpi;
      const int nimg = n0 + k, y = y0 + (t0 - k * a.rpi), x = x0 + c0;
      if (!((k < a.imgs) && (nimg < a.n) && (y + f <= a.h) && (x + f <= a.w_))) continue;
      float mx[EPV];
#pragma unroll
      for (int e = 0; e < EPV; ++e) mx[e] = -INFINITY;
      for (int i = 0; i < f; ++i)
        for (int j = 0; j < f; ++j) {
          const T* sp = ldsO + ((t0 + i) * TW + c0 + j) * OPITCH + vq * EPV;
#pragma unroll
          for (int e = 0; e < EPV; ++e) mx[e] = fmaxf(mx[e], (float)sp[e]);
        }
      T o[EPV];
#pragma unroll
      for (int e = 0; e < EPV; ++e) o[e] = (T)mx[e];
      *reinterpret_cast<uint4*>(pp + ((size_t)(nimg * hp + y / f) * wp + x / f) * a.pool_ld + cbase + vq * EPV) = *reinterpret_cast<const uint4*>(o);
    }
  }
}

// LDS row pitch (pixels) of the staged halo tile: chosen per tile width so that the lanes of one ds_read_b128 group fall on distinct
// 16-byte bank slots.  32x32x16 fragments: 32 lanes span 32 / TW tile rows, so rows are padded (TW = 16: to a multiple of 16 pixels = 256 B;
// TW = 8: to 8 modulo 16).  16x16x32 fragments (m16) with TW >= 16: a lane quarter reads 16 consecutive pixels of one row -- any pitch.
__host__ __device__ constexpr int igemm_pitch(int tw, int cl, bool m16) {
  return (tw == 32 || (m16 && tw == 16)) ? cl : (tw == 16 ? ((cl + 15) / 16) * 16 : (cl <= 8 ? 8 : ((cl - 8 + 15) / 16) * 16 + 8));
}

static inline int igemm_pick_tw(int w) {
  int best = 8, bestpad = cdiv(w, 8) * 8;
  const int cands[2] = {16, 32};
  for (int i = 0; i < 2; ++i) {
    int p = cdiv(w, cands[i]) * cands[i];
    if (p <= bestpad) { best = cands[i]; bestpad = p; }
  }
  return best;
}

// software-pipelined variant (conv_igemm_fast.hip); returns SATCV_ERR_UNSUPPORTED when the
// shape is outside its static limits so that the caller falls back to the generic kernel.
int igemm_fast_launch(IgemmArgs& a, int dtype, hipStream_t st, bool dry_run = false);
// the deep 3x3 tile on v_mfma_f32_16x16x32_bf16 (conv_igemm_m16.hip); SATCV_ERR_UNSUPPORTED outside its limits
int igemm_m16_launch(IgemmArgs& a, int dtype, hipStream_t st, bool dry);        // (dtype must be SATCV_BF16: the kernels reinterpret x / w / y as bf16)
// the same tile as a persistent, cross-tile pipelined kernel for the mid layers (conv_igemm_m16p.hip); SATCV_ERR_UNSUPPORTED outside its limits
int igemm_m16p_launch(IgemmArgs& a, int dtype, hipStream_t st, bool dry);
// persistent weights-stationary kernel of the thin 3x3 layers (conv_igemm_ws.hip); SATCV_ERR_UNSUPPORTED outside its limits
int igemm_ws_launch(IgemmArgs& a, int dtype, hipStream_t st, bool dry_run);
// the thin 3x3 layers with wave roles (conv_thin_roles.hip); SATCV_ERR_UNSUPPORTED outside its limits
int igemm_tr_launch(IgemmArgs& a, int dtype, hipStream_t st, bool dry);
// streaming kernel of the thin transposed convolutions (conv_transpose_thin.hip); SATCV_ERR_UNSUPPORTED outside its limits
int convt_thin_launch(const IgemmArgs& a, int dtype, hipStream_t st);
int convt_thin_dgrad_launch(const IgemmArgs& a, int dtype, hipStream_t st);
