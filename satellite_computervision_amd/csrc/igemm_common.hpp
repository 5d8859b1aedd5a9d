// Shared argument block and MFMA fragment helpers of the implicit-GEMM convolution kernels.
#pragma once
#include "common.hpp"

struct IgemmArgs {
  const void* x0; const void* x1;
  int c0, c1;
  const float* in_scale; const float* in_shift; int in_relu;
  const void* w; const float* bias; const float* out_scale;
  void* y; int ldy;
  float* stats; int stats_ld;
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
  int dbg;                         // ablation bits (env SATCV_DBG): 1 skip stores, 2 skip MFMA, 4 skip A loads, 8 skip B loads
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
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
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
