// Shared device/host helpers for the satcv HIP library (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp8.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/satcv.h"

typedef __bf16 bf16;
typedef __hip_fp8_e4m3 fp8;          // OCP e4m3fn on gfx950 (hardware v_cvt_pk_fp8_f32 / v_cvt_f32_fp8, saturating)
// same storage, but staged in 16-channel items for the block-scaled K=64 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, 2x the bf16 rate)
struct fp8s : public __hip_fp8_e4m3 { using __hip_fp8_e4m3::__hip_fp8_e4m3; };
using i32x8 = __attribute__((ext_vector_type(8))) int;
using i32x4 = __attribute__((ext_vector_type(4))) int;

// K granularity of the implicit-GEMM staging: EL elements per staged 16-byte (bf16 / fp8s), 32-byte (f32) or 8-byte (fp8) item,
// SUB items per lane half per MFMA step (a lane half feeds 8 k to the 32x32x16 forms, 32 k to the scaled 32x32x64 form)
template <typename T> struct KTraits { static constexpr int EL = 8, SUB = 1; };
template <> struct KTraits<fp8s> { static constexpr int EL = 16, SUB = 2; };
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using short4v = __attribute__((ext_vector_type(4))) short;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// thread-local last error (C ABI: int status + satcv_last_error()).
void satcv_set_error(const char* fmt, ...);
#define SATCV_CHECK(cond, ...)                        \
  do {                                                \
    if (!(cond)) {                                    \
      satcv_set_error(__VA_ARGS__);                   \
      return SATCV_ERR_INVALID;                       \
    }                                                 \
  } while (0)
#define SATCV_HIP(call)                                                        \
  do {                                                                         \
    hipError_t e__ = (call);                                                   \
    if (e__ != hipSuccess) {                                                   \
      satcv_set_error("%s failed: %s", #call, hipGetErrorString(e__));         \
      return SATCV_ERR_HIP;                                                    \
    }                                                                          \
  } while (0)

// opt a kernel into > 48 KB of dynamic LDS, once per (kernel, device); thread-safe (api.hip)
int satcv_ensure_dynamic_lds(const void* kern, size_t bytes);

__host__ __device__ static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
// n * h * w * f * f pixels (f = depth-to-space / space-to-depth factor or 1) stay below 2^31 -- evaluated without overflowing anything itself
static inline bool satcv_pixels_ok(long long n, long long h, long long w, long long f) {
  if (n < 1 || h < 1 || w < 1 || f < 0 || f > 16 || n > (1LL << 31) || h > (1LL << 31) || w > (1LL << 31)) return false;
  const long long hw = h * w;                            // < 2^62
  if (hw >= (1LL << 31)) return false;
  const long long t = n * hw;                            // < 2^62
  if (t >= (1LL << 31)) return false;
  return t * (f > 1 ? f * f : 1) < (1LL << 31);
}

// ---- 8-element vectors of the storage type (16 B for bf16, 32 B for f32) ----
template <typename T>
struct Vec8;
template <>
struct Vec8<bf16> {
  bf16x8 v;
};
template <>
struct Vec8<float> {
  float v[8];
};

template <typename T>
__device__ __forceinline__ void load8(const T* p, float (&out)[8]) {
  if constexpr (std::is_same<T, bf16>::value) {
    bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = (float)v[i];
  } else if constexpr (std::is_same<T, fp8>::value) {
    const uint2 q = *reinterpret_cast<const uint2*>(p);
    const fp8* b = reinterpret_cast<const fp8*>(&q);
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = (float)b[i];
  } else {
    float4 a = reinterpret_cast<const float4*>(p)[0];
    float4 b = reinterpret_cast<const float4*>(p)[1];
    out[0] = a.x; out[1] = a.y; out[2] = a.z; out[3] = a.w;
    out[4] = b.x; out[5] = b.y; out[6] = b.z; out[7] = b.w;
  }
}

template <typename T>
__device__ __forceinline__ void store8(T* p, const float (&in)[8]) {
  if constexpr (std::is_same<T, bf16>::value) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16)in[i];
    *reinterpret_cast<bf16x8*>(p) = v;
  } else if constexpr (std::is_same<T, fp8>::value) {
    uint2 q;
    fp8* b = reinterpret_cast<fp8*>(&q);
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = (fp8)in[i];
    *reinterpret_cast<uint2*>(p) = q;
  } else {
    reinterpret_cast<float4*>(p)[0] = make_float4(in[0], in[1], in[2], in[3]);
    reinterpret_cast<float4*>(p)[1] = make_float4(in[4], in[5], in[6], in[7]);
  }
}

// ---- raw 8-element (16/32-byte) moves and the in-register BN affine + ReLU ----
template <typename T>
struct Raw8 {
  static constexpr int NQ = sizeof(T) / 2;      // 16-byte quads per 8 elements
  uint4 q[NQ];
};
template <>
struct Raw8<fp8> {
  uint2 q;                                      // 8 bytes
};
template <>
struct Raw8<fp8s> {
  static constexpr int NQ = 1;                  // 16 elements = 16 bytes
  uint4 q[NQ];
};

template <typename T>
__device__ __forceinline__ Raw8<T> gload8(const T* p) {
  Raw8<T> r;
  if constexpr (std::is_same<T, fp8>::value) {
    r.q = *reinterpret_cast<const uint2*>(p);
  } else {
#pragma unroll
    for (int i = 0; i < Raw8<T>::NQ; ++i) r.q[i] = reinterpret_cast<const uint4*>(p)[i];
  }
  return r;
}
template <typename T>
__device__ __forceinline__ void lstore8(T* p, const Raw8<T>& r) {
  if constexpr (std::is_same<T, fp8>::value) {
    *reinterpret_cast<uint2*>(p) = r.q;
  } else {
#pragma unroll
    for (int i = 0; i < Raw8<T>::NQ; ++i) reinterpret_cast<uint4*>(p)[i] = r.q[i];
  }
}
template <typename T>
__device__ __forceinline__ Raw8<T> zero8() {
  Raw8<T> r;
  if constexpr (std::is_same<T, fp8>::value) {
    r.q = make_uint2(0, 0);
  } else {
#pragma unroll
    for (int i = 0; i < Raw8<T>::NQ; ++i) r.q[i] = make_uint4(0, 0, 0, 0);
  }
  return r;
}
// bf16 item: y = x * sc + sh in fp32, rounded to bf16, then the ReLU as a signed 16-bit maximum on the packed pairs (a negative float
// is a negative integer, -0 becomes +0): lim = 0 clamps, lim = 0x80008000 passes everything.  One v_pk_max_i16 per channel pair instead
// of a v_max_f32 AND a v_cndmask on the runtime relu flag per channel (the loaders of the thin layers are VALU-bound).
typedef short short2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned relu_pk_bf16(unsigned w, unsigned lim) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(short2v, w), __builtin_bit_cast(short2v, lim)));
}
__device__ __forceinline__ Raw8<bf16> affine8_lim(const Raw8<bf16>& r, const float* sc, const float* sh, unsigned lim) {
  bf16x8 v = __builtin_bit_cast(bf16x8, r.q[0]);
  bf16x8 w;
#pragma unroll
  for (int e = 0; e < 8; ++e) w[e] = (bf16)((float)v[e] * sc[e] + sh[e]);
  uint4 u = __builtin_bit_cast(uint4, w);
  Raw8<bf16> o;
  o.q[0] = make_uint4(relu_pk_bf16(u.x, lim), relu_pk_bf16(u.y, lim), relu_pk_bf16(u.z, lim), relu_pk_bf16(u.w, lim));
  return o;
}
template <typename T>
__device__ __forceinline__ Raw8<T> affine8(const Raw8<T>& r, const float* sc, const float* sh, int relu) {
  Raw8<T> o;
  if constexpr (std::is_same<T, bf16>::value) {
    o = affine8_lim(r, sc, sh, relu ? 0u : 0x80008000u);
  } else if constexpr (std::is_same<T, fp8>::value) {
    const fp8* f = reinterpret_cast<const fp8*>(&r.q);
    fp8* g = reinterpret_cast<fp8*>(&o.q);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = (float)f[e] * sc[e] + sh[e];
      g[e] = (fp8)(relu ? fmaxf(t, 0.f) : t);
    }
  } else if constexpr (std::is_same<T, fp8s>::value) {
    const fp8* f = reinterpret_cast<const fp8*>(&r.q[0]);
    fp8* g = reinterpret_cast<fp8*>(&o.q[0]);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float t = (float)f[e] * sc[e] + sh[e];
      g[e] = (fp8)(relu ? fmaxf(t, 0.f) : t);
    }
  } else {
    const float* f = reinterpret_cast<const float*>(&r.q[0]);
    float* g = reinterpret_cast<float*>(&o.q[0]);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = f[e] * sc[e] + sh[e];
      g[e] = relu ? fmaxf(t, 0.f) : t;
    }
  }
  return o;
}

// the same with the 8 scale + 8 shift values already in registers (rs[0..1] scale, rs[2..3] shift): bf16 / f32 / 8-element fp8 items
template <typename T>
__device__ __forceinline__ Raw8<T> affine8r(const Raw8<T>& r, const float4 (&rs)[4], int relu) {
  const float sc[8] = {rs[0].x, rs[0].y, rs[0].z, rs[0].w, rs[1].x, rs[1].y, rs[1].z, rs[1].w};
  const float sh[8] = {rs[2].x, rs[2].y, rs[2].z, rs[2].w, rs[3].x, rs[3].y, rs[3].z, rs[3].w};
  Raw8<T> o;
  if constexpr (std::is_same<T, bf16>::value) {
    o = affine8_lim(r, sc, sh, relu ? 0u : 0x80008000u);
  } else if constexpr (std::is_same<T, fp8>::value) {
    const fp8* f = reinterpret_cast<const fp8*>(&r.q);
    fp8* g = reinterpret_cast<fp8*>(&o.q);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = (float)f[e] * sc[e] + sh[e];
      g[e] = (fp8)(relu ? fmaxf(t, 0.f) : t);
    }
  } else {
    const float* f = reinterpret_cast<const float*>(&r.q[0]);
    float* g = reinterpret_cast<float*>(&o.q[0]);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = f[e] * sc[e] + sh[e];
      g[e] = relu ? fmaxf(t, 0.f) : t;
    }
  }
  return o;
}
// 8 stored elements held in registers -> fp32
template <typename T>
__device__ __forceinline__ void unpack8(const Raw8<T>& r, float (&out)[8]) {
  if constexpr (std::is_same<T, bf16>::value) {
    const bf16x8 v = __builtin_bit_cast(bf16x8, r.q[0]);
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = (float)v[i];
  } else if constexpr (std::is_same<T, float>::value) {
    const float* f = reinterpret_cast<const float*>(&r.q[0]);
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = f[i];
  } else {
    const fp8* b = reinterpret_cast<const fp8*>(&r.q);
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = (float)b[i];
  }
}
// ablation builds: keep a loaded item alive without storing it
template <typename T>
__device__ __forceinline__ void keep8(const Raw8<T>& r) {
  if constexpr (std::is_same<T, fp8>::value) { asm volatile("" :: "v"(r.q.x), "v"(r.q.y)); }
  else {
#pragma unroll
    for (int i = 0; i < Raw8<T>::NQ; ++i) asm volatile("" :: "v"(r.q[i].x), "v"(r.q[i].y), "v"(r.q[i].z), "v"(r.q[i].w));
  }
}
// keep ? r : 0 without a branch
template <typename T>
__device__ __forceinline__ Raw8<T> select8(bool keep, const Raw8<T>& r) {
  Raw8<T> o;
  if constexpr (std::is_same<T, fp8>::value) {
    o.q = make_uint2(keep ? r.q.x : 0u, keep ? r.q.y : 0u);
  } else {
#pragma unroll
    for (int i = 0; i < Raw8<T>::NQ; ++i) o.q[i] = make_uint4(keep ? r.q[i].x : 0u, keep ? r.q[i].y : 0u, keep ? r.q[i].z : 0u, keep ? r.q[i].w : 0u);
  }
  return o;
}

// value as it will read back from storage (bf16 rounding, identity for f32)
template <typename T>
__device__ __forceinline__ float round_to(float x) {
  if constexpr (std::is_same<T, bf16>::value) return (float)(bf16)x;
  else if constexpr (std::is_same<T, fp8>::value) return (float)(fp8)x;
  else return x;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- LDS-DMA (global_load_lds_dwordx4): one wave-instruction moves 64 x 16 bytes from per-lane global addresses to 1 KB of CONTIGUOUS
// LDS (lane l lands at lds_dst + 16 l), with no VGPR destination and no ds_write.  Inline asm so that hipcc neither counts it in its own
// s_waitcnt bookkeeping nor orders LDS reads behind it: the caller waits (dma_wait_all / a counted s_waitcnt vmcnt) and then passes a
// workgroup barrier before any wave reads the bytes (cdna_hip_programming.md section 5.7).  M0 carries the LDS base for the
// instruction and is restored.  gbase: wave-uniform base pointer (SGPR pair), voff: this lane's byte offset from it (32 bit).
typedef __attribute__((address_space(3))) char satcv_lds_char;
__device__ __forceinline__ unsigned lds_addr_of(const void* p) { return (unsigned)(uintptr_t)(satcv_lds_char*)p; }
__device__ __forceinline__ void lds_dma16(const void* gbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(gbase), "v"(voff), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// number of replica rows used for per-channel atomics (spreads contention; the
// consumer sums the rows in fixed order).
#define SATCV_STAT_REPL 32
