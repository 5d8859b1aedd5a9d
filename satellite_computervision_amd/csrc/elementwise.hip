// HBM-bound kernels of the U-Net path: ingest, weight packing, BatchNorm statistics /
// apply / backward, ReLU + max-pool, softmax/sigmoid head, losses, confusion matrix, Adam.
// All are vectorised over 8 channels (16 B bf16 / 32 B f32 per lane) of NHWC tensors.
#include "common.hpp"
#include <cstdlib>

#define EW_BLOCK 256
// grid-stride kernels: a bounded number of workgroups per CU.  Round 1 (weight gradients resident on every CU): 3 per CU, so that
// every workgroup is resident from the start -- with 8 per CU the kernels ran 1.6 "rounds" and crowded out the weight-gradient
// stream (5070 -> 5350 tiles/s).  With the weight gradients on 160 workgroups (round 2) the count hardly matters: 2 per CU +2.7 %
// step time, 3 / 6 / 8 / 12 / 16 within 0.3 %, 6 the best by a hair (9.50 vs 9.52 ms).  SATCV_EW_PER_CU overrides.
static inline int ew_grid(long long items, int cap = 0) {
  static const int dflt = [] { const char* e = getenv("SATCV_EW_PER_CU"); const int v = e ? atoi(e) : 6; return 256 * (v >= 1 ? v : 6); }();
  if (cap <= 0) cap = dflt;
  long long b = (items + EW_BLOCK - 1) / EW_BLOCK;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}
#define LAUNCH_OK(name)                                                          \
  do {                                                                           \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) {                                                     \
      satcv_set_error(name " launch: %s", hipGetErrorString(e__));               \
      return SATCV_ERR_HIP;                                                      \
    }                                                                            \
  } while (0)
#define DISPATCH_T(dtype, ...)                                                   \
  do {                                                                           \
    if ((dtype) == SATCV_BF16) { using T = bf16; __VA_ARGS__; }                  \
    else if ((dtype) == SATCV_F32) { using T = float; __VA_ARGS__; }             \
    else { satcv_set_error("bad dtype %d", (int)(dtype)); return SATCV_ERR_INVALID; } \
  } while (0)

// storage dtypes of the inference-only kernels (adds OCP fp8 e4m3)
#define DISPATCH_T8(dtype, ...)                                                  \
  do {                                                                           \
    if ((dtype) == SATCV_BF16) { using T = bf16; __VA_ARGS__; }                  \
    else if ((dtype) == SATCV_F32) { using T = float; __VA_ARGS__; }             \
    else if ((dtype) == SATCV_FP8) { using T = fp8; __VA_ARGS__; }               \
    else { satcv_set_error("bad dtype %d", (int)(dtype)); return SATCV_ERR_INVALID; } \
  } while (0)

// ------------------------------------------------------------------------ ingest
template <typename T>
__global__ void ingest_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, long long npix, int c, int cpad, float mul) {
  const int G = cpad / 8;
  const long long total = npix * G;
  for (long long it = blockIdx.x * (long long)blockDim.x + threadIdx.x; it < total; it += (long long)gridDim.x * blockDim.x) {
    const long long p = it / G; const int g = (int)(it % G);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { const int ch = g * 8 + e; v[e] = ch < c ? src[p * c + ch] * mul : 0.f; }
    store8<T>(dst + p * cpad + g * 8, v);
  }
}
extern "C" int satcv_ingest_nhwc(const float* src, void* dst, int64_t npix, int32_t c, int32_t cpad, int32_t dtype, void* stream) {
  SATCV_CHECK(src && dst && npix > 0 && c > 0 && cpad >= c && cpad % 8 == 0, "ingest_nhwc: bad args");
  DISPATCH_T8(dtype, hipLaunchKernelGGL(ingest_nhwc_kernel<T>, dim3(ew_grid(npix * (cpad / 8))), dim3(EW_BLOCK), 0, (hipStream_t)stream, src, (T*)dst, (long long)npix, c, cpad, 1.0f));
  LAUNCH_OK("ingest_nhwc");
  return SATCV_OK;
}
extern "C" int satcv_ingest_nhwc_scaled(const float* src, void* dst, int64_t npix, int32_t c, int32_t cpad, float mul, int32_t dtype, void* stream) {
  SATCV_CHECK(src && dst && npix > 0 && c > 0 && cpad >= c && cpad % 8 == 0, "ingest_nhwc_scaled: bad args");
  DISPATCH_T8(dtype, hipLaunchKernelGGL(ingest_nhwc_kernel<T>, dim3(ew_grid(npix * (cpad / 8))), dim3(EW_BLOCK), 0, (hipStream_t)stream, src, (T*)dst, (long long)npix, c, cpad, mul));
  LAUNCH_OK("ingest_nhwc_scaled");
  return SATCV_OK;
}

template <typename T, typename S>
__global__ void ingest_chw_kernel(const S* __restrict__ src, float scale, T* __restrict__ dst, int n, int c, int hw, int cpad) {
  const int G = cpad / 8;
  const long long total = (long long)n * hw * G;
  for (long long it = blockIdx.x * (long long)blockDim.x + threadIdx.x; it < total; it += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(it / ((long long)n * hw));           // group slowest: planes read coalesced
    const long long p = it % ((long long)n * hw);
    const long long img = p / hw; const int pix = (int)(p % hw);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ch = g * 8 + e;
      v[e] = ch < c ? (float)src[(img * c + ch) * hw + pix] * scale : 0.f;
    }
    store8<T>(dst + p * cpad + g * 8, v);
  }
}
extern "C" int satcv_ingest_chw(const void* src, int32_t src_kind, float scale, void* dst, int32_t n, int32_t c, int32_t h, int32_t w, int32_t cpad, int32_t dtype, void* stream) {
  SATCV_CHECK(src && dst && n > 0 && c > 0 && cpad >= c && cpad % 8 == 0, "ingest_chw: bad args");
  const int hw = h * w; const int grid = ew_grid((long long)n * hw * (cpad / 8));
  DISPATCH_T(dtype, {
    if (src_kind == 0) hipLaunchKernelGGL((ingest_chw_kernel<T, uint8_t>), dim3(grid), dim3(EW_BLOCK), 0, (hipStream_t)stream, (const uint8_t*)src, scale, (T*)dst, n, c, hw, cpad);
    else if (src_kind == 1) hipLaunchKernelGGL((ingest_chw_kernel<T, uint16_t>), dim3(grid), dim3(EW_BLOCK), 0, (hipStream_t)stream, (const uint16_t*)src, scale, (T*)dst, n, c, hw, cpad);
    else if (src_kind == 2) hipLaunchKernelGGL((ingest_chw_kernel<T, float>), dim3(grid), dim3(EW_BLOCK), 0, (hipStream_t)stream, (const float*)src, scale, (T*)dst, n, c, hw, cpad);
    else { satcv_set_error("ingest_chw: src_kind %d", src_kind); return SATCV_ERR_INVALID; }
  });
  LAUNCH_OK("ingest_chw");
  return SATCV_OK;
}

static inline int rup(int a, int b) { return (a + b - 1) / b * b; }

// ------------------------------------------------------------------ weight packing
// mode 0: conv fwd    dst[tap][k8][npad][8], k = ci, n = co          src HWIO (tap,ci,co)
// mode 1: conv dgrad  dst[tap'][k8][npad][8], k = co, n = ci, tap' flipped
// mode 2: convT fwd   dst[0][k8][npad][8], k = ci, n = (ij)*cout+o   src (ij,o,ci)
// mode 3: convT dgrad dst[0][k8][npad][8], k = (ij)*cout+o, n = ci
template <typename T, int EL = 8>
__global__ void pack_kernel(const float* __restrict__ src, T* __restrict__ dst, int mode, int taps, int cin, int cout, int kpad, int npad) {
  const int ptaps = (mode >= 2) ? 1 : taps;
  const long long total = (long long)ptaps * (kpad / EL) * npad;
  for (long long it = blockIdx.x * (long long)blockDim.x + threadIdx.x; it < total; it += (long long)gridDim.x * blockDim.x) {
    const int nn = (int)(it % npad);
    const int k8 = (int)((it / npad) % (kpad / EL));
    const int tap = (int)(it / ((long long)npad * (kpad / EL)));
    float v[EL];
#pragma unroll
    for (int e = 0; e < EL; ++e) {
      const int k = k8 * EL + e;
      float x = 0.f;
      if (mode == 0) { if (k < cin && nn < cout) x = src[((size_t)tap * cin + k) * cout + nn]; }
      else if (mode == 1) { if (k < cout && nn < cin) x = src[((size_t)(taps - 1 - tap) * cin + nn) * cout + k]; }
      else if (mode == 2) { if (k < cin && nn < taps * cout) x = src[(size_t)nn * cin + k]; }
      else { if (k < taps * cout && nn < cin) x = src[(size_t)k * cin + nn]; }
      v[e] = x;
    }
    if constexpr (EL == 8) {
      store8<T>(dst + it * 8, v);
    } else {
      float lo[8], hi[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { lo[e] = v[e]; hi[e] = v[8 + e]; }
      store8<T>(dst + it * 16, lo);
      store8<T>(dst + it * 16 + 8, hi);
    }
  }
}
// every layer's operand images in ONE launch (the per-layer form costs ~40 tiny launches on the critical path of a step):
// a device table of pack jobs + the exclusive prefix sum of their 16-byte output items; each thread finds its job by binary
// search and does what pack_kernel does.
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void pack_batched_kernel(const satcv_pack_job* __restrict__ jobs, const long long* __restrict__ prefix, int njobs, long long total) {
  // Round 6: the job of an item was found by a binary search PER THREAD (six dependent loads of the prefix table + the 48-byte job, ahead of the
  // eight gathered loads of every item: 89 us for 148 MB).  A job's item count is now rounded up to a multiple of the block size
  // (satcv_pack_job_items), so a block of 256 consecutive items belongs to ONE job: the search runs on block-uniform values (scalar loads),
  // once per block of items.
  for (long long g0 = (long long)blockIdx.x * EW_BLOCK; g0 < total; g0 += (long long)gridDim.x * EW_BLOCK) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (prefix[mid] <= g0) lo = mid; else hi = mid - 1; }
    lo = __builtin_amdgcn_readfirstlane(lo);
    const satcv_pack_job j = jobs[lo];
    long long it = g0 - prefix[lo] + threadIdx.x;
    const long long real = (long long)(j.mode >= 2 ? 1 : j.taps) * (j.kpad / 8) * j.npad;
    if (it >= real) continue;
    if (j.mode == 1 && (j.kpad / 8) % 4 == 0 && j.npad % 16 == 0) {
      // data-gradient image of a 3x3 kernel: k runs over Cout (contiguous in the Keras kernel), nn over Cin (stride Cout).  With nn on
      // the lanes every lane read its 32 bytes from another 128-byte line (4x the lines); here 4 lanes cover the 4 x 32 bytes of one
      // line (4 consecutive k8 of one nn) and 16 consecutive nn follow -- same items, another thread-to-item map
      const int nb = j.npad / 16, kb = (j.kpad / 8) / 4;
      const int l = (int)(it % 64); long long t = it / 64;
      const int nblk = (int)(t % nb); t /= nb;
      const int kblk = (int)(t % kb); const long long tap_ = t / kb;
      const int k8_ = kblk * 4 + (l & 3), nn_ = nblk * 16 + (l >> 2);
      it = (tap_ * (j.kpad / 8) + k8_) * j.npad + nn_;
    }
    const int nn = (int)(it % j.npad);
    const int k8 = (int)((it / j.npad) % (j.kpad / 8));
    const int tap = (int)(it / ((long long)j.npad * (j.kpad / 8)));
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = k8 * 8 + e;
      float x = 0.f;
      if (j.mode == 0) { if (k < j.cin && nn < j.cout) x = j.src[((size_t)tap * j.cin + k) * j.cout + nn]; }
      else if (j.mode == 1) { if (k < j.cout && nn < j.cin) x = j.src[((size_t)(j.taps - 1 - tap) * j.cin + nn) * j.cout + k]; }
      else if (j.mode == 2) { if (k < j.cin && nn < j.taps * j.cout) x = j.src[(size_t)nn * j.cin + k]; }
      else { if (k < j.taps * j.cout && nn < j.cin) x = j.src[(size_t)k * j.cin + nn]; }
      v[e] = x;
    }
    store8<T>(reinterpret_cast<T*>(j.dst) + it * 8, v);
  }
}
extern "C" int64_t satcv_pack_job_items(const satcv_pack_job* j) {
  if (!j || j->kpad <= 0 || j->kpad % 8 || j->npad <= 0 || j->mode < 0 || j->mode > 3) return -1;
  // (rounded up to the block size of the batched kernel: a block of items never straddles two jobs)
  const int64_t real = (int64_t)(j->mode >= 2 ? 1 : j->taps) * (j->kpad / 8) * j->npad;
  return (real + EW_BLOCK - 1) / EW_BLOCK * EW_BLOCK;
}
extern "C" int satcv_pack_weights_batched(const satcv_pack_job* jobs_dev, const int64_t* prefix_dev, int32_t njobs, int64_t total_items, int32_t dtype,
                                          void* stream) {
  SATCV_CHECK(jobs_dev && prefix_dev && njobs > 0 && total_items > 0, "pack_weights_batched: bad args");
  SATCV_CHECK(total_items % EW_BLOCK == 0, "pack_weights_batched: the prefix table must be built from satcv_pack_job_items (multiples of 256)");
  DISPATCH_T8(dtype, hipLaunchKernelGGL(pack_batched_kernel<T>, dim3(ew_grid(total_items)), dim3(EW_BLOCK), 0, (hipStream_t)stream, jobs_dev,
                                        (const long long*)prefix_dev, njobs, (long long)total_items));
  LAUNCH_OK("pack_weights_batched");
  return SATCV_OK;
}

extern "C" int satcv_pack_weights(const float* src, void* dst_fwd, void* dst_dgrad, int32_t kh, int32_t kw, int32_t cin, int32_t cout, int32_t cin_pad, int32_t transposed, int32_t dtype, void* stream) {
  SATCV_CHECK(src && cin > 0 && cout > 0 && cin_pad >= cin && cin_pad % 16 == 0, "pack_weights: bad args");
  const int taps = kh * kw;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SATCV_FP8X) {          // scaled fp8: 16-channel granules, forward images only (inference)
    SATCV_CHECK(dst_fwd && !dst_dgrad && cin_pad % 16 == 0, "pack_weights: the scaled fp8 image is forward-only, cin_pad %% 16 == 0");
    const int kp = cin_pad, np = rup(transposed ? taps * cout : cout, 32);
    const long long items = (long long)(transposed ? 1 : taps) * (kp / 16) * np;
    hipLaunchKernelGGL((pack_kernel<fp8, 16>), dim3(ew_grid(items)), dim3(EW_BLOCK), 0, st, src, (fp8*)dst_fwd, transposed ? 2 : 0, taps, cin, cout, kp, np);
    LAUNCH_OK("pack_weights");
    return SATCV_OK;
  }
  DISPATCH_T8(dtype, {
    if (!transposed) {
      if (dst_fwd) { const int kp = cin_pad, np = rup(cout, 32);
        hipLaunchKernelGGL(pack_kernel<T>, dim3(ew_grid((long long)taps * (kp / 8) * np)), dim3(EW_BLOCK), 0, st, src, (T*)dst_fwd, 0, taps, cin, cout, kp, np); }
      if (dst_dgrad) { const int kp = rup(cout, 16), np = rup(cin, 32);
        hipLaunchKernelGGL(pack_kernel<T>, dim3(ew_grid((long long)taps * (kp / 8) * np)), dim3(EW_BLOCK), 0, st, src, (T*)dst_dgrad, 1, taps, cin, cout, kp, np); }
    } else {
      if (dst_fwd) { const int kp = cin_pad, np = rup(taps * cout, 32);
        hipLaunchKernelGGL(pack_kernel<T>, dim3(ew_grid((long long)(kp / 8) * np)), dim3(EW_BLOCK), 0, st, src, (T*)dst_fwd, 2, taps, cin, cout, kp, np); }
      if (dst_dgrad) { const int kp = rup(taps * cout, 16), np = rup(cin, 32);
        hipLaunchKernelGGL(pack_kernel<T>, dim3(ew_grid((long long)(kp / 8) * np)), dim3(EW_BLOCK), 0, st, src, (T*)dst_dgrad, 3, taps, cin, cout, kp, np); }
    }
  });
  LAUNCH_OK("pack_weights");
  return SATCV_OK;
}

// -------------------------------------------------------------------- BN finalize
__global__ void bn_finalize_train_kernel(satcv_stat_t* stats, int ld, int c, float count, const float* gamma, const float* beta,
                                         float eps, float momentum, int updates, int bessel, float* mm, float* mv,
                                         float* scale, float* shift, float* mean_o, float* rstd_o) {
  // 32 lanes per channel, one replica row each (one load round trip instead of 32 dependent ones: the launch sits on the critical
  // path of every layer); the rows are combined by a fixed xor butterfly, i.e. in the same association every run
  static_assert(SATCV_STAT_ROWS == 32, "one lane per replica row");
  const int ch = blockIdx.x * (blockDim.x / 32) + threadIdx.x / 32, r = threadIdx.x & 31;
  double s1 = 0.0, s2 = 0.0;
  float gam = 0.f, bet = 0.f, mm0 = 0.f, mv0 = 0.f;       // the parameters travel with the replica rows: one load round trip, not two
  if (ch < c) {
    satcv_stat_t* row = stats + (size_t)r * 2 * ld;
    s1 = row[ch]; s2 = row[ld + ch];
    if (r == 0) { gam = gamma[ch]; bet = beta[ch]; if (mm) { mm0 = mm[ch]; mv0 = mv[ch]; } }
    row[ch] = 0.0; row[ld + ch] = 0.0;
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
  if (ch >= c || r != 0) return;
  const double mean_d = s1 / (double)count;
  const float mean = (float)mean_d;
  float var = (float)(s2 / (double)count - mean_d * mean_d);      // E[x^2] - E[x]^2 without the fp32 cancellation
  var = fmaxf(var, 0.f);
  const float rstd = rsqrtf(var + eps);
  const float sc = gam * rstd;
  scale[ch] = sc; shift[ch] = bet - mean * sc;
  mean_o[ch] = mean; rstd_o[ch] = rstd;
  if (mm) {
    const float vv = bessel ? var * (count / fmaxf(count - 1.f, 1.f)) : var;
    float a = mm0, b = mv0;
    for (int u = 0; u < updates; ++u) { a = a * momentum + mean * (1.f - momentum); b = b * momentum + vv * (1.f - momentum); }
    mm[ch] = a; mv[ch] = b;
  }
}
extern "C" int satcv_bn_finalize_train(satcv_stat_t* stats, int32_t stats_ld, int32_t c, float count, const float* gamma, const float* beta,
                                       float eps, float momentum, int32_t updates, int32_t bessel, float* moving_mean, float* moving_var,
                                       float* scale, float* shift, float* mean, float* rstd, void* stream) {
  SATCV_CHECK(stats && gamma && beta && scale && shift && mean && rstd && c > 0 && stats_ld >= c && count > 0, "bn_finalize_train: bad args");
  hipLaunchKernelGGL(bn_finalize_train_kernel, dim3(cdiv(c, 8)), dim3(256), 0, (hipStream_t)stream, stats, stats_ld, c, count, gamma, beta,
                     eps, momentum, updates, bessel, moving_mean, moving_var, scale, shift, mean, rstd);
  LAUNCH_OK("bn_finalize_train");
  return SATCV_OK;
}
__global__ void bn_affine_infer_kernel(const float* gamma, const float* beta, const float* mm, const float* mv, float eps, int c, float* scale, float* shift) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  const float sc = gamma[ch] / sqrtf(mv[ch] + eps);
  scale[ch] = sc; shift[ch] = beta[ch] - mm[ch] * sc;
}
extern "C" int satcv_bn_affine_infer(const float* gamma, const float* beta, const float* moving_mean, const float* moving_var, float eps, int32_t c, float* scale, float* shift, void* stream) {
  SATCV_CHECK(gamma && beta && moving_mean && moving_var && scale && shift && c > 0, "bn_affine_infer: bad args");
  hipLaunchKernelGGL(bn_affine_infer_kernel, dim3(cdiv(c, 128)), dim3(128), 0, (hipStream_t)stream, gamma, beta, moving_mean, moving_var, eps, c, scale, shift);
  LAUNCH_OK("bn_affine_infer");
  return SATCV_OK;
}

// every BatchNormalization of an inference plan in ONE launch (a ResNet-50 plan has 58 of them, ~5 us each as launches of their own:
// 6 % of the kernel time of a single 512 x 512 tile): one block per job
__global__ void bn_affine_infer_batched_kernel(const satcv_bn_affine_job* __restrict__ jobs, float eps) {
  const satcv_bn_affine_job j = jobs[blockIdx.x];
  for (int ch = threadIdx.x; ch < j.c; ch += blockDim.x) {
    const float sc = j.gamma[ch] / sqrtf(j.moving_var[ch] + eps);
    const float sh = j.beta[ch] - j.moving_mean[ch] * sc;
    j.scale[ch] = sc; j.shift[ch] = sh;
    if (j.bias_eff) j.bias_eff[ch] = sh + (j.conv_bias ? j.conv_bias[ch] * sc : 0.f);      // BN(acc + b) = sc * acc + (sc * b + sh)
  }
}
extern "C" int satcv_bn_affine_infer_batched(const satcv_bn_affine_job* jobs_device, int32_t njobs, float eps, void* stream) {
  SATCV_CHECK(jobs_device && njobs > 0, "bn_affine_infer_batched: bad args");
  hipLaunchKernelGGL(bn_affine_infer_batched_kernel, dim3(njobs), dim3(256), 0, (hipStream_t)stream, jobs_device, eps);
  LAUNCH_OK("bn_affine_infer_batched");
  return SATCV_OK;
}

// Block-level per-channel accumulation into LDS then one global atomic per channel per block.
// Threads iterate items = (window, group) with a stride that is a multiple of G so that a
// thread's channel group never changes.
// lds: EW_BLOCK * 16 floats.  Every thread parks its 16 partial sums; thread i then adds, in increasing thread order, the
// partials of the threads that own channel i's group (thread t owns group (first_group + t) % G).  A fixed order, so the value a
// workgroup contributes is reproducible; the replica rows are double (satcv_stat_t), where the arrival order of the workgroups
// only matters at 1e-16.
__device__ __forceinline__ void block_channel_reduce(float* lds, const float (&a)[8], const float (&b)[8], int g, bool active,
                                                     int c, satcv_stat_t* out, int out_ld, int c_out = -1) {
  const int G = c / 8;
  if (c_out < 0) c_out = c;                       // (c_out < c: only the first c_out channels own output rows -- the skip half of a concatenation)
  // [16][blockDim + 1]: a value row per partial sum, one column per thread -- the lanes of a wave write consecutive banks, and the
  // readers below (8 lanes per thread column, one per value row) are spread by the odd pitch.  (The [thread][16] form put every
  // fourth lane on the same bank: rocprofv3 counted 88 % of the LDS cycles of the BatchNorm kernels as bank conflicts.)
  // (round 6: pitch = block + 8, i.e. 8 modulo 32 banks -- the readers of one 32-lane half are 8 value rows x 4 consecutive thread columns: with
  //  the odd pitch of round 3 row e and column t met row e + 1 and column t - 1 on one bank, and rocprofv3 still counted half of these kernels' LDS
  //  cycles as conflicts; every launch site sizes the region (EW_BLOCK + 8) x 16 floats)
  const int pitch = blockDim.x + 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) { lds[e * pitch + threadIdx.x] = active ? a[e] : 0.f; lds[(8 + e) * pitch + threadIdx.x] = active ? b[e] : 0.f; }
  __syncthreads();
  const int g0 = (int)((blockIdx.x * (long long)blockDim.x) % G);          // group of thread 0
  satcv_stat_t* row = out + (size_t)(blockIdx.x % SATCV_STAT_ROWS) * 2 * out_ld;
  for (int i = threadIdx.x; i < 2 * c_out; i += blockDim.x) {
    const int which = i / c_out, ch = i - which * c_out, gg = ch / 8, e = ch % 8;
    float s = 0.f;
    for (int t = (gg - g0 + G) % G; t < (int)blockDim.x; t += G) s += lds[(which * 8 + e) * pitch + t];
    atomicAdd(row + which * out_ld + ch, (satcv_stat_t)s);
  }
}

// ------------------------------------------------------------- BN + ReLU + maxpool
// AMAX: also the index (i * f + j, first maximum in row-major window order -- the rule of bn_bwd_kernel) of every pooling window's maximum,
// one byte per pooled pixel and channel: the fused pooled backward (conv_bwd_fused.hip) routes the pooled gradient with it instead of
// re-reading the window
// FC: the pooling factor at compile time (0: the run-time argument) -- the window loops unroll and a window's loads are issued together
// instead of one dependent round trip per pixel
template <typename T, bool AMAX = false, int FC = 0>
__global__ void bn_relu_pool_kernel(const T* __restrict__ yraw, const float* __restrict__ scale, const float* __restrict__ shift,
                                    T* __restrict__ act, T* __restrict__ pooled, satcv_stat_t* stats, int stats_ld,
                                    int n, int h, int w, int c, int f_, int act_ld, unsigned char* __restrict__ amax = nullptr) {
  extern __shared__ float lds[];
  const int f = FC ? FC : f_;
  const int G = c / 8;
  const int hp = h / f, wp = w / f;               // 'valid' pooling
  const int hw_ = cdiv(h, f), ww = cdiv(w, f);    // windows incl. partial ones (act must cover every pixel)
  const long long total = (long long)n * hw_ * ww * G;
  const long long nthreads = (long long)gridDim.x * blockDim.x;
  const long long stride = nthreads / G * G;
  const long long gid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const int g = (int)(gid % G);
  float sc[8], sh[8], s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sc[e] = scale[g * 8 + e]; sh[e] = shift[g * 8 + e]; s1[e] = 0.f; s2[e] = 0.f; }
  const bool active = gid < stride;
  if (active) {
    const unsigned nwin = (unsigned)n * hw_ * ww, wstep = (unsigned)(stride / G);
    for (unsigned wdw0 = (unsigned)(gid / G); wdw0 < nwin; wdw0 += wstep) {
      unsigned wdw = wdw0;
      const int px = (int)(wdw % (unsigned)ww); wdw /= (unsigned)ww;
      const int py = (int)(wdw % (unsigned)hw_);
      const int img = (int)(wdw / (unsigned)hw_);
      float mx[8];
      unsigned char am[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { mx[e] = -INFINITY; am[e] = 0; }
      if constexpr (FC != 0) {
        // whole windows only (the launcher checks h % FC == 0 && w % FC == 0): all FC * FC loads first
        float v[FC * FC][8];
        const size_t off0 = ((size_t)(img * h + py * FC) * w + px * FC) * c + g * 8;
#pragma unroll
        for (int k = 0; k < FC * FC; ++k) load8<T>(yraw + off0 + ((size_t)(k / FC) * w + (k % FC)) * c, v[k]);
#pragma unroll
        for (int k = 0; k < FC * FC; ++k) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float a = fmaxf(v[k][e] * sc[e] + sh[e], 0.f);
            a = round_to<T>(a);
            v[k][e] = a; s1[e] += a; s2[e] += a * a;
            if (AMAX) { if (a > mx[e]) { mx[e] = a; am[e] = (unsigned char)k; } }
            else mx[e] = fmaxf(mx[e], a);
          }
          if (act) store8<T>(act + ((size_t)(img * h + py * FC + k / FC) * w + px * FC + k % FC) * act_ld + g * 8, v[k]);
        }
      } else
      for (int i = 0; i < f; ++i) {
        const int y = py * f + i; if (y >= h) break;
        for (int j = 0; j < f; ++j) {
          const int x = px * f + j; if (x >= w) break;
          const size_t off = ((size_t)(img * h + y) * w + x) * c + g * 8;
          float v[8];
          load8<T>(yraw + off, v);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float a = fmaxf(v[e] * sc[e] + sh[e], 0.f);
            a = round_to<T>(a);
            v[e] = a; s1[e] += a; s2[e] += a * a;
            if (AMAX) { if (a > mx[e]) { mx[e] = a; am[e] = (unsigned char)(i * f + j); } }
            else mx[e] = fmaxf(mx[e], a);
          }
          if (act) store8<T>(act + ((size_t)(img * h + y) * w + x) * act_ld + g * 8, v);
        }
      }
      if (pooled && py < hp && px < wp) {
        store8<T>(pooled + ((size_t)(img * hp + py) * wp + px) * c + g * 8, mx);
        if (AMAX) *reinterpret_cast<uint2*>(amax + ((size_t)(img * hp + py) * wp + px) * c + g * 8) = *reinterpret_cast<const uint2*>(am);
      }
    }
  }
  if (stats) block_channel_reduce(lds, s1, s2, g, active, c, stats, stats_ld);
}
extern "C" int satcv_bn_relu_pool(const void* yraw, const float* scale, const float* shift, void* act, int32_t act_ld, void* pooled, satcv_stat_t* stats,
                                  int32_t stats_ld, int32_t n, int32_t h, int32_t w_, int32_t c, int32_t f, int32_t dtype, void* stream) {
  if (act_ld <= 0) act_ld = c;
  SATCV_CHECK(yraw && scale && shift && n > 0 && h > 0 && w_ > 0 && c > 0 && c % 8 == 0 && c <= 2048 && f >= 1, "bn_relu_pool: bad args");
  const long long items = (long long)n * cdiv(h, f) * cdiv(w_, f) * (c / 8);
  if (f == 2 && h % 2 == 0 && w_ % 2 == 0) {
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_relu_pool_kernel<T, false, 2>), dim3(ew_grid(items)), dim3(EW_BLOCK), stats ? (EW_BLOCK + 8) * 16 * sizeof(float) : 0,
                                         (hipStream_t)stream, (const T*)yraw, scale, shift, (T*)act, (T*)pooled, stats, stats_ld, n, h, w_, c, f, act_ld));
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL(bn_relu_pool_kernel<T>, dim3(ew_grid(items)), dim3(EW_BLOCK), stats ? (EW_BLOCK + 8) * 16 * sizeof(float) : 0, (hipStream_t)stream,
                                         (const T*)yraw, scale, shift, (T*)act, (T*)pooled, stats, stats_ld, n, h, w_, c, f, act_ld));
  }
  LAUNCH_OK("bn_relu_pool");
  return SATCV_OK;
}
extern "C" int satcv_bn_relu_pool_amax(const void* yraw, const float* scale, const float* shift, void* act, int32_t act_ld, void* pooled, void* amax,
                                       satcv_stat_t* stats, int32_t stats_ld, int32_t n, int32_t h, int32_t w_, int32_t c, int32_t f, int32_t dtype, void* stream) {
  if (act_ld <= 0) act_ld = c;
  SATCV_CHECK(yraw && scale && shift && pooled && amax && n > 0 && h > 0 && w_ > 0 && c > 0 && c % 8 == 0 && c <= 2048 && f >= 1 && f * f <= 255,
              "bn_relu_pool_amax: bad args");
  const long long items = (long long)n * cdiv(h, f) * cdiv(w_, f) * (c / 8);
  if (f == 2 && h % 2 == 0 && w_ % 2 == 0) {
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_relu_pool_kernel<T, true, 2>), dim3(ew_grid(items)), dim3(EW_BLOCK), stats ? (EW_BLOCK + 8) * 16 * sizeof(float) : 0,
                                         (hipStream_t)stream, (const T*)yraw, scale, shift, (T*)act, (T*)pooled, stats, stats_ld, n, h, w_, c, f, act_ld,
                                         (unsigned char*)amax));
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_relu_pool_kernel<T, true>), dim3(ew_grid(items)), dim3(EW_BLOCK), stats ? (EW_BLOCK + 8) * 16 * sizeof(float) : 0,
                                         (hipStream_t)stream, (const T*)yraw, scale, shift, (T*)act, (T*)pooled, stats, stats_ld, n, h, w_, c, f, act_ld,
                                         (unsigned char*)amax));
  }
  LAUNCH_OK("bn_relu_pool_amax");
  return SATCV_OK;
}

// -------------------------------------------------------------------- BN backward
// APPLY=false: accumulate sum g, sum g*xhat.  APPLY=true: write dy, optional dbias.
template <typename T, bool APPLY>
__global__ void bn_bwd_kernel(const satcv_bnbwd_desc d) {
  extern __shared__ float lds[];
  const int c = d.c, f = d.dpool ? d.f : 1;
  const int G = c / 8;
  const int h = d.h, w = d.w_;
  const int hp = h / f, wp = w / f, hw_ = cdiv(h, f), ww = cdiv(w, f);
  const long long total = (long long)d.n * hw_ * ww * G;
  const long long nthreads = (long long)gridDim.x * blockDim.x;
  const long long stride = nthreads / G * G;
  const long long gid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const int g = (int)(gid % G);
  const T* da = (const T*)d.da; const T* dp = (const T*)d.dpool; const T* yr = (const T*)d.yraw; T* dy = (T*)d.dy;
  float sc[8], sh[8], mu[8], rs[8], c1[8], c2[8], s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ch = g * 8 + e;
    sc[e] = d.scale[ch]; sh[e] = d.shift[ch]; mu[e] = d.mean[ch]; rs[e] = d.rstd[ch];
    c1[e] = APPLY ? d.coef[ch] : 0.f; c2[e] = APPLY ? d.coef[c + ch] : 0.f;
    s1[e] = 0.f; s2[e] = 0.f;
  }
  const bool active = gid < stride;
  if (active) {
    const unsigned nwin = (unsigned)d.n * hw_ * ww, wstep = (unsigned)(stride / G);
    for (unsigned wdw0 = (unsigned)(gid / G); wdw0 < nwin; wdw0 += wstep) {
      unsigned wdw = wdw0;
      const int px = (int)(wdw % (unsigned)ww); wdw /= (unsigned)ww;
      const int py = (int)(wdw % (unsigned)hw_);
      const int img = (int)(wdw / (unsigned)hw_);
      // first arg-max of the activated window (only when routing a pooled gradient)
      int am[8]; float gp[8];
      const bool full = dp && py < hp && px < wp;
      if (!APPLY && f == 2 && full && da) {
        // (reduce pass only: the apply pass measured SLOWER in this form, packed or not: 208 -> 260 us at level 0)
        // the common case (MaxPooling2D((2, 2)), utils/model_tools.py:281) in ONE pass: the four raw values, the four dense gradients
        // and the pooled gradient are loaded once, back to back (9 loads in flight per thread instead of 4 + a dependent 9); they stay
        // PACKED in registers (the unpacked form cost the apply pass its occupancy: 198 -> 285 us)
        Raw8<T> v4[4], g4[4];
        const size_t p00 = (size_t)(img * h + py * 2) * w + px * 2;
        const size_t pixs[4] = {p00, p00 + 1, p00 + w, p00 + w + 1};
#pragma unroll
        for (int q = 0; q < 4; ++q) v4[q] = gload8<T>(yr + pixs[q] * d.ldy + g * 8);
#pragma unroll
        for (int q = 0; q < 4; ++q) g4[q] = gload8<T>(da + pixs[q] * d.ldda + g * 8);
        load8<T>(dp + ((size_t)(img * hp + py) * wp + px) * d.lddp + g * 8, gp);
        {
          float mx[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) { mx[e] = -INFINITY; am[e] = 0; }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float v[8];
            unpack8<T>(v4[q], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float a = round_to<T>(fmaxf(v[e] * sc[e] + sh[e], 0.f));
              if (a > mx[e]) { mx[e] = a; am[e] = q; }
            }
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[8], gr[8], o[8];
          unpack8<T>(v4[q], v);
          unpack8<T>(g4[q], gr);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float gg = gr[e];
            if (am[e] == q) gg += gp[e];
            const float a = v[e] * sc[e] + sh[e];
            gg = (a > 0.f || d.linear) ? gg : 0.f;
            const float xh = (v[e] - mu[e]) * rs[e];
            if (APPLY) { const float t = sc[e] * (gg - c1[e] - xh * c2[e]); o[e] = round_to<T>(t); s1[e] += o[e]; }
            else { s1[e] += gg; s2[e] += gg * xh; }
          }
          if (APPLY) store8<T>(dy + pixs[q] * d.lddy_out + g * 8, o);
        }
        continue;
      }
      if (full) {
        float mx[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { mx[e] = -INFINITY; am[e] = 0; }
        for (int i = 0; i < f; ++i) for (int j = 0; j < f; ++j) {
          float v[8];
          load8<T>(yr + ((size_t)(img * h + py * f + i) * w + px * f + j) * d.ldy + g * 8, v);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float a = round_to<T>(fmaxf(v[e] * sc[e] + sh[e], 0.f));
            if (a > mx[e]) { mx[e] = a; am[e] = i * f + j; }
          }
        }
        load8<T>(dp + ((size_t)(img * hp + py) * wp + px) * d.lddp + g * 8, gp);
      }
      for (int i = 0; i < f; ++i) {
        const int y = py * f + i; if (y >= h) break;
        for (int j = 0; j < f; ++j) {
          const int x = px * f + j; if (x >= w) break;
          const size_t pix = (size_t)(img * h + y) * w + x;
          float v[8], gr[8];
          load8<T>(yr + pix * d.ldy + g * 8, v);
          if (da) load8<T>(da + pix * d.ldda + g * 8, gr);
          else {
#pragma unroll
            for (int e = 0; e < 8; ++e) gr[e] = 0.f;
          }
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float gg = gr[e];
            if (full && am[e] == i * f + j) gg += gp[e];
            const float a = v[e] * sc[e] + sh[e];
            gg = (a > 0.f || d.linear) ? gg : 0.f;
            const float xh = (v[e] - mu[e]) * rs[e];
            if (APPLY) { const float t = sc[e] * (gg - c1[e] - xh * c2[e]); o[e] = round_to<T>(t); s1[e] += o[e]; }
            else { s1[e] += gg; s2[e] += gg * xh; }
          }
          if (APPLY) store8<T>(dy + pix * d.lddy_out + g * 8, o);
        }
      }
    }
  }
  if (!APPLY) block_channel_reduce(lds, s1, s2, g, active, c, d.sums, d.sums_ld);
  else if (d.dbias) {
    for (int i = threadIdx.x; i < c; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    if (active) {
#pragma unroll
      for (int e = 0; e < 8; ++e) atomicAdd(&lds[g * 8 + e], s1[e]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < c; i += blockDim.x) atomicAdd(d.dbias + i, lds[i]);
  }
}
// Dense variant (no pooled gradient): pure linear sweep, 32-bit incremental indexing.
template <typename T, bool APPLY>
__global__ __launch_bounds__(EW_BLOCK) void bn_bwd_dense_kernel(const satcv_bnbwd_desc d, const int rev, const int cld) {
  // cld: channel count the coefficient vector [c1 | c2] was built for (= d.c, or the whole concatenation when only its first part is applied)
  extern __shared__ float lds[];
  const int c = d.c, G = c / 8;
  const unsigned npix = (unsigned)d.n * d.h * d.w_;
  const unsigned nthreads = gridDim.x * blockDim.x;
  const unsigned per = nthreads / G;
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = gid < per * G;
  const int g = gid % G;
  // (second source of a concatenation: a thread's channel group, and with it its source / destination, never changes)
  const bool second = d.c_split > 0 && g * 8 >= d.c_split;
  const T* da = (const T*)d.da + g * 8;
  const T* yr = second ? (const T*)d.yraw1 + (g * 8 - d.c_split) : (const T*)d.yraw + g * 8;
  T* dy = second ? (T*)d.dy1 + (g * 8 - d.c_split) : (T*)d.dy + g * 8;
  const int ldy = second ? d.ldy1 : d.ldy, lddy = second ? d.lddy1 : d.lddy_out;
  const bool sk = APPLY && d.sk_sums != nullptr && !second;
  float sc[8], sh[8], mu[8], rs[8], c1[8], c2[8], s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ch = g * 8 + e;
    sc[e] = d.scale[ch]; sh[e] = d.shift[ch]; mu[e] = d.mean[ch]; rs[e] = d.rstd[ch];
    c1[e] = APPLY ? d.coef[ch] : 0.f; c2[e] = APPLY ? d.coef[cld + ch] : 0.f;
    s1[e] = 0.f; s2[e] = 0.f;
  }
  if (active) {
    // running pointers (one 64-bit add per tensor and pixel): the `(size_t)p * ld` form cost six quarter-rate 64-bit multiplies per
    // pixel because of the reversed order's select
    const unsigned pfirst = gid / G;
    const long long step = rev ? -(long long)per : (long long)per;
    const long long pstart = rev ? (long long)npix - 1 - pfirst : (long long)pfirst;
    const T* yq = yr + pstart * ldy;
    const T* dq = da + pstart * d.ldda;
    T* oq = dy + pstart * lddy;
    const long long ystep = step * ldy, dstep = step * d.ldda, ostep = step * lddy;
    const float lin_lo = d.linear ? -INFINITY : 0.f;      // (a > 0 || linear)  ==  a > lin_lo
#pragma unroll 4
    for (unsigned p0 = pfirst; p0 < npix; p0 += per) {
      float v[8], gr[8], o[8];
      load8<T>(yq, v);
      load8<T>(dq, gr);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float a = v[e] * sc[e] + sh[e];
        const float gg = (a > lin_lo) ? gr[e] : 0.f;
        const float xh = (v[e] - mu[e]) * rs[e];
        if (APPLY) {
          const float t = sc[e] * (gg - c1[e] - xh * c2[e]); o[e] = round_to<T>(t);
          // sk_sums: v is the ACTIVATED skip a = relu(BN(y_enc)) and o its gradient -- the first-source threads also form the sums of
          // THAT BatchNorm's backward in the activated form: sum o [a > 0], sum o a (satcv_bn_bwd_finalize2 converts)
          if (sk) { s1[e] += v[e] > 0.f ? o[e] : 0.f; s2[e] += o[e] * v[e]; }
          else s1[e] += o[e];
        }
        else { s1[e] += gg; s2[e] += gg * xh; }
      }
      if (APPLY) store8<T>(oq, o);
      yq += ystep; dq += dstep; oq += ostep;
    }
  }
  if (!APPLY) block_channel_reduce(lds, s1, s2, g, active, c, d.sums, d.sums_ld);
  else if (d.sk_sums) block_channel_reduce(lds, s1, s2, g, active && !second, c, d.sk_sums, d.sk_sums_ld, d.c_split > 0 ? d.c_split : -1);
  else if (d.dbias) {
    for (int i = threadIdx.x; i < c; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    if (active) {
#pragma unroll
      for (int e = 0; e < 8; ++e) atomicAdd(&lds[g * 8 + e], s1[e]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < c; i += blockDim.x) atomicAdd(d.dbias + i, lds[i]);
  }
}

// Round 6: the APPLY pass of the dense form rebuilt.  The loop above, instantiated with APPLY = true, compiled to two loads per iteration
// with `s_waitcnt vmcnt(0)` behind them (per-lane loop exits keep hipcc from hoisting the next pixels' loads), ~230 vector instructions per 8
// channels (the `sk` test is lane-dependent: an exec-mask branch per channel; every output was rounded on its own for a bias sum nobody asked
// for) -- at 12 KB per 920 SIMD cycles the pass was VALU-bound just under the HBM rate (3.9-5.2 TB/s against 5.9-6.3 of the reduce pass).
// Here: U pixels per iteration with all 2 U loads issued first (a wave-uniform trip count; the ragged tail runs clamped and predicated), the
// sums compiled in only where the launch wants them (SK: the skip half of a decoder concatenation, DB: SATCV_BN_BIAS_NOISE), and per channel
//   gm = (v sc + sh > lin_lo) ? g : 0;  t = sc gm - k1 - k2 (v - mu),  k1 = sc c1, k2 = sc rs c2
// (the centred form: v - mu first, as before -- no cancellation when |mean| >> std).
template <typename T, bool SK, bool DB, int U>
__global__ __launch_bounds__(EW_BLOCK) void bn_bwd_apply_dense_kernel(const satcv_bnbwd_desc d, const int rev, const int cld) {
  extern __shared__ float lds[];
  const int c = d.c, G = c / 8;
  const unsigned npix = (unsigned)d.n * d.h * d.w_;
  const unsigned nthreads = gridDim.x * blockDim.x;
  const unsigned per = nthreads / G;
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = gid < per * G;
  const int g = gid % G;
  const bool second = d.c_split > 0 && g * 8 >= d.c_split;
  const T* da = (const T*)d.da + g * 8;
  const T* yr = second ? (const T*)d.yraw1 + (g * 8 - d.c_split) : (const T*)d.yraw + g * 8;
  T* dy = second ? (T*)d.dy1 + (g * 8 - d.c_split) : (T*)d.dy + g * 8;
  const int ldy = second ? d.ldy1 : d.ldy, lddy = second ? d.lddy1 : d.lddy_out;
  float sc[8], sh[8], mu[8], k1[8], k2[8], s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ch = g * 8 + e;
    sc[e] = d.scale[ch]; sh[e] = d.shift[ch]; mu[e] = d.mean[ch];
    k1[e] = sc[e] * d.coef[ch]; k2[e] = sc[e] * d.rstd[ch] * d.coef[cld + ch];
    s1[e] = 0.f; s2[e] = 0.f;
  }
  const float lin_lo = d.linear ? -INFINITY : 0.f;      // (a > 0 || linear)  ==  a > lin_lo
  const unsigned pfirst = gid / G;
  // every thread of the grid runs the same number of iterations; pixels past the end (and the threads beyond per * G) are clamped to the
  // last pixel for their loads and store nothing
  const unsigned nit = (npix + per - 1) / per;
  // FULL iterations: every pixel of every thread exists (a grid-uniform test) -- no per-pixel predicates around the sums; the ragged last
  // iteration(s) run the same body with them
  auto body = [&](unsigned i0, auto FULLC) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(FULLC)::value;
    Raw8<T> rv[U], rg[U];
    unsigned pp[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned long long q = (unsigned long long)pfirst + (unsigned long long)(i0 + u) * per;
      ok[u] = FULL ? active : (active && q < npix);
      const unsigned pq = (FULL || q < npix) ? (unsigned)q : npix - 1;
      pp[u] = rev ? npix - 1 - pq : pq;
      rv[u] = gload8<T>(yr + (size_t)pp[u] * ldy);
      rg[u] = gload8<T>(da + (size_t)pp[u] * d.ldda);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float v[8], gr[8], o[8];
      unpack8<T>(rv[u], v);
      unpack8<T>(rg[u], gr);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float gm = (fmaf(v[e], sc[e], sh[e]) > lin_lo) ? gr[e] : 0.f;
        const float t = fmaf(-k2[e], v[e] - mu[e], fmaf(sc[e], gm, -k1[e]));
        if constexpr (SK || DB) {
          o[e] = round_to<T>(t);
          const float om = (FULL || ok[u]) ? o[e] : 0.f;      // (threads beyond per * G run clamped: their partial sums are dropped at the end)
          // SK: v is the ACTIVATED skip a = relu(BN(y_enc)) and o its gradient -- the first-source threads also form the sums of THAT
          // BatchNorm's backward in the activated form: sum o [a > 0], sum o a (satcv_bn_bwd_finalize2 converts)
          if constexpr (SK) { s1[e] += v[e] > 0.f ? om : 0.f; s2[e] = fmaf(om, v[e], s2[e]); }
          else s1[e] += om;
        } else {
          o[e] = t;
        }
      }
      if (ok[u]) store8<T>(dy + (size_t)pp[u] * lddy, o);
    }
  };
  // the largest pixel index any thread touches in iterations [i0, i0 + U) is per + (i0 + U - 1) per (the surplus threads beyond per * G start at
  // pixel `per`; they load like the others and store nothing)
  unsigned i0 = 0;
  for (; (unsigned long long)(i0 + U) * per < npix; i0 += U) body(i0, std::true_type{});
  for (; i0 < nit; i0 += U) body(i0, std::false_type{});
  if constexpr (SK) block_channel_reduce(lds, s1, s2, g, active && !second, c, d.sk_sums, d.sk_sums_ld, d.c_split > 0 ? d.c_split : -1);
  else if constexpr (DB) {
    for (int i = threadIdx.x; i < c; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    if (active) {
#pragma unroll
      for (int e = 0; e < 8; ++e) atomicAdd(&lds[g * 8 + e], s1[e]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < c; i += blockDim.x) atomicAdd(d.dbias + i, lds[i]);
  }
}
// (SATCV_BN_APPLY=0: the round-3 loop for every apply launch -- A/B switch)
template <typename T>
static void bn_bwd_apply_dense_launch(const satcv_bnbwd_desc& e, long long items, size_t lds_bytes, int rev, int cld, hipStream_t st) {
  static const int mode = getenv("SATCV_BN_APPLY") ? atoi(getenv("SATCV_BN_APPLY")) : 1;
  const dim3 grid(ew_grid(items)), block(EW_BLOCK);
  if (mode == 0) { hipLaunchKernelGGL((bn_bwd_dense_kernel<T, true>), grid, block, lds_bytes, st, e, rev, cld); return; }
  if (e.sk_sums) hipLaunchKernelGGL((bn_bwd_apply_dense_kernel<T, true, false, 4>), grid, block, lds_bytes, st, e, rev, cld);
  else if (e.dbias) hipLaunchKernelGGL((bn_bwd_apply_dense_kernel<T, false, true, 4>), grid, block, lds_bytes, st, e, rev, cld);
  else hipLaunchKernelGGL((bn_bwd_apply_dense_kernel<T, false, false, 4>), grid, block, lds_bytes, st, e, rev, cld);
}
static int bnbwd_check(const satcv_bnbwd_desc* d, bool apply) {
  SATCV_CHECK(d && d->yraw && d->scale && d->shift && d->mean && d->rstd, "bn_bwd: null pointer");
  SATCV_CHECK(d->da || d->dpool, "bn_bwd: no incoming gradient");
  SATCV_CHECK(d->c > 0 && d->c % 8 == 0 && d->c <= 2048 && d->n > 0 && d->h > 0 && d->w_ > 0, "bn_bwd: bad dims");
  SATCV_CHECK(!d->dpool || d->f >= 1, "bn_bwd: pool factor");
  SATCV_CHECK(!apply || !d->sk_sums || (d->c_split > 0 && !d->dbias && d->sk_sums_ld >= d->c_split), "bn_bwd_apply: sk_sums needs the two-source form");
  if (apply) SATCV_CHECK(d->coef && d->dy, "bn_bwd_apply: coef/dy missing");
  else SATCV_CHECK(d->sums && d->sums_ld >= d->c, "bn_bwd_reduce: sums missing");
  if (d->c_split > 0) {
    SATCV_CHECK(d->c_split % 8 == 0 && d->c_split < d->c && d->yraw1 && d->ldy1 >= d->c - d->c_split && !d->dpool && d->da && !d->dbias &&
                (long long)d->n * d->h * d->w_ < 0x7fffffffLL, "bn_bwd: second source needs the dense form (da, no dpool / dbias), c_split %% 8 == 0");
    if (apply) SATCV_CHECK(!d->dy1 || d->lddy1 >= d->c - d->c_split, "bn_bwd_apply: dy1 channel stride");      // (dy1 NULL: only the first source's channels are applied)
  }
  return SATCV_OK;
}
extern "C" int satcv_bn_bwd_reduce(const satcv_bnbwd_desc* d, void* stream) {
  int rc = bnbwd_check(d, false); if (rc) return rc;
  const int f = d->dpool ? d->f : 1;
  const long long items = (long long)d->n * cdiv(d->h, f) * cdiv(d->w_, f) * (d->c / 8);
  // every workgroup ends with 2 c double atomics into the replica rows: on the deep, small maps (64 x 8 x 8 x 1024: one item per thread on the
  // full grid) those -- 3 million of them -- were the launch (24 us for 17 MB, 0.09 of its roofline).  At least 8 items per thread.
  const int grid = ew_grid((items + 7) / 8);
  if (!d->dpool && d->da && (long long)d->n * d->h * d->w_ < 0x7fffffffLL) {
    DISPATCH_T(d->dtype, hipLaunchKernelGGL((bn_bwd_dense_kernel<T, false>), dim3(grid), dim3(EW_BLOCK), (EW_BLOCK + 8) * 16 * sizeof(float), (hipStream_t)stream, *d, 0, d->c));
  } else {
    DISPATCH_T(d->dtype, hipLaunchKernelGGL((bn_bwd_kernel<T, false>), dim3(grid), dim3(EW_BLOCK), (EW_BLOCK + 8) * 16 * sizeof(float), (hipStream_t)stream, *d));
  }
  LAUNCH_OK("bn_bwd_reduce");
  return SATCV_OK;
}
extern "C" int satcv_bn_bwd_apply(const satcv_bnbwd_desc* d, void* stream) {
  static const int rev = getenv("SATCV_BN_REV") ? atoi(getenv("SATCV_BN_REV")) : 1;
  int rc = bnbwd_check(d, true); if (rc) return rc;
  const int f = d->dpool ? d->f : 1;
  if (d->c_split > 0 && !d->dy1) {
    // two-source descriptor without a destination for the second source: the gradient of the `up` channels of a decoder concatenation is
    // formed by satcv_convt_bwd_fused in its loader -- apply the first c_split channels only (coefficients still indexed for all d->c)
    satcv_bnbwd_desc e = *d;
    e.c = d->c_split; e.c_split = 0; e.yraw1 = nullptr;
    const long long it1 = (long long)d->n * d->h * d->w_ * (e.c / 8);
    const size_t lds_1 = d->sk_sums ? (EW_BLOCK + 8) * 16 * sizeof(float) : e.c * sizeof(float);
    DISPATCH_T(d->dtype, bn_bwd_apply_dense_launch<T>(e, it1, lds_1, rev, d->c, (hipStream_t)stream));
    LAUNCH_OK("bn_bwd_apply");
    return SATCV_OK;
  }
  const long long items = (long long)d->n * cdiv(d->h, f) * cdiv(d->w_, f) * (d->c / 8);
  if (!d->dpool && d->da && (long long)d->n * d->h * d->w_ < 0x7fffffffLL) {
    const size_t lds_b = d->sk_sums ? (EW_BLOCK + 8) * 16 * sizeof(float) : d->c * sizeof(float);
    DISPATCH_T(d->dtype, bn_bwd_apply_dense_launch<T>(*d, items, lds_b, rev, d->c, (hipStream_t)stream));
  } else {
    DISPATCH_T(d->dtype, hipLaunchKernelGGL((bn_bwd_kernel<T, true>), dim3(ew_grid(items)), dim3(EW_BLOCK), d->c * sizeof(float), (hipStream_t)stream, *d));
  }
  LAUNCH_OK("bn_bwd_apply");
  return SATCV_OK;
}
__global__ void bn_bwd_finalize_kernel(satcv_stat_t* sums, int ld, int c, float count, float* dgamma, float* dbeta, float* coef, int accumulate) {
  const int ch = blockIdx.x * (blockDim.x / 32) + threadIdx.x / 32, r = threadIdx.x & 31;      // see bn_finalize_train_kernel
  double d1 = 0.0, d2 = 0.0;
  if (ch < c) {
    satcv_stat_t* row = sums + (size_t)r * 2 * ld;
    d1 = row[ch]; d2 = row[ld + ch];
    row[ch] = 0.0; row[ld + ch] = 0.0;
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) { d1 += __shfl_xor(d1, o, 64); d2 += __shfl_xor(d2, o, 64); }
  if (ch >= c || r != 0) return;
  const float s1 = (float)d1, s2 = (float)d2;
  if (dbeta) dbeta[ch] = accumulate ? dbeta[ch] + s1 : s1;
  if (dgamma) dgamma[ch] = accumulate ? dgamma[ch] + s2 : s2;
  coef[ch] = s1 / count; coef[c + ch] = s2 / count;
}
// sums from producers that only had the ACTIVATED output a = relu(scale * y + shift) of the layer at hand (rows [0]: sum g [a > 0], rows
// [1]: sum g a): sum g xhat = (sum g a - (shift + scale * mean) sum g) rstd / scale, since xhat = (a - shift - scale mean) rstd / scale
// wherever the mask is 1.  Added to the raw-form rows of a satcv_bn_bwd_reduce over whatever part of the gradient was not fused.
__global__ void bn_bwd_finalize2_kernel(satcv_stat_t* sums, int ld, satcv_stat_t* asums, int ald, int c, float count, const float* scale, const float* shift,
                                        const float* mean, const float* rstd, float* dgamma, float* dbeta, float* coef, int accumulate) {
  const int ch = blockIdx.x * (blockDim.x / 32) + threadIdx.x / 32, r = threadIdx.x & 31;
  double d1 = 0.0, d2 = 0.0, a1 = 0.0, a2 = 0.0;
  if (ch < c) {
    if (sums) { satcv_stat_t* row = sums + (size_t)r * 2 * ld; d1 = row[ch]; d2 = row[ld + ch]; row[ch] = 0.0; row[ld + ch] = 0.0; }
    satcv_stat_t* arow = asums + (size_t)r * 2 * ald;
    a1 = arow[ch]; a2 = arow[ald + ch]; arow[ch] = 0.0; arow[ald + ch] = 0.0;
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) {
    d1 += __shfl_xor(d1, o, 64); d2 += __shfl_xor(d2, o, 64); a1 += __shfl_xor(a1, o, 64); a2 += __shfl_xor(a2, o, 64);
  }
  if (ch >= c || r != 0) return;
  const double sc = (double)scale[ch], sh = (double)shift[ch], mu = (double)mean[ch], rs = (double)rstd[ch];
  const double a2x = sc != 0.0 ? (a2 - (sh + sc * mu) * a1) * (rs / sc) : 0.0;
  const float s1 = (float)(d1 + a1), s2 = (float)(d2 + a2x);
  if (dbeta) dbeta[ch] = accumulate ? dbeta[ch] + s1 : s1;
  if (dgamma) dgamma[ch] = accumulate ? dgamma[ch] + s2 : s2;
  coef[ch] = s1 / count; coef[c + ch] = s2 / count;
}
extern "C" int satcv_bn_bwd_finalize2(satcv_stat_t* sums, int32_t sums_ld, satcv_stat_t* act_sums, int32_t act_sums_ld, int32_t c, float count,
                                      const float* scale, const float* shift, const float* mean, const float* rstd, float* dgamma, float* dbeta,
                                      float* coef, int32_t accumulate, void* stream) {
  SATCV_CHECK(act_sums && coef && scale && shift && mean && rstd && c > 0 && act_sums_ld >= c && (!sums || sums_ld >= c) && count > 0, "bn_bwd_finalize2: bad args");
  hipLaunchKernelGGL(bn_bwd_finalize2_kernel, dim3(cdiv(c, 8)), dim3(256), 0, (hipStream_t)stream, sums, sums_ld, act_sums, act_sums_ld, c, count, scale, shift,
                     mean, rstd, dgamma, dbeta, coef, accumulate);
  LAUNCH_OK("bn_bwd_finalize2");
  return SATCV_OK;
}
extern "C" int satcv_bn_bwd_finalize(satcv_stat_t* sums, int32_t sums_ld, int32_t c, float count, float* dgamma, float* dbeta, float* coef, int32_t accumulate,
                                     void* stream) {
  SATCV_CHECK(sums && coef && c > 0 && sums_ld >= c && count > 0, "bn_bwd_finalize: bad args");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(c, 8)), dim3(256), 0, (hipStream_t)stream, sums, sums_ld, c, count, dgamma, dbeta, coef, accumulate);
  LAUNCH_OK("bn_bwd_finalize");
  return SATCV_OK;
}

// --------------------------------------------------------------------------- head
#define HEAD_NCMAX 8
template <typename T>
__global__ void head_fwd_kernel(const satcv_head_desc d) {
  extern __shared__ float lds[];                     // w[cin][ncls], b[ncls], scale[cin], shift[cin]
  const int cin = d.cin, nc = d.ncls;
  float* lw = lds; float* lb = lw + cin * nc; float* lsc = lb + nc; float* lsh = lsc + cin;
  for (int i = threadIdx.x; i < cin * nc; i += blockDim.x) lw[i] = d.w[i];
  for (int i = threadIdx.x; i < nc; i += blockDim.x) lb[i] = d.b[i];
  for (int i = threadIdx.x; i < cin; i += blockDim.x) { lsc[i] = d.in_scale ? d.in_scale[i] : 1.f; lsh[i] = d.in_shift ? d.in_shift[i] : 0.f; }
  __syncthreads();
  const T* x = (const T*)d.x;
  const bool tr = d.in_scale != nullptr;
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < d.npix; p += (long long)gridDim.x * blockDim.x) {
    float z[HEAD_NCMAX];
#pragma unroll
    for (int k = 0; k < HEAD_NCMAX; ++k) z[k] = k < nc ? lb[k] : -INFINITY;
    for (int g = 0; g < cin / 8; ++g) {
      float v[8];
      load8<T>(x + p * d.ldx + g * 8, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ch = g * 8 + e;
        const float a = tr ? fmaxf(v[e] * lsc[ch] + lsh[ch], 0.f) : v[e];
#pragma unroll
        for (int k = 0; k < HEAD_NCMAX; ++k) if (k < nc) z[k] += a * lw[ch * nc + k];
      }
    }
    if (d.activation == 2) {
#pragma unroll
      for (int k = 0; k < HEAD_NCMAX; ++k) if (k < nc) d.probs[p * nc + k] = z[k];
    } else if (d.activation == 0) {
      float mx = z[0]; int am = 0;
#pragma unroll
      for (int k = 1; k < HEAD_NCMAX; ++k) if (k < nc && z[k] > mx) { mx = z[k]; }
      float s = 0.f, ex[HEAD_NCMAX];
#pragma unroll
      for (int k = 0; k < HEAD_NCMAX; ++k) { ex[k] = k < nc ? expf(z[k] - mx) : 0.f; s += ex[k]; }
      const float inv = 1.f / s;
      float best = -1.f;
#pragma unroll
      for (int k = 0; k < HEAD_NCMAX; ++k) if (k < nc) {
        const float pr = ex[k] * inv;
        d.probs[p * nc + k] = pr;
        if (pr > best) { best = pr; am = k; }          // ties -> lowest index (tf.argmax)
      }
      if (d.classes) d.classes[p] = am;
    } else {
#pragma unroll
      for (int k = 0; k < HEAD_NCMAX; ++k) if (k < nc) {
        const float pr = 1.f / (1.f + expf(-z[k]));
        d.probs[p * nc + k] = pr;
        if (d.classes) d.classes[p * nc + k] = pr > d.thresh ? 1 : 0;
      }
    }
  }
}

// ---- register-resident variants for the common small heads (CIN * NC <= 128): every weight and every
// dW accumulator lives in registers, one pass over the pixels.
template <typename T, int NC, int CIN>
__global__ __launch_bounds__(EW_BLOCK) void head_fwd_fast_kernel(const satcv_head_desc d) {
  float w[CIN][NC], b[NC], sc[CIN], sh[CIN];
  const bool tr = d.in_scale != nullptr;
#pragma unroll
  for (int c = 0; c < CIN; ++c) {
    sc[c] = tr ? d.in_scale[c] : 1.f; sh[c] = tr ? d.in_shift[c] : 0.f;
#pragma unroll
    for (int k = 0; k < NC; ++k) w[c][k] = d.w[c * NC + k];
  }
#pragma unroll
  for (int k = 0; k < NC; ++k) b[k] = d.b[k];
  const T* x = (const T*)d.x;
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < d.npix; p += (long long)gridDim.x * blockDim.x) {
    float z[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) z[k] = b[k];
#pragma unroll
    for (int g = 0; g < CIN / 8; ++g) {
      float v[8];
      load8<T>(x + p * d.ldx + g * 8, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float a = tr ? fmaxf(v[e] * sc[g * 8 + e] + sh[g * 8 + e], 0.f) : v[e];
#pragma unroll
        for (int k = 0; k < NC; ++k) z[k] += a * w[g * 8 + e][k];
      }
    }
    if (d.activation == 2) {            // linear: raw logits (DeepLab head, upsampled later)
#pragma unroll
      for (int k = 0; k < NC; ++k) d.probs[p * NC + k] = z[k];
    } else if (d.activation == 0) {
      float mx = z[0];
#pragma unroll
      for (int k = 1; k < NC; ++k) mx = fmaxf(mx, z[k]);
      float s = 0.f, ex[NC];
#pragma unroll
      for (int k = 0; k < NC; ++k) { ex[k] = expf(z[k] - mx); s += ex[k]; }
      const float inv = 1.f / s;
      float best = -1.f; int am = 0;
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        const float pr = ex[k] * inv;
        d.probs[p * NC + k] = pr;
        if (pr > best) { best = pr; am = k; }
      }
      if (d.classes) d.classes[p] = am;
    } else {
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        const float pr = 1.f / (1.f + expf(-z[k]));
        d.probs[p * NC + k] = pr;
        if (d.classes) d.classes[p * NC + k] = pr > d.thresh ? 1 : 0;
      }
    }
  }
}

// One thread per (pixel, 8-channel group): the per-thread state (8 x NC weights and dW accumulators, 8 BN coefficients and sums) stays
// small enough for 4+ waves/SIMD.  (The first version kept all CIN channels of a pixel in one thread: 255 VGPRs, one wave per SIMD,
// 2.3 TB/s.)
template <typename T, int NC, int CIN, bool BNR>
__global__ __launch_bounds__(EW_BLOCK) void head_bwd_fast_kernel(const satcv_head_desc d) {
  constexpr int G = CIN / 8;                       // threads per pixel; EW_BLOCK * gridDim is a multiple of G, so a thread's group is fixed
  constexpr int NW = EW_BLOCK / 64, AWN = CIN * NC + NC + 2 * CIN;
  __shared__ float aw[NW][AWN];                    // one slot per wave: summed in wave order below (reproducible)
  const int g = threadIdx.x % G;
  float w[8][NC], sc[8], sh[8], acc[8][NC], accb[NC];
  float mu[BNR ? 8 : 1], rs[BNR ? 8 : 1], r1[BNR ? 8 : 1], r2[BNR ? 8 : 1];
  const bool tr = d.in_scale != nullptr;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = g * 8 + e;
    sc[e] = tr ? d.in_scale[c] : 1.f; sh[e] = tr ? d.in_shift[c] : 0.f;
    if constexpr (BNR) { mu[e] = d.bnr_mean[c]; rs[e] = d.bnr_rstd[c]; r1[e] = 0.f; r2[e] = 0.f; }
#pragma unroll
    for (int k = 0; k < NC; ++k) { w[e][k] = d.w[c * NC + k]; acc[e][k] = 0.f; }
  }
#pragma unroll
  for (int k = 0; k < NC; ++k) accb[k] = 0.f;
  const T* x = (const T*)d.x; T* dx = (T*)d.dx;
  const long long total = d.npix * G;
  for (long long it = blockIdx.x * (long long)blockDim.x + threadIdx.x; it < total; it += (long long)gridDim.x * blockDim.x) {
    const long long p = it / G;
    float dl[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) { dl[k] = d.dlogits[p * NC + k]; if (g == 0) accb[k] += dl[k]; }
    float v[8], o[8];
    load8<T>(x + p * d.ldx + g * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float a = tr ? fmaxf(v[e] * sc[e] + sh[e], 0.f) : v[e];
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < NC; ++k) { s += dl[k] * w[e][k]; acc[e][k] += a * dl[k]; }
      o[e] = s;
      if constexpr (BNR) {      // first pass of the BN+ReLU backward of x, on the value as stored
        const float gq = a > 0.f ? round_to<T>(s) : 0.f;
        r1[e] += gq; r2[e] += gq * ((v[e] - mu[e]) * rs[e]);
      }
    }
    if (dx) store8<T>(dx + p * d.lddx + g * 8, o);
  }
  // lanes with equal (lane % G) hold the same channel group: butterfly over the lane bits above log2(G)
  auto group_sum = [&](float val) {
#pragma unroll
    for (int o = 32; o >= G; o >>= 1) val += __shfl_xor(val, o, 64);
    return val;
  };
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const float s = group_sum(acc[e][k]);
      if (lane < G) aw[wave][(lane * 8 + e) * NC + k] = s;
    }
    if constexpr (BNR) {
      const float s1 = group_sum(r1[e]), s2 = group_sum(r2[e]);
      if (lane < G) { aw[wave][CIN * NC + NC + lane * 8 + e] = s1; aw[wave][CIN * NC + NC + CIN + lane * 8 + e] = s2; }
    }
  }
#pragma unroll
  for (int k = 0; k < NC; ++k) { const float s = wave_sum(accb[k]); if (lane == 0) aw[wave][CIN * NC + k] = s; }
  __syncthreads();
  auto tot = [&](int i) { float t = 0.f; for (int w = 0; w < NW; ++w) t += aw[w][i]; return t; };
  if (d.partials) {            // one row per workgroup, added in row order by satcv_head_bwd_finalize
    for (int i = threadIdx.x; i < CIN * NC + NC; i += blockDim.x) d.partials[(size_t)blockIdx.x * (CIN * NC + NC) + i] = tot(i);
  } else {
    if (d.dw) for (int i = threadIdx.x; i < CIN * NC; i += blockDim.x) atomicAdd(d.dw + i, tot(i));
    if (d.db) for (int i = threadIdx.x; i < NC; i += blockDim.x) atomicAdd(d.db + i, tot(CIN * NC + i));
  }
  if constexpr (BNR) {
    satcv_stat_t* rowp = d.bnr_sums + (size_t)(blockIdx.x % SATCV_STAT_ROWS) * 2 * d.bnr_sums_ld;
    for (int i = threadIdx.x; i < CIN; i += blockDim.x) {
      atomicAdd(rowp + i, (satcv_stat_t)tot(CIN * NC + NC + i));
      atomicAdd(rowp + d.bnr_sums_ld + i, (satcv_stat_t)tot(CIN * NC + NC + CIN + i));
    }
  }
}

template <typename T, bool BWD>
static bool head_fast_launch(const satcv_head_desc* d, hipStream_t st) {
  const int grid = BWD ? ew_grid(d->npix * (d->cin / 8)) : ew_grid(d->npix, 1024);      // backward: one thread per 8-channel group
#define HEAD_CASE(NC_, CIN_)                                                                                      \
  if (d->ncls == NC_ && d->cin == CIN_) {                                                                         \
    if constexpr (BWD) {                                                                                          \
      if (d->bnr_sums) hipLaunchKernelGGL((head_bwd_fast_kernel<T, NC_, CIN_, true>), dim3(grid), dim3(EW_BLOCK), 0, st, *d); \
      else hipLaunchKernelGGL((head_bwd_fast_kernel<T, NC_, CIN_, false>), dim3(grid), dim3(EW_BLOCK), 0, st, *d); \
    } else hipLaunchKernelGGL((head_fwd_fast_kernel<T, NC_, CIN_>), dim3(grid), dim3(EW_BLOCK), 0, st, *d);       \
    return true;                                                                                                  \
  }
  HEAD_CASE(1, 16) HEAD_CASE(2, 16) HEAD_CASE(1, 32) HEAD_CASE(2, 32) HEAD_CASE(3, 32) HEAD_CASE(4, 32) HEAD_CASE(1, 64) HEAD_CASE(2, 64)
#undef HEAD_CASE
  return false;
}
extern "C" int satcv_head_fwd(const satcv_head_desc* d, void* stream) {
  SATCV_CHECK(d && d->x && d->w && d->b && d->probs, "head_fwd: null pointer");
  SATCV_CHECK(d->cin > 0 && d->cin % 8 == 0 && d->ncls >= 1 && d->ncls <= HEAD_NCMAX && d->npix > 0, "head_fwd: bad dims (cin=%d ncls=%d)", d->cin, d->ncls);
  const size_t lds = (size_t)(d->cin * d->ncls + d->ncls + 2 * d->cin) * sizeof(float);
  DISPATCH_T8(d->dtype, { if (!head_fast_launch<T, false>(d, (hipStream_t)stream))
      hipLaunchKernelGGL(head_fwd_kernel<T>, dim3(ew_grid(d->npix)), dim3(EW_BLOCK), lds, (hipStream_t)stream, *d); });
  LAUNCH_OK("head_fwd");
  return SATCV_OK;
}

// dx[p][c] = sum_k dl[p][k] w[c][k];  dw[c][k] += sum_p a[p][c] dl[p][k];  db[k] += sum_p dl[p][k]
template <typename T>
__global__ void head_bwd_kernel(const satcv_head_desc d) {
  extern __shared__ float lds[];                     // w[cin][ncls], scale, shift, accw[cin][ncls], accb[ncls]
  const int cin = d.cin, nc = d.ncls;
  float* lw = lds; float* lsc = lw + cin * nc; float* lsh = lsc + cin; float* aw = lsh + cin; float* ab = aw + cin * nc;
  for (int i = threadIdx.x; i < cin * nc; i += blockDim.x) { lw[i] = d.w[i]; aw[i] = 0.f; }
  for (int i = threadIdx.x; i < nc; i += blockDim.x) ab[i] = 0.f;
  for (int i = threadIdx.x; i < cin; i += blockDim.x) { lsc[i] = d.in_scale ? d.in_scale[i] : 1.f; lsh[i] = d.in_shift ? d.in_shift[i] : 0.f; }
  __syncthreads();
  const T* x = (const T*)d.x; T* dx = (T*)d.dx;
  const bool tr = d.in_scale != nullptr;
  const int lane = threadIdx.x & 63;
  for (int g = 0; g < cin / 8; ++g) {
    float acc[8][HEAD_NCMAX];
    float accb[HEAD_NCMAX];
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int k = 0; k < HEAD_NCMAX; ++k) acc[e][k] = 0.f;
#pragma unroll
    for (int k = 0; k < HEAD_NCMAX; ++k) accb[k] = 0.f;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < d.npix; p += (long long)gridDim.x * blockDim.x) {
      float dl[HEAD_NCMAX];
#pragma unroll
      for (int k = 0; k < HEAD_NCMAX; ++k) dl[k] = k < nc ? d.dlogits[p * nc + k] : 0.f;
      float v[8], o[8];
      load8<T>(x + p * d.ldx + g * 8, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ch = g * 8 + e;
        const float a = tr ? fmaxf(v[e] * lsc[ch] + lsh[ch], 0.f) : v[e];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < HEAD_NCMAX; ++k) if (k < nc) { s += dl[k] * lw[ch * nc + k]; acc[e][k] += a * dl[k]; }
        o[e] = s;
      }
      if (dx) store8<T>(dx + p * d.lddx + g * 8, o);
      if (g == 0) {
#pragma unroll
        for (int k = 0; k < HEAD_NCMAX; ++k) accb[k] += dl[k];
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int k = 0; k < HEAD_NCMAX; ++k) if (k < nc) {
        const float s = wave_sum(acc[e][k]);
        if (lane == 0) atomicAdd(&aw[(g * 8 + e) * nc + k], s);
      }
    if (g == 0) {
#pragma unroll
      for (int k = 0; k < HEAD_NCMAX; ++k) if (k < nc) { const float s = wave_sum(accb[k]); if (lane == 0) atomicAdd(&ab[k], s); }
    }
  }
  __syncthreads();
  if (d.dw) for (int i = threadIdx.x; i < cin * nc; i += blockDim.x) atomicAdd(d.dw + i, aw[i]);
  if (d.db) for (int i = threadIdx.x; i < nc; i += blockDim.x) atomicAdd(d.db + i, ab[i]);
}
extern "C" int satcv_head_bwd(const satcv_head_desc* d, void* stream) {
  SATCV_CHECK(d && d->x && d->w && d->dlogits, "head_bwd: null pointer");
  SATCV_CHECK(!d->bnr_sums || (d->bnr_mean && d->bnr_rstd && d->in_scale && d->bnr_sums_ld >= d->cin), "head_bwd: incomplete bnr_* fields");
  SATCV_CHECK(d->cin > 0 && d->cin % 8 == 0 && d->ncls >= 1 && d->ncls <= HEAD_NCMAX && d->npix > 0, "head_bwd: bad dims");
  SATCV_CHECK(!d->partials || satcv_head_bwd_workspace(d) > 0, "head_bwd: partial rows only with the register-resident kernel");
  const size_t lds = (size_t)(2 * d->cin * d->ncls + d->ncls + 2 * d->cin) * sizeof(float);
  DISPATCH_T(d->dtype, { if (!head_fast_launch<T, true>(d, (hipStream_t)stream))
      { if (d->bnr_sums) { satcv_set_error("head_bwd: bnr fusion only on the register-resident kernel (cin in 16/32/64, small ncls)"); return SATCV_ERR_UNSUPPORTED; }
        hipLaunchKernelGGL(head_bwd_kernel<T>, dim3(ew_grid(d->npix, 1024)), dim3(EW_BLOCK), lds, (hipStream_t)stream, *d); } });
  LAUNCH_OK("head_bwd");
  return SATCV_OK;
}

static bool head_fast_shape(const satcv_head_desc* d) {
  static const int shapes[][2] = {{1, 16}, {2, 16}, {1, 32}, {2, 32}, {3, 32}, {4, 32}, {1, 64}, {2, 64}};
  for (auto& s : shapes) if (d->ncls == s[0] && d->cin == s[1]) return true;
  return false;
}
extern "C" int64_t satcv_head_bwd_workspace(const satcv_head_desc* d) {
  if (!d || d->npix <= 0 || d->cin <= 0 || d->cin % 8 || !head_fast_shape(d)) return 0;
  return (int64_t)ew_grid(d->npix * (d->cin / 8)) * (d->cin * d->ncls + d->ncls) * (int64_t)sizeof(float);
}
// dw[i] / db[i] += sum over the workgroup rows of `partials`: one wave per output element, lane l adds rows l, l+64, ... in order and
// the 64 lane sums are combined by the fixed butterfly of wave_sum -- the same association every run
__global__ __launch_bounds__(64) void head_bwd_finalize_kernel(const float* __restrict__ partials, int rows, int nw, int nb, float* dw, float* db) {
  const int i = blockIdx.x;
  float s = 0.f;
  for (int r = threadIdx.x; r < rows; r += 64) s += partials[(size_t)r * (nw + nb) + i];
  s = wave_sum(s);
  if (threadIdx.x == 0) {
    if (i < nw) { if (dw) dw[i] += s; }
    else if (db) db[i - nw] += s;
  }
}
extern "C" int satcv_head_bwd_finalize(const satcv_head_desc* d, void* stream) {
  SATCV_CHECK(d && d->partials && satcv_head_bwd_workspace(d) > 0, "head_bwd_finalize: no partial rows for this descriptor");
  const int nw = d->cin * d->ncls, nb = d->ncls, rows = ew_grid(d->npix * (d->cin / 8));
  hipLaunchKernelGGL(head_bwd_finalize_kernel, dim3(nw + nb), dim3(64), 0, (hipStream_t)stream, d->partials, rows, nw, nb, d->dw, d->db);
  LAUNCH_OK("head_bwd_finalize");
  return SATCV_OK;
}

// ------------------------------------------------- ResNet / DeepLab inference helpers (build-defined)
// general max pooling on an activated NHWC tensor: window k, stride s, symmetric padding pad (-inf padded)
template <typename T>
__global__ void maxpool_kernel(const T* __restrict__ x, T* __restrict__ out, int n, int h, int w, int c, int k, int s, int pad, int ho, int wo) {
  const int G = c / 8;
  const long long total = (long long)n * ho * wo * G;
  for (long long it = blockIdx.x * (long long)blockDim.x + threadIdx.x; it < total; it += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(it % G);
    long long p = it / G;
    const int ox = (int)(p % wo); p /= wo;
    const int oy = (int)(p % ho);
    const int img = (int)(p / ho);
    float mx[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) mx[e] = -INFINITY;
    for (int i = 0; i < k; ++i) {
      const int y = oy * s + i - pad; if (y < 0 || y >= h) continue;
      for (int j = 0; j < k; ++j) {
        const int xx = ox * s + j - pad; if (xx < 0 || xx >= w) continue;
        float v[8];
        load8<T>(x + ((size_t)(img * h + y) * w + xx) * c + g * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) mx[e] = fmaxf(mx[e], v[e]);
      }
    }
    store8<T>(out + ((size_t)(img * ho + oy) * wo + ox) * c + g * 8, mx);
  }
}
extern "C" int satcv_maxpool(const void* x, void* out, int32_t n, int32_t h, int32_t w_, int32_t c, int32_t k, int32_t s, int32_t pad, int32_t dtype,
                             void* stream) {
  SATCV_CHECK(x && out && n > 0 && h > 0 && w_ > 0 && c > 0 && c % 8 == 0 && k >= 1 && s >= 1 && pad >= 0, "maxpool: bad args");
  const int ho = (h + 2 * pad - k) / s + 1, wo = (w_ + 2 * pad - k) / s + 1;
  SATCV_CHECK(ho > 0 && wo > 0, "maxpool: empty output");
  DISPATCH_T8(dtype, hipLaunchKernelGGL(maxpool_kernel<T>, dim3(ew_grid((long long)n * ho * wo * (c / 8))), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                                       (const T*)x, (T*)out, n, h, w_, c, k, s, pad, ho, wo));
  LAUNCH_OK("maxpool");
  return SATCV_OK;
}
// out = Q(relu?(scale*x + shift)) with separate channel strides (skip half of the folded concat -> BN -> ReLU)
template <typename T, typename TO>
__global__ void affine_requant_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ scale, const float* __restrict__ shift, int relu,
                                      TO* __restrict__ out, int ldo, long long npix, int c) {
  const int G = c / 8;
  const long long total = npix * G;
  for (long long it = blockIdx.x * (long long)blockDim.x + threadIdx.x; it < total; it += (long long)gridDim.x * blockDim.x) {
    const long long p = it / G; const int g = (int)(it % G);
    float v[8];
    load8<T>(x + p * ldx + g * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float a = v[e] * scale[g * 8 + e] + shift[g * 8 + e];
      v[e] = relu ? fmaxf(a, 0.f) : a;
    }
    store8<TO>(out + p * ldo + g * 8, v);
  }
}
extern "C" int satcv_affine_requant(const void* x, int32_t ldx, const float* scale, const float* shift, int32_t relu, void* out, int32_t ldo,
                                    int64_t npix, int32_t c, int32_t dtype, int32_t dtype_out, void* stream) {
  SATCV_CHECK(x && scale && shift && out && npix > 0 && c > 0 && c % 8 == 0 && ldx >= c && ldo >= c, "affine_requant: bad args");
  SATCV_CHECK(dtype_out == dtype || dtype_out == SATCV_FP8 || (dtype == SATCV_FP8 && dtype_out == SATCV_BF16),
              "affine_requant: output dtype must equal the input dtype, be fp8, or be bf16 from fp8");
  if (dtype == SATCV_FP8 && dtype_out == SATCV_BF16) {       // de-quantisation where the hybrid inference graph leaves its fp8 levels
    hipLaunchKernelGGL((affine_requant_kernel<fp8, bf16>), dim3(ew_grid(npix * (c / 8))), dim3(EW_BLOCK), 0, (hipStream_t)stream, (const fp8*)x, ldx,
                       scale, shift, relu, (bf16*)out, ldo, (long long)npix, c);
  } else if (dtype_out == dtype) {
    DISPATCH_T8(dtype, hipLaunchKernelGGL((affine_requant_kernel<T, T>), dim3(ew_grid(npix * (c / 8))), dim3(EW_BLOCK), 0, (hipStream_t)stream, (const T*)x, ldx,
                                          scale, shift, relu, (T*)out, ldo, (long long)npix, c));
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL((affine_requant_kernel<T, fp8>), dim3(ew_grid(npix * (c / 8))), dim3(EW_BLOCK), 0, (hipStream_t)stream, (const T*)x, ldx,
                                         scale, shift, relu, (fp8*)out, ldo, (long long)npix, c));
  }
  LAUNCH_OK("affine_requant");
  return SATCV_OK;
}
// out = relu?( affine?(y) + affine?(res) ): the residual join of a bottleneck block
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void add_act_kernel(const T* __restrict__ y, const float* __restrict__ ysc, const float* __restrict__ ysh,
                                                           const T* __restrict__ res, const float* __restrict__ rsc, const float* __restrict__ rsh, int relu,
                                                           T* __restrict__ out, long long npix, int c) {
  // a thread keeps one 8-channel group for all its pixels: the four per-channel coefficients live in registers
  const int G = c / 8;
  const long long nthreads = (long long)gridDim.x * blockDim.x, per = nthreads / G;
  const long long gid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (gid >= per * G) return;
  const int g = (int)(gid % G);
  float s0[8], h0[8], s1[8], h1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ch = g * 8 + e;
    s0[e] = ysc ? ysc[ch] : 1.f; h0[e] = ysc ? ysh[ch] : 0.f;
    s1[e] = rsc ? rsc[ch] : 1.f; h1[e] = rsc ? rsh[ch] : 0.f;
  }
  const bool ya = ysc != nullptr, ra = rsc != nullptr;
#pragma unroll 4
  for (long long p = gid / G; p < npix; p += per) {
    float a[8], b[8];
    load8<T>(y + p * c + g * 8, a);
    load8<T>(res + p * c + g * 8, b);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float u = ya ? a[e] * s0[e] + h0[e] : a[e];
      const float r = ra ? b[e] * s1[e] + h1[e] : b[e];
      u += r;
      a[e] = relu ? fmaxf(u, 0.f) : u;
    }
    store8<T>(out + p * c + g * 8, a);
  }
}
extern "C" int satcv_add_act(const void* y, const float* y_scale, const float* y_shift, const void* res, const float* res_scale, const float* res_shift,
                             int32_t relu, void* out, int64_t npix, int32_t c, int32_t dtype, void* stream) {
  SATCV_CHECK(y && res && out && npix > 0 && c > 0 && c % 8 == 0 && c <= 2048, "add_act: bad args");
  DISPATCH_T(dtype, hipLaunchKernelGGL(add_act_kernel<T>, dim3(ew_grid(npix * (c / 8))), dim3(EW_BLOCK), 0, (hipStream_t)stream, (const T*)y, y_scale,
                                       y_shift, (const T*)res, res_scale, res_shift, relu, (T*)out, (long long)npix, c));
  LAUNCH_OK("add_act");
  return SATCV_OK;
}
template <typename T>
__global__ void relu_bwd_kernel(const T* __restrict__ act, T* __restrict__ g, long long nvec) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
    float a[8], v[8];
    load8<T>(act + i * 8, a); load8<T>(g + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = a[e] > 0.f ? v[e] : 0.f;
    store8<T>(g + i * 8, v);
  }
}
extern "C" int satcv_relu_bwd(const void* act, void* g, int64_t count, int32_t dtype, void* stream) {
  SATCV_CHECK(act && g && count > 0 && count % 8 == 0, "relu_bwd: bad args");
  DISPATCH_T(dtype, hipLaunchKernelGGL(relu_bwd_kernel<T>, dim3(ew_grid(count / 8)), dim3(EW_BLOCK), 0, (hipStream_t)stream, (const T*)act, (T*)g,
                                       (long long)(count / 8)));
  LAUNCH_OK("relu_bwd");
  return SATCV_OK;
}
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void bias_grad_kernel(const T* __restrict__ dy, int lddy, long long npix, int c, float* __restrict__ dbias,
                                                             float* __restrict__ partials) {
  extern __shared__ float lds[];
  const int G = c / 8;
  const long long nthreads = (long long)gridDim.x * blockDim.x, per = nthreads / G;
  const long long gid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const bool active = gid < per * G;
  const int g = (int)(gid % G);
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  if (active) {
    for (long long p = gid / G; p < npix; p += per) {
      float v[8];
      load8<T>(dy + p * lddy + g * 8, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += v[e];
    }
  }
  if (partials) {
    // reproducible form: the threads of a channel group park their sums and are added in thread order; one row per workgroup
    float* park = lds + c;                                   // [EW_BLOCK][8]
#pragma unroll
    for (int e = 0; e < 8; ++e) park[threadIdx.x * 8 + e] = active ? s[e] : 0.f;
    __syncthreads();
    for (int i = threadIdx.x; i < c; i += blockDim.x) {
      const int gg = i / 8, e = i % 8;
      // threads with (global id % G) == gg: local ids t with (blockIdx.x * blockDim.x + t) % G == gg
      const int first = (int)(((long long)gg - (long long)blockIdx.x * blockDim.x % G + G) % G);
      float t = 0.f;
      for (int th = first; th < (int)blockDim.x; th += G) t += park[th * 8 + e];
      partials[(size_t)blockIdx.x * c + i] = t;
    }
    return;
  }
  for (int i = threadIdx.x; i < c; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  if (active) {
#pragma unroll
    for (int e = 0; e < 8; ++e) atomicAdd(&lds[g * 8 + e], s[e]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < c; i += blockDim.x) atomicAdd(dbias + i, lds[i]);
}
// dbias[i] += sum of the rows: one wave per channel, fixed association (see head_bwd_finalize_kernel)
__global__ __launch_bounds__(64) void bias_grad_finalize_kernel(const float* __restrict__ partials, int rows, int c, float* dbias) {
  const int i = blockIdx.x;
  float s = 0.f;
  for (int r = threadIdx.x; r < rows; r += 64) s += partials[(size_t)r * c + i];
  s = wave_sum(s);
  if (threadIdx.x == 0) dbias[i] += s;
}
extern "C" int64_t satcv_bias_grad_workspace(int64_t npix, int32_t c) {
  if (npix <= 0 || c <= 0 || c % 8) return 0;
  return (int64_t)ew_grid(npix * (c / 8)) * c * (int64_t)sizeof(float);
}
extern "C" int satcv_bias_grad(const void* dy, int32_t lddy, int64_t npix, int32_t c, int32_t dtype, float* dbias, float* partials, void* stream) {
  SATCV_CHECK(dy && dbias && npix > 0 && c > 0 && c % 8 == 0 && c <= 2048 && lddy >= c, "bias_grad: bad args");
  const int grid = ew_grid(npix * (c / 8));
  const size_t lds = (size_t)c * sizeof(float) + (partials ? (size_t)EW_BLOCK * 8 * sizeof(float) : 0);
  DISPATCH_T(dtype, hipLaunchKernelGGL(bias_grad_kernel<T>, dim3(grid), dim3(EW_BLOCK), lds, (hipStream_t)stream, (const T*)dy,
                                       lddy, (long long)npix, c, dbias, partials));
  LAUNCH_OK("bias_grad");
  if (partials) {
    hipLaunchKernelGGL(bias_grad_finalize_kernel, dim3(c), dim3(64), 0, (hipStream_t)stream, partials, grid, c, dbias);
    LAUNCH_OK("bias_grad_finalize");
  }
  return SATCV_OK;
}
// bilinear upsampling (half-pixel centres, edge clamped -- tf.keras UpSampling2D(interpolation='bilinear')) of fp32 logits by an
// integer factor, then softmax + argmax (or sigmoid + threshold)
__global__ void upsample_head_kernel(const float* __restrict__ logits, int n, int h, int w, int nc, int f, int activation, float thresh,
                                     float* __restrict__ probs, int32_t* __restrict__ classes) {
  const int ho = h * f, wo = w * f;
  const long long total = (long long)n * ho * wo;
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < total; p += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(p % wo);
    const int oy = (int)((p / wo) % ho);
    const int img = (int)(p / ((long long)wo * ho));
    const float sy = fminf(fmaxf((oy + 0.5f) / f - 0.5f, 0.f), (float)(h - 1));
    const float sx = fminf(fmaxf((ox + 0.5f) / f - 0.5f, 0.f), (float)(w - 1));
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float fy = sy - y0, fx = sx - x0;
    const float* b = logits + (size_t)img * h * w * nc;
    float z[HEAD_NCMAX];
#pragma unroll
    for (int k = 0; k < HEAD_NCMAX; ++k) {
      z[k] = -INFINITY;
      if (k < nc) {
        const float v00 = b[((size_t)y0 * w + x0) * nc + k], v01 = b[((size_t)y0 * w + x1) * nc + k];
        const float v10 = b[((size_t)y1 * w + x0) * nc + k], v11 = b[((size_t)y1 * w + x1) * nc + k];
        z[k] = (v00 * (1.f - fx) + v01 * fx) * (1.f - fy) + (v10 * (1.f - fx) + v11 * fx) * fy;
      }
    }
    if (activation == 0) {
      float mx = z[0];
#pragma unroll
      for (int k = 1; k < HEAD_NCMAX; ++k) if (k < nc) mx = fmaxf(mx, z[k]);
      float s = 0.f, ex[HEAD_NCMAX];
#pragma unroll
      for (int k = 0; k < HEAD_NCMAX; ++k) { ex[k] = k < nc ? expf(z[k] - mx) : 0.f; s += ex[k]; }
      float best = -1.f; int am = 0;
#pragma unroll
      for (int k = 0; k < HEAD_NCMAX; ++k) if (k < nc) {
        const float pr = ex[k] / s;
        probs[p * nc + k] = pr;
        if (pr > best) { best = pr; am = k; }
      }
      if (classes) classes[p] = am;
    } else {
#pragma unroll
      for (int k = 0; k < HEAD_NCMAX; ++k) if (k < nc) {
        const float pr = 1.f / (1.f + expf(-z[k]));
        probs[p * nc + k] = pr;
        if (classes) classes[p * nc + k] = pr > thresh ? 1 : 0;
      }
    }
  }
}
extern "C" int satcv_upsample_head(const float* logits, int32_t n, int32_t h, int32_t w_, int32_t ncls, int32_t factor, int32_t activation, float thresh,
                                   float* probs, int32_t* classes, void* stream) {
  SATCV_CHECK(logits && probs && n > 0 && h > 0 && w_ > 0 && ncls >= 1 && ncls <= HEAD_NCMAX && factor >= 1, "upsample_head: bad args");
  hipLaunchKernelGGL(upsample_head_kernel, dim3(ew_grid((long long)n * h * factor * w_ * factor)), dim3(EW_BLOCK), 0, (hipStream_t)stream, logits, n, h, w_,
                     ncls, factor, activation, thresh, probs, classes);
  LAUNCH_OK("upsample_head");
  return SATCV_OK;
}

// ----------------------------------------------------------------------- dropout
// layers.SpatialDropout2D / layers.Dropout (utils/model_tools.py:311, 351, 363, 402), training only.
// mask values are 0 or 1/(1-rate); mode 0: one value per (image, channel) [SpatialDropout2D], mode 1: per element.
__device__ __forceinline__ float hash_uniform(unsigned long long seed, unsigned long long idx) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ULL * (idx + 1);          // splitmix64 finaliser (counter based)
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z = z ^ (z >> 31);
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}
__global__ void dropout_mask_kernel(unsigned long long seed, unsigned long long offset, float rate, long long count, float* mask) {
  const float keep = 1.f / (1.f - rate);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x)
    mask[i] = hash_uniform(seed, offset + (unsigned long long)i) >= rate ? keep : 0.f;
}
extern "C" int satcv_dropout_mask(uint64_t seed, uint64_t offset, float rate, int64_t count, float* mask, void* stream) {
  SATCV_CHECK(mask && count > 0 && rate >= 0.f && rate < 1.f, "dropout_mask: bad args");
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(ew_grid(count)), dim3(EW_BLOCK), 0, (hipStream_t)stream, (unsigned long long)seed,
                     (unsigned long long)offset, rate, (long long)count, mask);
  LAUNCH_OK("dropout_mask");
  return SATCV_OK;
}
// out = act(x) * mask, act(x) = relu?(scale*x+shift) when scale != NULL else x.  Also the backward (g*mask) with scale == NULL.
template <typename T>
__global__ void dropout_apply_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ scale, const float* __restrict__ shift, int relu,
                                     const float* __restrict__ mask, int ldm, int mode, T* __restrict__ out, int ldo, long long npix, int hw, int c) {
  const int G = c / 8;
  const long long total = npix * G;
  for (long long it = blockIdx.x * (long long)blockDim.x + threadIdx.x; it < total; it += (long long)gridDim.x * blockDim.x) {
    const long long p = it / G; const int g = (int)(it % G);
    float v[8];
    load8<T>(x + p * ldx + g * 8, v);
    const float* m = mode == 0 ? mask + (p / hw) * ldm + g * 8 : mask + p * ldm + g * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = v[e];
      if (scale) { a = a * scale[g * 8 + e] + shift[g * 8 + e]; if (relu) a = fmaxf(a, 0.f); }
      v[e] = a * m[e];
    }
    store8<T>(out + p * ldo + g * 8, v);
  }
}
extern "C" int satcv_dropout_apply(const void* x, int32_t ldx, const float* scale, const float* shift, int32_t relu, const float* mask,
                                   int32_t ldm, int32_t mask_mode, void* out, int32_t ldo, int32_t n, int32_t hw, int32_t c, int32_t dtype,
                                   void* stream) {
  SATCV_CHECK(x && mask && out && n > 0 && hw > 0 && c > 0 && c % 8 == 0 && ldm >= c && (mask_mode == 0 || mask_mode == 1), "dropout_apply: bad args");
  const long long npix = (long long)n * hw;
  DISPATCH_T(dtype, hipLaunchKernelGGL(dropout_apply_kernel<T>, dim3(ew_grid(npix * (c / 8))), dim3(EW_BLOCK), 0, (hipStream_t)stream, (const T*)x, ldx,
                                       scale, shift, relu, mask, ldm, mask_mode, (T*)out, ldo, npix, hw, c));
  LAUNCH_OK("dropout_apply");
  return SATCV_OK;
}

// ------------------------------------------------------------------------- losses
__global__ void loss_kernel(int kind, const float* __restrict__ probs, const float* __restrict__ yt, const float* __restrict__ wts,
                            int nc, int activation, long long npix, float grad_scale, float* loss_out, float* dlogits) {
  __shared__ float red[EW_BLOCK / 64];
  float lsum = 0.f;
  const float inv_n = 1.f / (float)npix;
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < npix; p += (long long)gridDim.x * blockDim.x) {
    float pr[HEAD_NCMAX], t[HEAD_NCMAX], gp[HEAD_NCMAX];
#pragma unroll
    for (int k = 0; k < HEAD_NCMAX; ++k) { pr[k] = k < nc ? probs[p * nc + k] : 0.f; t[k] = k < nc ? yt[p * nc + k] : 0.f; gp[k] = 0.f; }
    if (kind == 0) {            // weighted categorical cross-entropy, mean over pixels
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < HEAD_NCMAX; ++k) s += pr[k];
      const float lo = 1e-7f, hi = 1.f - 1e-7f;
      float G[HEAD_NCMAX], dot = 0.f;
#pragma unroll
      for (int k = 0; k < HEAD_NCMAX; ++k) {
        G[k] = 0.f;
        if (k < nc) {
          const float o = pr[k] / s;
          const float oc = fminf(fmaxf(o, lo), hi);
          lsum += -wts[k] * t[k] * logf(oc) * inv_n;
          const bool inside = (o >= lo) && (o <= hi);
          G[k] = inside ? -wts[k] * t[k] / oc * inv_n : 0.f;
          dot += G[k] * o;
        }
      }
#pragma unroll
      for (int k = 0; k < HEAD_NCMAX; ++k) gp[k] = (G[k] - dot) / s;
    } else {                    // weighted BCE on probabilities, mean over pixels*classes
      const float pw = wts[0], lo = 0.00001f, hi = 0.99999f;
      const float inv_e = inv_n / (float)nc;
#pragma unroll
      for (int k = 0; k < HEAD_NCMAX; ++k) if (k < nc) {
        const float yp = fminf(fmaxf(pr[k], lo), hi);
        lsum += (t[k] * -logf(yp) * pw + (1.f - t[k]) * -logf(1.f - yp)) * inv_e;
        const bool inside = (pr[k] >= lo) && (pr[k] <= hi);
        gp[k] = inside ? (-t[k] * pw / yp + (1.f - t[k]) / (1.f - yp)) * inv_e : 0.f;
      }
    }
    if (activation == 0) {      // softmax Jacobian
      float dot = 0.f;
#pragma unroll
      for (int k = 0; k < HEAD_NCMAX; ++k) dot += gp[k] * pr[k];
#pragma unroll
      for (int k = 0; k < HEAD_NCMAX; ++k) if (k < nc) dlogits[p * nc + k] = grad_scale * pr[k] * (gp[k] - dot);
    } else if (activation == 2) {   // linear head (regression): the outputs are the logits
#pragma unroll
      for (int k = 0; k < HEAD_NCMAX; ++k) if (k < nc) dlogits[p * nc + k] = grad_scale * gp[k];
    } else {
#pragma unroll
      for (int k = 0; k < HEAD_NCMAX; ++k) if (k < nc) dlogits[p * nc + k] = grad_scale * gp[k] * pr[k] * (1.f - pr[k]);
    }
  }
  lsum = wave_sum(lsum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) { float s = 0.f; for (int i = 0; i < EW_BLOCK / 64; ++i) s += red[i]; atomicAdd(loss_out, s); }
}
// Round 6: the same arithmetic, statement for statement, with the class count at compile time (2 or 4: one 8- / 16-byte load and store per
// tensor and pixel instead of a scalar one per class behind a run-time `k < nc`) for the case every U-Net of the reference trains with --
// weighted categorical cross-entropy on a softmax head (utils/model_tools.py:25, 414).  42 -> ~25 us per step at batch 64.
template <int NC>
__global__ __launch_bounds__(EW_BLOCK) void loss_cce_softmax_kernel(const float* __restrict__ probs, const float* __restrict__ yt, const float* __restrict__ wts,
                                                                    long long npix, float grad_scale, float* loss_out, float* dlogits) {
  typedef float fvec __attribute__((ext_vector_type(NC)));
  __shared__ float red[EW_BLOCK / 64];
  float lsum = 0.f;
  const float inv_n = 1.f / (float)npix;
  float wt[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) wt[k] = wts[k];
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < npix; p += (long long)gridDim.x * blockDim.x) {
    const fvec prv = reinterpret_cast<const fvec*>(probs)[p], tv = reinterpret_cast<const fvec*>(yt)[p];
    float pr[NC], t[NC], gp[NC], G[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) { pr[k] = prv[k]; t[k] = tv[k]; }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NC; ++k) s += pr[k];
    const float lo = 1e-7f, hi = 1.f - 1e-7f;
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const float o = pr[k] / s;
      const float oc = fminf(fmaxf(o, lo), hi);
      lsum += -wt[k] * t[k] * logf(oc) * inv_n;
      const bool inside = (o >= lo) && (o <= hi);
      G[k] = inside ? -wt[k] * t[k] / oc * inv_n : 0.f;
      dot += G[k] * o;
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) gp[k] = (G[k] - dot) / s;
    float dot2 = 0.f;                                   // softmax Jacobian
#pragma unroll
    for (int k = 0; k < NC; ++k) dot2 += gp[k] * pr[k];
    fvec dl;
#pragma unroll
    for (int k = 0; k < NC; ++k) dl[k] = grad_scale * pr[k] * (gp[k] - dot2);
    reinterpret_cast<fvec*>(dlogits)[p] = dl;
  }
  lsum = wave_sum(lsum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) { float s = 0.f; for (int i = 0; i < EW_BLOCK / 64; ++i) s += red[i]; atomicAdd(loss_out, s); }
}
extern "C" int satcv_loss_fwd_bwd(int32_t kind, const float* probs, const float* y_true, const float* weights, int32_t ncls, int32_t activation,
                                  int64_t npix, float grad_scale, float* loss_out, float* dlogits, void* stream) {
  SATCV_CHECK(probs && y_true && weights && loss_out && dlogits, "loss: null pointer");
  SATCV_CHECK((kind == 0 || kind == 1) && ncls >= 1 && ncls <= HEAD_NCMAX && npix > 0, "loss: bad args (kind=%d ncls=%d)", kind, ncls);
  static const bool fast = !(getenv("SATCV_LOSS_FAST") && atoi(getenv("SATCV_LOSS_FAST")) == 0);
  const bool al = ((uintptr_t)probs % (4 * ncls) == 0) && ((uintptr_t)y_true % (4 * ncls) == 0) && ((uintptr_t)dlogits % (4 * ncls) == 0);
  if (fast && kind == 0 && activation == 0 && (ncls == 2 || ncls == 4) && al) {
    if (ncls == 2) hipLaunchKernelGGL(loss_cce_softmax_kernel<2>, dim3(ew_grid(npix)), dim3(EW_BLOCK), 0, (hipStream_t)stream, probs, y_true, weights, (long long)npix, grad_scale, loss_out, dlogits);
    else hipLaunchKernelGGL(loss_cce_softmax_kernel<4>, dim3(ew_grid(npix)), dim3(EW_BLOCK), 0, (hipStream_t)stream, probs, y_true, weights, (long long)npix, grad_scale, loss_out, dlogits);
    LAUNCH_OK("loss");
    return SATCV_OK;
  }
  hipLaunchKernelGGL(loss_kernel, dim3(ew_grid(npix, 1024)), dim3(EW_BLOCK), 0, (hipStream_t)stream, kind, probs, y_true, weights, ncls, activation,
                     (long long)npix, grad_scale, loss_out, dlogits);
  LAUNCH_OK("loss");
  return SATCV_OK;
}


// ---- ratio-type losses: need global (or per-image) sums before the gradient is known
// ws layout: [nimg][3][ncls] = per image, per class: sum t*p, sum (t+p) [dice] / sum (t+(1-t)p) [iou], sum t (counts);
// mse_4d uses ws[0] = sum of squared finite errors, ws[1] = count of finite elements.
__global__ void loss_sums_kernel(int kind, const float* __restrict__ probs, const float* __restrict__ yt, int nc, long long ppi, int nimg, float* ws) {
  const int img = blockIdx.y;
  float a[HEAD_NCMAX], b[HEAD_NCMAX], c[HEAD_NCMAX];
#pragma unroll
  for (int k = 0; k < HEAD_NCMAX; ++k) { a[k] = 0.f; b[k] = 0.f; c[k] = 0.f; }
  for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < ppi; q += (long long)gridDim.x * blockDim.x) {
    const long long p = img * ppi + q;
#pragma unroll
    for (int k = 0; k < HEAD_NCMAX; ++k) if (k < nc) {
      const float pr = probs[p * nc + k], t = yt[p * nc + k];
      if (kind == 2) { a[k] += t * pr; b[k] += t + pr; c[k] += t; }
      else if (kind == 3) { a[k] += t * pr; b[k] += t + (1.f - t) * pr; }
      else { const float d = (pr - t) * (pr - t); if (isfinite(d)) { a[k] += d; b[k] += 1.f; } }
    }
  }
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < HEAD_NCMAX; ++k) if (k < nc) {
    const float sa = wave_sum(a[k]), sb = wave_sum(b[k]), sc = wave_sum(c[k]);
    if (lane == 0) {
      float* w = ws + ((size_t)img * 3) * nc;
      atomicAdd(w + k, sa); atomicAdd(w + nc + k, sb); atomicAdd(w + 2 * nc + k, sc);
    }
  }
}
__global__ void loss_global_grad_kernel(int kind, const float* __restrict__ probs, const float* __restrict__ yt, const float* __restrict__ gw,
                                        int nc, int activation, long long ppi, int nimg, float eps, float grad_scale, const float* __restrict__ ws,
                                        float* loss_out, float* dlogits) {
  const int img = blockIdx.y;
  // per-image (dice) or global (iou, mse) constants, recomputed by every thread from the tiny sums array
  float wk[HEAD_NCMAX], num = 0.f, den = 0.f, I = 0.f, U = 0.f, cnt = 0.f, sq = 0.f;
  if (kind == 2) {
#pragma unroll
    for (int k = 0; k < HEAD_NCMAX; ++k) {
      wk[k] = 0.f;
      if (k < nc) {
        const float* w = ws + ((size_t)img * 3) * nc;
        if (gw) wk[k] = gw[k];
        else { const float cn = w[2 * nc + k]; const float t = 1.f / (cn * cn); wk[k] = isfinite(t) ? t : eps; }
        num += wk[k] * w[k]; den += wk[k] * w[nc + k];
      }
    }
  } else {
    for (int i = 0; i < nimg; ++i) for (int k = 0; k < nc; ++k) {
      const float* w = ws + ((size_t)i * 3) * nc;
      if (kind == 3) { I += w[k]; U += w[nc + k]; } else { sq += w[k]; cnt += w[nc + k]; }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (kind == 2) atomicAdd(loss_out, (1.f - 2.f * num / den) / (float)nimg);
    else if (img == 0) atomicAdd(loss_out, kind == 3 ? 1.f - I / U : sq / cnt);
  }
  for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < ppi; q += (long long)gridDim.x * blockDim.x) {
    const long long p = img * ppi + q;
    float pr[HEAD_NCMAX], gp[HEAD_NCMAX];
#pragma unroll
    for (int k = 0; k < HEAD_NCMAX; ++k) {
      pr[k] = 0.f; gp[k] = 0.f;
      if (k < nc) {
        pr[k] = probs[p * nc + k];
        const float t = yt[p * nc + k];
        if (kind == 2) gp[k] = -2.f * wk[k] * (t * den - num) / (den * den) / (float)nimg;
        else if (kind == 3) gp[k] = -(t * U - I * (1.f - t)) / (U * U);
        else { const float d = pr[k] - t; gp[k] = isfinite(d * d) ? 2.f * d / cnt : 0.f; }
      }
    }
    if (activation == 0) {
      float dot = 0.f;
#pragma unroll
      for (int k = 0; k < HEAD_NCMAX; ++k) dot += gp[k] * pr[k];
#pragma unroll
      for (int k = 0; k < HEAD_NCMAX; ++k) if (k < nc) dlogits[p * nc + k] = grad_scale * pr[k] * (gp[k] - dot);
    } else if (activation == 2) {   // linear head (regression): the outputs are the logits
#pragma unroll
      for (int k = 0; k < HEAD_NCMAX; ++k) if (k < nc) dlogits[p * nc + k] = grad_scale * gp[k];
    } else {
#pragma unroll
      for (int k = 0; k < HEAD_NCMAX; ++k) if (k < nc) dlogits[p * nc + k] = grad_scale * gp[k] * pr[k] * (1.f - pr[k]);
    }
  }
}
extern "C" int satcv_loss_global_fwd_bwd(int32_t kind, const float* probs, const float* y_true, const float* class_weights, int32_t ncls,
                                         int32_t activation, int32_t nimg, int64_t pix_per_img, float eps, float grad_scale, float* workspace,
                                         float* loss_out, float* dlogits, void* stream) {
  SATCV_CHECK(probs && y_true && workspace && loss_out && dlogits, "loss_global: null pointer");
  SATCV_CHECK(kind >= 2 && kind <= 4 && ncls >= 1 && ncls <= HEAD_NCMAX && nimg > 0 && nimg <= 65535 && pix_per_img > 0, "loss_global: bad args");
  hipStream_t st = (hipStream_t)stream;
  SATCV_HIP(hipMemsetAsync(workspace, 0, (size_t)nimg * 3 * ncls * sizeof(float), st));
  int gx = (int)((pix_per_img + EW_BLOCK - 1) / EW_BLOCK); if (gx > 256) gx = 256; if (gx < 1) gx = 1;
  hipLaunchKernelGGL(loss_sums_kernel, dim3(gx, nimg), dim3(EW_BLOCK), 0, st, kind, probs, y_true, ncls, (long long)pix_per_img, nimg, workspace);
  hipLaunchKernelGGL(loss_global_grad_kernel, dim3(gx, nimg), dim3(EW_BLOCK), 0, st, kind, probs, y_true, class_weights, ncls, activation,
                     (long long)pix_per_img, nimg, eps, grad_scale, workspace, loss_out, dlogits);
  LAUNCH_OK("loss_global");
  return SATCV_OK;
}

// --------------------------------------------------------------------- confusion
__global__ void confusion_kernel(const int32_t* __restrict__ classes, const float* __restrict__ yt, int nc, long long npix, unsigned long long* conf) {
  __shared__ unsigned int cnt[HEAD_NCMAX * HEAD_NCMAX];
  for (int i = threadIdx.x; i < nc * nc; i += blockDim.x) cnt[i] = 0;
  __syncthreads();
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < npix; p += (long long)gridDim.x * blockDim.x) {
    int t = 0; float best = yt[p * nc];
    for (int k = 1; k < nc; ++k) { const float v = yt[p * nc + k]; if (v > best) { best = v; t = k; } }
    int pc = classes[p]; pc = pc < 0 ? 0 : (pc >= nc ? nc - 1 : pc);
    atomicAdd(&cnt[t * nc + pc], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nc * nc; i += blockDim.x) if (cnt[i]) atomicAdd(conf + i, (unsigned long long)cnt[i]);
}
extern "C" int satcv_confusion(const int32_t* classes, const float* y_true, int32_t ncls, int64_t npix, int64_t* confusion, void* stream) {
  SATCV_CHECK(classes && y_true && confusion && ncls >= 1 && ncls <= HEAD_NCMAX && npix > 0, "confusion: bad args");
  hipLaunchKernelGGL(confusion_kernel, dim3(ew_grid(npix, 512)), dim3(EW_BLOCK), 0, (hipStream_t)stream, classes, y_true, ncls, (long long)npix, (unsigned long long*)confusion);
  LAUNCH_OK("confusion");
  return SATCV_OK;
}

// -------------------------------------------------------------------------- Adam
// state = {lr, step_count, grad_scale, _}.  Block 0 / thread 0 bumps the step counter AFTER
// every block has read it?  No cross-block ordering exists, so the counter is advanced by a
// separate single-thread kernel launched after the update kernel on the same stream.
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long long n,
                            float b1, float b2, float eps, const float* __restrict__ state, const float* __restrict__ lr_mul) {
  const float lr = state[0], t = state[1] + 1.f, gs = state[2];
  const float alpha = lr * sqrtf(1.f - powf(b2, t)) / (1.f - powf(b1, t));
  const long long n4 = n / 4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float4 lm = lr_mul ? reinterpret_cast<const float4*>(lr_mul)[i] : make_float4(1.f, 1.f, 1.f, 1.f);
    float* P = &pp.x; const float* Gp = &gg.x; float* M = &mm.x; float* V = &vv.x; const float* L = &lm.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gr = Gp[e] * gs;
      const float mn = b1 * M[e] + (1.f - b1) * gr;
      const float vn = b2 * V[e] + (1.f - b2) * gr * gr;
      if (L[e] != 0.f) { M[e] = mn; V[e] = vn; P[e] -= L[e] * alpha * mn / (sqrtf(vn) + eps); }
    }
    reinterpret_cast<float4*>(p)[i] = pp; reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv;
  }
  // tail
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = n4 * 4 + threadIdx.x;
    const float l = lr_mul ? lr_mul[i] : 1.f;
    const float gr = g[i] * gs;
    const float mn = b1 * m[i] + (1.f - b1) * gr, vn = b2 * v[i] + (1.f - b2) * gr * gr;
    if (l != 0.f) { m[i] = mn; v[i] = vn; p[i] -= l * alpha * mn / (sqrtf(vn) + eps); }
  }
}
__global__ void adam_bump_kernel(float* state) { state[1] += 1.f; }
// the buffers a training step accumulates into (flat gradient, loss scalar), cleared by ONE launch of the library's own
__global__ __launch_bounds__(256) void zero2_kernel(uint4* __restrict__ a, long long na16, unsigned* __restrict__ b, long long nb4) {
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < na16; i += (long long)gridDim.x * blockDim.x) a[i] = z;
  if (blockIdx.x == 0) for (long long i = threadIdx.x; i < nb4; i += blockDim.x) b[i] = 0u;
}
extern "C" int satcv_zero2(void* a, int64_t bytes_a, void* b, int64_t bytes_b, void* stream) {
  SATCV_CHECK(bytes_a >= 0 && bytes_b >= 0 && (bytes_a == 0 || a) && (bytes_b == 0 || b), "zero2: bad args");
  SATCV_CHECK(((uintptr_t)a % 16 == 0) && bytes_a % 16 == 0 && ((uintptr_t)b % 4 == 0) && bytes_b % 4 == 0, "zero2: a must be 16-byte, b 4-byte aligned and sized");
  if (bytes_a == 0 && bytes_b == 0) return SATCV_OK;
  hipLaunchKernelGGL(zero2_kernel, dim3(ew_grid(bytes_a / 16 + 1, 4096)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<uint4*>(a), (long long)(bytes_a / 16),
                     reinterpret_cast<unsigned*>(b), (long long)(bytes_b / 4));
  LAUNCH_OK("zero2");
  return SATCV_OK;
}
extern "C" int satcv_adam_step_part(float* p, const float* g, float* m, float* v, int64_t n, float beta1, float beta2, float eps, float* state,
                                    const float* lr_mul, int32_t bump, void* stream) {
  SATCV_CHECK(p && g && m && v && state && n > 0, "adam: bad args");
  SATCV_CHECK(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0), "adam: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n / 4 + 1, 4096)), dim3(EW_BLOCK), 0, (hipStream_t)stream, p, g, m, v, (long long)n, beta1, beta2, eps, state, lr_mul);
  if (bump) hipLaunchKernelGGL(adam_bump_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state);
  LAUNCH_OK("adam");
  return SATCV_OK;
}
extern "C" int satcv_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float beta1, float beta2, float eps, float* state,
                               const float* lr_mul, void* stream) {
  SATCV_CHECK(p && g && m && v && state && n > 0, "adam: bad args");
  SATCV_CHECK(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0), "adam: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n / 4 + 1, 4096)), dim3(EW_BLOCK), 0, (hipStream_t)stream, p, g, m, v, (long long)n, beta1, beta2, eps, state, lr_mul);
  hipLaunchKernelGGL(adam_bump_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state);
  LAUNCH_OK("adam");
  return SATCV_OK;
}
