// ConvLSTM2D cell kernels and the small per-pixel dense heads of the LSTM model family (utils/model_tools.py:666-920, 1016-1060).
//
// A Keras ConvLSTM2D step is   z = conv(x_t, kernel) + bias + conv(h_{t-1}, recurrent_kernel)   followed by the gate arithmetic
//   i = ra(z_i), f = ra(z_f), g = act(z_c), o = ra(z_o);  c_t = f c_{t-1} + i g;  h_t = o act(c_t)        (gate order i, f, c, o)
// The two convolutions run on the implicit-GEMM kernels (the input convolution ONCE for all time steps: the sequence is stored time-major,
// (T, B, H, W, C), so "all steps" is a batch of T B images and a single step a contiguous slice of it); what is new here is
//   * satcv_ingest_seq            (B, T, H, W, C) float32 -> time-major storage type, channels padded
//   * satcv_convlstm_gates_fwd    z-slices -> c_t (float32), h_t (storage type), the post-activation gates kept for the backward pass,
//                                 BatchNorm statistics of the stored h_t (the reference normalises every ConvLSTM2D output)
//   * satcv_convlstm_gates_bwd    dh_t (up to two sources), dc_{t+1 -> t} -> dz_t (pre-activation gradient, the dy of both convolutions'
//                                 data / weight gradients), dc_{t -> t-1}
//   * satcv_dense_small_fwd/_bwd  Conv2D(k, 1x1) on one or two sources (concatenated), each optionally with a pending BatchNorm + ReLU and
//                                 a nearest-neighbour resize (tf.image.resize(..., 'nearest')), activation softmax / sigmoid / linear /
//                                 ReLU(max_value): the 1x1 `dense` layers and the three-way fusion head of get_hybrid_model.
#include "common.hpp"

#define EW_BLOCK 256
#define LSTM_OK(name)                                                                                  \
  do {                                                                                                 \
    hipError_t e__ = hipGetLastError();                                                                \
    if (e__ != hipSuccess) { satcv_set_error("%s launch: %s", name, hipGetErrorString(e__)); return SATCV_ERR_HIP; } \
  } while (0)
static inline int lstm_grid(long long items) {
  long long g = (items + EW_BLOCK - 1) / EW_BLOCK;
  if (g > 256 * 8) g = 256 * 8;
  return (int)(g < 1 ? 1 : g);
}

// ------------------------------------------------------------------------------------------------ ingest
template <typename T>
__global__ void ingest_seq_kernel(const float* __restrict__ src, T* __restrict__ dst, int B, int TT, long long hw, int c, int cpad) {
  const int groups = cpad / 8;
  const long long total = (long long)B * TT * hw * groups;
  for (long long it = blockIdx.x * (long long)blockDim.x + threadIdx.x; it < total; it += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(it % groups);
    long long p = it / groups;                         // destination pixel index: ((t * B + b) * hw + q)
    const long long q = p % hw; p /= hw;
    const int b = (int)(p % B), t = (int)(p / B);
    const float* s = src + (((long long)b * TT + t) * hw + q) * c;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { const int ch = g * 8 + e; v[e] = ch < c ? s[ch] : 0.f; }
    store8<T>(dst + (it / groups) * cpad + g * 8, v);
  }
}
extern "C" int satcv_ingest_seq(const float* src, void* dst, int32_t batch, int32_t steps, int32_t h, int32_t w_, int32_t c, int32_t cpad,
                                int32_t dtype, void* stream) {
  SATCV_CHECK(src && dst && batch > 0 && steps > 0 && h > 0 && w_ > 0 && c > 0 && cpad >= c && cpad % 8 == 0, "ingest_seq: bad args");
  const long long hw = (long long)h * w_, total = (long long)batch * steps * hw * (cpad / 8);
  if (dtype == SATCV_BF16) hipLaunchKernelGGL(ingest_seq_kernel<bf16>, dim3(lstm_grid(total)), dim3(EW_BLOCK), 0, (hipStream_t)stream, src, (bf16*)dst, batch, steps, hw, c, cpad);
  else if (dtype == SATCV_F32) hipLaunchKernelGGL(ingest_seq_kernel<float>, dim3(lstm_grid(total)), dim3(EW_BLOCK), 0, (hipStream_t)stream, src, (float*)dst, batch, steps, hw, c, cpad);
  else { satcv_set_error("ingest_seq: bad dtype"); return SATCV_ERR_INVALID; }
  LSTM_OK("ingest_seq");
  return SATCV_OK;
}

// ------------------------------------------------------------------------------------------------ gates
__device__ __forceinline__ float rec_act(float z, int kind) {
  return kind == 0 ? fminf(fmaxf(0.2f * z + 0.5f, 0.f), 1.f) : 1.f / (1.f + expf(-z));
}
__device__ __forceinline__ float rec_act_grad(float y, int kind) {      // from the VALUE: hard_sigmoid has slope 0.2 strictly inside (0, 1)
  return kind == 0 ? ((y > 0.f && y < 1.f) ? 0.2f : 0.f) : y * (1.f - y);
}

template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void convlstm_gates_fwd_kernel(const satcv_lstm_gates_desc d) {
  __shared__ float ssum[2][256];
  const int F = d.filters, G = F / 8;
  const long long total = d.npix * G;
  const T* xg = (const T*)d.xg; const T* hg = (const T*)d.hg;
  T* hout = (T*)d.h_out; T* gout = (T*)d.gates_out;
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  if (d.stats) { for (int i = threadIdx.x; i < 2 * 256; i += blockDim.x) (&ssum[0][0])[i] = 0.f; __syncthreads(); }
  // a thread keeps ONE channel group over its whole grid-stride walk (the stride is a multiple of G: EW_BLOCK % G == 0)
  const int g = threadIdx.x % G;
  for (long long it = blockIdx.x * (long long)blockDim.x + threadIdx.x; it < total; it += (long long)gridDim.x * blockDim.x) {
    const long long p = it / G;
    float z[4][8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      load8<T>(xg + p * d.ldx + k * F + g * 8, z[k]);
      if (hg) {
        float r[8];
        load8<T>(hg + p * d.ldh_g + k * F + g * 8, r);
#pragma unroll
        for (int e = 0; e < 8; ++e) z[k][e] += r[e];
      }
    }
    float cp[8], cn[8], hn[8], gi[8], gf[8], gg[8], go[8];
    if (d.c_prev) {
      const float4 a = *reinterpret_cast<const float4*>(d.c_prev + p * F + g * 8), b = *reinterpret_cast<const float4*>(d.c_prev + p * F + g * 8 + 4);
      cp[0] = a.x; cp[1] = a.y; cp[2] = a.z; cp[3] = a.w; cp[4] = b.x; cp[5] = b.y; cp[6] = b.z; cp[7] = b.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) cp[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      gi[e] = rec_act(z[0][e], d.rec_act); gf[e] = rec_act(z[1][e], d.rec_act); go[e] = rec_act(z[3][e], d.rec_act);
      gg[e] = d.act ? tanhf(z[2][e]) : z[2][e];
      cn[e] = gf[e] * cp[e] + gi[e] * gg[e];
      hn[e] = go[e] * (d.act ? tanhf(cn[e]) : cn[e]);
    }
    *reinterpret_cast<float4*>(d.c_out + p * F + g * 8) = make_float4(cn[0], cn[1], cn[2], cn[3]);
    *reinterpret_cast<float4*>(d.c_out + p * F + g * 8 + 4) = make_float4(cn[4], cn[5], cn[6], cn[7]);
    store8<T>(hout + p * d.ldh + g * 8, hn);
    if (gout) {
      store8<T>(gout + p * 4 * F + 0 * F + g * 8, gi); store8<T>(gout + p * 4 * F + 1 * F + g * 8, gf);
      store8<T>(gout + p * 4 * F + 2 * F + g * 8, gg); store8<T>(gout + p * 4 * F + 3 * F + g * 8, go);
    }
    if (d.stats) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float v = round_to<T>(hn[e]); s1[e] += v; s2[e] += v * v; }      // statistics of the STORED values
    }
  }
  if (d.stats) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { atomicAdd(&ssum[0][g * 8 + e], s1[e]); atomicAdd(&ssum[1][g * 8 + e], s2[e]); }
    __syncthreads();
    if ((int)threadIdx.x < F) {
      satcv_stat_t* row = d.stats + (size_t)(blockIdx.x % SATCV_STAT_ROWS) * 2 * d.stats_ld;
      atomicAdd(row + threadIdx.x, (satcv_stat_t)ssum[0][threadIdx.x]);
      atomicAdd(row + d.stats_ld + threadIdx.x, (satcv_stat_t)ssum[1][threadIdx.x]);
    }
  }
}

extern "C" int satcv_convlstm_gates_fwd(const satcv_lstm_gates_desc* d, void* stream) {
  SATCV_CHECK(d && d->xg && d->c_out && d->h_out && d->npix > 0, "lstm_gates_fwd: null pointer");
  SATCV_CHECK(d->filters >= 8 && d->filters <= 256 && d->filters % 8 == 0 && EW_BLOCK % (d->filters / 8) == 0, "lstm_gates_fwd: filters must be 8, 16, 32, 64, 128 or 256");
  SATCV_CHECK(d->ldx >= 4 * d->filters && d->ldx % 8 == 0 && d->ldh >= d->filters && d->ldh % 8 == 0 && (!d->hg || (d->ldh_g >= 4 * d->filters && d->ldh_g % 8 == 0)), "lstm_gates_fwd: bad leading dimensions");
  SATCV_CHECK(!d->stats || d->stats_ld >= d->filters, "lstm_gates_fwd: stats_ld");
  const int grid = lstm_grid(d->npix * (d->filters / 8));
  if (d->dtype == SATCV_BF16) hipLaunchKernelGGL(convlstm_gates_fwd_kernel<bf16>, dim3(grid), dim3(EW_BLOCK), 0, (hipStream_t)stream, *d);
  else if (d->dtype == SATCV_F32) hipLaunchKernelGGL(convlstm_gates_fwd_kernel<float>, dim3(grid), dim3(EW_BLOCK), 0, (hipStream_t)stream, *d);
  else { satcv_set_error("lstm_gates_fwd: bad dtype"); return SATCV_ERR_INVALID; }
  LSTM_OK("convlstm_gates_fwd");
  return SATCV_OK;
}

template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void convlstm_gates_bwd_kernel(const satcv_lstm_gates_desc d) {
  const int F = d.filters, G = F / 8;
  const long long total = d.npix * G;
  const T* dha = (const T*)d.dh_a; const T* dhb = (const T*)d.dh_b; const T* gates = (const T*)d.gates_out;
  T* dz = (T*)d.dz_out;
  for (long long it = blockIdx.x * (long long)blockDim.x + threadIdx.x; it < total; it += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(it % G);
    const long long p = it / G;
    float dh[8], tmp[8], gi[8], gf[8], gg[8], go[8], cp[8], cc[8], dcn[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { dh[e] = 0.f; cp[e] = 0.f; dcn[e] = 0.f; }
    if (dha) { load8<T>(dha + p * d.lddh_a + g * 8, tmp);
#pragma unroll
      for (int e = 0; e < 8; ++e) dh[e] += tmp[e]; }
    if (dhb) { load8<T>(dhb + p * d.lddh_b + g * 8, tmp);
#pragma unroll
      for (int e = 0; e < 8; ++e) dh[e] += tmp[e]; }
    load8<T>(gates + p * 4 * F + 0 * F + g * 8, gi); load8<T>(gates + p * 4 * F + 1 * F + g * 8, gf);
    load8<T>(gates + p * 4 * F + 2 * F + g * 8, gg); load8<T>(gates + p * 4 * F + 3 * F + g * 8, go);
    if (d.c_prev) { load8<float>(d.c_prev + p * F + g * 8, cp); }
    load8<float>(d.c_out + p * F + g * 8, cc);
    if (d.dc_next) { load8<float>(d.dc_next + p * F + g * 8, dcn); }
    float di[8], df[8], dg[8], dO[8], dcp[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float ac = d.act ? tanhf(cc[e]) : cc[e];
      const float dc = dcn[e] + dh[e] * go[e] * (d.act ? 1.f - ac * ac : 1.f);
      dO[e] = dh[e] * ac * rec_act_grad(go[e], d.rec_act);
      di[e] = dc * gg[e] * rec_act_grad(gi[e], d.rec_act);
      df[e] = dc * cp[e] * rec_act_grad(gf[e], d.rec_act);
      dg[e] = dc * gi[e] * (d.act ? 1.f - gg[e] * gg[e] : 1.f);
      dcp[e] = dc * gf[e];
    }
    store8<T>(dz + p * d.lddz + 0 * F + g * 8, di); store8<T>(dz + p * d.lddz + 1 * F + g * 8, df);
    store8<T>(dz + p * d.lddz + 2 * F + g * 8, dg); store8<T>(dz + p * d.lddz + 3 * F + g * 8, dO);
    store8<float>(d.dc_prev_out + p * F + g * 8, dcp);
  }
}

extern "C" int satcv_convlstm_gates_bwd(const satcv_lstm_gates_desc* d, void* stream) {
  SATCV_CHECK(d && d->gates_out && d->c_out && d->dz_out && d->dc_prev_out && (d->dh_a || d->dh_b) && d->npix > 0, "lstm_gates_bwd: null pointer");
  SATCV_CHECK(d->filters >= 8 && d->filters % 8 == 0 && d->lddz >= 4 * d->filters && d->lddz % 8 == 0, "lstm_gates_bwd: bad filters / lddz");
  SATCV_CHECK((!d->dh_a || d->lddh_a % 8 == 0) && (!d->dh_b || d->lddh_b % 8 == 0), "lstm_gates_bwd: bad dh leading dimensions");
  const int grid = lstm_grid(d->npix * (d->filters / 8));
  if (d->dtype == SATCV_BF16) hipLaunchKernelGGL(convlstm_gates_bwd_kernel<bf16>, dim3(grid), dim3(EW_BLOCK), 0, (hipStream_t)stream, *d);
  else if (d->dtype == SATCV_F32) hipLaunchKernelGGL(convlstm_gates_bwd_kernel<float>, dim3(grid), dim3(EW_BLOCK), 0, (hipStream_t)stream, *d);
  else { satcv_set_error("lstm_gates_bwd: bad dtype"); return SATCV_ERR_INVALID; }
  LSTM_OK("convlstm_gates_bwd");
  return SATCV_OK;
}

// ------------------------------------------------------------------------------------------------ small dense heads
// nearest-neighbour source index of destination index i (tf.image.resize 'nearest', half-pixel centres): min(floor((i + 0.5) in / out), in - 1)
__device__ __forceinline__ int nn_src(int i, int in, int out) {
  const int s = (int)floorf(((float)i + 0.5f) * ((float)in / (float)out));
  return s < in - 1 ? s : in - 1;
}
// activated value of element (pixel q, channel c) of a source: storage type (dtype 0 f32 / 1 bf16), optional pending BatchNorm + ReLU
__device__ __forceinline__ float dense_src_val(const satcv_dense_src& s, long long q, int c) {
  float v = s.dtype == SATCV_BF16 ? (float)reinterpret_cast<const bf16*>(s.x)[q * s.ld + c] : reinterpret_cast<const float*>(s.x)[q * s.ld + c];
  if (s.in_scale) { v = v * s.in_scale[c] + s.in_shift[c]; if (s.in_relu) v = fmaxf(v, 0.f); }
  return v;
}
// source pixel of output pixel p (n, y, x on the output grid h x w)
__device__ __forceinline__ long long dense_src_pix(const satcv_dense_src& s, long long p, int h, int w) {
  if (s.hs == 0) return p;
  const int x = (int)(p % w); long long r = p / w;
  const int y = (int)(r % h); const long long n = r / h;
  return (n * s.hs + nn_src(y, s.hs, h)) * s.ws + nn_src(x, s.ws, w);
}

#define DENSE_KMAX 16
__global__ __launch_bounds__(EW_BLOCK) void dense_small_fwd_kernel(const satcv_dense_desc d) {
  const int k = d.cout;
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < d.npix; p += (long long)gridDim.x * blockDim.x) {
    float z[DENSE_KMAX];
#pragma unroll
    for (int j = 0; j < DENSE_KMAX; ++j) z[j] = j < k ? d.b[j] : 0.f;
    int row = 0;
    for (int si = 0; si < d.nsrc; ++si) {
      const satcv_dense_src& s = d.src[si];
      const long long q = dense_src_pix(s, p, d.h, d.w_);
      for (int c = 0; c < s.cin; ++c) {
        const float a = dense_src_val(s, q, c);
#pragma unroll
        for (int j = 0; j < DENSE_KMAX; ++j) if (j < k) z[j] += a * d.w[(row + c) * k + j];
      }
      row += s.cin;
    }
    if (d.z_out) { for (int j = 0; j < k; ++j) d.z_out[p * k + j] = z[j]; }
    if (d.activation == 0) {
      float mx = z[0]; for (int j = 1; j < k; ++j) mx = fmaxf(mx, z[j]);
      float sum = 0.f, ex[DENSE_KMAX];
      for (int j = 0; j < k; ++j) { ex[j] = expf(z[j] - mx); sum += ex[j]; }
      float best = -1.f; int am = 0;
      for (int j = 0; j < k; ++j) { const float pr = ex[j] / sum; d.out[p * k + j] = pr; if (pr > best) { best = pr; am = j; } }
      if (d.classes) d.classes[p] = am;
    } else if (d.activation == 1) {
      for (int j = 0; j < k; ++j) d.out[p * k + j] = 1.f / (1.f + expf(-z[j]));
    } else if (d.activation == 2) {
      for (int j = 0; j < k; ++j) d.out[p * k + j] = z[j];
    } else {
      for (int j = 0; j < k; ++j) { float v = fmaxf(z[j], 0.f); if (d.max_value > 0.f) v = fminf(v, d.max_value); d.out[p * k + j] = v; }
    }
  }
}
extern "C" int satcv_dense_small_fwd(const satcv_dense_desc* d, void* stream) {
  SATCV_CHECK(d && d->w && d->b && d->out && d->npix > 0 && d->nsrc >= 1 && d->nsrc <= 2 && d->cout >= 1 && d->cout <= DENSE_KMAX, "dense_small_fwd: bad args");
  for (int i = 0; i < d->nsrc; ++i)
    SATCV_CHECK(d->src[i].x && d->src[i].cin > 0 && d->src[i].ld >= d->src[i].cin && (d->src[i].dtype == SATCV_F32 || d->src[i].dtype == SATCV_BF16) &&
                ((d->src[i].hs == 0) == (d->src[i].ws == 0)), "dense_small_fwd: bad source %d", i);
  SATCV_CHECK(d->h > 0 && d->w_ > 0 && d->npix % ((long long)d->h * d->w_) == 0, "dense_small_fwd: npix is not a whole number of h x w images");
  hipLaunchKernelGGL(dense_small_fwd_kernel, dim3(lstm_grid(d->npix)), dim3(EW_BLOCK), 0, (hipStream_t)stream, *d);
  LSTM_OK("dense_small_fwd");
  return SATCV_OK;
}

// stage 1: dz = dout * act'(z) (ReLU(max) / linear; softmax and sigmoid heads receive dlogits from the loss kernel: pass activation 2),
// dW and db, dx of the sources at output resolution.  stage 2 (resized sources): every SOURCE pixel gathers over its pre-image (deterministic).
//
// Round 5: ONE THREAD PER INPUT ROW.  Thread (pixel lane, row c) walks the lane's pixels, keeps dW[c][0 .. k) in registers and writes dx[p][c]; the
// extra row `rows` of a lane carries db (and writes dz_out).  Lanes are summed through LDS once per block, blocks by float atomics (order-dependent
// in the last bits, as before: these layers have a few hundred parameters).  The first form added every pixel's products to ONE LDS image with
// atomics -- 256 threads on the same addresses: 0.31 ms per launch, half of a get_lstm_model training step (profiles/r05_lstm_kernel_stats.csv).
__global__ __launch_bounds__(EW_BLOCK) void dense_small_bwd_kernel(const satcv_dense_desc d, const int rows, const int pl) {
  extern __shared__ float red[];                  // [pl][rows + 1][k]
  const int k = d.cout, rp = rows + 1;
  const int lanep = threadIdx.x / rp, c = threadIdx.x - lanep * rp;
  const bool live = lanep < pl;
  const int si = (c < rows && c >= d.src[0].cin) ? 1 : 0;
  const satcv_dense_src& s = d.src[si];
  const int cc = c - (si ? d.src[0].cin : 0);
  float acc[DENSE_KMAX], wr[DENSE_KMAX];
#pragma unroll
  for (int j = 0; j < DENSE_KMAX; ++j) { acc[j] = 0.f; wr[j] = (j < k && c < rows) ? d.w[c * k + j] : 0.f; }
  if (live) {
    for (long long p = (long long)blockIdx.x * pl + lanep; p < d.npix; p += (long long)gridDim.x * pl) {
      float dz[DENSE_KMAX];
#pragma unroll
      for (int j = 0; j < DENSE_KMAX; ++j) {
        float g = 0.f;
        if (j < k) {
          g = d.dout[p * k + j];
          if (d.activation == 3) { const float o = d.out[p * k + j]; g = (o > 0.f && (d.max_value <= 0.f || o < d.max_value)) ? g : 0.f; }
        }
        dz[j] = g;
      }
      if (c == rows) {
#pragma unroll
        for (int j = 0; j < DENSE_KMAX; ++j) if (j < k) { acc[j] += dz[j]; if (d.dz_out) d.dz_out[p * k + j] = dz[j]; }
      } else {
        const long long q = dense_src_pix(s, p, d.h, d.w_);
        const float a = dense_src_val(s, q, cc);
        float dx = 0.f;
#pragma unroll
        for (int j = 0; j < DENSE_KMAX; ++j) { acc[j] += a * dz[j]; dx += dz[j] * wr[j]; }
        if (s.dx && s.hs == 0) {
          if (s.dx_dtype == SATCV_BF16) reinterpret_cast<bf16*>(s.dx)[q * s.lddx + cc] = (bf16)dx;
          else reinterpret_cast<float*>(s.dx)[q * s.lddx + cc] = dx;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < DENSE_KMAX; ++j) if (j < k) red[(lanep * rp + c) * k + j] = acc[j];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < rp * k; i += blockDim.x) {
    float t = 0.f;
    for (int l = 0; l < pl; ++l) t += red[l * rp * k + i];
    if (i < rows * k) atomicAdd(d.dw + i, t);
    else atomicAdd(d.db + (i - rows * k), t);
  }
}
// (more than 255 input rows: every pixel's products added to one LDS image with atomics)
__global__ __launch_bounds__(EW_BLOCK) void dense_small_bwd_wide_kernel(const satcv_dense_desc d) {
  extern __shared__ float acc[];                  // [rows + 1][k]: dW rows then db
  const int k = d.cout;
  int rows = 0;
  for (int si = 0; si < d.nsrc; ++si) rows += d.src[si].cin;
  for (int i = threadIdx.x; i < (rows + 1) * k; i += blockDim.x) acc[i] = 0.f;
  __syncthreads();
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < d.npix; p += (long long)gridDim.x * blockDim.x) {
    float dz[DENSE_KMAX];
#pragma unroll
    for (int j = 0; j < DENSE_KMAX; ++j) dz[j] = 0.f;
    for (int j = 0; j < k; ++j) {
      float g = d.dout[p * k + j];
      if (d.activation == 3) { const float o = d.out[p * k + j]; g = (o > 0.f && (d.max_value <= 0.f || o < d.max_value)) ? g : 0.f; }
      dz[j] = g;
      if (d.dz_out) d.dz_out[p * k + j] = g;
      atomicAdd(&acc[rows * k + j], g);
    }
    int row = 0;
    for (int si = 0; si < d.nsrc; ++si) {
      const satcv_dense_src& s = d.src[si];
      const long long q = dense_src_pix(s, p, d.h, d.w_);
      for (int c = 0; c < s.cin; ++c) {
        const float a = dense_src_val(s, q, c);
        float dx = 0.f;
#pragma unroll
        for (int j = 0; j < DENSE_KMAX; ++j) if (j < k) { atomicAdd(&acc[(row + c) * k + j], a * dz[j]); dx += dz[j] * d.w[(row + c) * k + j]; }
        if (s.dx && s.hs == 0) {
          if (s.dx_dtype == SATCV_BF16) reinterpret_cast<bf16*>(s.dx)[q * s.lddx + c] = (bf16)dx;
          else reinterpret_cast<float*>(s.dx)[q * s.lddx + c] = dx;
        }
      }
      row += s.cin;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < rows * k; i += blockDim.x) atomicAdd(d.dw + i, acc[i]);
  for (int i = threadIdx.x; i < k; i += blockDim.x) atomicAdd(d.db + i, acc[rows * k + i]);
}
__global__ __launch_bounds__(EW_BLOCK) void dense_small_bwd_gather_kernel(const satcv_dense_desc d, int si, int row0) {
  const satcv_dense_src s = d.src[si];
  const int k = d.cout;
  const long long nimg = d.npix / ((long long)d.h * d.w_);
  const long long total = nimg * s.hs * s.ws;
  for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < total; q += (long long)gridDim.x * blockDim.x) {
    const int sx = (int)(q % s.ws); long long r = q / s.ws;
    const int sy = (int)(r % s.hs); const long long n = r / s.hs;
    // pre-image of (sy, sx): a contiguous block of output rows / columns (nn_src is monotone)
    int y0 = (int)floorf((float)sy * d.h / s.hs) - 1; if (y0 < 0) y0 = 0;
    int x0 = (int)floorf((float)sx * d.w_ / s.ws) - 1; if (x0 < 0) x0 = 0;
    float dzs[DENSE_KMAX];
#pragma unroll
    for (int j = 0; j < DENSE_KMAX; ++j) dzs[j] = 0.f;
    for (int y = y0; y < d.h; ++y) {
      const int yy = nn_src(y, s.hs, d.h);
      if (yy < sy) continue;
      if (yy > sy) break;
      for (int x = x0; x < d.w_; ++x) {
        const int xx = nn_src(x, s.ws, d.w_);
        if (xx < sx) continue;
        if (xx > sx) break;
        const long long p = (n * d.h + y) * d.w_ + x;
        for (int j = 0; j < k; ++j) dzs[j] += d.dz_out[p * k + j];
      }
    }
    for (int c = 0; c < s.cin; ++c) {
      float dx = 0.f;
      for (int j = 0; j < k; ++j) dx += dzs[j] * d.w[(row0 + c) * k + j];
      if (s.dx_dtype == SATCV_BF16) reinterpret_cast<bf16*>(s.dx)[q * s.lddx + c] = (bf16)dx;
      else reinterpret_cast<float*>(s.dx)[q * s.lddx + c] = dx;
    }
  }
}
extern "C" int satcv_dense_small_bwd(const satcv_dense_desc* d, void* stream) {
  SATCV_CHECK(d && d->w && d->dout && d->dw && d->db && d->npix > 0 && d->nsrc >= 1 && d->nsrc <= 2 && d->cout >= 1 && d->cout <= DENSE_KMAX, "dense_small_bwd: bad args");
  SATCV_CHECK(d->activation == 2 || (d->activation == 3 && d->out), "dense_small_bwd: activation must be linear (dlogits given) or ReLU (with the forward output)");
  int rows = 0;
  bool resized = false;
  for (int i = 0; i < d->nsrc; ++i) {
    rows += d->src[i].cin;
    if (d->src[i].hs && d->src[i].dx) resized = true;
  }
  SATCV_CHECK(!resized || d->dz_out, "dense_small_bwd: a resized source with a data gradient needs dz_out");
  SATCV_CHECK((size_t)(rows + 1) * d->cout * sizeof(float) <= 48 * 1024, "dense_small_bwd: too many input channels");
  hipStream_t st = (hipStream_t)stream;
  if (rows + 1 <= EW_BLOCK) {
    const int pl = EW_BLOCK / (rows + 1);
    long long grid = (d->npix + (long long)pl * 16 - 1) / ((long long)pl * 16);     // ~16 pixels per lane and block
    if (grid > 1024) grid = 1024;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(dense_small_bwd_kernel, dim3((unsigned)grid), dim3(EW_BLOCK), (size_t)pl * (rows + 1) * d->cout * sizeof(float), st, *d, rows, pl);
  } else {
    hipLaunchKernelGGL(dense_small_bwd_wide_kernel, dim3(lstm_grid(d->npix)), dim3(EW_BLOCK), (size_t)(rows + 1) * d->cout * sizeof(float), st, *d);
  }
  LSTM_OK("dense_small_bwd");
  int row0 = 0;
  for (int i = 0; i < d->nsrc; ++i) {
    if (d->src[i].hs && d->src[i].dx) {
      const long long total = d->npix / ((long long)d->h * d->w_) * d->src[i].hs * d->src[i].ws;
      hipLaunchKernelGGL(dense_small_bwd_gather_kernel, dim3(lstm_grid(total)), dim3(EW_BLOCK), 0, st, *d, i, row0);
      LSTM_OK("dense_small_bwd_gather");
    }
    row0 += d->src[i].cin;
  }
  return SATCV_OK;
}
