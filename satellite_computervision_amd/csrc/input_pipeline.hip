// Device-side tile input pipeline: what utils/processing.py:544-755 (UNETDataGenerator) and
// utils/array_tools.py:26-44, 159-213 do per batch on the host with NumPy, as HBM-bound HIP kernels that write the model's
// NHWC fp32 staging tensors directly.
//   planes (n, c, hin, win) of the on-disk dtype -> x / rescale_val (float64 like NumPy) -> [NaN / < -5000 mask channel,
//   flagged values replaced by N(0,1)] -> centre trim -> [colour augmentation about the per-image channel mean] ->
//   flip / flip / rot90 -> NHWC fp32 at a channel offset of the batch tensor;
//   labels (n, 1, hin, win) -> merge_classes look-up tables -> trim -> flip / flip / rot90 -> one-hot fp32.
// One thread per output pixel (all channels), so the NHWC stores are contiguous per thread and coalesced across x when the
// tile is not rotated; plane reads are coalesced across x.
#include "common.hpp"

namespace {

template <typename S>
__device__ __forceinline__ double ld(const void* p, size_t i) { return (double)reinterpret_cast<const S*>(p)[i]; }

__device__ __forceinline__ double load_src(const void* p, int kind, size_t i) {
  switch (kind) {
    case 0: return ld<uint8_t>(p, i);
    case 1: return ld<uint16_t>(p, i);
    case 2: return ld<float>(p, i);
    case 3: return ld<int16_t>(p, i);
    case 4: return ld<double>(p, i);
    case 5: return ld<int32_t>(p, i);
    default: return ld<long long>(p, i);
  }
}

// inverse of aug_array_morph (utils/array_tools.py:188-213): output pixel (yo, xo) of flip_v -> flip_h -> rot90(k) <- (y, x)
__device__ __forceinline__ void morph_src(int yo, int xo, int h, int w, int fv, int fh, int rot, int& y, int& x) {
  int a, b;                                   // coordinates in the flipped (pre-rotation) image m
  switch (rot & 3) {
    case 0: a = yo; b = xo; break;
    case 1: a = xo; b = w - 1 - yo; break;    // np.rot90(m)[i][j] = m[j][W-1-i]
    case 2: a = h - 1 - yo; b = w - 1 - xo; break;
    default: a = h - 1 - xo; b = yo; break;   // rot90(m, 3)[i][j] = m[H-1-j][i]
  }
  y = fv ? h - 1 - a : a;
  x = fh ? w - 1 - b : b;
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ double normal_draw(uint64_t seed, uint64_t idx) {
  const uint64_t r = splitmix64(seed ^ (idx * 0xD1342543DE82EF95ull));
  const double u1 = ((double)(uint32_t)(r >> 32) + 1.0) / 4294967297.0;
  const double u2 = (double)(uint32_t)r / 4294967296.0;
  return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

__device__ __forceinline__ double uniform_draw(uint64_t seed, uint64_t idx) {
  return (double)(splitmix64(seed ^ (idx * 0xD1342543DE82EF95ull)) >> 11) * (1.0 / 9007199254740992.0);      // [0, 1)
}

// nanmean over the trimmed window of x / rescale, one workgroup per (image, channel)
__global__ void tile_mean_kernel(const satcv_tile_desc d, double* __restrict__ out) {
  const int b = blockIdx.x / d.c, ch = blockIdx.x % d.c;
  const int ty = (d.hin - d.h) / 2, tx = (d.win - d.w_) / 2;
  const size_t plane = ((size_t)b * d.c + ch) * d.hin * d.win;
  double s = 0.0; long long cnt = 0;
  for (int i = threadIdx.x; i < d.h * d.w_; i += blockDim.x) {
    const int y = i / d.w_, x = i % d.w_;
    double v = load_src(d.src, d.src_kind, plane + (size_t)(y + ty) * d.win + x + tx);
    if (d.rescale != 0.0) v = d.src_kind == 2 ? (double)((float)v / (float)d.rescale) : v / d.rescale;
    // mask mode 2 (SiameseDataGenerator): NaNs were already replaced by U[0,1) draws when the colour augmentation takes its mean
    if (d.nan_mask == 2 && v != v) v = uniform_draw(d.seed, plane + (size_t)(y + ty) * d.win + x + tx);
    if (v == v) { s += v; ++cnt; }
  }
  __shared__ double ss[256]; __shared__ long long sc[256];
  ss[threadIdx.x] = s; sc[threadIdx.x] = cnt;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { ss[threadIdx.x] += ss[threadIdx.x + o]; sc[threadIdx.x] += sc[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = ss[0] / (double)sc[0];
}

// NumPy evaluates every operation separately (no fused multiply-add) and keeps float32 planes in float32 (integer planes
// become float64 when divided by the python float): both are reproduced so that the unaugmented paths are bit-exact.
#pragma clang fp contract(off)
__global__ void tile_ingest_kernel(const satcv_tile_desc d) {
  const bool f32 = d.src_kind == 2;
  const int ho = (d.rot & 1) ? d.w_ : d.h, wo = (d.rot & 1) ? d.h : d.w_;
  const long long total = (long long)d.n * ho * wo;
  const int ty = (d.hin - d.h) / 2, tx = (d.win - d.w_) / 2;
  for (long long it = blockIdx.x * (long long)blockDim.x + threadIdx.x; it < total; it += (long long)gridDim.x * blockDim.x) {
    const int xo = (int)(it % wo);
    const int yo = (int)((it / wo) % ho);
    const int b = (int)(it / ((long long)wo * ho));
    int y, x;
    morph_src(yo, xo, d.h, d.w_, d.flip_v, d.flip_h, d.rot, y, x);
    const size_t pix = (size_t)(y + ty) * d.win + x + tx;
    float* o = d.dst + (size_t)it * d.ldc + d.coff;
    bool flagged = false;
    for (int ch = 0; ch < d.c; ++ch) {
      const size_t plane = ((size_t)b * d.c + ch) * d.hin * d.win;
      double v = load_src(d.src, d.src_kind, plane + pix);
      if (d.rescale != 0.0) v = f32 ? (double)((float)v / (float)d.rescale) : v / d.rescale;
      if (d.nan_mask == 2) {
        // SiameseDataGenerator._get_unet_data (utils/processing.py:797-805): a pixel is valid when no band is NaN or below -1 (after the
        // rescale); only the NaN elements themselves are replaced, by U[0,1) draws
        const bool isn = v != v;
        flagged = flagged || isn || (v < -1.0);
        if (isn) v = uniform_draw(d.seed, plane + pix);
      } else if (d.nan_mask) {
        // the reference's mask ACCUMULATES over the channel order (utils/processing.py:561-566)
        flagged = flagged || (v != v) || (v < -5000.0);
        if (flagged && d.replace) v = normal_draw(d.seed, plane + pix);
      }
      if (d.ch_mean) {
        const double mu = d.ch_mean[b * d.c + ch];
        if (f32) {
          const float m = (float)mu, t1 = ((float)v - m) * (float)d.contra_mul, t2 = m * (float)d.bright_mul;
          v = (double)(t1 + t2);
        } else {
          const double t1 = (v - mu) * d.contra_mul, t2 = mu * d.bright_mul;
          v = t1 + t2;
        }
      }
      o[ch] = (float)v;
    }
    if (d.nan_mask == 2) o[d.c] = flagged ? 0.f : 1.f;                 // 1 = every band valid
    else if (d.nan_mask) o[d.c] = (flagged && d.replace) ? 1.f : 0.f;
  }
}

struct LabelArgs {
  const void* lc; int lc_kind; const int* lut; const void* lu; int lu_kind; const int* lu_lut;
  int n, hin, win, h, w, ncls, fv, fh, rot; float* dst; int ldc, coff;
};
__global__ void label_onehot_kernel(const LabelArgs a) {
  const int ho = (a.rot & 1) ? a.w : a.h, wo = (a.rot & 1) ? a.h : a.w;
  const long long total = (long long)a.n * ho * wo;
  const int ty = (a.hin - a.h) / 2, tx = (a.win - a.w) / 2;
  for (long long it = blockIdx.x * (long long)blockDim.x + threadIdx.x; it < total; it += (long long)gridDim.x * blockDim.x) {
    const int xo = (int)(it % wo);
    const int yo = (int)((it / wo) % ho);
    const int b = (int)(it / ((long long)wo * ho));
    int y, x;
    morph_src(yo, xo, a.h, a.w, a.fv, a.fh, a.rot, y, x);
    const size_t idx = (size_t)b * a.hin * a.win + (size_t)(y + ty) * a.win + x + tx;
    const long long v = (long long)load_src(a.lc, a.lc_kind, idx);          // .astype(int): truncation
    long long cls = v;
    if (a.lut && v >= 0 && v < 256 && a.lut[v] >= 0) cls = a.lut[v];
    if (a.lu) {
      const double u = load_src(a.lu, a.lu_kind, idx);
      const long long ui = (long long)u;
      if ((double)ui == u && ui >= 0 && ui < 256 && a.lu_lut[ui] >= 0) cls = a.lu_lut[ui];
    }
    float* o = a.dst + (size_t)it * a.ldc + a.coff;
    for (int k = 0; k < a.ncls; ++k) o[k] = (cls == k) ? 1.f : 0.f;          // tf.one_hot: out-of-range -> all zeros
  }
}

int grid_for(long long items) {
  long long g = (items + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 65536 ? 65536 : g));
}

int check_desc(const satcv_tile_desc* d, const char* who) {
  SATCV_CHECK(d && d->src && d->n > 0 && d->c > 0 && d->h > 0 && d->w_ > 0 && d->hin >= d->h && d->win >= d->w_, "%s: bad dims", who);
  SATCV_CHECK(d->src_kind >= 0 && d->src_kind <= 6, "%s: src_kind %d", who, d->src_kind);
  SATCV_CHECK(d->rot >= 0 && d->rot <= 3, "%s: rot must be 0..3", who);
  return SATCV_OK;
}

}  // namespace

extern "C" int satcv_tile_channel_mean(const satcv_tile_desc* d, double* mean_out, void* stream) {
  int rc = check_desc(d, "tile_channel_mean");
  if (rc) return rc;
  SATCV_CHECK(mean_out, "tile_channel_mean: null output");
  hipLaunchKernelGGL(tile_mean_kernel, dim3(d->n * d->c), dim3(256), 0, (hipStream_t)stream, *d, mean_out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { satcv_set_error("tile_channel_mean launch: %s", hipGetErrorString(e)); return SATCV_ERR_HIP; }
  return SATCV_OK;
}

extern "C" int satcv_tile_ingest(const satcv_tile_desc* d, void* stream) {
  int rc = check_desc(d, "tile_ingest");
  if (rc) return rc;
  SATCV_CHECK(d->dst && d->ldc >= d->coff + d->c + (d->nan_mask ? 1 : 0), "tile_ingest: destination channel range");
  SATCV_CHECK(!(d->nan_mask == 1 && d->ch_mean), "tile_ingest: UNETDataGenerator applies the colour augmentation only to unmasked sources");
  SATCV_CHECK(d->nan_mask >= 0 && d->nan_mask <= 2, "tile_ingest: nan_mask must be 0, 1 (UNETDataGenerator) or 2 (SiameseDataGenerator)");
  hipLaunchKernelGGL(tile_ingest_kernel, dim3(grid_for((long long)d->n * d->h * d->w_)), dim3(256), 0, (hipStream_t)stream, *d);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { satcv_set_error("tile_ingest launch: %s", hipGetErrorString(e)); return SATCV_ERR_HIP; }
  return SATCV_OK;
}

extern "C" int satcv_label_onehot(const void* lc, int32_t lc_kind, const int32_t* lut, const void* lu, int32_t lu_kind, const int32_t* lu_lut,
                                  int32_t n, int32_t hin, int32_t win, int32_t h, int32_t w_, int32_t nclasses, int32_t flip_v, int32_t flip_h,
                                  int32_t rot, float* dst, int32_t ldc, int32_t coff, void* stream) {
  SATCV_CHECK(lc && dst && n > 0 && h > 0 && w_ > 0 && hin >= h && win >= w_ && nclasses > 0 && ldc >= coff + nclasses, "label_onehot: bad args");
  SATCV_CHECK(lc_kind >= 0 && lc_kind <= 6 && (!lu || (lu_kind >= 0 && lu_kind <= 6 && lu_lut)) && rot >= 0 && rot <= 3, "label_onehot: bad kinds");
  LabelArgs a{lc, lc_kind, lut, lu, lu_kind, lu_lut, n, hin, win, h, w_, nclasses, flip_v, flip_h, rot, dst, ldc, coff};
  hipLaunchKernelGGL(label_onehot_kernel, dim3(grid_for((long long)n * h * w_)), dim3(256), 0, (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { satcv_set_error("label_onehot launch: %s", hipGetErrorString(e)); return SATCV_ERR_HIP; }
  return SATCV_OK;
}
