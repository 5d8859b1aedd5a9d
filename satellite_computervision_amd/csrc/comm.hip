// Collectives of the data-parallel step behind the C ABI (`allreduce_grads` + communicator init / destroy).
//
// One process per GPU; a communicator is created on the CURRENT device from a 128-byte RCCL unique id that rank 0 generates and the
// caller distributes (any side channel: torch.distributed's store, MPI, a file).  RCCL is bound at run time (dlopen), so the library
// loads on hosts without it and, inside a PyTorch process, shares the RCCL that process has already loaded instead of mapping a second
// copy.  Every call is asynchronous on the stream it is given; nothing here allocates device memory (the bf16 payload's staging
// buffer is the caller's).
#include "common.hpp"
#include <dlfcn.h>
#include <cstdlib>
#include <cstring>
#include <mutex>
// types and enums only: the entry points below are resolved with dlsym.  A ROCm install without the RCCL development headers still
// builds the library: the handful of ABI-stable declarations this file needs are restated (values as in nccl.h 2.x / rccl.h).
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7,
               ncclFloat64 = 8, ncclBfloat16 = 9 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3, ncclAvg = 4 } ncclRedOp_t;
}
#endif

namespace {
struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::once_flag g_rccl_once;
char g_rccl_err[256] = "";

void rccl_load() {
  const char* env = getenv("SATCV_RCCL_LIB");
  const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    if (!n || !*n) continue;
    g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.handle) break;
  }
  if (!g_rccl.handle) { snprintf(g_rccl_err, sizeof(g_rccl_err), "librccl not found (%s)", dlerror()); return; }
#define SYM(field, name)                                                                         \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.handle, name));           \
  if (!g_rccl.field) { snprintf(g_rccl_err, sizeof(g_rccl_err), "librccl: missing %s", name); g_rccl.handle = nullptr; return; }
  SYM(GetUniqueId, "ncclGetUniqueId")
  SYM(CommInitRank, "ncclCommInitRank")
  SYM(CommDestroy, "ncclCommDestroy")
  SYM(AllReduce, "ncclAllReduce")
  SYM(GroupStart, "ncclGroupStart")
  SYM(GroupEnd, "ncclGroupEnd")
  SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
}
int rccl_ready() {
  std::call_once(g_rccl_once, rccl_load);
  if (!g_rccl.handle) { satcv_set_error("comm: %s", g_rccl_err); return SATCV_ERR_UNSUPPORTED; }
  return SATCV_OK;
}
#define SATCV_NCCL(call)                                                                        \
  do {                                                                                          \
    ncclResult_t r__ = (call);                                                                  \
    if (r__ != ncclSuccess) { satcv_set_error("%s failed: %s", #call, g_rccl.GetErrorString(r__)); return SATCV_ERR_HIP; } \
  } while (0)
}  // namespace

struct satcv_comm { ncclComm_t comm; int rank, world, device; };

// fp32 <-> bf16 payload conversion (the gradient is summed in bf16 on the wire when the caller asks for the 37 MB payload)
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, bf16* __restrict__ dst, long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x * 4;
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      const float4 v = *reinterpret_cast<const float4*>(src + i);
      bf16x4 o; o[0] = (bf16)v.x; o[1] = (bf16)v.y; o[2] = (bf16)v.z; o[3] = (bf16)v.w;
      *reinterpret_cast<bf16x4*>(dst + i) = o;
    } else {
      for (long long k = i; k < n; ++k) dst[k] = (bf16)src[k];
    }
  }
}
__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const bf16* __restrict__ src, float* __restrict__ dst, long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x * 4;
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      const bf16x4 v = *reinterpret_cast<const bf16x4*>(src + i);
      *reinterpret_cast<float4*>(dst + i) = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    } else {
      for (long long k = i; k < n; ++k) dst[k] = (float)src[k];
    }
  }
}

extern "C" int satcv_comm_unique_id(void* id128) {
  SATCV_CHECK(id128, "comm_unique_id: null");
  int rc = rccl_ready(); if (rc) return rc;
  static_assert(sizeof(ncclUniqueId) == SATCV_COMM_ID_BYTES, "unique id size");
  ncclUniqueId id;
  SATCV_NCCL(g_rccl.GetUniqueId(&id));
  memcpy(id128, &id, sizeof(id));
  return SATCV_OK;
}

extern "C" int satcv_comm_init(satcv_comm** out, int32_t rank, int32_t world, const void* id128) {
  SATCV_CHECK(out && id128 && world >= 1 && rank >= 0 && rank < world, "comm_init: bad arguments (rank %d of %d)", rank, world);
  int rc = rccl_ready(); if (rc) return rc;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  satcv_comm* c = new satcv_comm{nullptr, rank, world, 0};
  if (hipGetDevice(&c->device) != hipSuccess) { delete c; satcv_set_error("comm_init: no current device"); return SATCV_ERR_HIP; }
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) { delete c; satcv_set_error("ncclCommInitRank failed: %s", g_rccl.GetErrorString(r)); return SATCV_ERR_HIP; }
  *out = c;
  return SATCV_OK;
}

extern "C" int satcv_comm_destroy(satcv_comm* c) {
  if (!c) return SATCV_OK;
  if (c->comm) SATCV_NCCL(g_rccl.CommDestroy(c->comm));
  delete c;
  return SATCV_OK;
}

extern "C" int satcv_comm_info(const satcv_comm* c, int32_t* rank, int32_t* world) {
  SATCV_CHECK(c && rank && world, "comm_info: null");
  *rank = c->rank; *world = c->world;
  return SATCV_OK;
}

extern "C" int satcv_allreduce(satcv_comm* c, void* buf, int64_t count, int32_t dtype, int32_t average, void* stream) {
  SATCV_CHECK(c && buf && count >= 0, "allreduce: bad arguments");
  ncclDataType_t dt;
  if (dtype == SATCV_F32) dt = ncclFloat32;
  else if (dtype == SATCV_BF16) dt = ncclBfloat16;
  else if (dtype == SATCV_F64) dt = ncclFloat64;
  else { satcv_set_error("allreduce: dtype %d", dtype); return SATCV_ERR_INVALID; }
  if (count == 0) return SATCV_OK;
  SATCV_NCCL(g_rccl.AllReduce(buf, buf, (size_t)count, dt, average ? ncclAvg : ncclSum, c->comm, reinterpret_cast<hipStream_t>(stream)));
  return SATCV_OK;
}

extern "C" int satcv_allreduce_grads(satcv_comm* c, float* grads, int64_t lo, int64_t hi, int64_t bucket_elems, int32_t payload,
                                     void* scratch, void* stream) {
  SATCV_CHECK(c && grads && lo >= 0 && hi >= lo && bucket_elems > 0, "allreduce_grads: bad range [%lld, %lld) / bucket %lld", (long long)lo,
              (long long)hi, (long long)bucket_elems);
  SATCV_CHECK(payload == SATCV_F32 || (payload == SATCV_BF16 && scratch), "allreduce_grads: payload %d (bf16 needs a staging buffer)", payload);
  if (hi == lo) return SATCV_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long long n = hi - lo;
  bf16* stage = reinterpret_cast<bf16*>(scratch);
  if (payload == SATCV_BF16) {
    const int grid = (int)((n / 4 + 255) / 256 < 2048 ? (n / 4 + 255) / 256 + 1 : 2048);
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid), dim3(256), 0, st, grads + lo, stage, n);
  }
  // buckets are cut from the END of the range (the backward pass finishes the layers in decreasing offset order, so a caller that
  // sends ranges as they become final and a caller that sends everything at once issue the same collectives); one group = one fused
  // launch sequence on the stream
  SATCV_NCCL(g_rccl.GroupStart());
  for (long long b_hi = n; b_hi > 0;) {
    const long long b_lo = b_hi > bucket_elems ? b_hi - bucket_elems : 0;
    ncclResult_t r;
    if (payload == SATCV_BF16) r = g_rccl.AllReduce(stage + b_lo, stage + b_lo, (size_t)(b_hi - b_lo), ncclBfloat16, ncclSum, c->comm, st);
    else r = g_rccl.AllReduce(grads + lo + b_lo, grads + lo + b_lo, (size_t)(b_hi - b_lo), ncclFloat32, ncclSum, c->comm, st);
    if (r != ncclSuccess) { (void)g_rccl.GroupEnd(); satcv_set_error("ncclAllReduce failed: %s", g_rccl.GetErrorString(r)); return SATCV_ERR_HIP; }
    b_hi = b_lo;
  }
  SATCV_NCCL(g_rccl.GroupEnd());
  if (payload == SATCV_BF16) {
    const int grid = (int)((n / 4 + 255) / 256 < 2048 ? (n / 4 + 255) / 256 + 1 : 2048);
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(grid), dim3(256), 0, st, stage, grads + lo, n);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { satcv_set_error("allreduce_grads: %s", hipGetErrorString(e)); return SATCV_ERR_HIP; }
  return SATCV_OK;
}
