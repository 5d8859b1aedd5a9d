// C-ABI plumbing: error string, device info, hipGraph capture helpers, per-kernel-class
// HIP-event profiling used by bench.py for the roofline numbers.
#include "common.hpp"
#include <cstring>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";
void satcv_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* satcv_version(void) { return "satcv 0.1 (gfx950)"; }
extern "C" const char* satcv_last_error(void) { return g_err; }

extern "C" int satcv_device_info(int32_t* out4) {
  SATCV_CHECK(out4, "device_info: null");
  int dev = 0;
  SATCV_HIP(hipGetDevice(&dev));
  hipDeviceProp_t p;
  SATCV_HIP(hipGetDeviceProperties(&p, dev));
  out4[0] = p.multiProcessorCount;
  out4[1] = (int32_t)p.maxSharedMemoryPerMultiProcessor;
  out4[2] = p.warpSize;
  int arch = 0;
  sscanf(p.gcnArchName, "gfx%d", &arch);
  out4[3] = arch;
  return SATCV_OK;
}

// ---------------------------------------------------------------- dynamic-LDS opt-in, once per (kernel, device)
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute of a kernel: the cache is keyed by both and guarded, so a
// process that drives several GPUs, or several threads with distinct streams, gets it on every device (re-entrant
// given distinct contexts / streams) without a driver call on the launch path of every layer.
#include <map>
static std::mutex g_lds_mu;
static std::map<std::pair<const void*, int>, size_t> g_lds_have;
int satcv_ensure_dynamic_lds(const void* kern, size_t bytes) {
  if (bytes <= 48 * 1024) return SATCV_OK;
  int dev = 0;
  SATCV_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_lds_mu);
  size_t& have = g_lds_have[std::make_pair(kern, dev)];
  if (bytes > have) {
    SATCV_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    have = bytes;
  }
  return SATCV_OK;
}

// ---------------------------------------------------------------- graph capture
extern "C" int satcv_graph_begin(void* stream) {
  SATCV_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
  return SATCV_OK;
}
extern "C" int satcv_graph_end(void* stream, void** graph_exec_out) {
  SATCV_CHECK(graph_exec_out, "graph_end: null out");
  hipGraph_t g = nullptr;
  SATCV_HIP(hipStreamEndCapture((hipStream_t)stream, &g));
  hipGraphExec_t ge = nullptr;
  hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) { satcv_set_error("hipGraphInstantiate: %s", hipGetErrorString(e)); return SATCV_ERR_HIP; }
  *graph_exec_out = (void*)ge;
  return SATCV_OK;
}
extern "C" int satcv_graph_launch(void* graph_exec, void* stream) {
  SATCV_CHECK(graph_exec, "graph_launch: null");
  SATCV_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
  return SATCV_OK;
}
extern "C" int satcv_graph_destroy(void* graph_exec) {
  if (graph_exec) SATCV_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
  return SATCV_OK;
}

// -------------------------------------------------------------------- profiling
// kind 0: 3x3 (and dilated) implicit-GEMM conv fwd/dgrad; 1: 1x1 / transposed-conv GEMMs;
// 2: weight-gradient kernel; 3: fused thin-layer backward (BatchNorm apply + data gradient + weight gradient).
#define PROF_KINDS 4
struct ProfRec { hipEvent_t a, b; double flops; };
static std::mutex g_prof_mu;
static int g_prof_mask = 0;
static std::vector<ProfRec> g_prof[PROF_KINDS];
static hipEvent_t g_prof_pending[PROF_KINDS];
static double g_prof_pending_flops[PROF_KINDS];

void satcv_prof_begin(int kind, double flops, hipStream_t st) {
  if (!(g_prof_mask & (1 << kind))) return;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  (void)hipEventRecord(e, st);
  g_prof_pending[kind] = e;
  g_prof_pending_flops[kind] = flops;
}
void satcv_prof_end(int kind, hipStream_t st) {
  if (!(g_prof_mask & (1 << kind))) return;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  (void)hipEventRecord(e, st);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof[kind].push_back({g_prof_pending[kind], e, g_prof_pending_flops[kind]});
}
extern "C" int satcv_prof_enable(int32_t kind_mask) {
  g_prof_mask = kind_mask;
  return SATCV_OK;
}
// Synchronises on the recorded events (call after the timed region) and clears them.
extern "C" int satcv_prof_collect(int32_t kind, double* total_ms, int64_t* launches, double* flops) {
  SATCV_CHECK(kind >= 0 && kind < PROF_KINDS && total_ms && launches && flops, "prof_collect: bad args");
  std::lock_guard<std::mutex> lk(g_prof_mu);
  double ms = 0, fl = 0;
  for (auto& r : g_prof[kind]) {
    (void)hipEventSynchronize(r.b);
    float t = 0;
    if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) ms += t;
    fl += r.flops;
    (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
  }
  *total_ms = ms; *launches = (int64_t)g_prof[kind].size(); *flops = fl;
  g_prof[kind].clear();
  return SATCV_OK;
}

// ------------------------------------------------------------------ kernel-selection knobs
// (environment defaults SATCV_DB / SATCV_THIN; satcv_set_option overrides them at run time)
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
int g_opt_igemm_db = env_int("SATCV_DB", 1);
int g_opt_wgrad_db = env_int("SATCV_WGRAD_DB", 1);
int g_opt_igemm_sched = env_int("SATCV_IGEMM_SCHED", 0);
int g_opt_igemm_thin = env_int("SATCV_THIN", 1);      // 0 off, 1 / 2 on wherever the shape limits allow (independent of the batch size)
int g_opt_igemm_m16 = env_int("SATCV_M16", 1);       // the 16x16x32 deep 3x3 tile: 0 off, 1 launches that produce statistics (training), 2 every eligible launch
int g_opt_m16p = env_int("SATCV_M16P", 1);               // conv_igemm_m16p.hip (persistent 16x16x32 tile): 0 off, 1 where a workgroup gets >= 2 tiles, 2 every eligible launch
extern int g_m16p_launches;
int g_tr_launches = 0;                                   // launches conv_thin_roles.hip took (tests: "path taken")
int g_opt_thin_roles = env_int("SATCV_THIN_ROLES", 1);   // conv_thin_roles.hip: 0 off, 1 the shapes it measured faster on, 2 every shape it serves
// conv_igemm_m16p.hip: s_setprio of the staging waves, decimal digits (plain launches)(launches with the fused input BatchNorm)(launches with the
// fused BatchNorm-backward sums), each 0 ... 3 -- e.g. 30 = priority 3 for the forward launches that transform their input, 0 elsewhere
int g_opt_m16p_prio = env_int("SATCV_M16P_PRIO", 30);      // (30: profiles/r06_ab_m16p_prio_step.txt)
int g_opt_wgrad_m16 = env_int("SATCV_WGRAD_M16", 0);   // wgrad_dma_kernel on v_mfma_f32_16x16x32_bf16 (round 6: built as the review asked, measured slower: off)
int g_opt_splitk = env_int("SATCV_SPLITK", 0);         // split-K of under-filled PLAIN (halo-tile / 1x1) launches: opt-in, see conv_igemm_fast.hip
static int* opt_slot(const char* key) {
  if (!key) return nullptr;
  if (!strcmp(key, "igemm_db")) return &g_opt_igemm_db;
  if (!strcmp(key, "igemm_thin")) return &g_opt_igemm_thin;
  if (!strcmp(key, "wgrad_db")) return &g_opt_wgrad_db;
  if (!strcmp(key, "igemm_sched")) return &g_opt_igemm_sched;
  if (!strcmp(key, "splitk")) return &g_opt_splitk;
  if (!strcmp(key, "igemm_m16")) return &g_opt_igemm_m16;
  if (!strcmp(key, "thin_roles")) return &g_opt_thin_roles;
  if (!strcmp(key, "thin_roles_launches")) return &g_tr_launches;
  if (!strcmp(key, "m16p")) return &g_opt_m16p;
  if (!strcmp(key, "m16p_prio")) return &g_opt_m16p_prio;
  if (!strcmp(key, "wgrad_m16")) return &g_opt_wgrad_m16;
  if (!strcmp(key, "m16p_launches")) return &g_m16p_launches;
  return nullptr;
}
extern "C" int satcv_set_option(const char* key, int32_t value) {
  int* p = opt_slot(key);
  SATCV_CHECK(p, "set_option: unknown key '%s'", key ? key : "(null)");
  *p = value;
  return SATCV_OK;
}
extern int g_ws_launches;          // conv_igemm_ws.hip
extern "C" int satcv_get_option(const char* key, int32_t* value) {
  if (key && value && !strcmp(key, "igemm_thin_launches")) { *value = g_ws_launches; return SATCV_OK; }
  int* p = opt_slot(key);
  SATCV_CHECK(p && value, "get_option: unknown key '%s'", key ? key : "(null)");
  *value = *p;
  return SATCV_OK;
}

// ------------------------------------------------------------------ CRC32C (host)
// TFRecord framing (utils/prediction_tools.py:221, 404; tf.io.TFRecordWriter / TFRecordDataset) protects the length and
// the payload of every record with a masked CRC-32C (Castagnoli, reflected polynomial 0x82F63B78).  Slice-by-8 tables.
static uint32_t g_crc_tab[8][256];
static bool g_crc_init = [] {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
    g_crc_tab[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; ++i)
    for (int t = 1; t < 8; ++t) g_crc_tab[t][i] = (g_crc_tab[t - 1][i] >> 8) ^ g_crc_tab[0][g_crc_tab[t - 1][i] & 0xff];
  return true;
}();
extern "C" uint32_t satcv_crc32c(const void* data, uint64_t nbytes, uint32_t crc_in) {
  const unsigned char* p = reinterpret_cast<const unsigned char*>(data);
  uint32_t c = ~crc_in;
  while (nbytes >= 8) {
    uint32_t lo, hi;
    memcpy(&lo, p, 4); memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = g_crc_tab[7][lo & 0xff] ^ g_crc_tab[6][(lo >> 8) & 0xff] ^ g_crc_tab[5][(lo >> 16) & 0xff] ^ g_crc_tab[4][lo >> 24] ^
        g_crc_tab[3][hi & 0xff] ^ g_crc_tab[2][(hi >> 8) & 0xff] ^ g_crc_tab[1][(hi >> 16) & 0xff] ^ g_crc_tab[0][hi >> 24];
    p += 8; nbytes -= 8;
  }
  while (nbytes--) c = (c >> 8) ^ g_crc_tab[0][(c ^ *p++) & 0xff];
  return ~c;
}
