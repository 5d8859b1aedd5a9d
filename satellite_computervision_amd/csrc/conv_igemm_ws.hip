// Weights-stationary 3x3 convolution for the THIN full- and half-resolution layers of the U-Net (Cin <= 64 stored channels,
// Cout 32 or 64, every pair of them; Conv2D of utils/model_tools.py:178, 312, 315 at decoder / encoder levels 0 and 1, forward and data gradient).
//
// Why a kernel of its own (s_memtime stamps of the general kernel on 64 -> 32 channels at 256 x 256, tools/stamp_probe.py): of the
// ~24,500 cycles a workgroup spent per 256-pixel tile only 3,500 were the MFMA phase; 4,300 went into the gather-table setup, 5,500 into
// issuing 11 loads per thread and 16-channel chunk, 2,300 into LDS stores, 4,700 into the epilogue, with two barriers per chunk.  Here:
//   * PERSISTENT workgroups (grid = resident workgroups): the setup runs once, the loop walks the tiles;
//   * the whole 9-tap weight tensor (<= 37 KB) is staged in LDS ONCE per workgroup instead of per tile and chunk (it was 46 % of the
//     bytes that went through the load path);
//   * the activation halo tile is staged for ALL input channels at once (one barrier pair per tile, no chunk loop); a thread always
//     handles the same 8 channels, so its BatchNorm scale / shift values live in registers for the whole launch;
//   * the next tile's loads are issued before the current tile is multiplied and stay in flight through its MFMA phase and epilogue;
//     2-4 workgroups per CU overlap one another's load, MFMA and store phases.
// LDS images and MFMA fragment addressing are those of conv_igemm_fast.hip ([slot][row][pitch][8] activations, [tap][slot][cout][8]
// weights, v_mfma_f32_32x32x16_bf16), the epilogue is the shared igemm_epilogue (bias, BN statistics, LDS-staged 16-byte stores).
#include "igemm_common.hpp"
#include <cstdlib>
extern int g_opt_igemm_thin;      // api.hip: satcv_set_option("igemm_thin", ...)
int g_ws_launches = 0;            // launches served by this kernel (satcv_get_option("igemm_thin_launches"): tests check the path taken)

#ifndef SATCV_ABLATE
#define SATCV_ABLATE 0
#endif
#define WABL(bit) ((SATCV_ABLATE & (bit)) != 0)
#ifdef SATCV_STAMP
__device__ unsigned long long g_stamp_ws[8][8][8];        // diagnostic build only
extern "C" int satcv_debug_read_stamps_ws(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp_ws), sizeof(g_stamp_ws)) == hipSuccess ? 0 : -1; }
#define STAMP(t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP_V(t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#endif

// WN = 2 (64 -> 64 channels): 8 waves, each pair of waves shares its 64 pixels and splits the 64 output channels -- the 74 KB weight
// tensor and the 44 KB tile leave ONE workgroup per CU, and 8 waves keep the two waves per SIMD of the other forms
// T: bf16, or fp8 (e4m3 storage, 8-byte items, v_mfma_f32_32x32x16_fp8_fp8: the folded fp8 inference graph's thin layers -- same MFMA rate,
// half the bytes of these HBM-bound layers)
// DIL: dilation rate (1, or 3: the dilated convolutions of the atrous CNNs, utils/model_tools.py:922-979 -- 16 / 32 channels at full resolution, which
// the tap-loop tile of conv_igemm_fast.hip served at a tenth of their HBM roofline, tools/family_time.py); the halo is DIL pixels wide
template <typename T, int CIN, int NT, int WPS, int WN = 1, int DIL = 1>
__global__ __launch_bounds__(256 * WN, WPS) void igemm_ws_kernel(const IgemmArgs a, const int total_tiles) {
  constexpr int TW = 32, TH = 8, WM = 4, MT = 2, BM = 256, BN = WN * NT * 32, NTHREADS = 256 * WN, EL = 8;
  constexpr int SLOTS = CIN / EL, CL = TW + 2 * DIL, PITCH = CL, RL = TH + 2 * DIL;
  constexpr int PLANE = RL * PITCH * EL;                                   // elements of one slot plane
  // 8 consecutive lanes store 8 / SLOTS pixels x SLOTS slots with one ds_write_b128 (serviced in groups of 8 lanes over 32 banks):
  // the plane stride must be 128 / SLOTS bytes modulo 128 for the eight 16-byte stores to fall on distinct banks
  constexpr int WANT = (SLOTS >= 8 ? 16 : 128 / SLOTS) / (int)sizeof(T), BANKSPAN = 128 / (int)sizeof(T);      // (elements)
  constexpr int SPAD = ((WANT - PLANE % BANKSPAN) % BANKSPAN + BANKSPAN) % BANKSPAN;
  constexpr int SLOT_STRIDE = PLANE + SPAD;
  constexpr int A_ITEMS = RL * CL * SLOTS, AI = (A_ITEMS + NTHREADS - 1) / NTHREADS;
  constexpr int W_ITEMS = 9 * SLOTS * BN;
  constexpr int OPITCH = BN + 16 / (int)sizeof(T);
  constexpr size_t A_BYTES = (size_t)SLOTS * SLOT_STRIDE * sizeof(T);
  constexpr size_t O_BYTES = (size_t)BM * OPITCH * sizeof(T) + (size_t)(WM + 1) * 2 * BN * sizeof(float);
  constexpr size_t R0_BYTES = ((A_BYTES > O_BYTES ? A_BYTES : O_BYTES) + 127) / 128 * 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* ldsA = reinterpret_cast<T*>(smem_raw);                                 // activation tile; re-used as the output staging tile
  T* ldsW = reinterpret_cast<T*>(smem_raw + R0_BYTES);                      // [tap][slot][BN][8], resident for the whole launch

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, hh = lane >> 5;
  const int slot_t = tid % SLOTS;                                           // NTHREADS % SLOTS == 0: a thread's channel slot is fixed
  const T* wp = reinterpret_cast<const T*>(a.w);

  // ---- once per workgroup: weights -> LDS, this thread's source / scale / shift, the gather table
  for (int it = tid; it < W_ITEMS; it += NTHREADS) {
    const int co = it % BN, run = it / BN;                                  // run = tap * SLOTS + slot
    const Raw8<T> v = gload8<T>(wp + ((size_t)run * a.cout_pad + co) * EL); // (cout_pad == BN: one N tile)
    lstore8<T>(ldsW + (size_t)it * EL, v);
  }
  const int ch0 = slot_t * EL;
  const bool second = ch0 >= a.c0;                                          // concat([skip, up]): which source holds this thread's channels
  const T* src = second ? reinterpret_cast<const T*>(a.x1) + (ch0 - a.c0) : reinterpret_cast<const T*>(a.x0) + ch0;
  const int cs = second ? a.c1 : a.c0;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sc[e] = a.in_scale ? a.in_scale[ch0 + e] : 1.f; sh[e] = a.in_scale ? a.in_shift[ch0 + e] : 0.f; }
  const bool aff = a.in_scale != nullptr;
  const unsigned relu_lim = a.in_relu != 0 ? 0u : 0x80008000u;            // ReLU as a packed 16-bit max (common.hpp: affine8_lim)

  // per staged item, tile-invariant: LDS destination, halo coordinates relative to the tile origin, element offset from the tile's
  // base pixel (the per-tile cost of an item is four compares against wave-uniform limits, a select and one 64-bit add: the
  // (n, y, x) -> address arithmetic with its quarter-rate multiplies was a third of the vector instructions of the tile loop)
  int a_l[AI], a_yx[AI], a_eoff[AI];
#pragma unroll
  for (int j = 0; j < AI; ++j) {
    const int it = tid + j * NTHREADS;
    const int pix = it / SLOTS, c = pix % CL, L = pix / CL;
    a_l[j] = it < A_ITEMS ? slot_t * SLOT_STRIDE + (L * PITCH + c) * EL : -1;
    a_yx[j] = ((L - DIL) << 16) | ((c - DIL) & 0xffff);
    a_eoff[j] = ((L - DIL) * a.w_ + (c - DIL)) * cs;
  }
  int a_off[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int q = (wm * MT + m) * 32 + r;
    a_off[m] = ((q / TW) * PITCH + (q % TW)) * EL;
  }

  auto tile_origin = [&](int v, int& n0, int& y0, int& x0) {
    const int tx = v % a.tiles_x; v /= a.tiles_x;
    const int ty = v % a.tiles_y;
    n0 = v / a.tiles_y; y0 = ty * TH; x0 = tx * TW;
  };
  auto next_origin = [&](int& n0, int& y0, int& x0) {                       // the tiles of a workgroup are consecutive: no division per tile
    x0 += TW;
    if (x0 >= a.w_) { x0 = 0; y0 += TH; if (y0 >= a.h) { y0 = 0; ++n0; } }
  };
  Raw8<T> ra[AI];
  unsigned valid = 0;                                                       // bit j: item j lies inside the image (else zero padding)
  auto issue_loads = [&](int n0, int y0, int x0) -> unsigned {
    unsigned vm = 0;
    const int ylo = -y0, yhi = a.h - y0, xlo = -x0, xhi = a.w_ - x0;        // wave-uniform limits of the halo coordinates
    const T* base = src + ((size_t)(n0 * a.h + y0) * a.w_ + x0) * cs;
#pragma unroll
    for (int j = 0; j < AI; ++j) {
      const int dy = a_yx[j] >> 16, dx = (int)(short)(a_yx[j] & 0xffff);
      const bool ok = a_l[j] >= 0 && dy >= ylo && dy < yhi && dx >= xlo && dx < xhi;
      vm |= (ok ? 1u : 0u) << j;                                            // (items outside load the base pixel and are zeroed below: no
      int off = ok ? a_eoff[j] : 0;                                         //  branch around a vector-memory instruction in the loop)
      asm volatile("" : "+v"(off));                                         // (select on 32 bits: the compiler otherwise keeps every offset sign-extended, 2 registers per item)
      if (!WABL(4)) ra[j] = gload8<T>(base + off);
      else ra[j] = zero8<T>();
    }
    return vm;
  };

  // ---- XCD-aware tile order: blocks b and b + 8 share an XCD; each XCD walks a contiguous range of tiles (neighbouring halos in its L2)
  const int G = gridDim.x;
  const int xcd = blockIdx.x & 7, nx = G >> 3, remx = G & 7;
  const int bid = (xcd < remx ? xcd * (nx + 1) : remx * (nx + 1) + (xcd - remx) * nx) + (blockIdx.x >> 3);
  // tiles of this workgroup: [t_lo, t_hi) -- contiguous, so consecutive tiles of one workgroup share halo rows / columns too
  const int per = total_tiles / G, extra = total_tiles % G;
  const int t_lo = bid * per + (bid < extra ? bid : extra), t_hi = t_lo + per + (bid < extra ? 1 : 0);

  // BN statistics of all tiles of this workgroup are summed in registers (thread c < BN owns channel c) and reach the replica rows
  // with ONE pair of atomics per channel at the end: per-tile atomics stay on the wave's vmcnt queue for thousands of cycles under
  // load and the next tile's loads could not be waited for without them
  double carry[2] = {0.0, 0.0};
  int n0, y0, x0;
  if (t_lo < t_hi) { tile_origin(t_lo, n0, y0, x0); valid = issue_loads(n0, y0, x0); }
  __syncthreads();                                                          // weights are in LDS
#ifdef SATCV_STAMP
  unsigned long long q0, q1, q2, q3, q4, q5, q6, q7, q8, z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  for (int t = t_lo; t < t_hi; ++t) {
#ifdef SATCV_STAMP
    STAMP(q0);
    STAMP_V(q1);                               // wait for this tile's loads
#endif
    // ---- registers -> LDS (BatchNorm affine + ReLU of the producing layer, zero padding AFTER it)
#pragma unroll
    for (int j = 0; j < AI; ++j) {
      Raw8<T> v = ra[j];
      if (aff) { if constexpr (std::is_same<T, bf16>::value) v = affine8_lim(v, sc, sh, relu_lim); else v = affine8<T>(v, sc, sh, a.in_relu); }
      v = select8<T>((valid >> j) & 1u, v);
      if (a_l[j] >= 0) lstore8<T>(ldsA + a_l[j], v);
    }
#ifdef SATCV_STAMP
    STAMP(q2);
#endif
    __syncthreads();
#ifdef SATCV_STAMP
    STAMP(q3);
#endif
    // ---- the next tile's loads: in flight during this tile's MFMA phase and epilogue
    const int tn0 = n0, ty0 = y0, tx0 = x0;
    if (t + 1 < t_hi) { next_origin(n0, y0, x0); valid = issue_loads(n0, y0, x0); }
#ifdef SATCV_STAMP
    STAMP(q4);
#endif
    // ---- 9 taps x CIN / 16 k-steps, fragment reads one step ahead of their MFMAs
    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
    {
      constexpr int KS = CIN / 16, STEPS = 9 * KS;
      FragT<T> af[2][MT], bf[2][NT];
      auto read_step = [&](int st, int buf) {
        const int tap = st / KS, ks = st % KS;
        const int tap_off = ((tap / 3) * DIL * PITCH + (tap % 3) * DIL) * EL;
        const int slot = ks * 2 + hh;
#pragma unroll
        for (int m = 0; m < MT; ++m) af[buf][m] = lds_frag<T>(ldsA + slot * SLOT_STRIDE + a_off[m] + tap_off);
#pragma unroll
        for (int n = 0; n < NT; ++n) bf[buf][n] = lds_frag<T>(ldsW + ((tap * SLOTS + slot) * BN + (wn * NT + n) * 32 + r) * EL);
      };
      read_step(0, 0);
#pragma unroll
      for (int st = 0; st < STEPS; ++st) {
        asm volatile("" ::: "memory");
        if (st + 1 < STEPS) read_step(st + 1, (st + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) { if (!WABL(2)) mma32<T>(acc[m][n], af[st & 1][m], bf[st & 1][n]); }
      }
    }
#ifdef SATCV_STAMP
    STAMP(q5);
#endif
    __syncthreads();                                                        // every wave is past its last fragment read: the tile region is free
#ifdef SATCV_STAMP
    STAMP(q6);
#endif
    igemm_epilogue<T, TW, WM, WN, MT, NT, WABL(1), true, false>(a, acc, tn0, ty0, tx0, 0, smem_raw, carry);
#ifdef SATCV_STAMP
    STAMP(q7);
#endif
    __syncthreads();                                                        // staged output read out before the next tile is written
#ifdef SATCV_STAMP
    STAMP(q8);
    z[0] += q1 - q0; z[1] += q2 - q1; z[2] += q3 - q2; z[3] += q4 - q3; z[4] += q5 - q4; z[5] += q6 - q5; z[6] += q7 - q6; z[7] += q8 - q7;
#endif
  }
  if (a.stats && tid < BN && tid < a.cout && t_lo < t_hi) {
    satcv_stat_t* rowp = a.stats + (size_t)(blockIdx.x % SATCV_STAT_ROWS) * 2 * a.stats_ld;
    atomicAdd(rowp + tid, (satcv_stat_t)carry[0]);
    atomicAdd(rowp + a.stats_ld + tid, (satcv_stat_t)carry[1]);
  }
#ifdef SATCV_STAMP
  if (blockIdx.x < 8 && lane == 0)
    for (int i = 0; i < 8; ++i) g_stamp_ws[blockIdx.x][wave][i] = z[i] / (unsigned long long)max(t_hi - t_lo, 1);
#endif
}

// ------------------------------------------------------------------ host side
template <typename T, int CIN, int NT, int WPS, int WN = 1, int DIL = 1>
static int ws_cfg(IgemmArgs& a, hipStream_t st, bool dry) {
  constexpr int TW = 32, TH = 8, BN = WN * NT * 32, SLOTS = CIN / 8, RL = TH + 2 * DIL, PITCH = TW + 2 * DIL;
  constexpr int ES = (int)sizeof(T);
  constexpr int PLANE = RL * PITCH * 8;
  constexpr int WANT = (SLOTS >= 8 ? 16 : 128 / SLOTS) / ES, BANKSPAN = 128 / ES;
  constexpr int SPAD = ((WANT - PLANE % BANKSPAN) % BANKSPAN + BANKSPAN) % BANKSPAN;
  constexpr size_t A_BYTES = (size_t)SLOTS * (PLANE + SPAD) * ES;
  constexpr size_t O_BYTES = (size_t)256 * (BN + 16 / ES) * ES + (size_t)5 * 2 * BN * 4;
  constexpr size_t R0 = ((A_BYTES > O_BYTES ? A_BYTES : O_BYTES) + 127) / 128 * 128;
  constexpr size_t LDS = R0 + (size_t)9 * SLOTS * BN * 8 * ES;
  static_assert(LDS <= 160 * 1024, "weights + tile exceed the LDS");
  // tiling fields the shared epilogue reads
  a.halh = a.halw = DIL;
  a.tiles_x = cdiv(a.w_, TW); a.tiles_y = cdiv(a.h, TH);
  a.rpi = TH; a.imgs = 1; a.ngroups = a.n; a.seg = RL; a.rl = RL; a.cl = PITCH; a.pitch = PITCH; a.n_tiles = 1;
  const long long total = (long long)a.n * a.tiles_y * a.tiles_x;
  if (total <= 0 || total > 0x7fffffffLL) return SATCV_ERR_UNSUPPORTED;
  if (dry) return SATCV_OK;
  auto kern = igemm_ws_kernel<T, CIN, NT, WPS, WN, DIL>;
  { const int rc = satcv_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS); if (rc) return rc; }
  static int ncu = 0;
  if (!ncu) {
    int dev = 0; hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) { satcv_set_error("ws: device query failed"); return SATCV_ERR_HIP; }
    ncu = p.multiProcessorCount;
  }
  // resident workgroups per CU: LDS- or wave-limited (4 waves each, WPS per SIMD requested)
  int per_cu = (int)((160 * 1024) / LDS);
  if (per_cu > WPS) per_cu = WPS;
  if (per_cu < 1) per_cu = 1;
  long long grid = (long long)ncu * per_cu;
  if (grid > total) grid = total;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256 * WN), LDS, st, a, (int)total);
  ++g_ws_launches;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { satcv_set_error("igemm_ws launch: %s", hipGetErrorString(e)); return SATCV_ERR_HIP; }
  return SATCV_OK;
}

// returns SATCV_ERR_UNSUPPORTED when the shape is outside this kernel's limits (the caller falls back to conv_igemm_fast.hip)
int igemm_ws_launch(IgemmArgs& a, int dtype, hipStream_t st, bool dry) {
  if (!g_opt_igemm_thin || (dtype != SATCV_BF16 && dtype != SATCV_FP8)) return SATCV_ERR_UNSUPPORTED;
  const int cin = a.c0 + a.c1;
  // (accumulate == 1, y += result, on the bf16 forms without statistics / fused pool: residual data gradients)
  if (a.kh != 3 || a.kw != 3 || (a.dil != 1 && a.dil != 3) || a.stride != 1 || a.mode_in || a.mode_out || a.bst_y) return SATCV_ERR_UNSUPPORTED;
  if (a.accumulate && (a.accumulate != 1 || dtype != SATCV_BF16 || a.stats || a.pool_y || a.out_relu || a.out_scale)) return SATCV_ERR_UNSUPPORTED;
  if (a.pool_y && (8 % a.pool_f != 0 || 32 % a.pool_f != 0)) return SATCV_ERR_UNSUPPORTED;      // pooling windows inside one 8 x 32 tile
  if (a.dil == 3) {
    // dilation 3 (atrous CNNs): 16 / 32 stored input channels -> 16 / 32 output channels (16: a 32-column tile whose upper half is zero weights;
    // the interior-tile epilogue skips the column groups beyond cout), bf16, no fused pool
    if (dtype != SATCV_BF16 || a.pool_y || !(cin == 16 || cin == 32) || !(a.cout == 16 || a.cout == 32) || a.cout_pad != 32 || a.cstat != a.cout) return SATCV_ERR_UNSUPPORTED;
    if (a.x1 && (a.c0 % 8 != 0)) return SATCV_ERR_UNSUPPORTED;
    if (a.h % 8 != 0 || a.w_ % 32 != 0 || a.ldy % 8 != 0 || ((uintptr_t)a.y % 16) != 0) return SATCV_ERR_UNSUPPORTED;
    return cin == 16 ? ws_cfg<bf16, 16, 1, 2, 1, 3>(a, st, dry) : ws_cfg<bf16, 32, 1, 2, 1, 3>(a, st, dry);
  }
  if (a.cout == 16 && dtype == SATCV_BF16 && !a.pool_y && (cin == 16 || cin == 32) && a.cout_pad == 32 && a.cstat == 16 && !(a.x1 && (a.c0 % 8 != 0)) &&
      a.h % 8 == 0 && a.w_ % 32 == 0 && a.ldy % 8 == 0 && ((uintptr_t)a.y % 16) == 0) {
    // 16 output channels (the 16-filter atrous CNN of utils/model_tools.py:992): a 32-column tile whose upper half is zero weights
    return cin == 16 ? ws_cfg<bf16, 16, 1, 3>(a, st, dry) : ws_cfg<bf16, 32, 1, 3>(a, st, dry);
  }
  if (!(cin == 16 || cin == 32 || cin == 64) || !(a.cout == 32 || a.cout == 64) || a.cout_pad != a.cout || a.cstat != a.cout) return SATCV_ERR_UNSUPPORTED;
  if (a.x1 && (a.c0 % 8 != 0)) return SATCV_ERR_UNSUPPORTED;
  const int epv = dtype == SATCV_FP8 ? 16 : 8;
  if (a.h % 8 != 0 || a.w_ % 32 != 0 || a.ldy % epv != 0 || ((uintptr_t)a.y % 16) != 0) return SATCV_ERR_UNSUPPORTED;      // whole 8 x 32 tiles: the kernel compiles the interior-tile epilogue only
  if (dtype == SATCV_FP8) {
    // e4m3 storage (folded fp8 inference graph): the 32- and 64-channel inputs; no statistics (inference only)
    if (a.stats || cin == 16 || (a.pool_y && a.pool_ld % 16 != 0)) return SATCV_ERR_UNSUPPORTED;
    if (a.cout == 32) return cin == 32 ? ws_cfg<fp8, 32, 1, 3>(a, st, dry) : ws_cfg<fp8, 64, 1, 2>(a, st, dry);
    return cin == 32 ? ws_cfg<fp8, 32, 2, 2>(a, st, dry) : ws_cfg<fp8, 64, 1, 1, 2>(a, st, dry);
  }
  // (no tile-count threshold: the kernel choice must not depend on the batch size -- inference is bit-identical across batch splits,
  //  tests/test_model_gpu.py::test_full_size_batch_invariance_property, and the two kernels sum K in different orders)
  if (a.cout == 32) {
    // three workgroups per CU where the registers allow (<= 168): 16 -> 32 at 256 x 256 132 -> 116 us, 32 -> 32 equal or better
    if (cin == 16) return ws_cfg<bf16, 16, 1, 3>(a, st, dry);
    if (cin == 32) return ws_cfg<bf16, 32, 1, 3>(a, st, dry);
    return ws_cfg<bf16, 64, 1, 2>(a, st, dry);
  }
  if (cin == 16) return ws_cfg<bf16, 16, 2, 2>(a, st, dry);
  if (cin == 32) return ws_cfg<bf16, 32, 2, 2>(a, st, dry);
  // 64 -> 64: weights (74 KB) + tile leave one workgroup per CU, so it has 8 waves (128 x 128 at batch 64: 154 -> 120 us with the fused
  // input BatchNorm; at batch 8, two tiles per workgroup, 23.4 -> 24.7 us)
  return ws_cfg<bf16, 64, 1, 1, 2>(a, st, dry);
}
