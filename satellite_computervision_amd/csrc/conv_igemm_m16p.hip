// Round 5: the 16x16x32 tile of conv_igemm_m16.hip as a PERSISTENT, cross-tile pipelined kernel for the mid 3x3 layers (forward and data
// gradient of Conv2D at /root/reference/utils/model_tools.py:178, 312, 315 with 64 ... 256 input channels: the 128 x 128 ... 32 x 32 levels).
//
// Why.  With 64 ... 256 input channels a 256-pixel x 128-channel tile is 6 ... 24 tap-row units of K; the one-tile-per-workgroup kernels pay
// their set-up, the first loads and the epilogue (statistics, LDS-staged stores) once per tile with nothing else in flight on the CU --
// 64 -> 128 channels at 128 x 128 ran at 560 TFLOP/s, 128 -> 256 at 64 x 64 at 0.24 of its roofline (profiles/r05_step_timeline.txt).
// Here a workgroup owns a contiguous range of tiles of ONE output-channel block and never stops:
//   * waves 8-11 (staging): the activation chunks (32 channels of the halo tile, BatchNorm + ReLU of the producing layer in registers, zero
//     padding) flow through two LDS stages as one stream ACROSS tiles -- chunk c + 1 is stored while chunk c is multiplied, chunk c + 3 is
//     in flight in registers --; between two such stores they drain the PREVIOUS tile from its LDS staging image: 16-byte global stores,
//     BatchNorm statistics of the stored values (or the fused BatchNorm-backward sums against the layer's raw outputs) in registers, summed
//     over all tiles of the workgroup and flushed once.
//   * waves 0-7 (matrix): fragment reads and MFMAs only, one tap-row unit (3 taps x 32 channels: 48 MFMAs per wave) per barrier.  Each wave
//     also moves its three 1-KB pieces of the NEXT unit's weights into the other slot of a two-slot ring with global_load_lds_dwordx4 (their
//     only vector-memory instructions, so `s_waitcnt vmcnt(0)` in front of the barrier waits for exactly those).  At the end of a tile a
//     wave rounds its 64 accumulators to bf16 into the staging image -- 16 ds_write_b64 -- and starts the next tile in the same interval.
// The MFMA operands are swapped with respect to conv_igemm_m16.hip (A = weights, B = pixels): a lane then holds FOUR CONSECUTIVE OUTPUT
// CHANNELS of one pixel per 16 x 16 block, i.e. 8 contiguous bytes of the NHWC line, instead of four pixels of one channel (2-byte stores).
// The bias is the accumulators' initial value.  Staging image [256 pixels][128 channels] without padding: the 8-byte granule g of pixel p
// lives at granule g ^ 2 (p & 15) of its 256-byte row -- the 32 lanes of a ds_write_b64 pass cover all 64 banks once, and the drain's
// ds_read_b128 of 16 lanes covers one whole row.
// LDS: 2 x 22 KB activation stages + 2 x 24 KB weight slots + 64 KB staging + bias / scale / shift tables (<= 3.5 KB) = 158.5 KB.
#include "igemm_common.hpp"
#include <cstdlib>

// compile-time ablation switches for profiling builds (tools/scripts/build_m16p_variants.sh; the product build has none set; timing only, wrong
// results): 1 the MFMAs, 2 the fragment reads, 4 the weight pieces (LDS-DMA), 8 the activation loads + stores of the stream (after the
// prologue), 16 the drain (image reads, global stores, sums), 32 the accumulator dump, 64 the activation LOADS only, 128 the barrier's
// vmcnt wait
#ifndef SATCV_M16P_ABL
#define SATCV_M16P_ABL 0
#endif
#define PABL(bit) ((SATCV_M16P_ABL & (bit)) != 0)

// diagnostic build (-DSATCV_STAMP_M16P, tools/m16p_stamp_probe.py): s_memtime sums per wave of the first 8 workgroups -- matrix waves: [0] fragment reads +
// MFMAs (+ dump), [1] the wait for their weight pieces, [2] the barrier; staging waves: [0] their interval's work, [2] the barrier; [3] intervals
#ifdef SATCV_STAMP_M16P
__device__ unsigned long long g_stamp_m16p[8][12][4];
extern "C" int satcv_debug_read_stamps_m16p(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp_m16p), sizeof(g_stamp_m16p)) == hipSuccess ? 0 : -1; }
#define PSTAMP(t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PSTAMP(t) do { } while (0)
#endif

__device__ __forceinline__ bf16x4 m16p_tr_read(const bf16* p) {
  typedef short short4v_ __attribute__((ext_vector_type(4)));
  short4v_ v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v_*)(p));
  return __builtin_bit_cast(bf16x4, v);
}

extern int g_opt_m16p_prio;            // api.hip
extern int g_opt_m16p;                 // api.hip: 0 off, 1 on where a workgroup gets at least two tiles
int g_m16p_launches = 0;               // launches taken here (satcv_get_option("m16p_launches"): tests assert the path)

// BN_: output channels per workgroup -- 128 (two 64-channel wave columns), or 64 for the layers with 64 filters (round 6: 64 -> 64 and
// 64 + 64 -> 64 at 128 x 128 and their data gradients; wave tile 64 pixels x 32 channels, half the accumulators, the same pixel tile and stream)
template <int BN_>
struct M16PGeom {
  static constexpr int TW = 32, TH = 8, BM = 256, BN = BN_, CL = TW + 2, RL = TH + 2;
  static constexpr int NPIX = ((RL * CL + 7) / 8) * 8;                       // halo pixels, padded to whole octets
  static constexpr int PLANE_E = (((NPIX * 16 + 255) / 256) * 256) / 2;      // elements per 8-channel slot plane (a multiple of 256 bytes)
  static constexpr int A_STAGE_E = 4 * PLANE_E;
  static constexpr int UNIT_E = 3 * 4 * BN * 8;                              // a tap row of one 32-channel chunk: [3 taps][4 slots][128][8]
  static constexpr int A_ITEMS = NPIX * 4;
  static constexpr int NS = 256, AI = (A_ITEMS + NS - 1) / NS;
  static constexpr int VPR = BN / 8, PPR = NS / VPR;                         // 16-byte groups per image row; pixels per drain round
  static constexpr int ROUNDS = BM * (BN / 8) / NS;                          // drain rounds of 256 x 16 bytes per tile
  static constexpr int NBW = BN / 32;                                        // 16-channel blocks per matrix wave (wave tile 64 pixels x BN / 2)
  static constexpr int SWM = BN == 128 ? 15 : 7;                             // image swizzle: 8-byte granule g of pixel p at g ^ 2 (p & SWM)
  static constexpr size_t FIXED_BYTES = (size_t)(2 * A_STAGE_E + 2 * UNIT_E + BM * BN) * 2 + BN * sizeof(float);
};

// BST: the launch carries the fused BatchNorm-backward sums (a data gradient: no input transform) -- two instantiations so that neither holds the
// other's per-channel constants in registers
template <bool BST, int BN_ = 128>
__global__ __launch_bounds__(768, 1) void igemm_m16p_kernel(const IgemmArgs a, const int m_total) {
  using T = bf16;
  using G = M16PGeom<BN_>;
  constexpr int BN = G::BN, CL = G::CL, TH = G::TH, TW = G::TW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* ldsA = reinterpret_cast<T*>(smem_raw);
  T* ldsW = ldsA + 2 * G::A_STAGE_E;
  T* ldsO = ldsW + 2 * G::UNIT_E;
  float* ldsB = reinterpret_cast<float*>(ldsO + G::BM * BN);                 // [128] bias of this output-channel block
  float* ldsT = ldsB + BN;                                                   // [2][cin] scale, shift of the fused input BatchNorm
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cin = a.c0 + a.c1, nch = cin / 32, upt = 3 * nch;

  // ---- this workgroup's tiles: XCD-aware id (blocks b and b + 8 share an XCD), output-channel block = id % n_tiles, a contiguous range
  // of pixel tiles (neighbouring ids = the n_tiles blocks of one range: same XCD, same activations at about the same time)
  int bid;
  {
    const int G_ = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q = G_ >> 3, rem = G_ & 7;
    bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (orig >> 3);
  }
  const int nbase = (bid % a.n_tiles) * BN;
  int m_lo, ntl;
  {
    const int rr = bid / a.n_tiles, R = gridDim.x / a.n_tiles;
    const int per = m_total / R, extra = m_total % R;
    m_lo = rr * per + (rr < extra ? rr : extra);
    ntl = per + (rr < extra ? 1 : 0);
  }
  if (ntl <= 0) return;
  for (int i = tid; i < BN; i += 768) ldsB[i] = a.bias ? a.bias[nbase + i] : 0.f;
  const bool xaff = !BST && a.in_scale != nullptr;
  if (xaff) {
    for (int i = tid; i < cin; i += 768) { ldsT[i] = a.in_scale[i]; ldsT[cin + i] = a.in_shift[i]; }
  }
  if constexpr (BST) {                 // (no input transform in these launches: the table holds the scale / shift of the BatchNorm whose sums are formed)
    for (int i = tid; i < BN; i += 768) { ldsT[i] = a.bst_scale[nbase + i]; ldsT[BN + i] = a.bst_shift[nbase + i]; }
  }
  auto tile_origin = [&](int v, int& n0, int& y0, int& x0) __attribute__((always_inline)) {
    const int tx = v % a.tiles_x; v /= a.tiles_x;
    const int ty = v % a.tiles_y;
    n0 = v / a.tiles_y; y0 = ty * TH; x0 = tx * TW;
  };

  if (wave >= 8) {
    // ================================================================ staging / draining waves (256 threads)
    constexpr int NS = G::NS, AI = G::AI;
    const int sid = tid - 512;
    // priority of the staging waves (option m16p_prio, per launch kind: a.dbg).  They are the youngest waves of their SIMDs: at equal priority the
    // matrix waves' MFMAs win the vector-issue arbitration (profiles/r06_m16p_stamp_variants.txt)
    if (a.dbg == 1) __builtin_amdgcn_s_setprio(1);
    else if (a.dbg == 2) __builtin_amdgcn_s_setprio(2);
    else if (a.dbg == 3) __builtin_amdgcn_s_setprio(3);
    const int nc = ntl * nch;                                                // chunks of this workgroup's stream
    // activation items: item it = (pixel octet it >> 5, lane-in-octet it & 7, slot ((it >> 3) + it) & 3): the 8 lanes of a ds_write_b128 group
    // hold 8 consecutive pixels of one plane, a wave-instruction's global loads cover 16 pixels x 64 contiguous bytes
    const int slot_t = ((sid >> 3) + sid) & 3;
    // (halo row / column of item j, recomputed where a tile's offsets are set up: six registers less across the stream)
    auto item_lc = [&](int j, int& L, int& c) __attribute__((always_inline)) -> bool {
      int it = sid + j * NS;
      asm volatile("" : "+v"(it));
      const int pix = ((it >> 5) << 3) | (it & 7);
      L = pix / CL; c = pix - L * CL;
      return it < G::A_ITEMS && L < G::RL;
    };
    Raw8<T> ra[2][AI];
    unsigned vmask[2] = {0u, 0u};
    // ---- load side of the chunk stream: cursor (l_v, l_x) = (pixel tile, chunk in the tile) of the chunk being loaded; per tile the items'
    // pixel offsets from the halo origin (outside the image: the tile's first pixel, zeroed at the LDS store -- no load in a lane-dependent
    // branch) and their validity; per chunk only the source pointer changes (two sources = the never materialised concat([skip, up]))
    int l_v = m_lo, l_x = 0;
    unsigned l_pix[AI], l_vm = 0;
    long long l_pb = 0;                                                     // halo origin of tile l_v in pixels (may lie before the tensor: never dereferenced)
    auto load_setup = [&]() __attribute__((always_inline)) {
      int n0, y0, x0; tile_origin(l_v, n0, y0, x0);
      const int ylo = 1 - y0, yhi = a.h - y0 + 1, xlo = 1 - x0, xhi = a.w_ - x0 + 1;          // halo rows / columns inside the image
      l_pb = (long long)(n0 * a.h + y0 - 1) * a.w_ + (x0 - 1);
      unsigned vm = 0;
#pragma unroll
      for (int j = 0; j < AI; ++j) {
        int L, c;
        const bool ok = item_lc(j, L, c) && L >= ylo && L < yhi && c >= xlo && c < xhi;
        vm |= (ok ? 1u : 0u) << j;
        l_pix[j] = ok ? (unsigned)(__mul24(L, a.w_) + c) : (unsigned)(a.w_ + 1);
      }
      l_vm = vm;
    };
    // items J0 ... J0 + NJ - 1 of the cursor's chunk -> register set SET
    auto load_items = [&](auto SET, auto J0C, auto NJC) __attribute__((always_inline)) {
      constexpr int S = decltype(SET)::value, J0 = decltype(J0C)::value, NJ = decltype(NJC)::value;
      const int cg0 = l_x * 32;
      const bool second = cg0 >= a.c0;
      const int cs = second ? a.c1 : a.c0;
      const T* src = second ? reinterpret_cast<const T*>(a.x1) + (cg0 - a.c0) : reinterpret_cast<const T*>(a.x0) + cg0;
      const T* xb = src + l_pb * cs;
#pragma unroll
      for (int j = J0; j < J0 + NJ; ++j) {
        unsigned off = (unsigned)__umul24(l_pix[j], (unsigned)cs) + (unsigned)slot_t * 8u;
        asm volatile("" : "+v"(off));
        ra[S][j] = gload8<T>(xb + off);
      }
      constexpr unsigned M = ((1u << NJ) - 1u) << J0;
      vmask[S] = (vmask[S] & ~M) | (l_vm & M);
    };
    auto load_advance = [&]() __attribute__((always_inline)) {
      if (++l_x == nch) { l_x = 0; ++l_v; load_setup(); }                    // (one tile beyond the range at the very end: computed, never loaded)
    };
    // ---- store side: items J0 ... of the chunk in set SET (chunk-in-tile x: its scale / shift) -> stage SET
    // (the scale / shift of a chunk's eight channels: read once per CHUNK -- at its first item pair -- and kept for its three intervals; read per
    //  interval they were a second serial LDS round trip in front of the stores, behind the matrix waves' fragment bursts)
    float4 rs[4];
    auto load_table = [&](int x) __attribute__((always_inline)) {
      if (xaff) {
        const float4* tp = reinterpret_cast<const float4*>(ldsT + x * 32 + slot_t * 8);
        const float4* hp = reinterpret_cast<const float4*>(ldsT + cin + x * 32 + slot_t * 8);
        rs[0] = tp[0]; rs[1] = tp[1]; rs[2] = hp[0]; rs[3] = hp[1];
      }
    };
    auto store_items = [&](int x, auto SET, auto J0C, auto NJC) __attribute__((always_inline)) {
      constexpr int S = decltype(SET)::value, J0 = decltype(J0C)::value, NJ = decltype(NJC)::value;
      if (J0 == 0) load_table(x);
      T* d = ldsA + S * G::A_STAGE_E + slot_t * G::PLANE_E;
#pragma unroll
      for (int j = J0; j < J0 + NJ; ++j) {
        Raw8<T> v = ra[S][j];
        if (xaff) v = affine8r<T>(v, rs, a.in_relu);
        v = select8<T>((vmask[S] >> j) & 1u, v);
        const int it = sid + j * NS;
        const int pix = ((it >> 5) << 3) | (it & 7);
        if (it < G::A_ITEMS) lstore8<T>(d + pix * 8, v);
      }
    };
    // ---- drain: this thread's 16-byte channel group vq of pixels (sid >> 4) + 16 round
    const int vq = sid % G::VPR, pq0 = sid / G::VPR;
    const int cg = nbase + vq * 8;
    float st1[8], st2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { st1[e] = 0.f; st2[e] = 0.f; }
    constexpr bool bst = BST;
    const bool bsec = BST && a.bst_y1 != nullptr && cg >= a.bst_split;
    const unsigned yld = BST ? (unsigned)(bsec ? a.bst_ld1 : a.bst_ld) : 0u;
    const unsigned ycol = BST ? (unsigned)(bsec ? cg - a.bst_split : cg) : 0u;   // this thread's first channel inside its raw-output tensor
    // (the sums run as sum gg and sum gg * v; sum gg * xhat = rstd * (sum gg v - mean * sum gg) is formed once per workgroup, in double, at
    //  the flush: the mean and 1 / std of 8 channels are 16 registers this wave does not have;
    //  the scale / shift of the mask test are re-read from LDS per round)
    const bool lin = a.bst_relu == 0;
    int d_tile = -1, d_round = G::ROUNDS;                                    // tile being drained, its next round
    // that tile's first pixel in the output tensor and in the raw-output tensor(s): wave-uniform pointers, the lane part is a 32-bit offset
    T* d_o = nullptr; const T* d_y0 = nullptr; const T* d_y1 = nullptr;
    uint4 yv[BST ? 4 : 1];
    // per-thread part of a round's addresses: pixel pq0 of the round's 16, channel group vq -- a kernel constant (32-bit element offsets);
    // the tile and the round only add wave-uniform terms (round r: tile row r >> 1, column half r & 1)
    const unsigned o_lane = (unsigned)pq0 * (unsigned)a.ldy + (unsigned)cg;
    const unsigned y_lane = BST ? (unsigned)pq0 * yld + ycol : 0u;
    const unsigned i_lane = (unsigned)(pq0 * BN + ((vq ^ (pq0 & G::SWM)) << 3));       // image: pixel q = pq0 + PPR r -> q & SWM = pq0 & SWM
    auto drain_begin = [&](int v) __attribute__((always_inline)) {
      int n0, y0, x0; tile_origin(v, n0, y0, x0);
      const size_t p0 = (size_t)(n0 * a.h + y0) * a.w_ + x0;
      d_o = reinterpret_cast<T*>(a.y) + p0 * a.ldy;
      if constexpr (BST) {
        d_y0 = reinterpret_cast<const T*>(a.bst_y) + p0 * a.bst_ld;
        d_y1 = a.bst_y1 ? reinterpret_cast<const T*>(a.bst_y1) + p0 * a.bst_ld1 : d_y0;
      }
      d_tile = v; d_round = 0;
    };
    auto round_off = [&](int r, unsigned ld) __attribute__((always_inline)) -> unsigned {        // wave-uniform element offset of round r
      return ((unsigned)((r * G::PPR) >> 5) * (unsigned)a.w_ + (unsigned)((r * G::PPR) & 31)) * ld;
    };
    // One interval's share of the drain, R rounds at a time (R = rpi is 1, 2 or 4 and divides the 16 rounds of a tile: no partial groups, no
    // per-round conditions -- the round-5 form with a run-time count kept its four 16-byte pieces in SCRATCH once nothing else pinned them in
    // registers).  rounds: store R rounds of the image (their raw outputs were requested one interval earlier); prefetch: request the raw outputs
    // of the next R rounds (fused BatchNorm-backward sums only).
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));      // (a first-class vector type: an array of uint4 structs that is only copied ended up in scratch)
    u32x4_ dv[4];
    // the interval's image pieces: requested FIRST in the interval (with the table reads of a chunk's first pair: one LDS round trip per interval)
    auto drain_read = [&](auto RC, bool rounds) __attribute__((always_inline)) {
      constexpr int R = decltype(RC)::value;
      if (rounds && d_round < G::ROUNDS) {
#pragma unroll
        for (int r = 0; r < R; ++r) dv[r] = *reinterpret_cast<const u32x4_*>(ldsO + (d_round + r) * G::PPR * BN + i_lane);
      }
    };
    auto drain_step = [&](auto RC, bool rounds, bool prefetch) __attribute__((always_inline)) {
      constexpr int R = decltype(RC)::value;
      if (rounds && d_round < G::ROUNDS) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          unsigned off = round_off(d_round + r, (unsigned)a.ldy) + o_lane;
          asm volatile("" : "+v"(off));
          *reinterpret_cast<u32x4_*>(d_o + off) = dv[r];
          if constexpr (BST) {
            const bf16x8 d8 = __builtin_bit_cast(bf16x8, dv[r]);
            const bf16x8 y8 = __builtin_bit_cast(bf16x8, yv[r]);
            int toff = vq * 8;
            asm volatile("" : "+v"(toff));
            const float4 c0 = *reinterpret_cast<const float4*>(ldsT + toff), c1 = *reinterpret_cast<const float4*>(ldsT + toff + 4);
            const float4 h0 = *reinterpret_cast<const float4*>(ldsT + BN + toff), h1 = *reinterpret_cast<const float4*>(ldsT + BN + toff + 4);
            const float bsc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, bsh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float fv = (float)d8[e], v = (float)y8[e];
              const float gg = (v * bsc[e] + bsh[e] > 0.f || lin) ? fv : 0.f;
              st1[e] += gg; st2[e] += gg * v;
            }
          }
          // (forward launches: the statistics of the stored values are formed by the MATRIX waves, as MFMAs on this image -- see stats_mfma)
        }
        d_round += R;
      }
      if constexpr (BST) {
        if (prefetch && d_round < G::ROUNDS) {
#pragma unroll
          for (int r = 0; r < R; ++r) {
            unsigned off = round_off(d_round + r, yld) + y_lane;
            asm volatile("" : "+v"(off));
            yv[r] = *reinterpret_cast<const uint4*>((bsec ? d_y1 : d_y0) + off);
          }
        }
      }
    };
    const std::integral_constant<int, 1> R1{};
    const std::integral_constant<int, 2> R2{};
    const std::integral_constant<int, 4> R4{};
    // rounds per interval: a tile's image is drained in the intervals 1 ... upt - 2 of the next tile (interval 0 only requests the first raw
    // outputs, the last interval is when the matrix waves write the next image)
    const int rpi = (G::ROUNDS + upt - 3) / (upt - 2);
    const std::integral_constant<int, 0> S0{};
    const std::integral_constant<int, 1> S1{};
    const std::integral_constant<int, 0> I0{};
    const std::integral_constant<int, 2> I2{};
    const std::integral_constant<int, 4> I4{};
    const std::integral_constant<int, AI> IALL{};
    static_assert(AI == 6, "three item pairs per chunk, one per interval");
    int s_x = 0;                                                             // chunk-in-tile of the next chunk to store
    load_setup();
    load_items(S0, I0, IALL); load_advance();
    load_items(S1, I0, IALL); load_advance();                               // (nc >= 2: cin % 64 == 0)
    __syncthreads();                                                         // (1) tables in LDS
    store_items(0, S0, I0, IALL); s_x = 1;
    if (2 < nc) { load_items(S0, I0, IALL); load_advance(); }
    __syncthreads();                                                         // (2) chunk 0 staged, weight unit 0 landed
    int gc = 0;                                                              // chunk being multiplied
#ifdef SATCV_STAMP_M16P
    unsigned long long z0, z1, z2, zs[4] = {0, 0, 0, 0};
    PSTAMP(z0);
#endif
    // one chunk = three intervals; PAR = parity of the chunk (= of the stage it is read from).  Chunk gc + 1 moves from its register set to
    // the other stage TWO ITEMS PER INTERVAL (the stage fell free at the barrier that started chunk gc), each pair reloaded at once with its
    // items of chunk gc + 3: the vector work of the stream is spread evenly over the intervals, like the drain's rounds
    // MAIN: the steady state -- every pair is stored and reloaded unconditionally.  hipcc sizes an `s_waitcnt vmcnt(N)` by the FEWEST
    // vector-memory operations on any path between a load and its use; with the stream's end handled by conditions inside the loop, a path
    // without reloads exists and every use waited for (nearly) everything in flight -- the drain's global stores of the interval before
    // included (`vmcnt(1)` in front of each reload, ~1 us per interval).  The steady-state loop has no such path (>= 10 younger loads when
    // a pair is used); the last chunks run in a second copy of the body with the conditions.
    auto chunk_intervals = [&](auto PAR, auto MAINC, int x, int t) __attribute__((always_inline)) {
      constexpr int P = decltype(PAR)::value;
      constexpr bool MAIN = decltype(MAINC)::value;
      const std::integral_constant<int, P ^ 1> SN{};
      const bool st = (MAIN || gc + 1 < nc) && !PABL(8), ld = (MAIN || gc + 3 < nc) && !PABL(8) && !PABL(64);
      auto pair = [&](auto J0C) __attribute__((always_inline)) {
        if (st) store_items(s_x, SN, J0C, I2);
        if (ld) load_items(SN, J0C, I2);
      };
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iv = x * 3 + ky;                                           // interval inside the tile
        const bool dr = d_tile >= 0 && !PABL(16) && iv > 0 && iv < upt - 1;
        if constexpr (BST) { if (rpi >= 2) drain_read(R2, dr); else drain_read(R1, dr); }
        else { if (rpi == 4) drain_read(R4, dr); else if (rpi == 2) drain_read(R2, dr); else drain_read(R1, dr); }
        if (ky == 0) pair(I0);
        else if (ky == 1) pair(I2);
        else {
          pair(I4);
          if (st && ++s_x == nch) s_x = 0;
          if (ld) load_advance();
        }
        if (d_tile >= 0 && !PABL(16) && iv < upt - 1) {
          // (interval 0 only requests the first raw outputs; the last interval is when the matrix waves write the next image)
          if constexpr (BST) {
            // (one compiled form of the sums: four rounds per interval -- 64 input channels -- run as two pairs)
            if (rpi >= 2) { drain_step(R2, iv > 0, true); if (rpi == 4) { drain_read(R2, iv > 0); drain_step(R2, iv > 0, true); } }
            else drain_step(R1, iv > 0, true);
          } else {
            if (rpi == 4) drain_step(R4, iv > 0, true);
            else if (rpi == 2) drain_step(R2, iv > 0, true);
            else drain_step(R1, iv > 0, true);
          }
        }
        PSTAMP(z1);
        __syncthreads();
        PSTAMP(z2);
#ifdef SATCV_STAMP_M16P
        zs[0] += z1 - z0; zs[2] += z2 - z1; zs[3] += 1; z0 = z2;
#endif
        if (iv == upt - 1) drain_begin(m_lo + t);                            // the barrier just passed published tile t's image
      }
      ++gc;
    };
    {
      int t = 0, x = 0;
      auto advance = [&]() __attribute__((always_inline)) { x += 2; if (x == nch) { x = 0; ++t; } };
      while (gc + 4 < nc) {                                                  // both chunks of the pair reload: gc + 1 + 3 < nc
        chunk_intervals(S0, std::true_type{}, x, t);
        chunk_intervals(S1, std::true_type{}, x + 1, t);
        advance();
      }
      while (gc < nc) {
        chunk_intervals(S0, std::false_type{}, x, t);
        chunk_intervals(S1, std::false_type{}, x + 1, t);
        advance();
      }
    }
#ifdef SATCV_STAMP_M16P
    if (blockIdx.x < 8 && lane == 0) for (int i = 0; i < 4; ++i) g_stamp_m16p[blockIdx.x][wave][i] = zs[i];
#endif
    // the last tile's image
    if (!PABL(16)) for (int r = 0; r < G::ROUNDS; ++r) { drain_step(R1, false, true); drain_read(R1, true); drain_step(R1, true, false); }
    if constexpr (!BST) return;                                              // (the forward statistics leave with the matrix waves)
    __syncthreads();                                                         // (3) staging waves only: the matrix waves do not take part
    if (a.stats) {
      // sums of this workgroup: the threads of a channel group summed through LDS in a fixed order, one pair of atomics per channel
      float* r2 = reinterpret_cast<float*>(smem_raw);                        // [16][256]
#pragma unroll
      for (int e = 0; e < 8; ++e) { r2[e * NS + sid] = st1[e]; r2[(8 + e) * NS + sid] = st2[e]; }
      __syncthreads();
      if (sid < BN) {
        const int gq = sid >> 3, e = sid & 7;
        double t1 = 0.0, t2 = 0.0;
        for (int k = 0; k < NS / G::VPR; ++k) { t1 += (double)r2[e * NS + k * G::VPR + gq]; t2 += (double)r2[(8 + e) * NS + k * G::VPR + gq]; }
        if constexpr (BST) t2 = (double)a.bst_rstd[nbase + sid] * (t2 - (double)a.bst_mean[nbase + sid] * t1);
        satcv_stat_t* rowp = a.stats + (size_t)(blockIdx.x % SATCV_STAT_ROWS) * 2 * a.stats_ld;
        atomicAdd(rowp + nbase + sid, (satcv_stat_t)t1);
        atomicAdd(rowp + a.stats_ld + nbase + sid, (satcv_stat_t)t2);
      }
    }
    return;
  }

  // ================================================================ matrix waves (512 threads): wave tile 64 pixels x 64 channels
  const int wm = wave >> 1, wn = wave & 1, g4 = lane >> 4, l16 = lane & 15;
  int a_off[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int q = wm * 64 + m * 16 + l16;
    a_off[m] = g4 * G::PLANE_E + ((q >> 5) * CL + (q & 31)) * 8;
  }
  constexpr int NBW = G::NBW;
  const int b_lane = (g4 * BN + wn * (BN / 2) + l16) * 8;
  const T* wp = reinterpret_cast<const T*>(a.w);
  const unsigned lds_w = lds_addr_of(ldsW);
  // this wave's pieces p = wave + 8 r of unit (chunk x, tap row ky): half (p & 1) of the 2-KB row of (tap-in-row, slot) = ((p >> 1) >> 2,
  // (p >> 1) & 3); contiguous in the packed image [tap][cin / 8][cout_pad][8] and in the LDS slot
  auto dma_unit = [&](int x, int ky, int slot) __attribute__((always_inline)) {
    constexpr int HALVES = BN / 64, NP = 12 * HALVES;                        // 1-KB pieces of a unit: 64 channels x 8 x 2 bytes each
#pragma unroll
    for (int r = 0; r < (NP + 7) / 8; ++r) {
      const int p = wave + 8 * r, run = p / HALVES;
      if (p >= NP) break;                                                    // (wave-uniform: 12 pieces on 8 waves)
      const int tap = ky * 3 + (run >> 2), sl = run & 3;
      const size_t off = ((size_t)(tap * (cin / 8) + x * 4 + sl) * a.cout_pad + nbase + (p % HALVES) * 64) * 8;
      // (wave-uniform by construction; the read-first-lane pair makes it so for the register allocator too -- an "s" operand)
      const unsigned long long ga = (unsigned long long)(uintptr_t)(wp + off);
      const unsigned long long gu_ = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(ga >> 32)) << 32) |
                                     (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)ga);
      lds_dma16(reinterpret_cast<const void*>((uintptr_t)gu_), (unsigned)lane * 16u,
                (unsigned)__builtin_amdgcn_readfirstlane(lds_w + (unsigned)(slot * G::UNIT_E * 2) + (unsigned)p * 1024u));
    }
  };
  f32x4 acc[4][NBW];
  auto acc_init = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int n = 0; n < NBW; ++n) {
      int boff = wn * (BN / 2) + n * 16 + 4 * g4;
      asm volatile("" : "+v"(boff));                                         // (re-read per tile: 16 registers less across the K loop)
      const float4 b4 = *reinterpret_cast<const float4*>(ldsB + boff);
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m][n] = f32x4{b4.x, b4.y, b4.z, b4.w};
    }
  };
  // ---- round 6: the forward launches' BatchNorm statistics (sum / sum of squares of the STORED bf16 values) as MFMAs on the staging image.
  // In the staging waves they were 16 vector instructions per drained 16 bytes -- a third of those waves' instruction stream, and the staging
  // waves' stream, not the matrix pipe, sets the pace of the forward launches (profiles/r05_m16p_interval_stamps.txt: 3 260 against 2 350 ticks per
  // interval with the fused input BatchNorm).  Matrix wave w owns channels 16 w ... 16 w + 15 of the block: per 32 pixels ONE fragment
  // F[c][p] (two transposing reads of the [pixel][channel] image, the drain's swizzle) feeds  G += F F^T  (diagonal = sum of squares) and
  // S += F 1  (row sums): 16 MFMAs and 16 reads per tile and wave (0.7 % of the tile's MFMAs), 8 accumulator registers, fp32 sums over the
  // workgroup's tiles, one pair of double atomics per channel at the end.
  f32x4 g_sq = {0.f, 0.f, 0.f, 0.f}, g_sm = {0.f, 0.f, 0.f, 0.f};
  const bool mstats = !BST && a.stats != nullptr && wave < BN / 16;
  auto stats_mfma = [&]() __attribute__((always_inline)) {
    const int q = l16 >> 2, pcol = l16 & 3;
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;
    const int gidx = wave * 4 + pcol;                                        // 8-byte granule of channels 16 w + 4 pcol ... + 3
#pragma unroll 2
    for (int kp = 0; kp < 8; ++kp) {
      const int p0 = kp * 32 + g4 * 8 + q, p1 = p0 + 4;
      const bf16x4 lo = m16p_tr_read(ldsO + p0 * BN + ((gidx ^ (2 * (p0 & G::SWM))) << 2));
      const bf16x4 hi = m16p_tr_read(ldsO + p1 * BN + ((gidx ^ (2 * (p1 & G::SWM))) << 2));
      const bf16x8 fr = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      g_sq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr, fr, g_sq, 0, 0, 0);
      g_sm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr, ones, g_sm, 0, 0, 0);
    }
  };
  dma_unit(0, 0, 0);
  dma_wait_all();
  __syncthreads();                                                           // (1)
  acc_init();
  __syncthreads();                                                           // (2)
  const int nu = ntl * upt;
  int gu = 0;
#ifdef SATCV_STAMP_M16P
  unsigned long long y0, y1, y2, y3, ys[4] = {0, 0, 0, 0};
  PSTAMP(y0);
#endif
  for (int t = 0; t < ntl; ++t) {
    for (int x = 0; x < nch; ++x) {
      if (mstats && x == 0 && t > 0) stats_mfma();                           // (the barrier that ended tile t - 1 published its image)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int slot = (x * 3 + ky) & 1;                                   // (upt is even: a tile starts on slot 0)
        if (gu + 1 < nu && !PABL(4)) {                                       // the next unit's weights: same output-channel block, next tap row / chunk / tile
          const int nky = ky == 2 ? 0 : ky + 1;
          const int nx = ky == 2 ? (x + 1 == nch ? 0 : x + 1) : x;
          dma_unit(nx, nky, slot ^ 1);
        }
        {
          FragT<T> af[2][4], bf[2][NBW];
          const T* Ab = ldsA + (x & 1) * G::A_STAGE_E + ky * CL * 8;
          const T* Wb = ldsW + slot * G::UNIT_E + b_lane;
          auto read_step = [&](int kx, int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int m = 0; m < 4; ++m) af[buf][m] = lds_frag<T>(Ab + a_off[m] + kx * 8);
#pragma unroll
            for (int n = 0; n < NBW; ++n) bf[buf][n] = lds_frag<T>(Wb + (kx * 4 * BN + n * 16) * 8);
          };
          if (!PABL(2) || gu == 0) read_step(0, 0);
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            asm volatile("" ::: "memory");
            if (kx + 1 < 3 && (!PABL(2) || gu == 0)) read_step(kx + 1, (kx + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
              for (int n = 0; n < NBW; ++n) {
                if constexpr (!PABL(1)) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[kx & 1][n].v, af[kx & 1][m].v, acc[m][n], 0, 0, 0);
                else asm volatile("" : "+v"(acc[m][n]) : "v"(bf[kx & 1][n].v), "v"(af[kx & 1][m].v));
              }
          }
        }
        if (ky == 2 && x + 1 == nch && !PABL(32)) {
          // tile done: rows 4 g4 ... + 3 of a block are four consecutive output channels of pixel l16 -> 8 bytes of the staging image
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const int px = wm * 64 + m * 16 + l16;
#pragma unroll
            for (int n = 0; n < NBW; ++n) {
              const int gr = (wn * (BN / 8) + n * 4 + g4) ^ (2 * (l16 & G::SWM));
              bf16x4 o;
#pragma unroll
              for (int j = 0; j < 4; ++j) o[j] = (bf16)acc[m][n][j];
              *reinterpret_cast<bf16x4*>(ldsO + px * BN + gr * 4) = o;
            }
          }
          acc_init();
        }
        PSTAMP(y1);
        if (!PABL(128)) dma_wait_all();
        PSTAMP(y2);
        __syncthreads();
        PSTAMP(y3);
#ifdef SATCV_STAMP_M16P
        ys[0] += y1 - y0; ys[1] += y2 - y1; ys[2] += y3 - y2; ys[3] += 1; y0 = y3;
#endif
        ++gu;
      }
    }
  }
#ifdef SATCV_STAMP_M16P
  if (blockIdx.x < 8 && lane == 0) for (int i = 0; i < 4; ++i) g_stamp_m16p[blockIdx.x][wave][i] = ys[i];
#endif
  if (mstats) {
    stats_mfma();                                                            // the last tile's image (nobody writes it any more)
    // G[c][c'] and S[c][*]: register r of lane (g4, l16) is row 4 g4 + r, column l16 -- the lanes with l16 >> 2 == g4 hold G's diagonal
    // element and the row sum of channel 16 w + l16 in register l16 & 3
    if ((l16 >> 2) == g4) {
      const int r = l16 & 3;
      const float s2 = r == 0 ? g_sq[0] : r == 1 ? g_sq[1] : r == 2 ? g_sq[2] : g_sq[3];
      const float s1 = r == 0 ? g_sm[0] : r == 1 ? g_sm[1] : r == 2 ? g_sm[2] : g_sm[3];
      satcv_stat_t* rowp = a.stats + (size_t)(blockIdx.x % SATCV_STAT_ROWS) * 2 * a.stats_ld;
      atomicAdd(rowp + nbase + wave * 16 + l16, (satcv_stat_t)s1);
      atomicAdd(rowp + a.stats_ld + nbase + wave * 16 + l16, (satcv_stat_t)s2);
    }
  }
}

// ------------------------------------------------------------------ host side
// bf16 3x3, dilation 1, plain NHWC in and out, whole 8 x 32 tiles, Cin % 64 == 0 (<= the LDS table), Cout % 128 == 0, bias + statistics or
// fused BatchNorm-backward sums, no ReLU / multiplier / pool / accumulation in the epilogue.  SATCV_ERR_UNSUPPORTED otherwise (the caller
// continues with the one-tile-per-workgroup kernels).
template <int BN>
static int m16p_launch_bn(IgemmArgs& a, hipStream_t st, bool dry) {
  using G = M16PGeom<BN>;
  if (!(a.kh == 3 && a.kw == 3 && a.dil == 1 && a.stride == 1 && a.mode_in == 0 && a.mode_out == 0 && !a.pool_y && !a.out_scale && !a.out_relu && !a.accumulate))
    return SATCV_ERR_UNSUPPORTED;
  const int cin = a.c0 + a.c1;
  if (a.h % G::TH != 0 || a.w_ % G::TW != 0) return SATCV_ERR_UNSUPPORTED;
  if (cin % 64 != 0 || (a.x1 && a.c0 % 32 != 0) || a.cout % BN != 0 || a.cout_pad % 64 != 0 || a.cout_pad < a.cout || a.cstat != a.cout) return SATCV_ERR_UNSUPPORTED;
  if (((uintptr_t)a.w % 16) != 0 || a.ldy % 8 != 0 || ((uintptr_t)a.y % 16) != 0 || ((uintptr_t)a.x0 % 16) != 0 || (a.x1 && ((uintptr_t)a.x1 % 16) != 0)) return SATCV_ERR_UNSUPPORTED;
  if (a.c0 % 8 != 0 || a.c1 % 8 != 0) return SATCV_ERR_UNSUPPORTED;
  if (a.bst_y) {
    if (a.bst_ld % 8 != 0 || ((uintptr_t)a.bst_y % 16) != 0 || (a.bst_y1 && (a.bst_split % 8 != 0 || a.bst_ld1 % 8 != 0 || ((uintptr_t)a.bst_y1 % 16) != 0))) return SATCV_ERR_UNSUPPORTED;
    if ((long long)G::TH * a.w_ * (a.bst_ld > a.bst_ld1 ? a.bst_ld : a.bst_ld1) >= (1LL << 31)) return SATCV_ERR_UNSUPPORTED;
  }
  if ((long long)(G::TH + 2) * a.w_ * (a.c0 > a.c1 ? a.c0 : a.c1) >= (1LL << 23)) return SATCV_ERR_UNSUPPORTED;      // 24-bit multiplies of the halo offsets
  if ((long long)G::TH * a.w_ * a.ldy >= (1LL << 31)) return SATCV_ERR_UNSUPPORTED;
  if (a.bst_y && a.in_scale) return SATCV_ERR_UNSUPPORTED;
  const size_t lds = G::FIXED_BYTES + (a.bst_y ? (size_t)2 * G::BN * sizeof(float) : a.in_scale ? (size_t)2 * cin * sizeof(float) : 0);
  if (lds > 160 * 1024) return SATCV_ERR_UNSUPPORTED;
  static int ncu = 0;
  if (!ncu) {
    int dev = 0; hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) { satcv_set_error("igemm_m16p: device query failed"); return SATCV_ERR_HIP; }
    ncu = p.multiProcessorCount;
  }
  a.halh = a.halw = 1;
  a.tiles_x = a.w_ / G::TW; a.tiles_y = a.h / G::TH;
  a.rpi = G::TH; a.imgs = 1; a.ngroups = a.n;
  a.seg = G::RL; a.rl = G::RL; a.cl = G::CL; a.pitch = G::CL;
  a.n_tiles = a.cout / BN;
  a.cpt = cin / 32; a.nchunks = a.cpt; a.taploop = 0; a.halh_tl = a.halw_tl = 1;
  a.ksplit = 1; a.kslab = nullptr;
  a.dbg = a.bst_y ? g_opt_m16p_prio % 10 : a.in_scale ? (g_opt_m16p_prio / 10) % 10 : (g_opt_m16p_prio / 100) % 10;
  if (BN == 64) {      // (half the MFMAs per staged byte: the staging waves set the pace of every launch kind -- SATCV_M16P_PRIO64, same digits)
    static const int p64 = [] { const char* e = getenv("SATCV_M16P_PRIO64"); return e ? atoi(e) : -1; }();
    if (p64 >= 0) a.dbg = a.bst_y ? p64 % 10 : a.in_scale ? (p64 / 10) % 10 : (p64 / 100) % 10;
  }
  const long long m_total = (long long)a.n * a.tiles_y * a.tiles_x;
  if (m_total <= 0 || m_total > 0x7fffffffLL || a.n_tiles > ncu) return SATCV_ERR_UNSUPPORTED;
  long long ranges = ncu / a.n_tiles;
  if (ranges > m_total) ranges = m_total;
  // (one tile per workgroup has nothing to pipeline across: the one-tile kernels, with their higher occupancy of waves per tile, keep those;
  //  option m16p = 2 sends every eligible launch here)
  if (g_opt_m16p < 2 && m_total < 2 * ranges) return SATCV_ERR_UNSUPPORTED;
  if (dry) return SATCV_OK;
  auto kern = a.bst_y ? igemm_m16p_kernel<true, BN> : igemm_m16p_kernel<false, BN>;
  { const int rc = satcv_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds); if (rc) return rc; }
  hipLaunchKernelGGL(kern, dim3((unsigned)(ranges * a.n_tiles)), dim3(768), lds, st, a, (int)m_total);
  ++g_m16p_launches;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { satcv_set_error("igemm_m16p launch: %s", hipGetErrorString(e)); return SATCV_ERR_HIP; }
  return SATCV_OK;
}

// SATCV_M16P_BN64=0: the 64-filter layers stay on the one-tile kernels (A/B switch of the round-6 64-channel block)
int igemm_m16p_launch(IgemmArgs& a, int dtype, hipStream_t st, bool dry) {
  if (dtype != SATCV_BF16) return SATCV_ERR_UNSUPPORTED;
  if (!g_opt_m16p) return SATCV_ERR_UNSUPPORTED;
  if (a.cout % 128 == 0) return m16p_launch_bn<128>(a, st, dry);
  static const int bn64 = [] { const char* e = getenv("SATCV_M16P_BN64"); return e ? atoi(e) : 1; }();
  if (bn64 && a.cout % 64 == 0) return m16p_launch_bn<64>(a, st, dry);
  return SATCV_ERR_UNSUPPORTED;
}
