/* Sanitizer driver of the HOST side of the C ABI (tests/test_host_cpu.py::test_host_side_of_the_c_abi_under_asan_ubsan).
 *
 * Linked against csrc/_obj_asan/libsatcv_hostasan.so -- every source of the library compiled with `--cuda-host-only
 * -fsanitize=address,undefined`: no kernel exists in it, nothing can launch -- and itself built with the same clang and sanitizers.
 * It drives the code that runs on the host before any launch: descriptor validation (null / zeroed / inconsistent descriptors must be
 * REFUSED with a message, never dereferenced), the tile / split-K / weight-gradient planning behind the dry-run and workspace queries over
 * a sweep of shapes (integer overflow, out-of-range table indices, misaligned pointers), the job-size queries of the batched launches,
 * the option table, and the CRC-32C of the TFRecord framing against a bitwise restatement at every length and alignment up to 70 bytes.
 * Exit code 0 and no sanitizer report = pass.  No GPU is touched. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "satcv.h"

/* the library was linked without device code: its module constructors must not hand (empty) device binaries to the HIP runtime.  The
 * executable's definitions of the registration entry points take precedence over libamdhip64's */
void** __hipRegisterFatBinary(const void* data) { static void* handle; (void)data; return &handle; }
void __hipUnregisterFatBinary(void** h) { (void)h; }
void __hipRegisterFunction(void** h, const void* f, char* df, const char* dn, unsigned tl, void* tid, void* bid, void* bd, void* gd, int* ws) {
  (void)h; (void)f; (void)df; (void)dn; (void)tl; (void)tid; (void)bid; (void)bd; (void)gd; (void)ws; }
void __hipRegisterVar(void** h, void* v, char* hv, char* dv, int ext, size_t size, int constant, int global) {
  (void)h; (void)v; (void)hv; (void)dv; (void)ext; (void)size; (void)constant; (void)global; }
void __hipRegisterManagedVar(void* h, void** p, void* init, const char* name, size_t size, unsigned align) {
  (void)h; (void)p; (void)init; (void)name; (void)size; (void)align; }

static int fails = 0;
#define EXPECT(cond, ...) do { if (!(cond)) { fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); ++fails; } } while (0)

static uint32_t crc32c_bitwise(const unsigned char* p, size_t n, uint32_t crc) {
  crc = ~crc;
  for (size_t i = 0; i < n; ++i) {
    crc ^= p[i];
    for (int k = 0; k < 8; ++k) crc = (crc >> 1) ^ (0x82F63B78u & (0u - (crc & 1u)));
  }
  return ~crc;
}

int main(void) {
  /* ---- identification and the error channel */
  EXPECT(satcv_version() && strlen(satcv_version()) > 0, "version string");
  EXPECT(satcv_last_error() != NULL, "last_error must never be NULL");

  /* ---- CRC-32C (masked CRC of the TFRecord length / payload fields: utils/prediction_tools.py:159-226 readers) */
  unsigned char buf[96];
  for (int i = 0; i < 96; ++i) buf[i] = (unsigned char)(i * 37 + 11);
  for (int off = 0; off < 9; ++off)
    for (int n = 0; n <= 70; ++n) {
      const uint32_t a = satcv_crc32c(buf + off, (uint64_t)n, 0), b = crc32c_bitwise(buf + off, (size_t)n, 0);
      EXPECT(a == b, "crc32c len %d offset %d: %08x vs %08x", n, off, a, b);
      const int h = n / 2;      /* continuation */
      EXPECT(satcv_crc32c(buf + off + h, (uint64_t)(n - h), satcv_crc32c(buf + off, (uint64_t)h, 0)) == b, "crc32c continuation len %d", n);
    }
  EXPECT(satcv_crc32c("123456789", 9, 0) == 0xE3069283u, "crc32c check value");

  /* ---- options */
  int32_t v = -1;
  EXPECT(satcv_set_option("no_such_option", 1) != 0, "unknown option must be refused");
  EXPECT(satcv_get_option("no_such_option", &v) != 0, "unknown option must be refused");
  EXPECT(satcv_get_option("igemm_m16", &v) == 0 && v >= 0, "igemm_m16 option");
  EXPECT(satcv_set_option("igemm_m16", 2) == 0 && satcv_get_option("igemm_m16", &v) == 0 && v == 2, "set/get round trip");
  EXPECT(satcv_set_option(NULL, 1) != 0, "NULL key");

  /* ---- null and zeroed descriptors: refused with a message, no dereference */
  satcv_conv_desc cd; memset(&cd, 0, sizeof(cd));
  EXPECT(satcv_conv2d_igemm(NULL, NULL) != 0, "NULL conv desc");
  EXPECT(satcv_conv2d_igemm(&cd, NULL) != 0 && strstr(satcv_last_error(), "null"), "zeroed conv desc: %s", satcv_last_error());
  EXPECT(satcv_conv2d_igemm_pipelined(NULL) == 0 && satcv_conv2d_igemm_pipelined(&cd) == 0, "dry run of an invalid descriptor answers 0");
  satcv_wgrad_desc wd; memset(&wd, 0, sizeof(wd));
  EXPECT(satcv_conv2d_wgrad(NULL, NULL) != 0 && satcv_conv2d_wgrad(&wd, NULL) != 0, "invalid wgrad desc");
  EXPECT(satcv_conv2d_wgrad_workspace(NULL) < 0, "wgrad workspace of NULL");
  satcv_bwdf_desc fd; memset(&fd, 0, sizeof(fd));
  EXPECT(satcv_conv2d_bwd_fused(NULL, NULL) != 0 && satcv_conv2d_bwd_fused(&fd, NULL) != 0, "invalid fused desc");
  EXPECT(satcv_conv2d_bwd_fused_workspace(NULL) < 0 && satcv_conv2d_bwd_fused_workspace(&fd) < 0, "fused workspace of an invalid desc");
  satcv_bnbwd_desc bd; memset(&bd, 0, sizeof(bd));
  EXPECT(satcv_bn_bwd_reduce(&bd, NULL) != 0 && satcv_bn_bwd_apply(&bd, NULL) != 0, "zeroed bn-backward desc");
  satcv_head_desc hd; memset(&hd, 0, sizeof(hd));
  EXPECT(satcv_head_fwd(&hd, NULL) != 0 && satcv_head_bwd(&hd, NULL) != 0, "zeroed head desc");
  satcv_reduce_job rj; memset(&rj, 0, sizeof(rj));
  EXPECT(satcv_reduce_job_items(NULL) < 0 && satcv_reduce_job_items(&rj) < 0, "invalid reduce job");
  EXPECT(satcv_reduce_slabs_batched(NULL, NULL, 0, 0, NULL) != 0, "empty batched reduce");
  EXPECT(satcv_conv2d_wgrad_reduce_job(&wd, &rj) != 0 && satcv_conv2d_bwd_fused_reduce_job(&fd, &rj) != 0, "reduce job of invalid descs");
  satcv_pack_job pj; memset(&pj, 0, sizeof(pj));
  (void)satcv_pack_job_items(&pj);
  EXPECT(satcv_pack_weights_batched(NULL, NULL, 0, 0, SATCV_BF16, NULL) != 0, "empty batched pack");

  /* ---- planning sweeps.  Pointers are fabricated (aligned, never dereferenced: dry runs and workspace queries only read the descriptor) */
  void* fake = (void*)(uintptr_t)0x7f0000001000ull;
  static const int HW[][2] = {{1, 1}, {4, 4}, {8, 8}, {12, 20}, {16, 16}, {20, 24}, {32, 32}, {40, 72}, {64, 64}, {100, 36}, {128, 128}, {256, 256}, {8, 264}, {512, 512}, {1024, 1024}};
  static const int CH[] = {16, 32, 48, 64, 96, 128, 192, 256, 512, 1024, 2048};
  static const int KD[][2] = {{1, 1}, {3, 1}, {3, 2}, {3, 3}, {3, 6}, {3, 12}, {3, 64}, {5, 1}, {7, 1}};
  long long planned = 0, accepted = 0;
  for (unsigned hi = 0; hi < sizeof(HW) / sizeof(HW[0]); ++hi)
    for (unsigned ci = 0; ci < sizeof(CH) / sizeof(CH[0]); ++ci)
      for (unsigned co = 0; co < sizeof(CH) / sizeof(CH[0]); ++co)
        for (unsigned ki = 0; ki < sizeof(KD) / sizeof(KD[0]); ++ki)
          for (int n = 1; n <= 64; n *= 8)
            for (int dt = 0; dt <= 3; ++dt) {
              satcv_conv_desc d; memset(&d, 0, sizeof(d));
              d.x0 = fake; d.w = fake; d.y = fake; d.c0 = CH[ci]; d.ldy = CH[co]; d.n = n; d.h = HW[hi][0]; d.w_ = HW[hi][1];
              d.cout = CH[co]; d.cout_pad = (CH[co] + 63) / 64 * 64; d.kh = d.kw = KD[ki][0]; d.dil = KD[ki][1]; d.cstat = CH[co]; d.dtype = dt;
              d.stride = 1; d.f = 1;
              if (dt == 3 && (d.c0 % 64)) continue;
              accepted += satcv_conv2d_igemm_pipelined(&d);
              ++planned;
              /* two-source input (decoder_block's concat) and a statistics target: other planning branches */
              d.x1 = fake; d.c1 = CH[ci]; d.stats = (satcv_stat_t*)fake; d.stats_ld = CH[co];
              if (!(dt == 3 && (d.c1 % 64))) { accepted += satcv_conv2d_igemm_pipelined(&d); ++planned; }
              if (dt <= 1 && (KD[ki][0] == 1 || KD[ki][0] == 3)) {
                satcv_wgrad_desc w; memset(&w, 0, sizeof(w));
                w.x0 = fake; w.c0 = CH[ci]; w.dy = fake; w.lddy = CH[co]; w.dw = (float*)fake; w.cin = CH[ci]; w.cout = CH[co];
                w.n = n; w.h = HW[hi][0]; w.w_ = HW[hi][1]; w.kh = w.kw = KD[ki][0]; w.dil = KD[ki][1]; w.f = 1; w.dtype = dt;
                const int64_t nb = satcv_conv2d_wgrad_workspace(&w);
                EXPECT(nb != 0, "wgrad workspace 0 bytes");
                if (nb > 0) {
                  w.workspace = (float*)fake; w.workspace_bytes = nb;
                  satcv_reduce_job j;
                  if (satcv_conv2d_wgrad_reduce_job(&w, &j) == 0) {
                    EXPECT(j.nslab >= 1 && j.lanes >= 1 && j.lanes <= 16 && j.kpad >= w.cin && j.npad >= w.cout, "reduce job geometry");
                    EXPECT(satcv_reduce_job_items(&j) > 0, "reduce job items");
                  }
                }
                ++planned;
              }
            }
  EXPECT(planned > 10000 && accepted > 1000, "planning sweep ran (%lld planned, %lld on the pipelined kernel)", planned, accepted);
  /* extreme extents: must be refused or planned without overflow */
  {
    satcv_conv_desc d; memset(&d, 0, sizeof(d));
    d.x0 = fake; d.w = fake; d.y = fake; d.c0 = 1 << 20; d.ldy = 1 << 20; d.n = 1 << 15; d.h = 1 << 15; d.w_ = 1 << 15; d.cout = 1 << 20; d.cout_pad = 1 << 20;
    d.kh = d.kw = 3; d.dil = 1; d.cstat = 1 << 20; d.dtype = SATCV_BF16; d.stride = 1; d.f = 1;
    (void)satcv_conv2d_igemm_pipelined(&d);
    d.n = 0x7fffffff; d.h = 0x7fffffff; d.w_ = 0x7fffffff;
    (void)satcv_conv2d_igemm_pipelined(&d);
  }
  printf("host ABI driver: %lld plans, %lld accepted by the pipelined kernel, %d failures\n", planned, accepted, fails);
  return fails ? 1 : 0;
}
