// Probe: operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x fp8 e4m3, unit scales) on gfx950.
// Hypothesis H1: lane l holds A[row l&31][k = 32*(l>>5) + j], B[k = 32*(l>>5) + j][col l&31], j = byte index 0..31 of the
// 8-dword operand.  Hypothesis H2: k = 16*(l>>5) + j for j<16 and 32 + 16*(l>>5) + (j-16) for j>=16 (two K=32 halves).
#include <hip/hip_runtime.h>
#include <hip/hip_fp8.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __hip_fp8_e4m3 fp8;
using i32x8 = __attribute__((ext_vector_type(8))) int;
using f32x16 = __attribute__((ext_vector_type(16))) float;

__global__ void k(const unsigned char* A, const unsigned char* B, float* C, int hyp) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  union { i32x8 v; unsigned char b[32]; } a, b;
  for (int j = 0; j < 32; ++j) {
    const int kk = hyp == 1 ? 32 * h + j : (j < 16 ? 16 * h + j : 32 + 16 * h + (j - 16));
    a.b[j] = A[r * 64 + kk];          // A[row][k]
    b.b[j] = B[kk * 32 + r];          // B[k][col]
  }
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a.v, b.v, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}

int main() {
  std::vector<unsigned char> A(32 * 64), B(64 * 32);
  std::vector<float> Af(32 * 64), Bf(64 * 32);
  srand(1);
  for (int i = 0; i < 32 * 64; ++i) { float v = (float)(rand() % 7 - 3); Af[i] = v; fp8 q(v); A[i] = *reinterpret_cast<unsigned char*>(&q); }
  for (int i = 0; i < 64 * 32; ++i) { float v = (float)(rand() % 5 - 2); Bf[i] = v; fp8 q(v); B[i] = *reinterpret_cast<unsigned char*>(&q); }
  unsigned char *dA, *dB; float* dC;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dC, 32 * 32 * 4);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  for (int hyp = 1; hyp <= 2; ++hyp) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, hyp);
    std::vector<float> C(32 * 32);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      float s = 0; for (int kk = 0; kk < 64; ++kk) s += Af[i * 64 + kk] * Bf[kk * 32 + j];
      if (s != C[i * 32 + j]) ++bad;
    }
    printf("hypothesis %d: %d mismatches of 1024\n", hyp, bad);
  }
  return 0;
}
