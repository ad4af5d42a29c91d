// Probe: is v_mfma_f32_32x32x2_f32 bit-identical to a k-ordered fmaf chain starting from C?
// Also probes: f32 divide and sqrt are correctly rounded in default hipcc device mode.
// Build: hipcc --offload-arch=gfx950 -O2 -o mfma_chain_probe mfma_chain_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A: [32][K] row-major, B: [K][32] row-major, C0: [32][32]; D: [32][32]
__global__ void k_mfma(const float* A, const float* B, const float* C0, float* D, int K) {
  int l = threadIdx.x; int r = l & 31, h = l >> 5;
  f32x16 acc;
  for (int i = 0; i < 16; i++) { int row = (i & 3) + 8 * (i >> 2) + 4 * h; acc[i] = C0[row * 32 + r]; }
  for (int k = 0; k < K; k += 2) {
    float a = A[r * K + k + h];
    float b = B[(k + h) * 32 + r];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  for (int i = 0; i < 16; i++) { int row = (i & 3) + 8 * (i >> 2) + 4 * h; D[row * 32 + r] = acc[i]; }
}
__global__ void k_divsqrt(const float* x, const float* y, float* q, float* s, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { q[i] = x[i] / y[i]; s[i] = sqrtf(x[i]); }
}
int main() {
  const int K = 576;
  std::vector<float> A(32 * K), B(K * 32), C(32 * 32), D(32 * 32), R(32 * 32);
  srand(7);
  auto rnd = []() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  for (auto& v : A) v = rnd();
  for (auto& v : B) v = rnd() * 0.1f;
  for (auto& v : C) v = rnd();
  for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) {
    float acc = C[i * 32 + j];
    for (int k = 0; k < K; k++) acc = fmaf(A[i * K + k], B[k * 32 + j], acc);
    R[i * 32 + j] = acc;
  }
  float *dA, *dB, *dC, *dD;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
  k_mfma<<<1, 64>>>(dA, dB, dC, dD, K);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 1024; i++) if (memcmp(&D[i], &R[i], 4)) bad++;
  printf("mfma_32x32x2 vs fmaf chain (K=%d): %d / 1024 mismatching bit patterns\n", K, bad);
  // div / sqrt
  const int n = 1 << 20; std::vector<float> x(n), y(n), q(n), s(n);
  for (int i = 0; i < n; i++) { x[i] = rnd() * rnd() * 100.f; if (x[i] < 0) x[i] = -x[i]; y[i] = fabsf(rnd()) + 1e-3f; }
  float *dx, *dy, *dq, *ds; hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4); hipMalloc(&dq, n * 4); hipMalloc(&ds, n * 4);
  hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dy, y.data(), n * 4, hipMemcpyHostToDevice);
  k_divsqrt<<<n / 256, 256>>>(dx, dy, dq, ds, n);
  hipMemcpy(q.data(), dq, n * 4, hipMemcpyDeviceToHost); hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost);
  int bq = 0, bs = 0;
  for (int i = 0; i < n; i++) { float qq = x[i] / y[i], ss = sqrtf(x[i]); if (memcmp(&qq, &q[i], 4)) bq++; if (memcmp(&ss, &s[i], 4)) bs++; }
  printf("div mismatches %d / %d, sqrt mismatches %d / %d\n", bq, n, bs, n);
  return (bad || bq || bs) ? 1 : 0;
}
