// Probe: (1) is v_mfma_f32_16x16x4_f32 bit-identical to a k-ordered fmaf chain (k0..k3 inside one instruction)?
//        (2) sustained f32 MFMA rate on random data, every CU issuing, for the 32x32x2 and the 16x16x4 shape at the same
//            FLOP per wave (MI355X_MICROARCH.md, DVFS give-back item 7: the clock a chip holds can depend on the shape).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o mfma_shape_probe mfma_shape_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// 16x16x4: lane l holds A[row l&15][k = l>>4], B[k = l>>4][col l&15]; D: col = l&15, row = (l>>4)*4 + reg
__global__ void k_chain16(const float* A, const float* B, const float* C0, float* D, int K) {
  const int l = threadIdx.x, c = l & 15, g = l >> 4;
  f32x4 acc;
  for (int i = 0; i < 4; i++) acc[i] = C0[(g * 4 + i) * 16 + c];
  for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c * K + k + g], B[(k + g) * 16 + c], acc, 0, 0, 0);
  for (int i = 0; i < 4; i++) D[(g * 4 + i) * 16 + c] = acc[i];
}

template <int SHAPE, int NACC>
__global__ __launch_bounds__(256) void k_rate(const float* in, float* out, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  float a[4], b[4];
  for (int i = 0; i < 4; i++) { a[i] = in[(tid * 8 + i) & 0xFFFFF]; b[i] = in[(tid * 8 + 4 + i) & 0xFFFFF]; }
  if constexpr (SHAPE == 32) {
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; n++) for (int r = 0; r < 16; r++) acc[n][r] = 0.0f;
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int n = 0; n < NACC; n++) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + n) & 3], acc[n], 0, 0, 0);
    float s = 0.0f;
    for (int n = 0; n < NACC; n++) for (int r = 0; r < 16; r++) s += acc[n][r];
    out[tid] = s;
  } else {
    f32x4 acc[NACC];
    for (int n = 0; n < NACC; n++) for (int r = 0; r < 4; r++) acc[n][r] = 0.0f;
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int n = 0; n < NACC; n++) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[(u + n) & 3], acc[n], 0, 0, 0);
    float s = 0.0f;
    for (int n = 0; n < NACC; n++) for (int r = 0; r < 4; r++) s += acc[n][r];
    out[tid] = s;
  }
}

template <int SHAPE, int NACC>
static void rate(const char* name, const float* din, float* dout, int wg_per_cu) {
  const int iters = SHAPE == 32 ? 1500 : 3000;     // same FLOP per wave: 32x32x2 = 4096 FLOP, 16x16x4 = 2048 FLOP
  const int grid = 256 * wg_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; w++) k_rate<SHAPE, NACC><<<grid, 256>>>(din, dout, iters);
  hipDeviceSynchronize();
  std::vector<double> ts;
  for (int rep = 0; rep < 5; rep++) {
    hipEventRecord(e0);
    for (int i = 0; i < 20; i++) k_rate<SHAPE, NACC><<<grid, 256>>>(din, dout, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms / 20.0);
  }
  std::sort(ts.begin(), ts.end());
  const double flop = (double)grid * 4 * iters * 4 * NACC * (SHAPE == 32 ? 4096.0 : 2048.0);
  printf("%-44s %d WG/CU: %8.3f ms  %7.1f TFLOP/s\n", name, wg_per_cu, ts[2], flop / (ts[2] * 1e-3) / 1e12);
}

int main() {
  const int K = 576;
  std::vector<float> A(16 * K), B(K * 16), C(256), D(256), R(256);
  srand(7);
  auto rnd = []() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  for (auto& v : A) v = rnd();
  for (auto& v : B) v = rnd() * 0.1f;
  for (auto& v : C) v = rnd();
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
    float acc = C[i * 16 + j];
    for (int k = 0; k < K; k++) acc = fmaf(A[i * K + k], B[k * 16 + j], acc);
    R[i * 16 + j] = acc;
  }
  float *dA, *dB, *dC, *dD;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
  k_chain16<<<1, 64>>>(dA, dB, dC, dD, K);
  hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 256; i++) if (memcmp(&D[i], &R[i], 4)) bad++;
  printf("mfma_16x16x4 vs k-ordered fmaf chain (K=%d): %d / 256 mismatching bit patterns\n", K, bad);

  std::vector<float> h(1 << 20);
  for (auto& v : h) v = rnd();
  float *din, *dout; hipMalloc(&din, h.size() * 4); hipMalloc(&dout, 256 * 8 * 256 * 4);
  hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  // workgroups of 256 threads: WG/CU = waves per SIMD
  for (int wg : {1, 2, 3, 4, 5, 6, 8}) {
    rate<32, 1>("v_mfma_f32_32x32x2_f32, 1 accumulator  / wave", din, dout, wg);
    rate<32, 2>("v_mfma_f32_32x32x2_f32, 2 accumulators / wave", din, dout, wg);
    if (wg <= 6) rate<32, 4>("v_mfma_f32_32x32x2_f32, 4 accumulators / wave", din, dout, wg);
    rate<16, 2>("v_mfma_f32_16x16x4_f32, 2 accumulators / wave", din, dout, wg);
    rate<16, 4>("v_mfma_f32_16x16x4_f32, 4 accumulators / wave", din, dout, wg);
    rate<16, 8>("v_mfma_f32_16x16x4_f32, 8 accumulators / wave", din, dout, wg);
    if (wg <= 6) rate<16, 16>("v_mfma_f32_16x16x4_f32, 16 accumulators / wave", din, dout, wg);
  }
  return bad ? 1 : 0;
}
