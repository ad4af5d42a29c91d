// Probe: where does linear_fm_kernel<64> (dani_slam_amd/csrc/lg_kernels.hip) lose time?  A copy of the kernel with
// compile-time ablation flags, timed on the ffn.0 shape (512 <- 256+256, 16416 tokens) and the Wqkv shape.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o lin_probe lin_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
struct LinArgs { const float* x0; const float* x1; int K0, K1; const float* w; const float* bias; float* y; int OUT, T, ld; };
enum { F_NOSTAGE = 1, F_NOBAR = 2, F_NOEPI = 4, F_NOLOAD = 8, F_NOMFMA = 16, F_KC32 = 32 };
constexpr int NCO = 64, TT = 64;

template <int OCC, int FLAGS>
__global__ __launch_bounds__(256, OCC) void lin_probe_kernel(LinArgs a) {
  constexpr int KC = (FLAGS & F_KC32) ? 32 : 64;
  constexpr int NX = KC * TT / 4 / 256, NWF = KC * NCO / 4 / 256;
  __shared__ __attribute__((aligned(16))) float lds[KC * TT + KC * NCO];
  float* sX = lds;
  float* sW = lds + KC * TT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
  const int otiles = (a.OUT + NCO - 1) / NCO, ntiles = (a.T + TT - 1) / TT;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tile = (slot / otiles) * 8 + xcd, ot = slot % otiles;
  if (tile >= ntiles) return;
  const int tok0 = tile * TT;
  const int K = a.K0 + a.K1, nchunks = K / KC;
  const float* wt = a.w + (size_t)ot * K * NCO;
  const int wtok = (wave & 1) * 32, wout = (wave >> 1) * 32;
  f32x16 acc, acc2;
#pragma unroll
  for (int r = 0; r < 16; r++) { acc[r] = a.bias[ot * NCO + wout + acc_row(r, h)]; acc2[r] = 0.0f; }
  f32x4 px[NX], pw[NWF];
  int xoff[NX];
#pragma unroll
  for (int u = 0; u < NX; u++) { const int e = tid + 256 * u; xoff[u] = (e / (TT / 4)) * a.ld + tok0 + 4 * (e % (TT / 4)); }
#define PREFETCH(ch_)                                                                                \
  {                                                                                                  \
    const int k0_ = (ch_) * KC;                                                                      \
    const float* src_ = (k0_ < a.K0) ? a.x0 + (size_t)k0_ * a.ld : a.x1 + (size_t)(k0_ - a.K0) * a.ld; \
    _Pragma("unroll") for (int u = 0; u < NX; u++) px[u] = *reinterpret_cast<const f32x4*>(src_ + xoff[u]); \
    const f32x4* ws_ = reinterpret_cast<const f32x4*>(wt + (size_t)(ch_) * (KC * NCO));             \
    _Pragma("unroll") for (int u = 0; u < NWF; u++) pw[u] = ws_[tid + 256 * u];                      \
  }
  const float* bA = sW + h * NCO + wout + j;
  const float* bB = sX + h * TT + wtok + j;
  PREFETCH(0);
  for (int ch = 0; ch < nchunks; ch++) {
    if (!(FLAGS & F_NOSTAGE) || ch == 0) {
      if (!(FLAGS & F_NOBAR)) __syncthreads();
#pragma unroll
      for (int u = 0; u < NX; u++) reinterpret_cast<f32x4*>(sX)[tid + 256 * u] = px[u];
#pragma unroll
      for (int u = 0; u < NWF; u++) reinterpret_cast<f32x4*>(sW)[tid + 256 * u] = pw[u];
      if (!(FLAGS & F_NOBAR)) __syncthreads();
      if (!(FLAGS & F_NOLOAD) && ch + 1 < nchunks) PREFETCH(ch + 1);
    }
    if (!(FLAGS & F_NOMFMA)) {
      constexpr int G = 2, NG = KC / 2 / G;
      float av[NG][G], bv[NG][G];
      auto rd = [&](int g) {
#pragma unroll
        for (int q = 0; q < G; q++) { const int kp = g * G + q; av[g][q] = bA[(kp * 2) * NCO]; bv[g][q] = bB[(kp * 2) * TT]; }
      };
      rd(0);
#pragma unroll
      for (int g = 0; g < NG; g++) {
        if (g + 1 < NG) rd(g + 1);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g][0], bv[g][0], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g][1], bv[g][1], acc2, 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      }
    } else {
      acc[0] += bA[ch] + bB[ch];
    }
  }
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] += acc2[r];
  const int tok = tok0 + wtok + j;
  const bool ok = tok < a.T;
  if (!(FLAGS & F_NOEPI)) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int row = ot * NCO + wout + acc_row(r, h);
      if (ok && row < a.OUT) a.y[(size_t)row * a.ld + tok] = acc[r];
    }
  } else if (acc[3] == 12345.678f) a.y[tok] = acc[5];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef void (*KernelFn)(LinArgs);
struct Variant { std::string name; KernelFn fn; int occ; std::vector<double> us; };

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 16416, ld = 17920;
  float *X, *Y, *Wt, *Bs;
  CK(hipMalloc(&X, (size_t)1024 * ld * 4)); CK(hipMalloc(&Y, (size_t)768 * ld * 4)); CK(hipMalloc(&Wt, (size_t)768 * 512 * 4)); CK(hipMalloc(&Bs, 768 * 4));
  CK(hipMemset(X, 0, (size_t)1024 * ld * 4)); CK(hipMemset(Wt, 0, (size_t)768 * 512 * 4)); CK(hipMemset(Bs, 0, 768 * 4));
  std::vector<Variant> vs;
#define RUN(occ, fl, nm) vs.push_back(Variant{nm, lin_probe_kernel<occ, fl>, occ, {}});
  RUN(2, 0, "baseline")
  RUN(3, 0, "baseline, min 3 WG/CU")
  RUN(4, 0, "baseline, min 4 WG/CU")
  RUN(2, F_NOEPI, "no epilogue stores")
  RUN(2, F_NOBAR, "no barriers (racy)")
  RUN(2, F_NOLOAD, "no global loads after chunk 0")
  RUN(2, F_NOSTAGE, "no staging after chunk 0")
  RUN(2, F_NOSTAGE | F_NOEPI, "MFMA + LDS reads only")
  RUN(2, F_NOMFMA, "staging only (no MFMA)")
  RUN(2, F_KC32, "KC = 32")
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  struct Shape { const char* name; int out, k0, k1; } shapes[] = {{"ffn.0 512 <- 256+256", 512, 256, 256}, {"Wqkv  768 <- 256", 768, 256, 0}, {"ffn.3 256 <- 512", 256, 512, 0}};
  for (const Shape& sh : shapes) {
    LinArgs la{X, X + (size_t)512 * ld, sh.k0, sh.k1, Wt, Bs, Y, sh.out, T, ld};
    const int otiles = sh.out / 64, grid = ((T + 63) / 64 + 7) / 8 * 8 * otiles;
    for (auto& v : vs) v.us.clear();
    for (int i = 0; i < 100; i++) hipLaunchKernelGGL(vs[0].fn, dim3(grid), dim3(256), 0, 0, la);
    for (int rd = 0; rd < 7; rd++)
      for (auto& v : vs) {
        hipLaunchKernelGGL(v.fn, dim3(grid), dim3(256), 0, 0, la);
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < 10; i++) hipLaunchKernelGGL(v.fn, dim3(grid), dim3(256), 0, 0, la);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        v.us.push_back(ms * 100.0);
      }
    const double fl = 2.0 * sh.out * (sh.k0 + sh.k1) * (double)T;
    printf("%s, T = %d\n", sh.name, T);
    for (auto& v : vs) {
      std::sort(v.us.begin(), v.us.end());
      printf("  %-34s median %7.1f us  min %7.1f   %6.1f TFLOP/s\n", v.name.c_str(), v.us[3], v.us[0], fl / (v.us[3] * 1e-6) / 1e12);
    }
  }
  return 0;
}
