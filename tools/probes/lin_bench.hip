// Times the PRODUCT feature-major linear launch (dani_slam_amd/csrc/lg_kernels.hip, included as source) on the
// LightGlue layer shapes for several token counts.  Usage: lin_bench [T ...]
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I ../../dani_slam_amd/csrc -o lin_bench lin_bench.hip
#include "../../dani_slam_amd/csrc/lg_kernels.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace dani;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  std::vector<int> Ts;
  for (int i = 1; i < argc; i++) Ts.push_back(atoi(argv[i]));
  if (Ts.empty()) Ts = {16384, 16416, 16448, 16896};
  const int ld = 70000;
  float *X, *Y, *Wt, *Bs, *C;
  CK(hipMalloc(&X, (size_t)1024 * ld * 4)); CK(hipMalloc(&Y, (size_t)768 * ld * 4)); CK(hipMalloc(&Wt, (size_t)768 * 512 * 4));
  CK(hipMalloc(&Bs, 768 * 4)); CK(hipMalloc(&C, (size_t)64 * ld * 4));
  CK(hipMemset(X, 0, (size_t)1024 * ld * 4)); CK(hipMemset(Wt, 0, (size_t)768 * 512 * 4)); CK(hipMemset(Bs, 0, 768 * 4)); CK(hipMemset(C, 0, (size_t)64 * ld * 4));
  struct Shape { const char* name; int out, k0, k1, rot; bool res; } shapes[] = {
      {"Wqkv   768 <- 256 (+rotary)", 768, 256, 0, 512, false}, {"qk|v   512 <- 256", 512, 256, 0, 0, false},
      {"ffn.0  512 <- 256+256", 512, 256, 256, 0, false},        {"ffn.3  256 <- 512 (+res)", 256, 512, 0, 0, true}};
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 300; i++) launch_linear_fm(X, 256, X + (size_t)256 * ld, 256, Wt, Bs, Y, 512, 16384, ld, nullptr, C, C + (size_t)32 * ld, 0, 1.0f, 0);
  CK(hipDeviceSynchronize());
  for (const Shape& sh : shapes)
    for (int T : Ts)
    for (int tt : {0, 64, 128, 256}) {
      g_lin_force_tt = tt;
      std::vector<double> us;
      for (int rd = 0; rd < 7; rd++) {
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < 10; i++)
          launch_linear_fm(X, sh.k0, sh.k1 ? X + (size_t)512 * ld : nullptr, sh.k1, Wt, Bs, Y, sh.out, T, ld, sh.res ? Y : nullptr, C, C + (size_t)32 * ld, sh.rot, 1.0f, 0);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        us.push_back(ms * 100.0);
      }
      std::sort(us.begin(), us.end());
      const double fl = 2.0 * sh.out * (sh.k0 + sh.k1) * (double)T;
      printf("%-30s T=%6d TT=%3d  median %7.1f us  min %7.1f   %6.1f TFLOP/s\n", sh.name, T, tt, us[3], us[0], fl / (us[3] * 1e-6) / 1e12);
    }
  return 0;
}
