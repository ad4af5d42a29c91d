// Times the PRODUCT feature-major linear launch (dani_slam_amd/csrc/lg_kernels.hip, included as source) on the
// LightGlue layer shapes for several token counts.  Usage: lin_bench [T ...]
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I ../../dani_slam_amd/csrc -o lin_bench lin_bench.hip
//        (add -DDANI_CONV_STAMPS for the core clock of workgroup 0 and the entry / first-chunk / loop-done / exit timeline of every workgroup)
#include "../../dani_slam_amd/csrc/lg_kernels.hip"
#include "lin_ring_kernels.inc"
#include "lin_rs_kernels.inc"
#include <algorithm>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace dani;
// tt: 0 / 64 / 128 / 256 = the product launcher (with its tuning hook), >= 10000 = a ring variant (lin_ring_kernels.inc)
static float* g_wq = nullptr;            // weights in linear_rsw_kernel's layout (re-packed per shape by prepare_wq)
static std::vector<float> g_wt_host;    // the packed weights of the shipped kernel, on the host
static bool prepare_wq(int OUT, int K) {
  std::vector<float> wq((size_t)OUT * K);
  repack_rsw(g_wt_host.data(), OUT, K, wq.data());
  return hipMemcpy(g_wq, wq.data(), wq.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
}
static bool launch_any(int tt, const float* x0, int K0, const float* x1, int K1, const float* w, const float* b, float* y, int OUT, int T,
                       int ld, const float* res, const float* cosb, const float* sinb, int rot, float alpha) {
  if (tt >= 50000) {   // the product launcher with the register-streaming form: 50000 = its own choice, 5000N = NCB forced
    g_lin_force_tt = 0; g_lin_force_rs = tt == 50000 ? -1 : tt - 50000;
    const bool ok = launch_linear_fm(x0, K0, x1, K1, w, b, y, OUT, T, ld, res, cosb, sinb, rot, alpha, 0, g_wq);
    g_lin_force_rs = 0;
    return ok;
  }
  if (tt >= 40000) {
    LinArgs a;
    a.x0 = x0; a.x1 = x1 ? x1 : x0; a.K0 = K0; a.K1 = K1; a.w = w; a.bias = b; a.y = y; a.res = res;
    a.cosb = cosb; a.sinb = sinb; a.OUT = OUT; a.T = T; a.ld = ld; a.rot_rows = rot; a.alpha = alpha; a.ksplit = 0;
    return launch_linear_rsw(tt, a, g_wq, 0);
  }
  if (tt >= 30000) {
    LinArgs a;
    a.x0 = x0; a.x1 = x1 ? x1 : x0; a.K0 = K0; a.K1 = K1; a.w = w; a.bias = b; a.y = y; a.res = res;
    a.cosb = cosb; a.sinb = sinb; a.OUT = OUT; a.T = T; a.ld = ld; a.rot_rows = rot; a.alpha = alpha; a.ksplit = 0;
    return launch_linear_rs(tt, a, 0);
  }
  if (tt >= 10000) return launch_linear_variant(tt, x0, K0, x1, K1, w, b, y, OUT, T, ld, res, cosb, sinb, rot, alpha, 0);
  g_lin_force_tt = tt; g_lin_force_rs = 0;
  return launch_linear_fm(x0, K0, x1, K1, w, b, y, OUT, T, ld, res, cosb, sinb, rot, alpha, 0);
}
__global__ void fill_rand(float* p, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u ^ (seed * 0x9E3779B9u); x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    p[i] = ((int)(x & 0xFFFF) - 32768) * (1.0f / 65536.0f);
  }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  std::vector<int> Ts, variants;
  for (int i = 1; i < argc; i++) Ts.push_back(atoi(argv[i]));
  if (const char* v = getenv("LIN_VARIANTS")) { for (const char* q = v; *q;) { variants.push_back(atoi(q)); while (*q && *q != ',') q++; if (*q) q++; } }
  if (variants.empty()) variants = {0};
  if (Ts.empty()) Ts = {16384, 16416, 16448, 16896};
  const int ld = 70000;
  float *X, *Y, *Wt, *Bs, *C;
  CK(hipMalloc(&X, (size_t)1024 * ld * 4)); CK(hipMalloc(&Y, (size_t)768 * ld * 4)); CK(hipMalloc(&Wt, (size_t)768 * 512 * 4));
  CK(hipMalloc(&Bs, 768 * 4)); CK(hipMalloc(&C, (size_t)64 * ld * 4));
  CK(hipMemset(X, 0, (size_t)1024 * ld * 4)); CK(hipMemset(Wt, 0, (size_t)768 * 512 * 4)); CK(hipMemset(Bs, 0, 768 * 4)); CK(hipMemset(C, 0, (size_t)64 * ld * 4));
  // random operands (zero-filled ones read high: the MFMA units draw less power on zeros)
  hipLaunchKernelGGL(fill_rand, dim3(4096), dim3(256), 0, 0, X, (size_t)1024 * ld, 1u);
  hipLaunchKernelGGL(fill_rand, dim3(1024), dim3(256), 0, 0, Wt, (size_t)768 * 512, 2u);
  hipLaunchKernelGGL(fill_rand, dim3(1), dim3(256), 0, 0, Bs, (size_t)768, 3u);
  hipLaunchKernelGGL(fill_rand, dim3(1024), dim3(256), 0, 0, C, (size_t)64 * ld, 4u);
  CK(hipDeviceSynchronize());
  CK(hipMalloc(&g_wq, (size_t)768 * 512 * 4));
  g_wt_host.resize((size_t)768 * 512);
  CK(hipMemcpy(g_wt_host.data(), Wt, g_wt_host.size() * 4, hipMemcpyDeviceToHost));
  if (getenv("LIN_CHECK")) {   // ring variants against the shipped kernel, bit for bit, on every shape
    const int T = 16416;
    float* Y2; CK(hipMalloc(&Y2, (size_t)768 * ld * 4));
    std::vector<float> r0((size_t)768 * T), r1((size_t)768 * T);
    struct S2 { int out, k0, k1, rot; bool res; } sh2[] = {{768, 256, 0, 512, false}, {512, 256, 256, 0, false}, {256, 512, 0, 0, true}};
    for (const S2& sh : sh2)
      for (int tt : variants) {
        if (tt == 0) continue;
        prepare_wq(sh.out, sh.k0 + sh.k1);
        launch_any(0, X, sh.k0, sh.k1 ? X + (size_t)512 * ld : nullptr, sh.k1, Wt, Bs, Y, sh.out, T, ld, sh.res ? X + (size_t)600 * ld : nullptr, C, C + (size_t)32 * ld, sh.rot, 0.7f);
        launch_any(tt, X, sh.k0, sh.k1 ? X + (size_t)512 * ld : nullptr, sh.k1, Wt, Bs, Y2, sh.out, T, ld, sh.res ? X + (size_t)600 * ld : nullptr, C, C + (size_t)32 * ld, sh.rot, 0.7f);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy2D(r0.data(), (size_t)T * 4, Y, (size_t)ld * 4, (size_t)T * 4, sh.out, hipMemcpyDeviceToHost));
        CK(hipMemcpy2D(r1.data(), (size_t)T * 4, Y2, (size_t)ld * 4, (size_t)T * 4, sh.out, hipMemcpyDeviceToHost));
        size_t bad = 0; double sum = 0, maxd = 0;
        for (size_t i = 0; i < (size_t)sh.out * T; i++) { bad += memcmp(&r0[i], &r1[i], 4) != 0; sum += fabs(r0[i]); maxd = std::max(maxd, (double)fabs(r0[i] - r1[i])); }
        printf("check out=%d K=%d variant %d: %zu of %zu differ (mean |y| %.3f, max |diff| %.3g)\n", sh.out, sh.k0 + sh.k1, tt, bad, (size_t)sh.out * T, sum / ((double)sh.out * T), maxd);
      }
    g_lin_force_tt = 0;
  }
  struct Shape { const char* name; int out, k0, k1, rot; bool res; } shapes[] = {
      {"Wqkv   768 <- 256 (+rotary)", 768, 256, 0, 512, false}, {"qk|v   512 <- 256", 512, 256, 0, 0, false},
      {"ffn.0  512 <- 256+256", 512, 256, 256, 0, false},        {"ffn.3  256 <- 512 (+res)", 256, 512, 0, 0, true}};
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 300; i++) launch_linear_fm(X, 256, X + (size_t)256 * ld, 256, Wt, Bs, Y, 512, 16384, ld, nullptr, C, C + (size_t)32 * ld, 0, 1.0f, 0);
  CK(hipDeviceSynchronize());
  for (const Shape& sh : shapes)
    for (int T : Ts) if (prepare_wq(sh.out, sh.k0 + sh.k1))
    for (int tt : variants) {
      std::vector<double> us;
      if (!launch_any(tt, X, sh.k0, sh.k1 ? X + (size_t)512 * ld : nullptr, sh.k1, Wt, Bs, Y, sh.out, T, ld, sh.res ? Y : nullptr, C, C + (size_t)32 * ld, sh.rot, 1.0f)) continue;
      for (int rd = 0; rd < 7; rd++) {
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < 10; i++)
          launch_any(tt, X, sh.k0, sh.k1 ? X + (size_t)512 * ld : nullptr, sh.k1, Wt, Bs, Y, sh.out, T, ld, sh.res ? Y : nullptr, C, C + (size_t)32 * ld, sh.rot, 1.0f);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        us.push_back(ms * 100.0);
      }
      std::sort(us.begin(), us.end());
      const double fl = 2.0 * sh.out * (sh.k0 + sh.k1) * (double)T;
      printf("%-30s T=%6d TT=%5d  median %7.1f us  min %7.1f   %6.1f TFLOP/s", sh.name, T, tt, us[3], us[0], fl / (us[3] * 1e-6) / 1e12);
#ifdef DANI_CONV_STAMPS
      if (tt < 10000) {   // workgroup 0 of the last launch: its main loop in core clocks against the 100 MHz wall clock
        unsigned long long st[4];
        CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(dani::g_lin_stamps), sizeof(st)));
        const double wall_us = (double)(st[3] - st[0]) / 100.0;
        printf("   wg0 main loop %.1f us at %.2f GHz", wall_us, (double)(st[2] - st[1]) / wall_us / 1e3);
        // every workgroup of the last launch: when it entered, had its first chunk staged, left the K loop, exited (us after the first entry)
        const int nwg = std::min(8192, (int)(((T + 63) / 64 + 7) / 8 * 8 * ((sh.out + 63) / 64)));
        std::vector<unsigned> tr((size_t)8192 * 4);
        CK(hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(dani::g_lin_trace), tr.size() * 4));
        std::vector<double> col[4];
        unsigned t0 = 0xffffffffu;
        const int ntl = (T + 63) / 64, otl = (sh.out + 63) / 64;
        auto live = [&](int b) { const int slot = b >> 3, tile = (slot / otl) * 8 + (b & 7); return tile < ntl; };
        for (int b = 0; b < nwg; b++) if (live(b)) t0 = std::min(t0, tr[(size_t)b * 4]);
        for (int b = 0; b < nwg; b++) if (live(b)) for (int k = 0; k < 4; k++) col[k].push_back((double)(tr[(size_t)b * 4 + k] - t0) / 100.0);
        printf("\n      %zu workgroups; us after the first entry, min / p10 / p50 / p90 / max:", col[0].size());
        const char* nm[4] = {"entry", "first chunk staged", "K loop done", "exit"};
        for (int k = 0; k < 4; k++) {
          std::sort(col[k].begin(), col[k].end());
          const size_t n = col[k].size();
          printf("\n        %-20s %6.1f %6.1f %6.1f %6.1f %6.1f", nm[k], col[k][0], col[k][n / 10], col[k][n / 2], col[k][n * 9 / 10], col[k][n - 1]);
        }
      }
#endif
      printf("\n");
    }
  return 0;
}
