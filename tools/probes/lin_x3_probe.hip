// Checks and times the bf16x3 linear kernel (dani_slam_amd/csrc/lg_x3.hip) against the f32 MFMA one (lg_kernels.hip),
// both included as source.  Accuracy is measured against a double-precision host contraction on sampled outputs.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I ../../dani_slam_amd/csrc -o lin_x3_probe lin_x3_probe.hip
#include "../../dani_slam_amd/csrc/lg_kernels.hip"
#include "../../dani_slam_amd/csrc/lg_x3.hip"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
using namespace dani;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
  const int T = 16416, ld = 16640;
  struct Shape { const char* name; int out, k0, k1, rot; bool res; float alpha; } shapes[] = {
      {"Wqkv   768 <- 256 (+rotary)", 768, 256, 0, 512, false, 1.0f}, {"qk|v   512 <- 256", 512, 256, 0, 0, false, 1.0f},
      {"ffn.0  512 <- 256+256", 512, 256, 256, 0, false, 1.0f},        {"ffn.3  256 <- 512 (+res)", 256, 512, 0, 0, true, 1.0f},
      {"final  256 <- 256 (alpha)", 256, 256, 0, 0, false, 0.25f}};
  std::mt19937 rng(7);
  std::normal_distribution<float> nd(0.0f, 1.0f);
  std::vector<float> hx((size_t)512 * ld), hx1((size_t)256 * ld), hres((size_t)768 * ld), hc((size_t)32 * ld), hs((size_t)32 * ld);
  for (auto& v : hx) v = nd(rng) * 3.0f;
  for (auto& v : hx1) v = nd(rng);
  for (auto& v : hres) v = nd(rng);
  for (size_t i = 0; i < hc.size(); i++) { const float ang = nd(rng) * 3.0f; hc[i] = cosf(ang); hs[i] = sinf(ang); }
  float *X, *X1, *R, *C, *S, *Y32, *Y3, *Wp, *Bp; void* W3;
  CK(hipMalloc(&X, hx.size() * 4)); CK(hipMalloc(&X1, hx1.size() * 4)); CK(hipMalloc(&R, hres.size() * 4));
  CK(hipMalloc(&C, hc.size() * 4)); CK(hipMalloc(&S, hs.size() * 4));
  CK(hipMalloc(&Y32, (size_t)768 * ld * 4)); CK(hipMalloc(&Y3, (size_t)768 * ld * 4));
  CK(hipMalloc(&Wp, (size_t)768 * 512 * 4)); CK(hipMalloc(&Bp, 768 * 4)); CK(hipMalloc(&W3, (size_t)768 * 512 * 6));
  CK(hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(X1, hx1.data(), hx1.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(R, hres.data(), hres.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(C, hc.data(), hc.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(S, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t ea, eb;
  CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
  for (const Shape& sh : shapes) {
    const int K = sh.k0 + sh.k1;
    std::vector<float> w((size_t)sh.out * K), b(sh.out);
    for (auto& v : w) v = nd(rng) * 0.06f;
    for (auto& v : b) v = nd(rng) * 0.1f;
    const int tiles = (sh.out + 63) / 64;
    std::vector<float> wp((size_t)tiles * 64 * K), bp((size_t)tiles * 64);
    for (int ct = 0; ct < tiles; ct++)   // 1x1 case of pack_conv_weights: [otile][k][64]
      for (int k = 0; k < K; k++)
        for (int co = 0; co < 64; co++) wp[((size_t)ct * K + k) * 64 + co] = (ct * 64 + co < sh.out) ? w[(size_t)(ct * 64 + co) * K + k] : 0.0f;
    for (int i = 0; i < tiles * 64; i++) bp[i] = i < sh.out ? b[i] : 0.0f;
    std::vector<uint16_t> w3(x3_packed_u16(sh.out, K));
    pack_linear_x3(w.data(), sh.out, K, w3.data());
    CK(hipMemcpy(Wp, wp.data(), wp.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(Bp, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W3, w3.data(), w3.size() * 2, hipMemcpyHostToDevice));
    const float* x1 = sh.k1 ? X1 : nullptr;
    auto run32 = [&]() { launch_linear_fm(X, sh.k0, x1, sh.k1, Wp, Bp, Y32, sh.out, T, ld, sh.res ? R : nullptr, C, S, sh.rot, sh.alpha, 0); };
    auto run3 = [&]() { launch_linear_x3(X, sh.k0, x1, sh.k1, W3, Bp, Y3, sh.out, T, ld, sh.res ? R : nullptr, C, S, sh.rot, sh.alpha, 0); };
    CK(hipMemset(Y32, 0, (size_t)768 * ld * 4)); CK(hipMemset(Y3, 0, (size_t)768 * ld * 4));
    run32(); run3();
    CK(hipDeviceSynchronize());
    std::vector<float> y32((size_t)sh.out * ld), y3((size_t)sh.out * ld);
    CK(hipMemcpy(y32.data(), Y32, y32.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(y3.data(), Y3, y3.size() * 4, hipMemcpyDeviceToHost));
    // double reference on sampled (row pair, token)
    double e32 = 0, e3 = 0, d = 0, ymax = 0;
    std::uniform_int_distribution<int> rt(0, T - 1), rr(0, sh.out / 2 - 1);
    for (int it = 0; it < 4000; it++) {
      const int tok = (it < 64) ? T - 1 - it : rt(rng), r0 = 2 * rr(rng);
      double v[2];
      for (int q = 0; q < 2; q++) {
        double acc = b[r0 + q];
        for (int k = 0; k < K; k++) acc += (double)w[(size_t)(r0 + q) * K + k] * (double)(k < sh.k0 ? hx[(size_t)k * ld + tok] : hx1[(size_t)(k - sh.k0) * ld + tok]);
        v[q] = acc;
      }
      if (r0 < sh.rot) {
        const int ip = (r0 & 63) >> 1;
        const double c = hc[(size_t)ip * ld + tok], s = hs[(size_t)ip * ld + tok];
        const double a0 = v[0] * c - v[1] * s, a1 = v[1] * c + v[0] * s;
        v[0] = a0; v[1] = a1;
      }
      for (int q = 0; q < 2; q++) {
        double ref = v[q] * sh.alpha + (sh.res ? hres[(size_t)(r0 + q) * ld + tok] : 0.0);
        const size_t o = (size_t)(r0 + q) * ld + tok;
        e32 = std::max(e32, std::fabs(y32[o] - ref)); e3 = std::max(e3, std::fabs(y3[o] - ref));
        d = std::max(d, (double)std::fabs(y32[o] - y3[o])); ymax = std::max(ymax, std::fabs(ref));
      }
    }
    // columns >= T untouched
    bool clean = true;
    for (int r = 0; r < sh.out && clean; r++) for (int t = T; t < ld; t++) if (y3[(size_t)r * ld + t] != 0.0f) { clean = false; break; }
    // timing, interleaved
    std::vector<double> t32, t3, t3n1;
    for (int i = 0; i < 50; i++) { run32(); run3(); }
    for (int rd = 0; rd < 9; rd++) {
      float ms;
      CK(hipEventRecord(ea, 0)); for (int i = 0; i < 10; i++) run32(); CK(hipEventRecord(eb, 0)); CK(hipEventSynchronize(eb));
      CK(hipEventElapsedTime(&ms, ea, eb)); t32.push_back(ms * 100.0);
      CK(hipEventRecord(ea, 0)); for (int i = 0; i < 10; i++) run3(); CK(hipEventRecord(eb, 0)); CK(hipEventSynchronize(eb));
      CK(hipEventElapsedTime(&ms, ea, eb)); t3.push_back(ms * 100.0);
      g_x3_force_ncb = 2;
      CK(hipEventRecord(ea, 0)); for (int i = 0; i < 10; i++) run3(); CK(hipEventRecord(eb, 0)); CK(hipEventSynchronize(eb));
      CK(hipEventElapsedTime(&ms, ea, eb)); t3n1.push_back(ms * 100.0);
      g_x3_force_ncb = 0;
    }
    std::sort(t3n1.begin(), t3n1.end());
    std::sort(t32.begin(), t32.end()); std::sort(t3.begin(), t3.end());
    const double fl = 2.0 * sh.out * K * (double)T, bytes = 4.0 * T * (K + sh.out * (sh.res ? 2 : 1));
    printf("%-30s |y|max %6.2f  err vs f64: f32-mfma %.3e  bf16x3 %.3e  |f32-x3| %.3e  pad %s   f32 %6.1f us  x3[128-out] %6.1f us  x3 %6.1f us (%.1f TFLOP/s eq, %.2f TB/s alg)\n",
           sh.name, ymax, e32, e3, d, clean ? "clean" : "DIRTY", t32[4], t3n1[4], t3[4], fl / (t3[4] * 1e-6) / 1e12, bytes / (t3[4] * 1e-6) / 1e12);
  }
  return 0;
}
