// Times the PRODUCT attention launch (dani_slam_amd/csrc/lg_kernels.hip, included as source) on the bench's shape,
// with the engine's main/tail token layout.  Usage: attn_bench [n ...]   (default 1024 1025 1000 1030)
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I ../../dani_slam_amd/csrc -o attn_bench attn_bench.hip
#include "../../dani_slam_amd/csrc/lg_kernels.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace dani;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  std::vector<int> ns;
  for (int i = 1; i < argc; i++) ns.push_back(atoi(argv[i]));
  if (ns.empty()) ns = {1024, 1025, 1024, 1025, 1027, 1000, 1030};
  const int nsets = 16, ld = nsets * 1100 + 512;
  std::vector<float> hq((size_t)768 * ld);
  srand(3);
  for (auto& v : hq) v = (float)rand() / (float)RAND_MAX - 0.5f;
  float *QKV, *O; LgSet* ds;
  CK(hipMalloc(&QKV, hq.size() * 4)); CK(hipMalloc(&O, (size_t)256 * ld * 4)); CK(hipMalloc(&ds, sizeof(LgSet) * nsets));
  CK(hipMemcpy(QKV, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  bool warm = false;
  for (int n : ns) {
    std::vector<LgSet> hs(nsets);
    int off = 0, ntail = 0, max_mp = 0, any_tail = 0;
    for (int i = 0; i < nsets; i++) {
      LgSet& s = hs[i];
      const int t = n % 32;
      s.n = n; s.nmain = (n >= 32 && t > 0 && t <= LG_TAIL_MAX) ? n - t : n;
      s.tok_off = off; off += round_up(std::max(s.nmain, 1), 32);
      s.tail_off = ntail; ntail += n - s.nmain;
      max_mp = std::max(max_mp, round_up(s.nmain, 32));
      any_tail = std::max(any_tail, n - s.nmain);
    }
    for (auto& s : hs) s.tail_off += off;
    for (int i = 0; i < nsets; i++) { LgSet& s = hs[i]; const LgSet& p = hs[i ^ 1]; s.partner_off = p.tok_off; s.partner_n = p.n; s.partner_nmain = p.nmain; s.partner_tail_off = p.tail_off; }
    CK(hipMemcpy(ds, hs.data(), sizeof(LgSet) * nsets, hipMemcpyHostToDevice));
    const double flops = (double)nsets * 4.0 * 256.0 * n * (double)n;
    if (!warm) { for (int i = 0; i < 200; i++) launch_attention(QKV, QKV + (size_t)256 * ld, QKV + (size_t)512 * ld, O, ld, ds, nsets, max_mp, any_tail, 0, 0.125f, 0); CK(hipDeviceSynchronize()); warm = true; }
    for (int force : {2, 6, 7, 8})
    for (int cross = 0; cross < 2; cross++) {
      g_attn_force = force;
      std::vector<double> us;
      for (int rd = 0; rd < 7; rd++) {
        launch_attention(QKV, QKV + (size_t)256 * ld, QKV + (size_t)512 * ld, O, ld, ds, nsets, max_mp, any_tail, cross, 0.125f, 0);
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < 10; i++) launch_attention(QKV, QKV + (size_t)256 * ld, QKV + (size_t)512 * ld, O, ld, ds, nsets, max_mp, any_tail, cross, 0.125f, 0);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        us.push_back(ms * 100.0);
      }
#ifdef ARS_CLOCK
      if (force == 6) { unsigned long long ck[2]; CK(hipMemcpyFromSymbol(ck, HIP_SYMBOL(dani::g_ars_clock), sizeof(ck))); printf("  workgroup 0: %.1f us at %.3f GHz core clock\n", ck[1] / 100.0, (double)ck[0] / ck[1] / 10.0); }
#endif
      std::sort(us.begin(), us.end());
      printf("n=%5d %s kernel %s  main tiles %d any_tail %d   median %7.1f us  min %7.1f   %6.1f TFLOP/s\n", n, cross ? "cross" : "self ", force == 1 ? "ks  " : force == 2 ? "nw4 " : force == 3 ? "x3  " : force == 6 ? "rs8 " : force == 7 ? "rs4 " : force == 8 ? "rs16" : "x3w ", max_mp / 64, any_tail,
             us[3], us[0], flops / (us[3] * 1e-6) / 1e12);
    }
  }
  return 0;
}
