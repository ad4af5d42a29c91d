// Probe: the register-resident FFN chain (csrc/ffn_chain.h) against the product's four launches
//   linear (ffn.0) -> ln_gelu -> linear (ffn.3 + residual) -> linear (next projection, rotary)
// on random operands: values (max |diff| per output tensor) and time per chain.
// Usage: ffn_chain_probe [T ...]      Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o ffn_chain_probe ffn_chain_probe.hip
#include "../../dani_slam_amd/csrc/lg_kernels.hip"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace dani;
#ifndef NT_BLOCKS
#define NT_BLOCKS 0   // the last NT_BLOCKS 16-token blocks are left out (the shadow experiment gave them to the first workgroups: tools/probes/ffn_chain_shadow_null.h.txt)
#endif

__global__ void fill_rand(float* p, size_t n, unsigned seed, float scale, float shift) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u ^ (seed * 0x9E3779B9u); x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    p[i] = ((int)(x & 0xFFFF) - 32768) * (1.0f / 65536.0f) * scale + shift;
  }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Lin { float *w, *wp, *bp, *wrs, *pc, *b; int out, in; };   // row-major, linear_fm packing, lin_rs packing, chain packing, raw bias

static int make_lin(Lin& L, int out, int in, unsigned seed, float gain, bool acc_order, bool proj) {
  L.out = out; L.in = in;
  CK(hipMalloc(&L.w, (size_t)out * in * 4)); CK(hipMalloc(&L.b, (size_t)out * 4));
  hipLaunchKernelGGL(fill_rand, dim3(512), dim3(256), 0, 0, L.w, (size_t)out * in, seed, gain, 0.0f);
  hipLaunchKernelGGL(fill_rand, dim3(1), dim3(256), 0, 0, L.b, (size_t)out, seed + 77, 0.5f, 0.0f);
  CK(hipDeviceSynchronize());
  std::vector<float> w((size_t)out * in), b(out);
  CK(hipMemcpy(w.data(), L.w, w.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), L.b, b.size() * 4, hipMemcpyDeviceToHost));
  const int tiles = (out + 63) / 64;
  std::vector<float> wp((size_t)tiles * 64 * in), bp((size_t)tiles * 64), wrs((size_t)out * in), pc((size_t)out * in + LG_CHAIN_SLACK_FLOATS, 0.0f);
  pack_conv_weights(w.data(), b.data(), in, out, 1, 32, 64, wp.data(), bp.data());
  pack_linear_rs_host(w.data(), out, in, wrs.data());
  if (proj) pack_chain_proj(w.data(), out, in, pc.data()); else pack_chain_a(w.data(), out, in, acc_order, pc.data());
  CK(hipMalloc(&L.wp, wp.size() * 4)); CK(hipMalloc(&L.bp, bp.size() * 4)); CK(hipMalloc(&L.wrs, wrs.size() * 4)); CK(hipMalloc(&L.pc, pc.size() * 4));
  CK(hipMemcpy(L.wp, wp.data(), wp.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(L.bp, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(L.wrs, wrs.data(), wrs.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(L.pc, pc.data(), pc.size() * 4, hipMemcpyHostToDevice));
  return 0;
}

// pack_conv_weights lives in conv_mfma.hip in the product; the probe links nothing, so restate the 1x1 case used for linears
namespace dani {
void pack_conv_weights(const float* w, const float* bias, int Cin, int Cout, int ks, int CC, int NCO, float* wp, float* bp) {
  const int T = ks * ks, cout_tiles = (Cout + NCO - 1) / NCO, nch = (Cin + CC - 1) / CC;
  size_t o = 0;
  for (int ct = 0; ct < cout_tiles; ct++)
    for (int ch = 0; ch < nch; ch++)
      for (int cp = 0; cp < CC / 2; cp++)
        for (int t = 0; t < T; t++)
          for (int cc = 0; cc < 2; cc++)
            for (int co = 0; co < NCO; co++) {
              const int cout = ct * NCO + co, cin = ch * CC + 2 * cp + cc;
              wp[o++] = (cout < Cout && cin < Cin) ? w[((size_t)cout * Cin + cin) * T + t] : 0.0f;
            }
  for (int i = 0; i < cout_tiles * NCO; i++) bp[i] = (i < Cout) ? bias[i] : 0.0f;
}
}

int main(int argc, char** argv) {
  std::vector<int> Ts;
  for (int i = 1; i < argc; i++) Ts.push_back(atoi(argv[i]));
  if (Ts.empty()) Ts = {16384};
  const int ld = 17408;
  float *X, *X0, *CTX, *H, *Y, *XR, *YR, *G, *Be, *COS, *SIN;
  CK(hipMalloc(&X, (size_t)256 * ld * 4)); CK(hipMalloc(&X0, (size_t)256 * ld * 4)); CK(hipMalloc(&CTX, (size_t)256 * ld * 4));
  CK(hipMalloc(&H, (size_t)512 * ld * 4)); CK(hipMalloc(&Y, (size_t)768 * ld * 4)); CK(hipMalloc(&XR, (size_t)256 * ld * 4)); CK(hipMalloc(&YR, (size_t)768 * ld * 4));
  CK(hipMalloc(&G, 512 * 4)); CK(hipMalloc(&Be, 512 * 4)); CK(hipMalloc(&COS, (size_t)32 * ld * 4)); CK(hipMalloc(&SIN, (size_t)32 * ld * 4));
  hipLaunchKernelGGL(fill_rand, dim3(4096), dim3(256), 0, 0, X0, (size_t)256 * ld, 5u, 2.0f, 0.0f);
  hipLaunchKernelGGL(fill_rand, dim3(4096), dim3(256), 0, 0, CTX, (size_t)256 * ld, 9u, 2.0f, 0.1f);
  hipLaunchKernelGGL(fill_rand, dim3(1), dim3(256), 0, 0, G, (size_t)512, 6u, 1.0f, 1.0f);
  hipLaunchKernelGGL(fill_rand, dim3(1), dim3(256), 0, 0, Be, (size_t)512, 7u, 0.5f, 0.0f);
  hipLaunchKernelGGL(fill_rand, dim3(512), dim3(256), 0, 0, COS, (size_t)32 * ld, 11u, 2.0f, 0.0f);
  hipLaunchKernelGGL(fill_rand, dim3(512), dim3(256), 0, 0, SIN, (size_t)32 * ld, 12u, 2.0f, 0.0f);
  Lin f0, f3, pq, pv;
  if (make_lin(f0, 512, 512, 21u, 0.12f, false, false) || make_lin(f3, 256, 512, 22u, 0.12f, true, false) ||
      make_lin(pq, 768, 256, 23u, 0.2f, true, true) || make_lin(pv, 512, 256, 24u, 0.2f, true, true)) return 1;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int T : Ts) {
    if (T % 16 || T > ld - 128) { printf("T=%d skipped\n", T); continue; }
    for (int which = 0; which < 2; which++) {
      const Lin& pd = which ? pv : pq;
      const int rot = which ? 0 : 512;
      const float alpha = which ? 1.0f : 0.5f;
      auto reset = [&]() { (void)hipMemcpyAsync(X, X0, (size_t)256 * ld * 4, hipMemcpyDeviceToDevice, 0); };
      auto four = [&]() {
        bool ok = launch_linear_fm(X, 256, CTX, 256, f0.wp, f0.bp, H, 512, T, ld, nullptr, nullptr, nullptr, 0, 1.0f, 0, f0.wrs);
        launch_ln_gelu(H, G, Be, T, ld, 0);
        ok = launch_linear_fm(H, 512, nullptr, 0, f3.wp, f3.bp, X, 256, T, ld, X, nullptr, nullptr, 0, 1.0f, 0, f3.wrs) && ok;
        ok = launch_linear_fm(X, 256, nullptr, 0, pd.wp, pd.bp, Y, pd.out, T, ld, nullptr, COS, SIN, rot, alpha, 0, pd.wrs) && ok;
        return ok;
      };
      auto one = [&]() {
        FfnChainArgs a;
        a.x = X; a.ctx = CTX; a.p0 = f0.pc; a.b0 = f0.b; a.gamma = G; a.beta = Be; a.p3 = f3.pc; a.b3 = f3.b; a.pd = pd.pc; a.bd = pd.b;
        a.y = Y; a.cosb = COS; a.sinb = SIN; a.OUTD = pd.out; a.rot_rows = rot; a.alpha = alpha; a.ld = ld; a.nblocks = T / 16 - NT_BLOCKS;
        return launch_ffn_chain(a, 0);
      };
      reset(); if (!four()) { printf("linear refused\n"); return 1; }
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(XR, X, (size_t)256 * ld * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(YR, Y, (size_t)768 * ld * 4, hipMemcpyDeviceToDevice));
      CK(hipMemset(Y, 0, (size_t)768 * ld * 4));
      reset(); if (!one()) { printf("chain refused\n"); return 1; }
      CK(hipDeviceSynchronize()); CK(hipGetLastError());
      std::vector<float> a((size_t)256 * ld), b((size_t)256 * ld), c((size_t)768 * ld), d((size_t)768 * ld);
      CK(hipMemcpy(a.data(), XR, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), X, b.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(c.data(), YR, c.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(d.data(), Y, d.size() * 4, hipMemcpyDeviceToHost));
      double mx = 0, my = 0, sx = 0, sy = 0, mxt = 0, myt = 0;
      const int Tm = T - 16 * NT_BLOCKS;     // columns >= Tm are shadow tokens
      for (int r = 0; r < 256; r++) for (int t = 0; t < T; t++) { const double e = fabs(a[(size_t)r * ld + t] - b[(size_t)r * ld + t]); if (t < Tm) mx = std::max(mx, e); else mxt = std::max(mxt, e); sx += fabs(a[(size_t)r * ld + t]); }
      for (int r = 0; r < pd.out; r++) for (int t = 0; t < T; t++) { const double e = fabs(c[(size_t)r * ld + t] - d[(size_t)r * ld + t]); if (t < Tm) my = std::max(my, e); else myt = std::max(myt, e); sy += fabs(c[(size_t)r * ld + t]); }
      if (NT_BLOCKS) {
        printf("   shadow columns: max |dx'| %.3g  max |dy| %.3g;  x' col %d rows 0..5 ref/got:", mxt, myt, Tm);
        for (int r = 0; r < 6; r++) printf(" %.4f/%.4f", a[(size_t)r * ld + Tm], b[(size_t)r * ld + Tm]);
        printf("\n   y col %d rows 0..5 ref/got:", Tm);
        for (int r = 0; r < 6; r++) printf(" %.4f/%.4f", c[(size_t)r * ld + Tm], d[(size_t)r * ld + Tm]);
        printf("\n");
      }
      printf("T=%d proj %d (rot %d): chain vs four launches: max |dx'| %.3g (mean |x'| %.3f)  max |dy| %.3g (mean |y| %.3f)\n", T, pd.out, rot, mx, sx / (256.0 * T), my, sy / ((double)pd.out * T));
      for (int form = 0; form < 2; form++) {
        std::vector<double> us;
        for (int rd = 0; rd < 7; rd++) {
          reset();
          CK(hipEventRecord(e0, 0));
          for (int i = 0; i < 10; i++) { if (form) one(); else four(); }
          CK(hipEventRecord(e1, 0));
          CK(hipEventSynchronize(e1));
          float ms = 0;
          CK(hipEventElapsedTime(&ms, e0, e1));
          us.push_back(ms * 100.0);
        }
        std::sort(us.begin(), us.end());
        printf("T=%6d proj %3d %-46s median %7.1f us  min %7.1f\n", T, pd.out, form ? "ffn_chain_kernel (one launch)" : "linear + ln_gelu + linear + linear (four)", us[3], us[0]);
      }
    }
  }
  return 0;
}
