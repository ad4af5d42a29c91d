// Probe: LayerNorm + GELU + ffn.3 (+ residual) as ONE kernel against the product's two launches (ln_gelu_kernel, then
// linear_rs_kernel<1,4>).  A workgroup owns 64 tokens: it loads their 512 hidden rows once, normalises them in registers,
// leaves GELU(LN(h)) in LDS (128 KB) and runs ffn.3 with its B operands from LDS and its weights through the register ring of
// lin_rs.h.  Saves ln_gelu's write + ffn.3's read of the hidden buffer (67 MB) and one launch.
// Usage: ffn3_ln_bench [T ...]   Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I ../../dani_slam_amd/csrc -o ffn3_ln_bench ffn3_ln_bench.hip
#include "../../dani_slam_amd/csrc/lg_kernels.hip"
#include <algorithm>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace dani;

__device__ unsigned g_f3_trace[512 * 8];
typedef float f3x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f3x2 f3_fma(f3x2 a, f3x2 b, f3x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f3x2 f3_s(float v) { f3x2 r = {v, v}; return r; }
// lg_erff on two values at once: the same operations in the same order per element (v_pk_fma_f32 / v_pk_mul_f32 are IEEE)
__device__ __forceinline__ f3x2 f3_erf2(f3x2 a) {
  f3x2 t; t[0] = fabsf(a[0]); t[1] = fabsf(a[1]);
  const f3x2 s = a * a;
  f3x2 r = f3_fma(f3_s(-1.72853470e-5f), t, f3_s(3.83197126e-4f));
  const f3x2 u = f3_fma(f3_s(-3.88396438e-3f), t, f3_s(2.42546219e-2f));
  r = f3_fma(r, s, u);
  r = f3_fma(r, t, f3_s(-1.06777877e-1f));
  r = f3_fma(r, t, f3_s(-6.34846687e-1f));
  r = f3_fma(r, t, f3_s(-1.28717512e-1f));
  r = f3_fma(r, t, -t);
  const f3x2 e = r * f3_s(1.44269504088896341f);
  f3x2 rr; rr[0] = copysignf(1.0f - __builtin_amdgcn_exp2f(e[0]), a[0]); rr[1] = copysignf(1.0f - __builtin_amdgcn_exp2f(e[1]), a[1]);
  f3x2 q = f3_s(-5.96761703e-4f);
  q = f3_fma(q, s, f3_s(4.99119423e-3f));
  q = f3_fma(q, s, f3_s(-2.67681349e-2f));
  q = f3_fma(q, s, f3_s(1.12819925e-1f));
  q = f3_fma(q, s, f3_s(-3.76125336e-1f));
  q = f3_fma(q, s, f3_s(1.28379166e-1f));
  q = f3_fma(q, a, a);
  f3x2 o; o[0] = t[0] > 0.927734375f ? rr[0] : q[0]; o[1] = t[1] > 0.927734375f ? rr[1] : q[1];
  return o;
}
struct Ffn3Args { const float* h; const float* gamma; const float* beta; const float* wq; const float* bias; float* x; int T, ld; };

template <int UQ>
__global__ __launch_bounds__(256, 1) void ffn3_ln_kernel(Ffn3Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* sB = lds;                       // [512][64]: GELU(LN(h)) of the tile, row = hidden feature
  float* red = lds + 512 * 64;           // [2][4][64]
  float* sG = red + 512;                 // gamma[512] | beta[512]: read back per row (as global loads hipcc waits for each pair in turn)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t0 = blockIdx.x * 64;
#define F3_STAMP(k_) if (tid == 0) g_f3_trace[blockIdx.x * 8 + (k_)] = (unsigned)__builtin_amdgcn_s_memrealtime()
  F3_STAMP(0);
  constexpr int NCB = 2, nq = 64;
  const int out0 = wave * 64;
  // ---- weights: ring prologue first, they arrive under phase A
  rs_f32x4 av[UQ][NCB];
  const unsigned aoff = (unsigned)lane * 16u;
  auto load_a = [&](int s, int q) {
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
      const float* ab = a.wq + ((size_t)(out0 / 32 + cb) * nq + q) * 256;
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(av[s][cb]) : "v"(aoff), "s"(ab));
    }
  };
#pragma unroll
  for (int s = 0; s < UQ; s++) load_a(s, s);
  // ---- phase A: lane = (token quad tq, row group rg); wave w holds rows 128 w + 4 i + rg, i = 0 .. 31
  {
    sG[tid] = a.gamma[tid]; sG[256 + tid] = a.gamma[256 + tid]; sG[512 + tid] = a.beta[tid]; sG[768 + tid] = a.beta[256 + tid];
    const int tq = lane & 15, rg = lane >> 4;
    rs_f32x4 v[32];
    const float* hp = a.h + (size_t)(128 * wave + rg) * a.ld + t0 + 4 * tq;
#pragma unroll
    for (int i = 0; i < 32; i++) v[i] = *reinterpret_cast<const rs_f32x4*>(hp + (size_t)(4 * i) * a.ld);
    rs_f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 32; i++) s += v[i];
#pragma unroll
    for (int c = 0; c < 4; c++) { s[c] += __shfl_xor(s[c], 16); s[c] += __shfl_xor(s[c], 32); }
    if (rg == 0) *reinterpret_cast<rs_f32x4*>(&red[wave * 64 + 4 * tq]) = s;
    __syncthreads();
    F3_STAMP(1);
    rs_f32x4 mean = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int w = 0; w < 4; w++) mean += *reinterpret_cast<const rs_f32x4*>(&red[w * 64 + 4 * tq]);
    mean *= (1.0f / 512.0f);
    rs_f32x4 s2 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 32; i++) { const rs_f32x4 d = v[i] - mean; s2 += d * d; }
#pragma unroll
    for (int c = 0; c < 4; c++) { s2[c] += __shfl_xor(s2[c], 16); s2[c] += __shfl_xor(s2[c], 32); }
    if (rg == 0) *reinterpret_cast<rs_f32x4*>(&red[256 + wave * 64 + 4 * tq]) = s2;
    __syncthreads();
    rs_f32x4 var = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int w = 0; w < 4; w++) var += *reinterpret_cast<const rs_f32x4*>(&red[256 + w * 64 + 4 * tq]);
    rs_f32x4 rstd;
#pragma unroll
    for (int c = 0; c < 4; c++) rstd[c] = 1.0f / sqrtf(var[c] * (1.0f / 512.0f) + 1e-5f);
#pragma unroll
    for (int i = 0; i < 32; i++) {
      const int f = 128 * wave + 4 * i + rg;
      const float g = sG[f], b = sG[512 + f];
      rs_f32x4 y;
#pragma unroll
      for (int c = 0; c < 4; c += 2) {
        f3x2 vv = {v[i][c], v[i][c + 1]}, mm = {mean[c], mean[c + 1]}, rs2 = {rstd[c], rstd[c + 1]};
        const f3x2 t = (vv - mm) * rs2 * f3_s(g) + f3_s(b);
        const f3x2 o = f3_s(0.5f) * t * (f3_s(1.0f) + f3_erf2(t * f3_s(0.70710678118654752440f)));
        y[c] = o[0]; y[c + 1] = o[1];
      }
      *reinterpret_cast<rs_f32x4*>(&sB[f * 64 + 4 * tq]) = y;
    }
  }
  __syncthreads();
  F3_STAMP(2);
  // ---- phase B: wave = 64 outputs x 64 tokens (two token blocks interleaved by token: block m = tokens 2 j + m)
  const int j = lane & 31, h = lane >> 5;
  rs_f32x16 acc[NCB][2];
#pragma unroll
  for (int cb = 0; cb < NCB; cb++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const float bv = a.bias[out0 + cb * 32 + rs_acc_row(r, h)];
      acc[cb][0][r] = bv; acc[cb][1][r] = bv;
    }
  const float* bl = sB + h * 64 + 2 * j;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 bq[2][4];   // B operands of the current and the next k-quad (read from LDS one quad ahead)
  auto read_b = [&](int buf, int q) {
#pragma unroll
    for (int i = 0; i < 4; i++) bq[buf][i] = *reinterpret_cast<const f32x2*>(bl + (size_t)(4 * q + i) * 128);
  };
  read_b(0, 0);
  auto quad = [&](int s, int q) {
    const int cur = s & 1;   // UQ is even: the parity of q is the parity of its slot
    read_b(cur ^ 1, min(q + 1, nq - 1));
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
      for (int cb = 0; cb < NCB; cb++) {
        acc[cb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][cb][i], bq[cur][i][0], acc[cb][0], 0, 0, 0);
        acc[cb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][cb][i], bq[cur][i][1], acc[cb][1], 0, 0, 0);
      }
    }
  };
#define F3_WAIT(N_, s_) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(av[s_][0]), "+v"(av[s_][1]) : "n"(N_))
  int q0 = 0;
  for (; q0 + UQ < nq; q0 += UQ) {
#pragma unroll
    for (int s = 0; s < UQ; s++) {
      F3_WAIT((UQ - 1) * NCB, s);
      quad(s, q0 + s);
      int qn = min(q0 + s + UQ, nq - 1);
      asm volatile("" : "+s"(qn));
      load_a(s, qn);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  auto last = [&](auto sc) {
    constexpr int s = decltype(sc)::value;
    if constexpr (s < UQ) {
      F3_WAIT((UQ - 1 - s) * NCB, s);
      quad(s, q0 + s);
    }
  };
  last(std::integral_constant<int, 0>()); last(std::integral_constant<int, 1>());
  last(std::integral_constant<int, 2>()); last(std::integral_constant<int, 3>());
#undef F3_WAIT
  F3_STAMP(3);
  // ---- epilogue: + residual, in place on x; lane (j, h) holds tokens t0 + 2 j, + 1 of the rows acc_row(r, h)
#pragma unroll
  for (int cb = 0; cb < NCB; cb++) {
    f32x2 rv[16];
#pragma unroll
    for (int r = 0; r < 16; r++) rv[r] = *reinterpret_cast<const f32x2*>(a.x + (size_t)(out0 + cb * 32 + rs_acc_row(r, h)) * a.ld + t0 + 2 * j);
#pragma unroll
    for (int r = 0; r < 16; r++) {
      f32x2 o; o[0] = acc[cb][0][r] + rv[r][0]; o[1] = acc[cb][1][r] + rv[r][1];
      *reinterpret_cast<f32x2*>(a.x + (size_t)(out0 + cb * 32 + rs_acc_row(r, h)) * a.ld + t0 + 2 * j) = o;
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  F3_STAMP(4);
}

__global__ void fill_rand(float* p, size_t n, unsigned seed, float scale, float shift) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u ^ (seed * 0x9E3779B9u); x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    p[i] = ((int)(x & 0xFFFF) - 32768) * (1.0f / 65536.0f) * scale + shift;
  }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  std::vector<int> Ts;
  for (int i = 1; i < argc; i++) Ts.push_back(atoi(argv[i]));
  if (Ts.empty()) Ts = {16384};
  const int ld = 17408;
  float *H, *H0, *X, *X0, *Wt, *Wq, *Bs, *G, *Be;
  CK(hipMalloc(&H, (size_t)512 * ld * 4)); CK(hipMalloc(&H0, (size_t)512 * ld * 4)); CK(hipMalloc(&X, (size_t)256 * ld * 4)); CK(hipMalloc(&X0, (size_t)256 * ld * 4));
  CK(hipMalloc(&Wt, (size_t)256 * 512 * 4)); CK(hipMalloc(&Wq, (size_t)256 * 512 * 4)); CK(hipMalloc(&Bs, 256 * 4)); CK(hipMalloc(&G, 512 * 4)); CK(hipMalloc(&Be, 512 * 4));
  hipLaunchKernelGGL(fill_rand, dim3(4096), dim3(256), 0, 0, H0, (size_t)512 * ld, 1u, 4.0f, 0.3f);
  hipLaunchKernelGGL(fill_rand, dim3(4096), dim3(256), 0, 0, X0, (size_t)256 * ld, 5u, 2.0f, 0.0f);
  hipLaunchKernelGGL(fill_rand, dim3(512), dim3(256), 0, 0, Wt, (size_t)256 * 512, 2u, 0.2f, 0.0f);
  hipLaunchKernelGGL(fill_rand, dim3(1), dim3(256), 0, 0, Bs, (size_t)256, 3u, 1.0f, 0.0f);
  hipLaunchKernelGGL(fill_rand, dim3(1), dim3(256), 0, 0, G, (size_t)512, 6u, 1.0f, 1.0f);
  hipLaunchKernelGGL(fill_rand, dim3(1), dim3(256), 0, 0, Be, (size_t)512, 7u, 0.5f, 0.0f);
  CK(hipDeviceSynchronize());
  std::vector<float> wt((size_t)256 * 512), wq((size_t)256 * 512);
  CK(hipMemcpy(wt.data(), Wt, wt.size() * 4, hipMemcpyDeviceToHost));
  pack_linear_rs(wt.data(), 256, 512, wq.data());
  CK(hipMemcpy(Wq, wq.data(), wq.size() * 4, hipMemcpyHostToDevice));
  const size_t ldsb = (size_t)(512 * 64 + 512 + 1024) * 4;
  CK(hipFuncSetAttribute((const void*)ffn3_ln_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int T : Ts) {
    const int Tm = T / 64 * 64;
    auto reset = [&]() { (void)hipMemcpyAsync(H, H0, (size_t)512 * ld * 4, hipMemcpyDeviceToDevice, 0); (void)hipMemcpyAsync(X, X0, (size_t)256 * ld * 4, hipMemcpyDeviceToDevice, 0); };
    auto two = [&]() {
      launch_ln_gelu(H, G, Be, T, ld, 0);
      return launch_linear_fm(H, 512, nullptr, 0, Wt, Bs, X, 256, T, ld, X, nullptr, nullptr, 0, 1.0f, 0, Wq);
    };
    auto one = [&]() {
      Ffn3Args fa{H, G, Be, Wq, Bs, X, Tm, ld};
      hipLaunchKernelGGL(ffn3_ln_kernel<4>, dim3(Tm / 64), dim3(256), ldsb, 0, fa);
    };
    // values
    std::vector<float> r0((size_t)256 * Tm), r1((size_t)256 * Tm);
    reset(); if (!two()) { printf("linear refused\n"); return 1; }
    CK(hipDeviceSynchronize());
    CK(hipMemcpy2D(r0.data(), (size_t)Tm * 4, X, (size_t)ld * 4, (size_t)Tm * 4, 256, hipMemcpyDeviceToHost));
    reset(); one();
    CK(hipDeviceSynchronize());
    CK(hipMemcpy2D(r1.data(), (size_t)Tm * 4, X, (size_t)ld * 4, (size_t)Tm * 4, 256, hipMemcpyDeviceToHost));
    double maxd = 0, sum = 0;
    for (size_t i = 0; i < r0.size(); i++) { maxd = std::max(maxd, (double)fabs(r0[i] - r1[i])); sum += fabs(r0[i]); }
    printf("T=%d: fused vs two launches: max |diff| %.3g (mean |y| %.3f)\n", T, maxd, sum / r0.size());
    for (int which = 0; which < 2; which++) {
      std::vector<double> us;
      for (int rd = 0; rd < 7; rd++) {
        reset();
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < 10; i++) { if (which) one(); else two(); }
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        us.push_back(ms * 100.0);
      }
      std::sort(us.begin(), us.end());
      printf("T=%6d %-34s median %7.1f us  min %7.1f\n", T, which ? "fused ffn3_ln_kernel" : "ln_gelu + linear_rs (ffn.3)", us[3], us[0]);
      if (which) {
        std::vector<unsigned> tr(512 * 8);
        CK(hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(g_f3_trace), tr.size() * 4));
        const int nwg = Tm / 64;
        unsigned e0 = 0xffffffffu;
        for (int b = 0; b < nwg; b++) e0 = std::min(e0, tr[b * 8]);
        const char* nm[5] = {"entry", "rows loaded + summed", "GELU(LN) in LDS", "K loop done", "exit (stores drained)"};
        for (int k = 0; k < 5; k++) {
          std::vector<double> c;
          for (int b = 0; b < nwg; b++) c.push_back((double)(tr[b * 8 + k] - e0) / 100.0);
          std::sort(c.begin(), c.end());
          printf("        %-24s us after the first entry: min %5.1f  p50 %5.1f  max %5.1f\n", nm[k], c[0], c[c.size() / 2], c.back());
        }
      }
    }
  }
  return 0;
}
