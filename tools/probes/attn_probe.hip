// Probe: where does the LightGlue attention kernel (attn_ks_kernel in dani_slam_amd/csrc/lg_kernels.hip) lose time?
// A copy of the kernel with compile-time ablation flags, timed on the bench's shape (16 sets x 1000 keypoints, 4 heads).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o attn_probe attn_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct LgSet { int tok_off, n, partner_off, partner_n; float size_w, size_h; const float* kp; const float* desc; };
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
constexpr int AT_STRIDE = 33;

enum { F_XCD = 1, F_NOSOFTMAX = 2, F_NOSTAGE = 4, F_NOPV = 8, F_NOS = 16, F_NOBAR = 32, F_PRIO_SLOT = 64, F_PRIO_ODD = 128, F_PRIO_PHASE = 256, F_PRIO_PV = 512, F_PRIO_LOW1 = 1024, F_PIPE = 2048, F_DBUF = 4096 };

template <int OCC, int FLAGS>
__global__ __launch_bounds__(256, OCC) void attn_probe_kernel(const float* Q, const float* K, const float* V, float* O, int ld,
                                                              const LgSet* sets, int cross, float scale) {
  extern __shared__ float sKV_dyn[];
  float* sKV = sKV_dyn;   // [2 buffers (F_DBUF)][key half][K | V][64 d][33]
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  {
    const int L0 = bx + gridDim.x * (by + gridDim.y * bz);
    // static asymmetric wave priority: the two workgroups sharing a CU should not run their phases in lockstep
    if (FLAGS & F_PRIO_SLOT) { if (((L0 >> 3) >> 5) & 1) __builtin_amdgcn_s_setprio((FLAGS & F_PRIO_LOW1) ? 1 : 2); }
    if (FLAGS & F_PRIO_ODD) { if ((L0 >> 3) & 1) __builtin_amdgcn_s_setprio(2); }
  }
  if (FLAGS & F_XCD) {   // consecutive logical ids on one XCD: all query tiles of a (set, head) share that XCD's L2
    const int total = gridDim.x * gridDim.y * gridDim.z;
    const int L = bx + gridDim.x * (by + gridDim.y * bz);
    const int q = total / 8, r = total % 8, xcd = L % 8, slot = L / 8;
    const int Lp = xcd < r ? xcd * (q + 1) + slot : r * (q + 1) + (xcd - r) * q + slot;
    bx = Lp % gridDim.x; by = (Lp / gridDim.x) % gridDim.y; bz = Lp / (gridDim.x * gridDim.y);
  }
  const LgSet st = sets[bz];
  const int np = (st.n + 31) & ~31;
  const int q0 = bx * 64;
  if (q0 >= np) return;
  const int head = by;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
  const int qb = wave & 1, kh = wave >> 1;
  const int k_off = cross ? st.partner_off : st.tok_off, k_n = cross ? st.partner_n : st.n;
  const int qi = q0 + qb * 32 + j;
  const bool wave_active = (q0 + qb * 32) < np;
  const int qtok = st.tok_off + min(qi, np - 1);
  const float* Qh = Q + (size_t)head * 64 * ld;
  const float* Kh = K + (size_t)head * 64 * ld + k_off;
  const float* Vh = V + (size_t)head * 64 * ld + k_off;
  float* sK = sKV + (size_t)kh * 2 * 64 * AT_STRIDE;
  float* sV = sK + 64 * AT_STRIDE;

  float qreg[32];
#pragma unroll
  for (int t = 0; t < 32; t++) qreg[t] = Qh[(size_t)(2 * t + h) * ld + qtok] * (scale * 1.44269504088896341f);
  f32x16 o[2];
#pragma unroll
  for (int r = 0; r < 16; r++) { o[0][r] = 0.0f; o[1][r] = 0.0f; }
  float m = -1e30f, l = 0.0f;

  const int ntiles = (k_n + 31) / 32, half = (ntiles + 1) / 2;
  f32x4 pf[8];
  const float* gsrc[8];
  int loff[8], hsel[8];
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const int e = tid + 256 * u;
    const int q4 = e & 7, d = (e >> 3) & 63, kv = (e >> 9) & 1, hh = e >> 10;
    gsrc[u] = (kv ? Vh : Kh) + (size_t)d * ld + 4 * q4;
    loff[u] = (hh * 2 + kv) * 64 * AT_STRIDE + d * AT_STRIDE + 4 * q4;
    hsel[u] = hh * half;
  }
#define ATK_PREFETCH(it_)                                                                          \
  {                                                                                                \
    _Pragma("unroll") for (int u = 0; u < 8; u++)                                                  \
      pf[u] = *reinterpret_cast<const f32x4*>(gsrc[u] + min(hsel[u] + (it_), ntiles - 1) * 32);    \
  }
  ATK_PREFETCH(0);
  constexpr int BUF = 4 * 64 * AT_STRIDE;
  if (FLAGS & F_DBUF) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
#pragma unroll
      for (int c = 0; c < 4; c++) sKV[loff[u] + c] = pf[u][c];
    }
    __syncthreads();
    if (1 < half) ATK_PREFETCH(1);
  }
  for (int it = 0; it < half; it++) {
    if (FLAGS & F_DBUF) {
      // stage tile it + 1 into the other buffer (its registers arrived during the previous tile's compute), start the
      // load of tile it + 2, then compute on tile it; ONE barrier per iteration
      if (it + 1 < half) {
        float* dst = sKV + ((it + 1) & 1) * BUF;
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
          for (int c = 0; c < 4; c++) dst[loff[u] + c] = pf[u][c];
        }
        if (it + 2 < half) ATK_PREFETCH(it + 2);
      }
      sK = sKV + (it & 1) * BUF + (size_t)kh * 2 * 64 * AT_STRIDE;
      sV = sK + 64 * AT_STRIDE;
    } else
    if (!(FLAGS & F_NOSTAGE) || it == 0) {
      if (!(FLAGS & F_NOBAR)) __syncthreads();
#pragma unroll
      for (int u = 0; u < 8; u++) {
#pragma unroll
        for (int c = 0; c < 4; c++) sKV[loff[u] + c] = pf[u][c];
      }
      if (!(FLAGS & F_NOBAR)) __syncthreads();
      if (!(FLAGS & 0) && it + 1 < half) ATK_PREFETCH(it + 1);
    }
    const int kt = kh * half + it;
    if (wave_active && kt < ntiles) {
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; r++) s[r] = (FLAGS & F_NOS) ? qreg[r] : 0.0f;
      if (FLAGS & F_PRIO_PHASE) __builtin_amdgcn_s_setprio(3);
      if (!(FLAGS & F_NOS)) {
        if (FLAGS & F_PIPE) {   // operand reads one group (2 k-steps) ahead of the MFMAs
          float kv[16][2];
          kv[0][0] = sK[(0 + h) * AT_STRIDE + j]; kv[0][1] = sK[(2 + h) * AT_STRIDE + j];
#pragma unroll
          for (int g = 0; g < 16; g++) {
            if (g + 1 < 16) { kv[g + 1][0] = sK[(4 * (g + 1) + h) * AT_STRIDE + j]; kv[g + 1][1] = sK[(4 * (g + 1) + 2 + h) * AT_STRIDE + j]; }
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[g][0], qreg[2 * g], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[g][1], qreg[2 * g + 1], s, 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          }
        } else {
#pragma unroll
        for (int t = 0; t < 32; t++) s = __builtin_amdgcn_mfma_f32_32x32x2f32(sK[(2 * t + h) * AT_STRIDE + j], qreg[t], s, 0, 0, 0);
        }
      }
      if (FLAGS & F_PRIO_PHASE) __builtin_amdgcn_s_setprio(0);
      if (!(FLAGS & F_NOSOFTMAX)) {
        const int kbase = kt * 32;
        if (kbase + 32 > k_n) {
#pragma unroll
          for (int r = 0; r < 16; r++)
            if (kbase + acc_row(r, h) >= k_n) s[r] = -1e30f;
        }
        float mx = s[0];
#pragma unroll
        for (int r = 1; r < 16; r++) mx = fmaxf(mx, s[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m, mx);
        float ls = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r++) { s[r] = __builtin_amdgcn_exp2f(s[r] - m_new); ls += s[r]; }
        if (__ballot(m_new > m)) {
          const float alpha = __builtin_amdgcn_exp2f(m - m_new);
          l *= alpha;
#pragma unroll
          for (int r = 0; r < 16; r++) { o[0][r] *= alpha; o[1][r] *= alpha; }
        }
        l += ls;
        m = m_new;
      } else {
        l += s[0];
      }
      if (FLAGS & F_PRIO_PV) __builtin_amdgcn_s_setprio(3);
      if (!(FLAGS & F_NOPV)) {
        if (FLAGS & F_PIPE) {
          float vv[16][2];
          vv[0][0] = sV[j * AT_STRIDE + acc_row(0, h)]; vv[0][1] = sV[(32 + j) * AT_STRIDE + acc_row(0, h)];
#pragma unroll
          for (int r = 0; r < 16; r++) {
            if (r + 1 < 16) { vv[r + 1][0] = sV[j * AT_STRIDE + acc_row(r + 1, h)]; vv[r + 1][1] = sV[(32 + j) * AT_STRIDE + acc_row(r + 1, h)]; }
            o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[r][0], s[r], o[0], 0, 0, 0);
            o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[r][1], s[r], o[1], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          }
        } else {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int key = acc_row(r, h);
          o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(sV[j * AT_STRIDE + key], s[r], o[0], 0, 0, 0);
          o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(sV[(32 + j) * AT_STRIDE + key], s[r], o[1], 0, 0, 0);
        }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; r++) { o[0][r] += s[r]; }
      }
    }
    if (FLAGS & F_DBUF) __syncthreads();
  }
#undef ATK_PREFETCH
  __syncthreads();
  float* xch = sKV + (size_t)qb * (64 * 34);
  if (kh == 1) {
    float* p = xch + lane * 34;
    p[0] = m; p[1] = l;
#pragma unroll
    for (int r = 0; r < 16; r++) { p[2 + r] = o[0][r]; p[18 + r] = o[1][r]; }
  }
  __syncthreads();
  if (kh == 0 && qi < np) {
    const float* p = xch + lane * 34;
    const float m1 = p[0], l1 = p[1];
    const float mm = fmaxf(m, m1);
    const float a0 = __builtin_amdgcn_exp2f(m - mm), a1 = __builtin_amdgcn_exp2f(m1 - mm);
    const float lt0 = l * a0 + l1 * a1;
    const float lt = lt0 + __shfl_xor(lt0, 32);
    const float inv = (lt > 0.0f && qi < st.n) ? 1.0f / lt : 0.0f;
    float* Oh = O + (size_t)head * 64 * ld + st.tok_off + qi;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      Oh[(size_t)acc_row(r, h) * ld] = (o[0][r] * a0 + p[2 + r] * a1) * inv;
      Oh[(size_t)(32 + acc_row(r, h)) * ld] = (o[1][r] * a0 + p[18 + r] * a1) * inv;
    }
  }
}

#define VARIANTS \
  RUN(2, F_PIPE, "reads ahead (product)") \
  RUN(2, F_PIPE | F_DBUF, "reads ahead + double-buffered LDS, 1 barrier") \
  RUN(1, F_PIPE | F_DBUF, "same, 1 WG / CU") \
  RUN(2, F_PIPE, "reads ahead (product) again")

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#include <algorithm>
#include <string>
typedef void (*KernelFn)(const float*, const float*, const float*, float*, int, const LgSet*, int, float);
struct Variant { std::string name; KernelFn fn; int occ; std::vector<double> us; };

int main(int argc, char** argv) {
  const int nsets = 16, n = argc > 1 ? atoi(argv[1]) : 1000, np = (n + 31) & ~31;
  const int ld = nsets * np + 256;
  std::vector<LgSet> hs(nsets);
  for (int i = 0; i < nsets; i++) { hs[i] = LgSet{i * np, n, (i ^ 1) * np, n, 640.f, 480.f, nullptr, nullptr}; }
  std::vector<float> hq((size_t)256 * ld);
  srand(3);
  for (auto& v : hq) v = (float)rand() / (float)RAND_MAX - 0.5f;
  float *Q, *K, *V, *O; LgSet* ds;
  CK(hipMalloc(&Q, hq.size() * 4)); CK(hipMalloc(&K, hq.size() * 4)); CK(hipMalloc(&V, hq.size() * 4)); CK(hipMalloc(&O, hq.size() * 4));
  CK(hipMalloc(&ds, sizeof(LgSet) * nsets));
  CK(hipMemcpy(Q, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
  for (auto& v : hq) v = (float)rand() / (float)RAND_MAX - 0.5f;
  CK(hipMemcpy(K, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
  for (auto& v : hq) v = (float)rand() / (float)RAND_MAX - 0.5f;
  CK(hipMemcpy(V, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(ds, hs.data(), sizeof(LgSet) * nsets, hipMemcpyHostToDevice));
  const double flops = (double)nsets * 4.0 * 256.0 * n * (double)n;   // 4 heads x (QK^T + PV) x 2 flop x 64 d
  printf("attention probe: %d sets x %d keypoints, grid %d x 4 x %d\n", nsets, n, (np + 63) / 64, nsets);
  std::vector<Variant> vs;
#define RUN(occ, fl, nm) vs.push_back(Variant{nm, attn_probe_kernel<occ, fl>, occ, {}});
  VARIANTS
  for (auto& v : vs) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(v.fn), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 4 * 64 * AT_STRIDE * 4));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  dim3 grid((np + 63) / 64, 4, nsets);
  const int rounds = 7, reps = 10;
  for (int rd = 0; rd < rounds; rd++)
    for (auto& v : vs) {
      hipLaunchKernelGGL(v.fn, grid, dim3(256), 2 * 4 * 64 * AT_STRIDE * 4, 0, Q, K, V, O, ld, ds, 0, 0.125f);
      CK(hipEventRecord(a, 0));
      for (int i = 0; i < reps; i++) hipLaunchKernelGGL(v.fn, grid, dim3(256), 2 * 4 * 64 * AT_STRIDE * 4, 0, Q, K, V, O, ld, ds, 0, 0.125f);
      CK(hipEventRecord(b, 0));
      CK(hipEventSynchronize(b));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, a, b));
      v.us.push_back(ms * 1e3 / reps);
    }
  for (auto& v : vs) {
    std::sort(v.us.begin(), v.us.end());
    const double med = v.us[v.us.size() / 2];
    printf("%-52s occ=%d  median %7.1f us  min %7.1f  max %7.1f   %6.1f TFLOP/s\n", v.name.c_str(), v.occ, med, v.us.front(), v.us.back(),
           flops / (med * 1e-6) / 1e12);
  }
  return 0;
}
