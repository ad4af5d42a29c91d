// lg_kernels.hip -- LightGlue on the gfx950 FP32 matrix cores.
//
// Replaces the TorchScript forward the reference runs on LibTorch-CPU inside LightGlue::matchDescriptors
// (/root/reference/src/LightGlue.cpp:84; the network itself is the absent blob lightglue_scripted_v2.pt, so the
// architecture is the published LightGlue/SuperPoint model restated in oracle/lg_oracle.py -- parity unpinned).
//
// Layout: every activation is FEATURE-MAJOR  [features][ld]  (ld = padded token count of the whole batch, all
// keypoint sets concatenated, each set padded to a multiple of 32 tokens).  With tokens on the fast axis
//   * a linear layer  Y^T[out][tok] = W[out][k] X^T[k][tok]  is exactly the 1x1-conv GEMM shape: weights are the
//     MFMA A operand (one conflict-free ds_read_b32 per lane), tokens are the B operand, and the 32x32 accumulator
//     has tokens on the lanes -> coalesced 128-B stores;
//   * attention computes S^T = K Q^T so that a query's softmax statistics live on ONE lane pair (max / sum are
//     16 in-lane ops + one shfl_xor 32), and the probabilities in the S^T accumulator ARE the B operand of the
//     O^T = V^T P^T product register for register (no LDS round trip, no transposes).
// All contractions use v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate); bound: MFMA.
//
// Ragged sets: a set with n = 32k + t keypoints, t <= LG_TAIL_MAX, keeps 32k tokens in its main region and its t tail
// tokens in a region shared by all sets (common.h, LgSet).  The quadtree returns one or two keypoints more than asked
// for, and with whole-block padding that single keypoint cost every launch a third, almost empty round of workgroups
// (1088 instead of 1024 on 512 slots: attention 193 us instead of 150 us per launch at 16 x 1025 keypoints).
#include "common.h"
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include "lin_epilogue.h"
#include "lin_rs.h"
#include "x3_common.h"

namespace dani {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
}
__device__ __forceinline__ float wave_sumf(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// ---------------------------------------------------------------------------------------------
// linear_fm_kernel:  Y[o][t] = epilogue( sum_k W[o][k] * X[k][t] + b[o] ),  X = concat(X0 rows, X1 rows).
// Workgroup tile 64 outputs x TT tokens (TT = 64 is what launch_linear_fm ships; the larger tiles stay for the tuning
// hook g_lin_force_tt).  K is streamed through LDS in chunks of KC rows with register prefetch (same pipeline as
// conv_mfma_kernel).
// Epilogues: rotary embedding on rows < rot_rows (Wqkv with rows pre-permuted to [q|k|v][head][d]),
//            scale alpha, residual add.
// ---------------------------------------------------------------------------------------------
struct LinArgs {
  const float* x0; const float* x1; int K0, K1;
  const float* w; const float* bias;
  float* y; const float* res; const float* cosb; const float* sinb;
  int OUT, T, ld, rot_rows;
  float alpha;
  int ksplit;   // 64-token kernel: split K between the wave pairs of a tail tile of <= 32 tokens (DANI_LIN_NO_KSPLIT=1 turns it off)
};

constexpr int LIN_KC = 32, LIN_NCO = 64;

#ifdef DANI_CONV_STAMPS
// Debug build only (tools/probes/lin_bench.hip with -DDANI_CONV_STAMPS): wall clock (100 MHz) and core clock of workgroup 0,
// and per workgroup the wall-clock ticks (10 ns) at entry, first chunk staged, K loop done, exit.
__device__ unsigned long long g_lin_stamps[4];
__device__ unsigned int g_lin_trace[8192 * 4];
#define LIN_TRACE(k_) do { if (tid == 0 && blockIdx.x < 8192) g_lin_trace[blockIdx.x * 4 + (k_)] = (unsigned)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define LIN_TRACE(k_) do { } while (0)
#endif

// TT = tokens per workgroup.  256: wave = 64 out x 64 tok (2x2 accumulators); 128: wave = 64 out x 32 tok (2x1);
// 64: waves arranged 2x2, wave = 32 out x 32 tok (1x1).  Smaller tiles trade operand reuse for more workgroups:
// with ~17k tokens and 256 outputs a 256-token tile gives only 264 workgroups for 256 CUs (two rounds, 50 % idle).
template <int TT>
__global__ __launch_bounds__(256, 2) void linear_fm_kernel(LinArgs a) {
  constexpr int NCB = (TT >= 128) ? 2 : 1;          // cout blocks per wave
  constexpr int NTB = (TT == 256) ? 2 : 1;          // token blocks per wave
  constexpr int KC = (TT <= 128) ? 64 : LIN_KC;      // rows of X per LDS chunk (deeper for the small tiles: fewer barriers)
  constexpr int NX = KC * TT / 4 / 256;             // float4 of X per thread per chunk
  constexpr int NWF = KC * LIN_NCO / 4 / 256;       // float4 of W per thread per chunk
  __shared__ __attribute__((aligned(16))) float lds[KC * TT + KC * LIN_NCO];
  float* sX = lds;
  float* sW = lds + KC * TT;
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform, and known to be: row bases go to the scalar unit
  // 1-D grid, XCD-aware: workgroup ids are dealt round-robin to the 8 XCDs (id % 8), each with its own L2.  All output
  // tiles of one token tile go to the same XCD, back to back, so that the X tile they share is fetched into one L2
  // once instead of into all eight.
  const int otiles = (a.OUT + LIN_NCO - 1) / LIN_NCO, ntiles = (a.T + TT - 1) / TT;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tile = (slot / otiles) * 8 + xcd, ot = slot % otiles;
  if (tile >= ntiles) return;
  const int tok0 = tile * TT;
#ifdef DANI_CONV_STAMPS
  if (blockIdx.x == 0 && tid == 0) { g_lin_stamps[0] = __builtin_amdgcn_s_memrealtime(); g_lin_stamps[1] = __builtin_readcyclecounter(); }
#endif
  LIN_TRACE(0);
  const int K = a.K0 + a.K1, nchunks = K / KC;
  const float* wt = a.w + (size_t)ot * K * LIN_NCO;
  // Tail tile of the 64-token kernel (at most 32 tokens left: the shared tail region of a batch of ~1025-keypoint sets):
  // its two token-half waves would idle while the other two run the whole K loop, and the workgroup, which lands as a
  // FIFTH one on a CU that already holds four, stretches that CU's launch by 25 %.  Instead both wave pairs take the first
  // 32 tokens and split every K chunk between them (k-pair groups alternate); the halves meet in LDS at the end.
  const bool ksplit = (TT == 64) && a.ksplit && (a.T - tok0 <= 32);    // uniform per workgroup
  const int kh = ksplit ? (wave & 1) : 0;
  const int wtok = (TT >= 128) ? wave * (TT / 4) : (ksplit ? 0 : (wave & 1) * 32);   // wave's first token inside the tile
  const int wout = (TT >= 128) ? 0 : (wave >> 1) * 32;                 // wave's first output row inside the 64

  f32x16 acc[NCB][NTB];
  f32x16 acc2;   // TT == 64 only: second accumulator chain (odd k-pairs), see the note in attn_kernel
#pragma unroll
  for (int cb = 0; cb < NCB; cb++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const float bv = kh ? 0.0f : a.bias[ot * LIN_NCO + wout + cb * 32 + acc_row(r, h)];
#pragma unroll
      for (int tb = 0; tb < NTB; tb++) acc[cb][tb][r] = bv;
      acc2[r] = 0.0f;
    }

  f32x4 px[NX], pw[NWF];
  int xoff[NX];
#pragma unroll
  for (int u = 0; u < NX; u++) {
    const int e = tid + 256 * u;
    xoff[u] = (e / (TT / 4)) * a.ld + tok0 + 4 * (e % (TT / 4));
  }
#define LIN_PREFETCH(ch_)                                                                            \
  {                                                                                                  \
    const int k0_ = (ch_) * KC;                                                                      \
    const float* src_ = (k0_ < a.K0) ? a.x0 + (size_t)k0_ * a.ld : a.x1 + (size_t)(k0_ - a.K0) * a.ld; \
    _Pragma("unroll") for (int u = 0; u < NX; u++) px[u] = *reinterpret_cast<const f32x4*>(src_ + xoff[u]); \
    const f32x4* ws_ = reinterpret_cast<const f32x4*>(wt + (size_t)(ch_) * (KC * LIN_NCO));         \
    _Pragma("unroll") for (int u = 0; u < NWF; u++) pw[u] = ws_[tid + 256 * u];                      \
  }
  const float* bA = sW + h * LIN_NCO + wout + j;
  const float* bB = sX + h * TT + wtok + j;

  LIN_PREFETCH(0);
  for (int ch = 0; ch < nchunks; ch++) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NX; u++) reinterpret_cast<f32x4*>(sX)[tid + 256 * u] = px[u];
#pragma unroll
    for (int u = 0; u < NWF; u++) reinterpret_cast<f32x4*>(sW)[tid + 256 * u] = pw[u];
    __syncthreads();
    if (ch == 0) LIN_TRACE(1);
    if (ch + 1 < nchunks) LIN_PREFETCH(ch + 1);
    // Operand reads run PD k-pair groups ahead of the MFMAs that use them (groups of 2 k-pairs: the compiler merges
    // their reads into ds_read2st64): left alone, hipcc sinks every read next to its MFMA and waits on it.
    // STEP = 2: the k-split tail tile, this wave takes every second group starting at its own (pointers pre-offset).
    auto chunk = [&](auto step_tag, const float* bA_, const float* bB_) {
      constexpr int STEP = decltype(step_tag)::value;
      constexpr int G = 2, NG = KC / 2 / G / STEP, PD = 1;   // measured best of G in {1, 2, 4, 8} x PD in {1, 2} (lin_bench)
      float av[NG][G][NCB], bv[NG][G][NTB];
      auto rd = [&](int g) {
#pragma unroll
        for (int q = 0; q < G; q++) {
          const int kp = g * STEP * G + q;
#pragma unroll
          for (int cb = 0; cb < NCB; cb++) av[g][q][cb] = bA_[(kp * 2) * LIN_NCO + cb * 32];
#pragma unroll
          for (int tb = 0; tb < NTB; tb++) bv[g][q][tb] = bB_[(kp * 2) * TT + tb * 32];
        }
      };
#pragma unroll
      for (int g = 0; g < PD; g++) rd(g);
#pragma unroll
      for (int g = 0; g < NG; g++) {
        if (g + PD < NG) rd(g + PD);
#pragma unroll
        for (int q = 0; q < G; q++) {
          if constexpr (NCB * NTB == 1) {
            if (q & 1) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g][q][0], bv[g][q][0], acc2, 0, 0, 0);
            else acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g][q][0], bv[g][q][0], acc[0][0], 0, 0, 0);
          } else {
#pragma unroll
            for (int cb = 0; cb < NCB; cb++)
#pragma unroll
              for (int tb = 0; tb < NTB; tb++)
                acc[cb][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g][q][cb], bv[g][q][tb], acc[cb][tb], 0, 0, 0);
          }
        }
        // one group of LDS reads (next group's operands), then this group's MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, G * (NCB + NTB), 0);
        __builtin_amdgcn_sched_group_barrier(0x008, G * NCB * NTB, 0);
      }
    };
    if constexpr (TT == 64) {
      if (ksplit) chunk(std::integral_constant<int, 2>(), bA + kh * 4 * LIN_NCO, bB + kh * 4 * TT);
      else chunk(std::integral_constant<int, 1>(), bA, bB);
    } else {
      chunk(std::integral_constant<int, 1>(), bA, bB);
    }
  }
#undef LIN_PREFETCH
  if constexpr (NCB * NTB == 1) {
#pragma unroll
    for (int r = 0; r < 16; r++) acc[0][0][r] += acc2[r];
    if (ksplit) {   // fold the second K half into the first through LDS (the chunk buffers are free after the barrier)
      __syncthreads();
      float* xch = lds + ((wave >> 1) * 64 + lane) * 17;
      if (kh) {
#pragma unroll
        for (int r = 0; r < 16; r++) xch[r] = acc[0][0][r];
      }
      __syncthreads();
      if (kh) return;
#pragma unroll
      for (int r = 0; r < 16; r++) acc[0][0][r] += xch[r];
    }
  }

#ifdef DANI_CONV_STAMPS
  if (blockIdx.x == 0 && tid == 0) { g_lin_stamps[2] = __builtin_readcyclecounter(); g_lin_stamps[3] = __builtin_amdgcn_s_memrealtime(); }
#endif
  LIN_TRACE(2);
  LinEpi ep;
  ep.y = a.y; ep.res = a.res; ep.cosb = a.cosb; ep.sinb = a.sinb; ep.bias = nullptr; ep.ld = a.ld; ep.rot_rows = a.rot_rows; ep.alpha = a.alpha;
#pragma unroll
  for (int tb = 0; tb < NTB; tb++) {
    const int tok = tok0 + wtok + tb * 32 + j;
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) lin_epilogue<false>(acc[cb][tb], ot * LIN_NCO + wout + cb * 32, h, tok, tok < a.T, ep);
  }
  LIN_TRACE(3);
}

std::atomic<int> g_lin_force_tt{0};   // tuning hook of tools/probes/lin_bench.hip: force the token tile (0 = the 64-token tile)

void pack_linear_rs_host(const float* w_row_major, int out, int in, float* dst) { pack_linear_rs(w_row_major, out, in, dst); }

std::atomic<int> g_lin_force_rs{-1};  // tuning hook: -1 = choose per shape (linear_rs_choose), 0 = never, 1..3 = linear_rs_kernel with that NCB

bool launch_linear_fm(const float* x0, int K0, const float* x1, int K1, const float* w_packed, const float* bias_packed,
                      float* y, int OUT, int T, int ld, const float* res, const float* cosb, const float* sinb,
                      int rot_rows, float alpha, hipStream_t s, const float* w_rs) {
  if (OUT % 64 || rot_rows % 64 || (size_t)OUT * ld >= ((size_t)1 << 30)) return false;   // contract of lin_epilogue.h: refused, not launched
  // register-streaming form (lin_rs.h) where the grid it gives fills the chip at least as well as the LDS-tiled one
  static const int rs_env = getenv("DANI_LIN_RS") ? atoi(getenv("DANI_LIN_RS")) : -1;
  const int force_rs = g_lin_force_rs.load(std::memory_order_relaxed), force_tt = g_lin_force_tt.load(std::memory_order_relaxed);   // one read per launch (dani_lg_set_kernel_option may run on another thread)
  const int rs_mode = force_rs >= 0 ? force_rs : rs_env;
  if (w_rs && rs_mode != 0 && force_tt == 0) {
    const int ncb = rs_mode > 0 ? rs_mode : linear_rs_choose(OUT, K0 + K1, T, ld);
    if (ncb > 0) {
      LinRsArgs r;
      r.x0 = x0; r.x1 = x1 ? x1 : x0; r.K0 = K0; r.K1 = K1; r.wq = w_rs; r.bias = bias_packed; r.y = y; r.res = res;
      r.cosb = cosb; r.sinb = sinb; r.OUT = OUT; r.T = T; r.ld = ld; r.rot_rows = rot_rows; r.alpha = alpha; r.n_tail_wg = 0; r.tail_tok0 = 0;
      if (launch_linear_rs(ncb, r, s)) return true;
    }
  }
  LinArgs a;
  a.x0 = x0; a.x1 = x1 ? x1 : x0; a.K0 = K0; a.K1 = K1; a.w = w_packed; a.bias = bias_packed; a.y = y; a.res = res;
  a.cosb = cosb; a.sinb = sinb; a.OUT = OUT; a.T = T; a.ld = ld; a.rot_rows = rot_rows; a.alpha = alpha;
  static const int ksplit_on = getenv("DANI_LIN_NO_KSPLIT") ? 0 : 1;
  a.ksplit = ksplit_on;
  const int otiles = (OUT + LIN_NCO - 1) / LIN_NCO;
  // The 64-token tile (90 VGPRs, 32 KB LDS, 5 workgroups per CU) beats the 128- and 256-token ones (128 / 148 VGPRs,
  // 48 KB, 3 per CU) on every LightGlue shape from 16 k to 66 k tokens by 7-17 % (tools/probes/lin_bench.hip), so it is
  // always used; the larger instantiations stay reachable through the probe's tuning hook.
  auto grid = [&](int tt) { return dim3((unsigned)(((T + tt - 1) / tt + 7) / 8 * 8 * otiles)); };   // see the kernel: 8 XCDs
  if (force_tt == 256) hipLaunchKernelGGL(linear_fm_kernel<256>, grid(256), dim3(256), 0, s, a);
  else if (force_tt == 128) hipLaunchKernelGGL(linear_fm_kernel<128>, grid(128), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(linear_fm_kernel<64>, grid(64), dim3(256), 0, s, a);
  return true;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm(512, eps 1e-5, affine) + exact GELU, in place on H [512][ld].
// ---------------------------------------------------------------------------------------------
// Block = 64 tokens x 16 feature groups (1024 threads): wave g holds features [32g, 32g+32) of 64 tokens in 32
// VGPRs (high occupancy, one read + one write of the tensor); statistics are reduced across the 16 waves via LDS.
// erf to < 1e-7 absolute (two minimax branches: a polynomial in a^2 below 0.9277, 1 - exp(polynomial) above; both
// are evaluated and selected, 22 instructions with one v_exp_f32).  The library erff costs ~50 instructions per
// element with its branches, which made this HBM-shaped kernel half VALU-bound (1691 VALU instructions per thread for
// 32 elements).  Checked against scipy.special.erf on 2 M points in [-6, 6]: max |error| 7.5e-8.
__device__ __forceinline__ float lg_erff(float a) {
  const float t = fabsf(a), s = a * a;
  float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
  const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
  r = fmaf(r, s, u);
  r = fmaf(r, t, -1.06777877e-1f);
  r = fmaf(r, t, -6.34846687e-1f);
  r = fmaf(r, t, -1.28717512e-1f);
  r = fmaf(r, t, -t);
  r = copysignf(1.0f - __builtin_amdgcn_exp2f(r * 1.44269504088896341f), a);
  float q = -5.96761703e-4f;
  q = fmaf(q, s, 4.99119423e-3f);
  q = fmaf(q, s, -2.67681349e-2f);
  q = fmaf(q, s, 1.12819925e-1f);
  q = fmaf(q, s, -3.76125336e-1f);
  q = fmaf(q, s, 1.28379166e-1f);
  q = fmaf(q, a, a);
  return t > 0.927734375f ? r : q;
}

__global__ __launch_bounds__(1024) void ln_gelu_kernel(float* hbuf, const float* gamma, const float* beta, int T, int ld) {
  __shared__ float red[16][64];
  const int lp = threadIdx.x & 63, g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: row bases on the scalar unit
  const int tok = blockIdx.x * 64 + lp;
  const bool ok = tok < T;
  float v[32];
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 32; i++) {
    v[i] = ok ? (hbuf + (size_t)(g * 32 + i) * ld)[tok] : 0.0f;
    s += v[i];
  }
  red[g][lp] = s;
  __syncthreads();
  float tot = 0.0f;
#pragma unroll
  for (int k = 0; k < 16; k++) tot += red[k][lp];
  const float mean = tot * (1.0f / 512.0f);
  __syncthreads();
  float s2 = 0.0f;
#pragma unroll
  for (int i = 0; i < 32; i++) { const float d = v[i] - mean; s2 += d * d; }
  red[g][lp] = s2;
  __syncthreads();
  float tv = 0.0f;
#pragma unroll
  for (int k = 0; k < 16; k++) tv += red[k][lp];
  const float rstd = 1.0f / sqrtf(tv * (1.0f / 512.0f) + 1e-5f);
  if (!ok) return;
#pragma unroll
  for (int i = 0; i < 32; i++) {
    const int f = g * 32 + i;
    const float y = (v[i] - mean) * rstd * gamma[f] + beta[f];
    (hbuf + (size_t)f * ld)[tok] = 0.5f * y * (1.0f + lg_erff(y * 0.70710678118654752440f));
  }
}

// TOK = 16 or 32 tokens per workgroup instead of 64.  Lane = (token l % TOK, feature group l / TOK), wave g holds features
// [32g + R q, 32g + R q + R) of its TOK tokens in R = 32 TOK / 64 VGPRs; the statistics are reduced over the 64 / TOK groups by
// shuffles and over the 16 waves through LDS.  Same arithmetic per element; the order of the two reductions differs from
// ln_gelu_kernel's (LightGlue is held to 1e-3).  16: small token counts (one or two pairs are 33-66 of the 64-token workgroups on
// 256 CUs).  32: the large batches -- a 64-token workgroup runs load -> reduce -> erf -> store in lock step, and 257 of them on 256
// CUs leave nothing to overlap those phases with; 513 workgroups put two on every CU.
template <int TOK>
__global__ __launch_bounds__(1024) void ln_gelu_tok_kernel(float* hbuf, const float* gamma, const float* beta, int T, int ld) {
  constexpr int NQ = 64 / TOK, R = 32 / NQ;
  __shared__ float red[16][TOK];
  const int l = threadIdx.x & 63, tl = l & (TOK - 1), q = l / TOK, g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tok = blockIdx.x * TOK + tl;
  const bool ok = tok < T;
  float v[R];
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < R; i++) {
    v[i] = ok ? (hbuf + (size_t)(g * 32 + q * R + i) * ld)[tok] : 0.0f;
    s += v[i];
  }
#pragma unroll
  for (int o = TOK; o < 64; o <<= 1) s += __shfl_xor(s, o);
  if (q == 0) red[g][tl] = s;
  __syncthreads();
  float tot = 0.0f;
#pragma unroll
  for (int k = 0; k < 16; k++) tot += red[k][tl];
  const float mean = tot * (1.0f / 512.0f);
  __syncthreads();
  float s2 = 0.0f;
#pragma unroll
  for (int i = 0; i < R; i++) { const float d = v[i] - mean; s2 += d * d; }
#pragma unroll
  for (int o = TOK; o < 64; o <<= 1) s2 += __shfl_xor(s2, o);
  if (q == 0) red[g][tl] = s2;
  __syncthreads();
  float tv = 0.0f;
#pragma unroll
  for (int k = 0; k < 16; k++) tv += red[k][tl];
  const float rstd = 1.0f / sqrtf(tv * (1.0f / 512.0f) + 1e-5f);
  if (!ok) return;
#pragma unroll
  for (int i = 0; i < R; i++) {
    const int f = g * 32 + q * R + i;
    const float y = (v[i] - mean) * rstd * gamma[f] + beta[f];
    (hbuf + (size_t)f * ld)[tok] = 0.5f * y * (1.0f + lg_erff(y * 0.70710678118654752440f));
  }
}

void launch_ln_gelu(float* hbuf, const float* gamma, const float* beta, int T, int ld, hipStream_t s) {
  static const int small_t = getenv("DANI_LN_SMALL_T") ? atoi(getenv("DANI_LN_SMALL_T")) : 64 * 128;   // fewer than 128 of the 64-token workgroups
  static const int big_tok = getenv("DANI_LN_TOK") ? atoi(getenv("DANI_LN_TOK")) : 32;                 // tokens per workgroup for the large batches (32 or 64)
  if (T < small_t) hipLaunchKernelGGL(ln_gelu_tok_kernel<16>, dim3((T + 15) / 16), dim3(1024), 0, s, hbuf, gamma, beta, T, ld);
  else if (big_tok == 32) hipLaunchKernelGGL(ln_gelu_tok_kernel<32>, dim3((T + 31) / 32), dim3(1024), 0, s, hbuf, gamma, beta, T, ld);
  else if (big_tok == 16) hipLaunchKernelGGL(ln_gelu_tok_kernel<16>, dim3((T + 15) / 16), dim3(1024), 0, s, hbuf, gamma, beta, T, ld);
  else hipLaunchKernelGGL(ln_gelu_kernel, dim3((T + 63) / 64), dim3(1024), 0, s, hbuf, gamma, beta, T, ld);
}

// ---------------------------------------------------------------------------------------------
// Learnable-Fourier positional encoding: proj = Wr (32x2) . kpt_n ; tables cos/sin [32][ld].
// Also performs LightGlue::normalizeKeypoints (src/LightGlue.cpp:28-51): k = (k - size/2) / (max(size)/2).
// ---------------------------------------------------------------------------------------------
__global__ void posenc_kernel(const LgSet* sets, int nsets, const float* wr, float* cosb, float* sinb, int ld) {
  const int set = blockIdx.y;
  const LgSet st = sets[set];
  const int i = blockIdx.x * 64 + threadIdx.x;   // position in the set's span: padded main region, then tail tokens
  const int f0 = threadIdx.y * 8;                // 4 feature groups of 8 per token: 64 library sin / cos calls in one thread made
                                                 // this the longest of LightGlue's small kernels (29 us for one pair)
  const int mp = (st.nmain + 31) & ~31, nt = st.n - st.nmain;
  int tok, kpi = -1;
  if (i < mp) { tok = st.tok_off + i; if (i < st.nmain) kpi = i; }
  else if (i - mp < nt) { tok = st.tail_off + (i - mp); kpi = st.nmain + (i - mp); }
  else return;
  const float sx = st.size_w / 2.0f, sy = st.size_h / 2.0f;
  const float scale = fmaxf(st.size_w, st.size_h) / 2.0f;
  float x = 0.0f, y = 0.0f;
  if (kpi >= 0) { x = (st.kp[2 * kpi] - sx) / scale; y = (st.kp[2 * kpi + 1] - sy) / scale; }
#pragma unroll 4
  for (int f = f0; f < f0 + 8; f++) {
    const float p = wr[2 * f] * x + wr[2 * f + 1] * y;
    cosb[(size_t)f * ld + tok] = cosf(p);
    sinb[(size_t)f * ld + tok] = sinf(p);
  }
}

void launch_posenc(const LgSet* sets, int nsets, int max_span, const float* wr, float* cosb, float* sinb, int ld, hipStream_t s) {
  hipLaunchKernelGGL(posenc_kernel, dim3((max_span + 63) / 64, nsets), dim3(64, 4), 0, s, sets, nsets, wr, cosb, sinb, ld);
}

// each set's token-major [n][256] descriptors (LgSet::desc) -> feature-major [256][ld] (zero for padding tokens)
__global__ __launch_bounds__(256) void desc_to_fm_kernel(const LgSet* sets, float* x, int ld) {
  __shared__ float tile[64][65];
  const int set = blockIdx.z;
  const LgSet st = sets[set];
  const int mp = (st.nmain + 31) & ~31, nt = st.n - st.nmain;
  const int t0 = blockIdx.x * 64, f0 = blockIdx.y * 64;   // t0: position in the span (padded main region, then tail)
  if (t0 >= mp + nt) return;
  const int c = threadIdx.x & 63, rq = threadIdx.x >> 6;
  for (int k = 0; k < 16; k++) {
    const int tl = rq + 4 * k, ti = t0 + tl;
    int kpi = -1;
    if (ti < mp) { if (ti < st.nmain) kpi = ti; }
    else if (ti - mp < nt) kpi = st.nmain + (ti - mp);
    tile[tl][c] = (kpi >= 0) ? st.desc[(size_t)kpi * 256 + f0 + c] : 0.0f;
  }
  __syncthreads();
  const int ti = t0 + c;
  int tok = -1;
  if (ti < mp) tok = st.tok_off + ti;
  else if (ti - mp < nt) tok = st.tail_off + (ti - mp);
  if (tok < 0) return;
  for (int k = 0; k < 16; k++) {
    const int fl = rq + 4 * k;
    x[(size_t)(f0 + fl) * ld + tok] = tile[c][fl];
  }
}

void launch_desc_to_fm(const LgSet* sets, int nsets, int max_span, float* x, int ld, hipStream_t s) {
  hipLaunchKernelGGL(desc_to_fm_kernel, dim3((max_span + 63) / 64, 4, nsets), dim3(256), 0, s, sets, x, ld);
}

// ---------------------------------------------------------------------------------------------
// Flash-style attention in f32:  O = softmax(scale * Q K^T) V  for one head, one (query set, key set) pair per
// blockIdx.z; self-attention uses the set's own tokens as keys, cross-attention the partner set's.
// Workgroup = 128 queries (wave = 32 queries), key tiles of 32 staged through LDS (row stride 33: the K^T read is
// key-contiguous, the V^T read walks d with stride 33 -> both conflict-free).
// ---------------------------------------------------------------------------------------------
constexpr int AT_STRIDE = 33;

// Workgroup roles of an attention launch with tail queries: the grid has `tail_slots` extra blockIdx.x values per
// (head, set), one per tail query; the linear workgroup id (dispatch order) is re-read so that ALL main tiles come
// first -- their count fills the chip's slots in whole rounds -- and the short tail-query workgroups run last, on the
// CUs that finish first.  Returns the tail query index, or -1 for a main tile (bx = its query tile).
//
// XCD locality: workgroup ids are dealt round-robin to the 8 XCDs (id % 8), each with its own 4 MiB L2.  In plain grid
// order the gx query tiles of one (head, set) -- which all read that pair's K and V, 512 KiB at 1024 keys -- land on 8
// different XCDs, so every L2 fetches every K/V: 305 MB of HBM/fabric traffic per launch against 67 MB algorithmic at
// 16 x 1025 keypoints (profiles/r01_pmc_summary.json).  The remap below gives all query tiles of a (head, set) group to
// ONE XCD, in consecutive dispatch slots (so they are resident together and share the tiles in that L2): with
// L = 8 * slot + xcd, group = 8 * (slot / gx) + xcd and tile = slot % gx.  Groups beyond the last multiple of 8 keep the
// plain order.  Speed only: any placement computes the same values.
__device__ __forceinline__ int attn_role(int tail_slots_signed, int& bx, int& by, int& bz) {
  bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
  // Sign of the argument: > 0 = the tail-query workgroups take the LAST ids of the launch (large batches: the main tiles
  // fill every resident slot, and short workgroups dispatched first delay as many MFMA workgroups: 150 -> 156 us per launch
  // at 16 x 1025 keypoints); < 0 = the FIRST ids (attn_k4_kernel: about one main workgroup per CU, so the tail queries'
  // ~15 us of VALU work run beside the main tiles from the start instead of after them).
  const bool tail_first = tail_slots_signed < 0;
  const int tail_slots = tail_first ? -tail_slots_signed : tail_slots_signed;
  const int gx = gridDim.x - tail_slots, gy = gridDim.y;
  const int ngroups = gy * gridDim.z, nmain = gx * ngroups, ntail = tail_slots * ngroups;
  int L = blockIdx.x + gridDim.x * (blockIdx.y + gy * blockIdx.z);
  int t = -1;
  if (tail_first) { if (L < ntail) t = L; else L -= ntail; }
  else if (L >= nmain) t = L - nmain;
  if (t < 0) {
    // (tail_first with ntail % 8 != 0: id L runs on XCD (L + ntail) % 8 -- still one XCD per value of L % 8)
    const int full = (ngroups & ~7) * gx;      // ids covered by whole octets of groups
    int g, tl;
    if (L < full) { const int slot = L >> 3; g = ((slot / gx) << 3) + (L & 7); tl = slot % gx; }
    else { const int r = L - full; g = (ngroups & ~7) + r / gx; tl = r % gx; }
    bx = tl; by = g % gy; bz = g / gy;
    return -1;
  }
  bx = 0; by = (t / tail_slots) % gy; bz = t / (tail_slots * gy);
  return t % tail_slots;
}

// key side of a (query set, key set) pair
struct AttnKeys { int off, n, nmain, tail_off; };
__device__ __forceinline__ AttnKeys attn_keys(const LgSet& st, int cross) {
  AttnKeys k;
  k.off = cross ? st.partner_off : st.tok_off; k.n = cross ? st.partner_n : st.n;
  k.nmain = cross ? st.partner_nmain : st.nmain; k.tail_off = cross ? st.partner_tail_off : st.tail_off;
  return k;
}

// S^T[key][query] += K[key][d] Q[query][d] over the 64 d of one staged tile, and O^T[d][query] += V[key][d] P[query][key]:
// the LDS operand reads of group g + 1 are issued before the MFMAs of group g (left alone, hipcc sinks every read next
// to its MFMA and waits on it; measured 151 -> 145.5 us per launch at 16 x 1024 keypoints, tools/probes/attn_probe.hip).
__device__ __forceinline__ void attn_qk_tile(const float* sK, const float (&qreg)[32], f32x16& s, int j, int h) {
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
}
__device__ __forceinline__ void attn_pv_tile(const float* sV, const f32x16& s, f32x16 (&o)[2], int j, int h) {
  float vv[16][2];   // P^T accumulator register r is the B operand of step r
  vv[0][0] = sV[j * AT_STRIDE + acc_row(0, h)]; vv[0][1] = sV[(32 + j) * AT_STRIDE + acc_row(0, h)];
#pragma unroll
  for (int r = 0; r < 16; r++) {
    if (r + 1 < 16) { vv[r + 1][0] = sV[j * AT_STRIDE + acc_row(r + 1, h)]; vv[r + 1][1] = sV[(32 + j) * AT_STRIDE + acc_row(r + 1, h)]; }
    o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[r][0], s[r], o[0], 0, 0, 0);
    o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[r][1], s[r], o[1], 0, 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
  }
}

// The key set's tail keys (at most LG_TAIL_MAX), folded into a 32-query wave's online-softmax state on the VALU:
// one key costs ~130 VALU/LDS ops against the 64 MFMAs of a whole key tile.  Their K / V columns are staged once per
// workgroup in LDS (`tk`: [key][K 64 | V 64], loaded with the query prologue); every read below is a broadcast.
// qreg / o / m / l as in the MFMA loop (base-2 logits; lane (j, h) holds query j, the d = 2t + h half of q and the
// acc_row(r, h) rows of O^T).
__device__ __forceinline__ void attn_stage_tail_keys(const float* Kh, const float* Vh, int ld, int ktok0, int nt, float* tk) {
  for (int e = threadIdx.x; e < nt * 128; e += blockDim.x) {
    const int key = e >> 7, c = e & 127;
    tk[e] = (c < 64 ? Kh : Vh)[(size_t)(c & 63) * ld + ktok0 + key];
  }
}
__device__ __forceinline__ void attn_tail_keys(const float* tk, int nt, const float (&qreg)[32], f32x16 (&o)[2], float& m, float& l,
                                               int h) {
  for (int k = 0; k < nt; k++) {
    const float* kc = tk + k * 128;
    float sp = 0.0f;
#pragma unroll
    for (int t = 0; t < 32; t++) sp = fmaf(qreg[t], kc[2 * t + h], sp);
    const float sv = sp + __shfl_xor(sp, 32);
    const float m_new = fmaxf(m, sv);
    const float alpha = __builtin_amdgcn_exp2f(m - m_new), pr = __builtin_amdgcn_exp2f(sv - m_new);
    l = l * alpha + (h == 0 ? pr : 0.0f);          // l is a per-half partial sum: count the key once
#pragma unroll
    for (int r = 0; r < 16; r++) {
      o[0][r] = fmaf(pr, kc[64 + acc_row(r, h)], o[0][r] * alpha);
      o[1][r] = fmaf(pr, kc[96 + acc_row(r, h)], o[1][r] * alpha);
    }
    m = m_new;
  }
}

// One tail query of a set (at most LG_TAIL_MAX per set), all VALU: keys on the lanes, the workgroup's waves take the
// 64-key chunks round-robin (K and V columns loaded in batches of 64 independent loads) and their partial (m, l, O)
// states are merged through LDS (`xch`, >= nw * 66 floats).  These workgroups run LAST in the attention launch
// (attn_role), one per (tail query, head, set), ~15 us each on a chip that is emptying.
__device__ __forceinline__ void attn_tail_query(const float* Q, const float* K, const float* V, float* O, int ld,
                                                const LgSet& st, int qi, int head, int cross, float scale, float* xch) {
  if (qi >= st.n - st.nmain) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const AttnKeys ks = attn_keys(st, cross);
  const float* Kh = K + (size_t)head * 64 * ld;
  const float* Vh = V + (size_t)head * 64 * ld;
  {
    const int qt = st.tail_off + qi;
    const int qbits = __float_as_int(Q[(size_t)(head * 64 + lane) * ld + qt] * (scale * 1.44269504088896341f));   // lane = d
    float o[64];
#pragma unroll
    for (int d = 0; d < 64; d++) o[d] = 0.0f;
    float m = -1e30f, l = 0.0f;
    for (int k0 = wave * 64; k0 < ks.n; k0 += nw * 64) {
      const int key = k0 + lane;
      const bool valid = key < ks.n;
      const int kk = min(key, ks.n - 1);
      const int tok = kk < ks.nmain ? ks.off + kk : ks.tail_off + (kk - ks.nmain);
      float col[64];
#pragma unroll
      for (int d = 0; d < 64; d++) col[d] = Kh[(size_t)d * ld + tok];
      float sv = 0.0f;
#pragma unroll
      for (int d = 0; d < 64; d++) sv = fmaf(__int_as_float(__builtin_amdgcn_readlane(qbits, d)), col[d], sv);
#pragma unroll
      for (int d = 0; d < 64; d++) col[d] = Vh[(size_t)d * ld + tok];
      if (!valid) sv = -1e30f;
      const float m_new = fmaxf(m, sv);
      const float alpha = __builtin_amdgcn_exp2f(m - m_new), pr = valid ? __builtin_amdgcn_exp2f(sv - m_new) : 0.0f;
      l = l * alpha + pr;
#pragma unroll
      for (int d = 0; d < 64; d++) o[d] = fmaf(pr, col[d], o[d] * alpha);
      m = m_new;
    }
    // wave-level reduction: lane d ends up with O[d] of this wave's keys.  Reduce-scatter butterfly: at width w a lane keeps
    // the half of its vector that its bit w selects and receives the partner's copy of that half -- 63 exchanges instead of
    // the 64 x 6 of one full wave sum per d (these workgroups share their SIMDs with MFMA-bound main waves: every vector
    // instruction here is an issue slot taken from them).
    const float M = wave_max(m);
    const float w = __builtin_amdgcn_exp2f(m - M);
    const float Lw = wave_sumf(l * w);
#pragma unroll
    for (int d = 0; d < 64; d++) o[d] *= w;
#pragma unroll
    for (int wd = 32; wd >= 1; wd >>= 1) {
      const bool up = (lane & wd) != 0;
#pragma unroll
      for (int i = 0; i < wd; i++) {
        const float keep = up ? o[i + wd] : o[i], send = up ? o[i] : o[i + wd];
        o[i] = keep + __shfl_xor(send, wd);
      }
    }
    const float out = o[0];
    xch[wave * 66 + lane] = out;
    if (lane == 0) { xch[wave * 66 + 64] = M; xch[wave * 66 + 65] = Lw; }
    __syncthreads();
    if (wave == 0) {
      float MM = -1e30f;
      for (int k = 0; k < nw; k++) MM = fmaxf(MM, xch[k * 66 + 64]);
      float acc = 0.0f, LL = 0.0f;
      for (int k = 0; k < nw; k++) {
        const float f = __builtin_amdgcn_exp2f(xch[k * 66 + 64] - MM);
        acc += xch[k * 66 + lane] * f; LL += xch[k * 66 + 65] * f;
      }
      O[(size_t)(head * 64 + lane) * ld + qt] = LL > 0.0f ? acc / LL : 0.0f;
    }
  }
}

template <int NW>   // waves (= 32-query blocks) per workgroup
__global__ __launch_bounds__(64 * NW, 2) void attn_kernel(const float* Q, const float* K, const float* V, float* O, int ld,
                                                          const LgSet* sets, int tail_slots, int cross, float scale) {
  __shared__ float sK[64 * AT_STRIDE];
  __shared__ float sV[64 * AT_STRIDE];
  __shared__ float sTK[LG_TAIL_MAX * 128];
  int bx, head, bz;
  const int tail_q = attn_role(tail_slots, bx, head, bz);
  const LgSet st = sets[bz];
  if (tail_q >= 0) { attn_tail_query(Q, K, V, O, ld, st, tail_q, head, cross, scale, sK); return; }
  const int mp = (st.nmain + 31) & ~31;
  const int q0 = bx * (32 * NW);
  if (q0 >= mp) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
  const AttnKeys ks = attn_keys(st, cross);
  const int k_off = ks.off, k_n = ks.nmain;                // the MFMA loop covers the key set's main region
  const int qi = q0 + wave * 32 + j;                       // may exceed nmain (padding rows: computed, not stored)
  const int qtok = st.tok_off + min(qi, mp - 1);
  const float* Qh = Q + (size_t)head * 64 * ld;
  const float* Kh = K + (size_t)head * 64 * ld + k_off;
  const float* Vh = V + (size_t)head * 64 * ld + k_off;
  if (ks.n > ks.nmain)   // visible to all waves after the first barrier of the tile loop
    attn_stage_tail_keys(K + (size_t)head * 64 * ld, V + (size_t)head * 64 * ld, ld, ks.tail_off, ks.n - ks.nmain, sTK);

  float qreg[32];
#pragma unroll
  // softmax runs in base 2: log2(e) is folded into the query scale so that exp() is a bare v_exp_f32
  for (int t = 0; t < 32; t++) qreg[t] = Qh[(size_t)(2 * t + h) * ld + qtok] * (scale * 1.44269504088896341f);

  f32x16 o[2];
#pragma unroll
  for (int r = 0; r < 16; r++) { o[0][r] = 0.0f; o[1][r] = 0.0f; }
  float m = -1e30f, l = 0.0f;

  const int ntiles = (k_n + 31) / 32;
  // staging: tile = 64 d x 32 keys; 512 float4 per matrix, NS per thread
  constexpr int NS = 512 / (64 * NW);
  f32x4 pk[NS], pv[NS];
  int soff[NS], loff[NS];
#pragma unroll
  for (int u = 0; u < NS; u++) {
    const int e = tid + 64 * NW * u, d = e >> 3, q4 = e & 7;
    soff[u] = d * ld + 4 * q4;
    loff[u] = d * AT_STRIDE + 4 * q4;
  }
#define AT_PREFETCH(kt_)                                                               \
  {                                                                                    \
    _Pragma("unroll") for (int u = 0; u < NS; u++) {                                   \
      pk[u] = *reinterpret_cast<const f32x4*>(Kh + (size_t)soff[u] + (kt_) * 32);      \
      pv[u] = *reinterpret_cast<const f32x4*>(Vh + (size_t)soff[u] + (kt_) * 32);      \
    }                                                                                  \
  }
  AT_PREFETCH(0);
  for (int kt = 0; kt < ntiles; kt++) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NS; u++) {
#pragma unroll
      for (int c = 0; c < 4; c++) { sK[loff[u] + c] = pk[u][c]; sV[loff[u] + c] = pv[u][c]; }
    }
    __syncthreads();
    if (kt + 1 < ntiles) AT_PREFETCH(kt + 1);

    // S^T[key][query] = sum_d K[key][d] Q[query][d]
    // (hoisting ALL 32 operand reads + two accumulator chains cost 199 VGPRs and was slower; reading ONE group ahead,
    //  attn_qk_tile, is the version that pays)
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; r++) s[r] = 0.0f;
    attn_qk_tile(sK, qreg, s, j, h);

    const int kbase = kt * 32;
    float mx = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      if (kbase + acc_row(r, h) >= k_n) s[r] = -1e30f;
      mx = fmaxf(mx, s[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m, mx);
    const float alpha = __builtin_amdgcn_exp2f(m - m_new);
    float ls = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; r++) { s[r] = __builtin_amdgcn_exp2f(s[r] - m_new); ls += s[r]; }
    l = l * alpha + ls;
    m = m_new;
#pragma unroll
    for (int r = 0; r < 16; r++) { o[0][r] *= alpha; o[1][r] *= alpha; }
    // O^T[d][query] += sum_key V[key][d] P[query][key]; P^T accumulator register r is the B operand of step r
    attn_pv_tile(sV, s, o, j, h);
  }
#undef AT_PREFETCH
  if (ks.n > ks.nmain) {
    if (ntiles == 0) __syncthreads();   // no tile loop ran: make the staged tail keys visible
    attn_tail_keys(sTK, ks.n - ks.nmain, qreg, o, m, l, h);
  }
  const float lt = l + __shfl_xor(l, 32);
  const float inv = lt > 0.0f ? 1.0f / lt : 0.0f;            // empty key set -> zero message
  // padding queries (nmain <= qi < padded nmain) get zeros so that every column of the set stays finite: masked keys
  // are multiplied by p = 0 in later layers, and 0 * NaN would poison valid rows
  if (qi < mp) {
    const float keep = qi < st.nmain ? inv : 0.0f;
    float* Oh = O + (size_t)head * 64 * ld + st.tok_off + qi;
#pragma unroll
    for (int db = 0; db < 2; db++)
#pragma unroll
      for (int r = 0; r < 16; r++) Oh[(size_t)(db * 32 + acc_row(r, h)) * ld] = qi < st.nmain ? o[db][r] * keep : 0.0f;
  }
}

// ---------------------------------------------------------------------------------------------
// Key-split variant: workgroup = 64 queries = 2 query blocks x 2 key halves (wave w: query block w&1, key half w>>1).
// Each wave runs the online softmax over its half of the keys; the two partial (m, l, O) states of a query block are
// merged through LDS at the end.  Twice as many, half as long waves: with ~2100 query blocks per launch on 1024
// SIMDs the un-split kernel needs 3 rounds for 2.06 waves per SIMD (69 % balance), this one 5 half-rounds for 4.25
// (82 %).  The two query blocks share every staged K/V tile.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_ks_kernel(const float* Q, const float* K, const float* V, float* O, int ld,
                                                         const LgSet* sets, int tail_slots, int cross, float scale) {
  __shared__ float sKV[4 * 64 * AT_STRIDE];   // [key half][K | V][64 d][33]
  __shared__ float sTK[LG_TAIL_MAX * 128];
  int bx, head, bz;
  const int tail_q = attn_role(tail_slots, bx, head, bz);
  const LgSet st = sets[bz];
  if (tail_q >= 0) { attn_tail_query(Q, K, V, O, ld, st, tail_q, head, cross, scale, sKV); return; }
  const int np = (st.nmain + 31) & ~31;                    // queries of the main region
  const int q0 = bx * 64;
  if (q0 >= np) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
  const int qb = wave & 1, kh = wave >> 1;
  const AttnKeys ks = attn_keys(st, cross);
  const int k_off = ks.off, k_n = ks.nmain;                // the MFMA loop covers the key set's main region
  const int qi = q0 + qb * 32 + j;
  const bool wave_active = (q0 + qb * 32) < np;              // wave-uniform: the last tile may hold one block only
  const int qtok = st.tok_off + min(qi, np - 1);
  const float* Qh = Q + (size_t)head * 64 * ld;
  const float* Kh = K + (size_t)head * 64 * ld + k_off;
  const float* Vh = V + (size_t)head * 64 * ld + k_off;
  float* sK = sKV + (size_t)kh * 2 * 64 * AT_STRIDE;
  float* sV = sK + 64 * AT_STRIDE;
  if (ks.n > ks.nmain)   // visible to all waves after the first barrier of the tile loop
    attn_stage_tail_keys(K + (size_t)head * 64 * ld, V + (size_t)head * 64 * ld, ld, ks.tail_off, ks.n - ks.nmain, sTK);

  float qreg[32];
#pragma unroll
  // softmax runs in base 2: log2(e) is folded into the query scale so that exp() is a bare v_exp_f32
  for (int t = 0; t < 32; t++) qreg[t] = Qh[(size_t)(2 * t + h) * ld + qtok] * (scale * 1.44269504088896341f);
  f32x16 o[2];
#pragma unroll
  for (int r = 0; r < 16; r++) { o[0][r] = 0.0f; o[1][r] = 0.0f; }
  float m = -1e30f, l = 0.0f;

  const int ntiles = (k_n + 31) / 32, half = (ntiles + 1) / 2;
  // staging: per iteration 2 key tiles x (K, V) x 512 float4 = 2048 float4, 8 per thread
  f32x4 pf[8];
  const float* gsrc[8];      // row base of each staged float4 (K or V, row d, column 4*q4)
  int loff[8], hsel[8];
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const int e = tid + 256 * u;             // [key half (2)][K/V (2)][d (64)][q4 (8)]
    const int q4 = e & 7, d = (e >> 3) & 63, kv = (e >> 9) & 1, hh = e >> 10;
    gsrc[u] = (kv ? Vh : Kh) + (size_t)d * ld + 4 * q4;
    loff[u] = (hh * 2 + kv) * 64 * AT_STRIDE + d * AT_STRIDE + 4 * q4;
    hsel[u] = hh * half;
  }
  // a key half may hold one tile fewer than the other: its extra load is clamped to the last tile and never used
  // (and to tile 0 for an empty key set, whose padded region is still 32 tokens wide)
#define ATK_PREFETCH(it_)                                                                          \
  {                                                                                                \
    _Pragma("unroll") for (int u = 0; u < 8; u++)                                                  \
      pf[u] = *reinterpret_cast<const f32x4*>(gsrc[u] + max(min(hsel[u] + (it_), ntiles - 1), 0) * 32); \
  }
  ATK_PREFETCH(0);
  for (int it = 0; it < half; it++) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; u++) {
#pragma unroll
      for (int c = 0; c < 4; c++) sKV[loff[u] + c] = pf[u][c];
    }
    __syncthreads();
    if (it + 1 < half) ATK_PREFETCH(it + 1);
    const int kt = kh * half + it;                        // this wave's key tile
    if (wave_active && kt < ntiles) {
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; r++) s[r] = 0.0f;
      attn_qk_tile(sK, qreg, s, j, h);
      const int kbase = kt * 32;
      if (kbase + 32 > k_n) {                               // only the last key tile can hold padding keys
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
      if (__ballot(m_new > m)) {                            // rescale only when some query's running max moved
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);
        l *= alpha;
#pragma unroll
        for (int r = 0; r < 16; r++) { o[0][r] *= alpha; o[1][r] *= alpha; }
      }
      l += ls;
      m = m_new;
      attn_pv_tile(sV, s, o, j, h);
    }
  }
#undef ATK_PREFETCH
  if (ks.n > ks.nmain) {                                  // the (shorter) second key half also takes the tail keys
    if (half == 0) __syncthreads();                       // no tile loop ran: make the staged tail keys visible
    if (kh == 1 && wave_active) attn_tail_keys(sTK, ks.n - ks.nmain, qreg, o, m, l, h);
  }
  // ---- merge the two key halves of each query block through LDS (reusing the tile buffers)
  __syncthreads();
  float* xch = sKV + (size_t)qb * (64 * 34);           // per query block: 64 lanes x 34 floats
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
    const float inv = (lt > 0.0f && qi < st.nmain) ? 1.0f / lt : 0.0f;   // padding queries and empty key sets -> 0
    float* Oh = O + (size_t)head * 64 * ld + st.tok_off + qi;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      Oh[(size_t)acc_row(r, h) * ld] = (o[0][r] * a0 + p[2 + r] * a1) * inv;
      Oh[(size_t)(32 + acc_row(r, h)) * ld] = (o[1][r] * a0 + p[18 + r] * a1) * inv;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Four-way key split for small problems (one or two pairs: fewer key-split workgroups than CUs): workgroup = ONE block of
// 32 queries x 4 key quarters, wave w runs the online softmax over quarter w.  Half the serial length of attn_ks_kernel's
// waves, twice the workgroups (264 at 2 x 1025 keypoints instead of 136 on 256 CUs).  No tile is shared between waves, so
// each wave stages its own K / V tiles in its own LDS region and the tile loop has NO workgroup barrier; the four partial
// (m, l, O) states meet once, at the end.  Dynamic LDS (4 x 2 tiles x 8448 B = 66 KiB, over the static limit).
// ---------------------------------------------------------------------------------------------
template <int NKQ> constexpr int atk4_lds_floats() { return NKQ * 2 * 64 * AT_STRIDE + LG_TAIL_MAX * 128; }
template <int NKQ>   // key parts = waves per workgroup: 4, or 8 (two waves per SIMD, half the serial length again)
__global__ __launch_bounds__(64 * NKQ) void attn_k4_kernel(const float* Q, const float* K, const float* V, float* O, int ld,
                                                         const LgSet* sets, int tail_slots, int cross, float scale) {
  extern __shared__ __attribute__((aligned(16))) float k4_lds[];
  float* sKV = k4_lds;                                      // [key quarter][K | V][64 d][33]
  float* sTK = k4_lds + NKQ * 2 * 64 * AT_STRIDE;
  int bx, head, bz;
  const int tail_q = attn_role(tail_slots, bx, head, bz);
  const LgSet st = sets[bz];
  if (tail_q >= 0) { attn_tail_query(Q, K, V, O, ld, st, tail_q, head, cross, scale, sKV); return; }
  const int np = (st.nmain + 31) & ~31;
  const int q0 = bx * 32;
  if (q0 >= np) return;
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
  const int kq = __builtin_amdgcn_readfirstlane(tid >> 6);
  const AttnKeys ks = attn_keys(st, cross);
  const int k_off = ks.off, k_n = ks.nmain;
  const int qi = q0 + j;
  const int qtok = st.tok_off + qi;                          // q0 + 31 < np: every query of the block lies in the padded main region
  const float* Qh = Q + (size_t)head * 64 * ld;
  const float* Kh = K + (size_t)head * 64 * ld + k_off;
  const float* Vh = V + (size_t)head * 64 * ld + k_off;
  float* sK = sKV + (size_t)kq * 2 * 64 * AT_STRIDE;
  float* sV = sK + 64 * AT_STRIDE;
  const int ntail = ks.n - ks.nmain;
  if (ntail > 0) {
    attn_stage_tail_keys(K + (size_t)head * 64 * ld, V + (size_t)head * 64 * ld, ld, ks.tail_off, ntail, sTK);
    __syncthreads();
  }
  float qreg[32];
#pragma unroll
  for (int t = 0; t < 32; t++) qreg[t] = Qh[(size_t)(2 * t + h) * ld + qtok] * (scale * 1.44269504088896341f);
  f32x16 o[2];
#pragma unroll
  for (int r = 0; r < 16; r++) { o[0][r] = 0.0f; o[1][r] = 0.0f; }
  float m = -1e30f, l = 0.0f;

  const int ntiles = (k_n + 31) / 32, quarter = (ntiles + NKQ - 1) / NKQ;
  const int kt0 = kq * quarter, kt1 = min(kt0 + quarter, ntiles);
  // this wave's staging of one key tile: K and V, 64 d x 32 keys each = 512 float4 per matrix, 8 + 8 per lane
  f32x4 pk[8], pv[8];
  int soff[8], loff[8];
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const int e = lane + 64 * u, d = e >> 3, q4 = e & 7;
    soff[u] = d * ld + 4 * q4;
    loff[u] = d * AT_STRIDE + 4 * q4;
  }
#define ATK4_WAVE_SYNC()                                                               \
  {                                                                                    \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                             \
    __builtin_amdgcn_wave_barrier();                                                   \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                             \
  }
#define ATK4_PREFETCH(kt_)                                                             \
  {                                                                                    \
    _Pragma("unroll") for (int u = 0; u < 8; u++) {                                    \
      pk[u] = *reinterpret_cast<const f32x4*>(Kh + (size_t)soff[u] + (kt_) * 32);      \
      pv[u] = *reinterpret_cast<const f32x4*>(Vh + (size_t)soff[u] + (kt_) * 32);      \
    }                                                                                  \
  }
  if (kt0 < kt1) ATK4_PREFETCH(kt0);
  for (int kt = kt0; kt < kt1; kt++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
#pragma unroll
      for (int c = 0; c < 4; c++) { sK[loff[u] + c] = pk[u][c]; sV[loff[u] + c] = pv[u][c]; }
    }
    ATK4_WAVE_SYNC();   // the tile is written by this wave's lanes and read across them: order the ds ops (no instruction emitted)
    if (kt + 1 < kt1) ATK4_PREFETCH(kt + 1);
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; r++) s[r] = 0.0f;
    attn_qk_tile(sK, qreg, s, j, h);
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
    attn_pv_tile(sV, s, o, j, h);
    ATK4_WAVE_SYNC();   // reads of this tile before the next tile's (or the merge state's) stores
  }
#undef ATK4_PREFETCH
#undef ATK4_WAVE_SYNC
  if (ntail > 0 && kq == NKQ - 1) attn_tail_keys(sTK, ntail, qreg, o, m, l, h);   // the last quarter is the shortest
  // ---- merge: quarters 1..3 leave their state in their OWN tile region (64 lanes x 34 floats), quarter 0 folds them in
  if (kq != 0) {
    float* p = sK + lane * 34;
    p[0] = m; p[1] = l;
#pragma unroll
    for (int r = 0; r < 16; r++) { p[2 + r] = o[0][r]; p[18 + r] = o[1][r]; }
  }
  __syncthreads();
  if (kq == 0) {
#pragma unroll
    for (int w = 1; w < NKQ; w++) {
      const float* p = sKV + (size_t)w * 2 * 64 * AT_STRIDE + lane * 34;
      const float m1 = p[0], l1 = p[1];
      const float mm = fmaxf(m, m1);
      const float a0 = __builtin_amdgcn_exp2f(m - mm), a1 = __builtin_amdgcn_exp2f(m1 - mm);
      l = l * a0 + l1 * a1;
#pragma unroll
      for (int r = 0; r < 16; r++) { o[0][r] = o[0][r] * a0 + p[2 + r] * a1; o[1][r] = o[1][r] * a0 + p[18 + r] * a1; }
      m = mm;
    }
    const float lt = l + __shfl_xor(l, 32);
    const float inv = (lt > 0.0f && qi < st.nmain) ? 1.0f / lt : 0.0f;   // padding queries and empty key sets -> 0
    float* Oh = O + (size_t)head * 64 * ld + st.tok_off + qi;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      Oh[(size_t)acc_row(r, h) * ld] = o[0][r] * inv;
      Oh[(size_t)(32 + acc_row(r, h)) * ld] = o[1][r] * inv;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// attn_x3_kernel: attn_ks_kernel with both contractions on the bf16 matrix cores at f32 accuracy (x3_common.h; the
// DANI_LG_GEMM_BF16X3_ALL option).  Same workgroup shape (64 queries = 2 query blocks x 2 key halves), same online
// softmax in f32, same tail-key / tail-query / merge code.  What changes:
//   * Q is split once into B fragments in registers (lane = query, 8 consecutive d per k-step and lane half);
//   * a K tile is split while staging into three [key][d] bf16 planes (128-B rows, 16-B slots XOR-swizzled by key
//     pair) -- the A operand of S^T = K Q^T;
//   * a V tile is split into three [d][key] bf16 planes whose 32 keys are stored in the order the S^T accumulator
//     holds them: register r = 8t + e of lane half h is key 16t + 4h + e (e < 4) or 16t + 8 + 4h + (e - 4), so fragment
//     (t, h) of a V row is [keys 16t+4h..+3 | keys 16t+8+4h..+3] = two of the staging thread's float4 -- and the
//     probabilities, split in place, ARE the B operand of O^T = V^T P^T, register for register (no shuffle, no LDS);
//   * per key tile and wave 24 + 24 bf16 MFMAs (32 cycles each) replace 32 + 32 f32 MFMAs (64 cycles each).
// ---------------------------------------------------------------------------------------------
constexpr int AX_KPLANE = 32 * 128, AX_VPLANE = 64 * 64, AX_HALF = 3 * AX_KPLANE + 3 * AX_VPLANE;   // bytes

__global__ __launch_bounds__(256, 2) void attn_x3_kernel(const float* Q, const float* K, const float* V, float* O, int ld,
                                                         const LgSet* sets, int tail_slots, int cross, float scale) {
  __shared__ __attribute__((aligned(16))) unsigned char sKV[2 * AX_HALF];   // [key half][K hi|mid|lo][V hi|mid|lo]
  __shared__ float sTK[LG_TAIL_MAX * 128];
  int bx, head, bz;
  const int tail_q = attn_role(tail_slots, bx, head, bz);
  const LgSet st = sets[bz];
  if (tail_q >= 0) { attn_tail_query(Q, K, V, O, ld, st, tail_q, head, cross, scale, reinterpret_cast<float*>(sKV)); return; }
  const int np = (st.nmain + 31) & ~31;
  const int q0 = bx * 64;
  if (q0 >= np) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
  const int qb = wave & 1, kh = wave >> 1;
  const AttnKeys ks = attn_keys(st, cross);
  const int k_off = ks.off, k_n = ks.nmain;
  const int qi = q0 + qb * 32 + j;
  const bool wave_active = (q0 + qb * 32) < np;
  const int qtok = st.tok_off + min(qi, np - 1);
  const float* Qh = Q + (size_t)head * 64 * ld;
  const float* Kh = K + (size_t)head * 64 * ld + k_off;
  const float* Vh = V + (size_t)head * 64 * ld + k_off;
  const unsigned char* sK = sKV + kh * AX_HALF;
  const unsigned char* sV = sK + 3 * AX_KPLANE;
  if (ks.n > ks.nmain)
    attn_stage_tail_keys(K + (size_t)head * 64 * ld, V + (size_t)head * 64 * ld, ld, ks.tail_off, ks.n - ks.nmain, sTK);

  // Q fragments: k-step s (16 d), plane p; lane (query j, half h) holds d = 16 s + 8 h + 0..7
  const float qscale = scale * 1.44269504088896341f;   // base-2 softmax
  u32x4 qf[4][3];
#pragma unroll
  for (int s4 = 0; s4 < 4; s4++)
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      const float a0 = Qh[(size_t)(16 * s4 + 8 * h + e) * ld + qtok] * qscale;
      const float a1 = Qh[(size_t)(16 * s4 + 8 * h + e + 1) * ld + qtok] * qscale;
      unsigned hi, mid, lo;
      x3_split2(a0, a1, hi, mid, lo);
      qf[s4][0][e >> 1] = hi; qf[s4][1][e >> 1] = mid; qf[s4][2][e >> 1] = lo;
    }
  f32x16 o[2];
#pragma unroll
  for (int r = 0; r < 16; r++) { o[0][r] = 0.0f; o[1][r] = 0.0f; }
  float m = -1e30f, l = 0.0f;

  const int ntiles = (k_n + 31) / 32, half = (ntiles + 1) / 2;
  // staging roles (per iteration: one key tile per key half).  Threads 0..127 serve key half 0, 128..255 half 1.
  //   K: 4 consecutive d rows x 4 consecutive keys per thread (transposed into [key][d]: one 8-B store per key and plane);
  //      a 16-lane store group = 4 key groups x 4 d groups -> 16 different 8-B columns of the 128-B bank row
  //   V: 4 float4 (one d row, 4 consecutive keys each) per thread, stored as they are (one 8-B store per plane)
  const int hh = tid >> 7, idx = tid & 127;
  const int kq4 = ((idx >> 4) & 1) * 4 + (idx & 3), kdg = (idx >> 5) * 4 + ((idx >> 2) & 3);
  f32x4 pk[4], pv[4];
  const float* ksrc = Kh + (size_t)(4 * kdg) * ld + 4 * kq4;
  const float* vsrc[4];
  unsigned voff[4];
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int e2 = idx + 128 * u, q4 = e2 & 7, d = e2 >> 3;
    vsrc[u] = Vh + (size_t)d * ld + 4 * q4;
    const int slot = (q4 >> 2) * 2 + (q4 & 1), part = (q4 >> 1) & 1;
    voff[u] = (unsigned)(hh * AX_HALF + 3 * AX_KPLANE + d * 64 + (((slot ^ ((d >> 2) & 3)) << 4) | (part << 3)));
  }
  unsigned char* kdst = sKV + hh * AX_HALF;
#define AX_PREFETCH(it_)                                                                       \
  {                                                                                            \
    const int tl_ = max(min(hh * half + (it_), ntiles - 1), 0) * 32;                           \
    _Pragma("unroll") for (int u = 0; u < 4; u++) {                                            \
      pk[u] = *reinterpret_cast<const f32x4*>(ksrc + (size_t)u * ld + tl_);                    \
      pv[u] = *reinterpret_cast<const f32x4*>(vsrc[u] + tl_);                                  \
    }                                                                                          \
  }
  AX_PREFETCH(0);
  for (int it = 0; it < half; it++) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; c++) {          // K: key 4 kq4 + c, d = 4 kdg .. +3
      unsigned h0, m0, l0, h1, m1, l1;
      x3_split2(pk[0][c], pk[1][c], h0, m0, l0);
      x3_split2(pk[2][c], pk[3][c], h1, m1, l1);
      const int key = 4 * kq4 + c;
      unsigned char* dst = kdst + key * 128 + ((((kdg >> 1) ^ ((key >> 1) & 7)) << 4) | ((kdg & 1) << 3));
      *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(dst + AX_KPLANE) = u32x2{m0, m1};
      *reinterpret_cast<u32x2*>(dst + 2 * AX_KPLANE) = u32x2{l0, l1};
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {          // V: one d row, 4 consecutive keys
      unsigned h0, m0, l0, h1, m1, l1;
      x3_split2(pv[u][0], pv[u][1], h0, m0, l0);
      x3_split2(pv[u][2], pv[u][3], h1, m1, l1);
      unsigned char* dst = sKV + voff[u];
      *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(dst + AX_VPLANE) = u32x2{m0, m1};
      *reinterpret_cast<u32x2*>(dst + 2 * AX_VPLANE) = u32x2{l0, l1};
    }
    __syncthreads();
    AX_PREFETCH(min(it + 1, max(half - 1, 0)));
    __builtin_amdgcn_sched_barrier(0);
    const int kt = kh * half + it;
    if (wave_active && kt < ntiles) {
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; r++) s[r] = 0.0f;
      // S^T[key][query] = K[key][:] . Q[query][:]
#pragma unroll
      for (int s4 = 0; s4 < 4; s4++) {
        const unsigned char* ap = sK + j * 128 + (((2 * s4 + h) ^ ((j >> 1) & 7)) << 4);
        const bf16x8 a_h = *reinterpret_cast<const bf16x8*>(ap);
        const bf16x8 a_m = *reinterpret_cast<const bf16x8*>(ap + AX_KPLANE);
        const bf16x8 a_l = *reinterpret_cast<const bf16x8*>(ap + 2 * AX_KPLANE);
        x3_mfma6(s, a_h, a_m, a_l, __builtin_bit_cast(bf16x8, qf[s4][0]), __builtin_bit_cast(bf16x8, qf[s4][1]),
                 __builtin_bit_cast(bf16x8, qf[s4][2]));
      }
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
      // O^T[d][query] += V[key][d] P[query][key]: registers 8t .. 8t+7 of the S^T accumulator are k-step t of the B operand
#pragma unroll
      for (int t = 0; t < 2; t++) {
        u32x4 ph, pm, pl;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          unsigned hi, mid, lo;
          x3_split2(s[8 * t + e], s[8 * t + e + 1], hi, mid, lo);
          ph[e >> 1] = hi; pm[e >> 1] = mid; pl[e >> 1] = lo;
        }
#pragma unroll
        for (int db = 0; db < 2; db++) {
          const int d = db * 32 + j;
          const unsigned char* vp = sV + d * 64 + (((2 * t + h) ^ ((d >> 2) & 3)) << 4);
          const bf16x8 v_h = *reinterpret_cast<const bf16x8*>(vp);
          const bf16x8 v_m = *reinterpret_cast<const bf16x8*>(vp + AX_VPLANE);
          const bf16x8 v_l = *reinterpret_cast<const bf16x8*>(vp + 2 * AX_VPLANE);
          x3_mfma6(o[db], v_h, v_m, v_l, __builtin_bit_cast(bf16x8, ph), __builtin_bit_cast(bf16x8, pm), __builtin_bit_cast(bf16x8, pl));
        }
      }
    }
  }
#undef AX_PREFETCH
  if (ks.n > ks.nmain) {                                  // tail keys on the VALU, with the f32 query in attn_ks_kernel's layout
    if (half == 0) __syncthreads();
    if (kh == 1 && wave_active) {
      float qreg[32];
#pragma unroll
      for (int t = 0; t < 32; t++) qreg[t] = Qh[(size_t)(2 * t + h) * ld + qtok] * qscale;
      attn_tail_keys(sTK, ks.n - ks.nmain, qreg, o, m, l, h);
    }
  }
  // ---- merge the two key halves of each query block through LDS (reusing the tile buffers)
  __syncthreads();
  float* xch = reinterpret_cast<float*>(sKV) + (size_t)qb * (64 * 34);
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
    const float inv = (lt > 0.0f && qi < st.nmain) ? 1.0f / lt : 0.0f;
    float* Oh = O + (size_t)head * 64 * ld + st.tok_off + qi;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      Oh[(size_t)acc_row(r, h) * ld] = (o[0][r] * a0 + p[2 + r] * a1) * inv;
      Oh[(size_t)(32 + acc_row(r, h)) * ld] = (o[1][r] * a0 + p[18 + r] * a1) * inv;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// attn_x3w_kernel: the 128-query form of attn_x3_kernel (attn_kernel<4>'s structure: 4 waves = 4 query blocks, every
// wave consumes every staged key tile, no key split and no merge).  One K/V tile is split per iteration by all 256
// threads -- half the split work per MFMA of the key-split form -- and used by 4 waves.  Same fragments and LDS layouts.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_x3w_kernel(const float* Q, const float* K, const float* V, float* O, int ld,
                                                          const LgSet* sets, int tail_slots, int cross, float scale) {
  __shared__ __attribute__((aligned(16))) unsigned char sKV[AX_HALF];   // [K hi|mid|lo][V hi|mid|lo]
  __shared__ float sTK[LG_TAIL_MAX * 128];
  int bx, head, bz;
  const int tail_q = attn_role(tail_slots, bx, head, bz);
  const LgSet st = sets[bz];
  if (tail_q >= 0) { attn_tail_query(Q, K, V, O, ld, st, tail_q, head, cross, scale, reinterpret_cast<float*>(sKV)); return; }
  const int mp = (st.nmain + 31) & ~31;
  const int q0 = bx * 128;
  if (q0 >= mp) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
  const AttnKeys ks = attn_keys(st, cross);
  const int k_off = ks.off, k_n = ks.nmain;
  const int qi = q0 + wave * 32 + j;
  const int qtok = st.tok_off + min(qi, mp - 1);
  const float* Qh = Q + (size_t)head * 64 * ld;
  const float* Kh = K + (size_t)head * 64 * ld + k_off;
  const float* Vh = V + (size_t)head * 64 * ld + k_off;
  const unsigned char* sK = sKV;
  const unsigned char* sV = sKV + 3 * AX_KPLANE;
  if (ks.n > ks.nmain)
    attn_stage_tail_keys(K + (size_t)head * 64 * ld, V + (size_t)head * 64 * ld, ld, ks.tail_off, ks.n - ks.nmain, sTK);

  const float qscale = scale * 1.44269504088896341f;
  u32x4 qf[4][3];
#pragma unroll
  for (int s4 = 0; s4 < 4; s4++)
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      const float a0 = Qh[(size_t)(16 * s4 + 8 * h + e) * ld + qtok] * qscale;
      const float a1 = Qh[(size_t)(16 * s4 + 8 * h + e + 1) * ld + qtok] * qscale;
      unsigned hi, mid, lo;
      x3_split2(a0, a1, hi, mid, lo);
      qf[s4][0][e >> 1] = hi; qf[s4][1][e >> 1] = mid; qf[s4][2][e >> 1] = lo;
    }
  f32x16 o[2];
#pragma unroll
  for (int r = 0; r < 16; r++) { o[0][r] = 0.0f; o[1][r] = 0.0f; }
  float m = -1e30f, l = 0.0f;

  const int ntiles = (k_n + 31) / 32;
  // staging roles, 256 threads for one tile.  K: 2 consecutive d rows x 4 consecutive keys per thread -> one 4-B store
  // per key and plane; a 32-lane store group = 4 key groups x 8 d pairs -> 32 different dwords of the 128-B bank row.
  // V: 2 float4 (one d row, 4 consecutive keys each) per thread, one 8-B store per plane.
  const int kq4 = ((tid >> 5) & 1) * 4 + (tid & 3), kdg = (tid >> 6) * 8 + ((tid >> 2) & 7);   // d = 2 kdg, 2 kdg + 1
  f32x4 pk[2], pv[2];
  const float* ksrc = Kh + (size_t)(2 * kdg) * ld + 4 * kq4;
  const float* vsrc[2];
  unsigned voff[2];
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int e2 = tid + 256 * u, q4 = e2 & 7, d = e2 >> 3;
    vsrc[u] = Vh + (size_t)d * ld + 4 * q4;
    const int slot = (q4 >> 2) * 2 + (q4 & 1), part = (q4 >> 1) & 1;
    voff[u] = (unsigned)(3 * AX_KPLANE + d * 64 + (((slot ^ ((d >> 2) & 3)) << 4) | (part << 3)));
  }
#define AXW_PREFETCH(kt_)                                                                      \
  {                                                                                            \
    const int tl_ = max(min((kt_), ntiles - 1), 0) * 32;                                       \
    _Pragma("unroll") for (int u = 0; u < 2; u++) {                                            \
      pk[u] = *reinterpret_cast<const f32x4*>(ksrc + (size_t)u * ld + tl_);                    \
      pv[u] = *reinterpret_cast<const f32x4*>(vsrc[u] + tl_);                                  \
    }                                                                                          \
  }
  AXW_PREFETCH(0);
  for (int kt = 0; kt < ntiles; kt++) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; c++) {          // K: key 4 kq4 + c, d = 2 kdg, 2 kdg + 1
      unsigned h0, m0, l0;
      x3_split2(pk[0][c], pk[1][c], h0, m0, l0);
      const int key = 4 * kq4 + c;
      unsigned char* dst = sKV + key * 128 + ((((kdg >> 2) ^ ((key >> 1) & 7)) << 4) | ((kdg & 3) << 2));
      *reinterpret_cast<unsigned*>(dst) = h0;
      *reinterpret_cast<unsigned*>(dst + AX_KPLANE) = m0;
      *reinterpret_cast<unsigned*>(dst + 2 * AX_KPLANE) = l0;
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      unsigned h0, m0, l0, h1, m1, l1;
      x3_split2(pv[u][0], pv[u][1], h0, m0, l0);
      x3_split2(pv[u][2], pv[u][3], h1, m1, l1);
      unsigned char* dst = sKV + voff[u];
      *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(dst + AX_VPLANE) = u32x2{m0, m1};
      *reinterpret_cast<u32x2*>(dst + 2 * AX_VPLANE) = u32x2{l0, l1};
    }
    __syncthreads();
    AXW_PREFETCH(kt + 1);
    __builtin_amdgcn_sched_barrier(0);
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; r++) s[r] = 0.0f;
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++) {
      const unsigned char* ap = sK + j * 128 + (((2 * s4 + h) ^ ((j >> 1) & 7)) << 4);
      const bf16x8 a_h = *reinterpret_cast<const bf16x8*>(ap);
      const bf16x8 a_m = *reinterpret_cast<const bf16x8*>(ap + AX_KPLANE);
      const bf16x8 a_l = *reinterpret_cast<const bf16x8*>(ap + 2 * AX_KPLANE);
      x3_mfma6(s, a_h, a_m, a_l, __builtin_bit_cast(bf16x8, qf[s4][0]), __builtin_bit_cast(bf16x8, qf[s4][1]),
               __builtin_bit_cast(bf16x8, qf[s4][2]));
    }
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
    // P^T split for both k-steps first, so that the second one's VALU work is there to be interleaved with the first
    // step's MFMAs (one MFMA, then up to five VALU instructions: the MFMA holds the vector issue for 8 of its 32 cycles)
    u32x4 pp[2][3];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        unsigned hi, mid, lo;
        x3_split2(s[8 * t + e], s[8 * t + e + 1], hi, mid, lo);
        pp[t][0][e >> 1] = hi; pp[t][1][e >> 1] = mid; pp[t][2][e >> 1] = lo;
      }
#pragma unroll
    for (int t = 0; t < 2; t++) {
#pragma unroll
      for (int db = 0; db < 2; db++) {
        const int d = db * 32 + j;
        const unsigned char* vp = sV + d * 64 + (((2 * t + h) ^ ((d >> 2) & 3)) << 4);
        const bf16x8 v_h = *reinterpret_cast<const bf16x8*>(vp);
        const bf16x8 v_m = *reinterpret_cast<const bf16x8*>(vp + AX_VPLANE);
        const bf16x8 v_l = *reinterpret_cast<const bf16x8*>(vp + 2 * AX_VPLANE);
        x3_mfma6(o[db], v_h, v_m, v_l, __builtin_bit_cast(bf16x8, pp[t][0]), __builtin_bit_cast(bf16x8, pp[t][1]), __builtin_bit_cast(bf16x8, pp[t][2]));
      }
      if (t == 0) {
#pragma unroll
        for (int q = 0; q < 12; q++) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        }
      }
    }
  }
#undef AXW_PREFETCH
  if (ks.n > ks.nmain) {
    if (ntiles == 0) __syncthreads();
    float qreg[32];
#pragma unroll
    for (int t = 0; t < 32; t++) qreg[t] = Qh[(size_t)(2 * t + h) * ld + qtok] * qscale;
    attn_tail_keys(sTK, ks.n - ks.nmain, qreg, o, m, l, h);
  }
  const float lt = l + __shfl_xor(l, 32);
  const float inv = lt > 0.0f ? 1.0f / lt : 0.0f;
  if (qi < mp) {
    const float keep = qi < st.nmain ? inv : 0.0f;
    float* Oh = O + (size_t)head * 64 * ld + st.tok_off + qi;
#pragma unroll
    for (int db = 0; db < 2; db++)
#pragma unroll
      for (int r = 0; r < 16; r++) Oh[(size_t)(db * 32 + acc_row(r, h)) * ld] = qi < st.nmain ? o[db][r] * keep : 0.0f;
  }
}

}  // namespace dani
#include "attn_rs.h"
#include "ffn_chain.h"
namespace dani {

bool launch_ffn_chain_fm(float* x, const float* ctx, const float* p0, const float* b0, const float* gamma, const float* beta,
                         const float* p3, const float* b3, const float* pd, const float* bd, int OUTD, float* y, const float* cosb,
                         const float* sinb, int rot_rows, float alpha, int nblocks, int ld, hipStream_t s) {
  FfnChainArgs a;
  a.x = x; a.ctx = ctx; a.p0 = p0; a.b0 = b0; a.gamma = gamma; a.beta = beta; a.p3 = p3; a.b3 = b3; a.pd = pd; a.bd = bd;
  a.y = y; a.cosb = cosb; a.sinb = sinb; a.OUTD = OUTD; a.rot_rows = rot_rows; a.alpha = alpha; a.ld = ld; a.nblocks = nblocks;
  return launch_ffn_chain(a, s);
}
void pack_chain_weights_host(const float* w, int out, int in, int role, float* dst) {
  if (role == 3) pack_chain_proj(w, out, in, dst);
  else pack_chain_a(w, out, in, role == 2, dst);
}

std::atomic<int> g_ffn_chain{-1};   // dani_lg_set_kernel_option("ffn_chain"): -1 = per shape (lg_forward), 0 = never, 1 = always
std::atomic<int> g_attn_force{0};   // tuning hook of tools/probes/attn_bench.hip: 1 = attn_ks_kernel, 2 = attn_kernel<4>, 3 = attn_x3_kernel, 4 = attn_x3w_kernel, 5 = attn_k4_kernel, 6 = attn_rs_kernel

void launch_attention(const float* Q, const float* K, const float* V, float* O, int ld, const LgSet* sets, int nsets,
                      int max_mp, int max_tail, int cross, float scale, hipStream_t s, int x3) {
  const int force = g_attn_force.load(std::memory_order_relaxed);   // one read per launch (dani_lg_set_kernel_option may run on another thread)
  // grid.x: query tiles of the main regions + max_tail slots per (head, set) for the tail-query workgroups (attn_role)
  if ((x3 && force == 0 && ((max_mp + 127) / 128) * 4 * nsets >= 512) || force == 4)   // as for the f32 pair below
    hipLaunchKernelGGL(attn_x3w_kernel, dim3((max_mp + 127) / 128 + max_tail, 4, nsets), dim3(256), 0, s, Q, K, V, O, ld, sets, max_tail, cross, scale);
  else if (x3 || force == 3)
    hipLaunchKernelGGL(attn_x3_kernel, dim3((max_mp + 63) / 64 + max_tail, 4, nsets), dim3(256), 0, s, Q, K, V, O, ld, sets, max_tail, cross, scale);
  // 128-query workgroups (no key split, half the tile staging per MFMA, no merge) once they fill the chip's 512 resident
  // slots in whole rounds; below that the key-split kernel's twice as many, half as long workgroups balance better.
  // At 16 x 1025 keypoints (512 workgroups = one round): 150.1 vs 156.0 us (tools/probes/attn_bench.hip).
  else if (force != 1 && (force == 2 || force == 6 || force == 7 || force == 8 || ((max_mp + 127) / 128) * 4 * nsets >= 512)) {
    // register-streaming form (attn_rs.h) unless DANI_ATTN_RS=0 or the probe asks for the LDS-tiled kernel (force 2)
    static const bool rs_off = getenv("DANI_ATTN_RS") && atoi(getenv("DANI_ATTN_RS")) == 0;
    static const int rs_tail_sign = getenv("DANI_ATTN_RS_TAIL_FIRST") ? -1 : 1;   // probe knob: tail-query workgroups first in dispatch order
    // ... and unless every key set is shorter than one 128-key super-tile: a batch of many small sets (dani_lg_match_groups:
    // ~200 sets of 40-60 keypoints) would run a whole super-tile -- 256 MFMAs, 64 16-byte loads -- per wave for 40-60 valid
    // keys, 2-3x the work of attn_kernel<4>'s 32-key tiles
    if (force == 6 || (force == 0 && !rs_off && max_mp >= 128))
      // ring of 8 loads, 2 waves per SIMD: equal to the ring of 4 at 3 waves per SIMD without tail queries, 2-3 us faster with
      // them (tools/probes/attn_bench.hip, profiles/r03_attn_rs_probe.txt; force 7 = the ring of 4)
      hipLaunchKernelGGL(attn_rs_kernel<8>, dim3((max_mp + 127) / 128 + max_tail, 4, nsets), dim3(256), 0, s, Q, K, V, O, ld, sets, rs_tail_sign * max_tail, cross, scale);
    else if (force == 8)   // probe: ring of 16 loads
      hipLaunchKernelGGL(attn_rs_kernel<16>, dim3((max_mp + 127) / 128 + max_tail, 4, nsets), dim3(256), 0, s, Q, K, V, O, ld, sets, rs_tail_sign * max_tail, cross, scale);
    else if (force == 7)
      hipLaunchKernelGGL(attn_rs_kernel<4>, dim3((max_mp + 127) / 128 + max_tail, 4, nsets), dim3(256), 0, s, Q, K, V, O, ld, sets, rs_tail_sign * max_tail, cross, scale);
    else
    hipLaunchKernelGGL(attn_kernel<4>, dim3((max_mp + 127) / 128 + max_tail, 4, nsets), dim3(256), 0, s, Q, K, V, O, ld, sets, max_tail, cross, scale);
  }
  // fewer key-split workgroups than CUs (one pair of 1024 keypoints: 136): four key quarters per 32-query block instead
  else if (force == 5 || (force == 0 && ((max_mp + 63) / 64) * 4 * nsets <= 256)) {
    static bool attr_set_dev[64] = {};   // per device; idempotent: a race between two first callers sets the same values twice
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    bool& attr_set = attr_set_dev[dev_ & 63];
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)attn_k4_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, atk4_lds_floats<4>() * 4);
      (void)hipFuncSetAttribute((const void*)attn_k4_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, atk4_lds_floats<8>() * 4);
      attr_set = true;
    }
    static const int parts = getenv("DANI_ATTN_K4_PARTS") ? atoi(getenv("DANI_ATTN_K4_PARTS")) : 4;
    static const int tail_sign = getenv("DANI_ATTN_K4_TAIL_LAST") ? 1 : -1;
    const int k4_tail = tail_sign * max_tail;
    if (parts == 8 && max_mp >= 8 * 32 * 2)   // at least two key tiles per part
      hipLaunchKernelGGL(attn_k4_kernel<8>, dim3((max_mp + 31) / 32 + max_tail, 4, nsets), dim3(512), (size_t)atk4_lds_floats<8>() * 4, s, Q, K, V, O, ld, sets, k4_tail, cross, scale);
    else
      hipLaunchKernelGGL(attn_k4_kernel<4>, dim3((max_mp + 31) / 32 + max_tail, 4, nsets), dim3(256), (size_t)atk4_lds_floats<4>() * 4, s, Q, K, V, O, ld, sets, k4_tail, cross, scale);
  } else
    hipLaunchKernelGGL(attn_ks_kernel, dim3((max_mp + 63) / 64 + max_tail, 4, nsets), dim3(256), 0, s, Q, K, V, O, ld, sets, max_tail, cross, scale);
}

// ---------------------------------------------------------------------------------------------
// MatchAssignment: matchability logit -> logsigmoid;  sim = md0^T md1 per pair;  log-double-softmax statistics;
// row / column argmax of  scores = log_softmax_row(sim) + log_softmax_col(sim) + logsig(z0) + logsig(z1).
// ---------------------------------------------------------------------------------------------
// The matchability term and the dense copy in one pass (rounds 1-3: a thread per token walking 256 rows alone, 21.7 us, then 70 000
// workgroups of 64 threads copying one row segment each, 18.4 us).  The assignment stage works on a dense copy of the projected
// descriptors and matchability terms: set s occupies [dense_off, dense_off + n) (padded to 32), gathered from its main region
// and its tail tokens.  Workgroup = 64 tokens of one set x 8 feature groups of 32: each thread copies its 32 rows of md and
// accumulates its share of z = b + w . x; the eight partial sums meet in LDS in group order.
__global__ __launch_bounds__(512) void match_densify_kernel(const float* x, const float* w, float b, const float* md, const LgSet* sets,
                                                            float* md_dense, float* logsig_dense, int ld) {
  __shared__ float part[8][64];
  const LgSet st = sets[blockIdx.y];
  const int tl = threadIdx.x & 63, g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = blockIdx.x * 64 + tl;
  const bool ok = i < st.n;
  const int tok = !ok ? 0 : (i < st.nmain ? st.tok_off + i : st.tail_off + (i - st.nmain));
  const int dpos = st.dense_off + i;
  float z = 0.0f;
#pragma unroll 8
  for (int k = 0; k < 32; k++) {
    const int f = g * 32 + k;
    const float xv = ok ? x[(size_t)f * ld + tok] : 0.0f;
    if (ok) md_dense[(size_t)f * ld + dpos] = md[(size_t)f * ld + tok];
    z = __builtin_fmaf(w[f], xv, z);
  }
  part[g][tl] = z;
  __syncthreads();
  if (g == 0 && ok) {
    float zz = b;
#pragma unroll
    for (int k = 0; k < 8; k++) zz += part[k][tl];
    logsig_dense[dpos] = -(fmaxf(-zz, 0.0f) + log1pf(__expf(-fabsf(zz))));
  }
}
void launch_match_densify(const float* x, const float* w, float b, const float* md, const LgSet* sets, int nsets, int max_n, float* md_dense,
                          float* logsig_dense, int ld, hipStream_t s) {
  hipLaunchKernelGGL(match_densify_kernel, dim3((max_n + 63) / 64, nsets), dim3(512), 0, s, x, w, b, md, sets, md_dense, logsig_dense, ld);
}

// sim[pair][i][j] = sum_d md[d][off0+i] * md[d][off1+j]; tile 64 (i) x 128 (j): wave w -> j block pair? 4 waves = 2x2 of 32x64
__global__ __launch_bounds__(256, 2) void sim_kernel(const float* md, int ld, const LgSet* sets, float* sim, int sim_ld,
                                                     size_t sim_pair_stride) {
  __shared__ __attribute__((aligned(16))) float lds[32 * 64 + 32 * 128];
  float* sA = lds;            // [32 k][64 i]
  float* sB = lds + 32 * 64;  // [32 k][128 j]
  const int pair = blockIdx.z;
  const LgSet s0 = sets[2 * pair], s1 = sets[2 * pair + 1];
  const int np0 = (s0.n + 31) & ~31, np1 = (s1.n + 31) & ~31;
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 128;
  if (i0 >= np0 || j0 >= np1) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
  const int wi = wave >> 1, wj = wave & 1;   // wave tile: 32 i x 64 j
  f32x16 acc[2];
#pragma unroll
  for (int r = 0; r < 16; r++) { acc[0][r] = 0.0f; acc[1][r] = 0.0f; }
  // (chunk k0 + 32 prefetched into registers under chunk k0's MFMAs was measured and changed nothing: 55.0 against 56.1 us -- two
  //  workgroups per CU already cover each other's loads; the launch is 2.4 rounds of 512 resident workgroups with 11 % of the tile
  //  area past the sets' ends)
  for (int k0 = 0; k0 < 256; k0 += 32) {
    __syncthreads();
    // A: 32 x 64 floats = 512 float4 (2/thread); B: 32 x 128 = 1024 float4 (4/thread)
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int e = tid + 256 * u, kk = e >> 4, c4 = e & 15;
      reinterpret_cast<f32x4*>(sA)[e] = *reinterpret_cast<const f32x4*>(md + (size_t)(k0 + kk) * ld + s0.dense_off + i0 + 4 * c4);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int e = tid + 256 * u, kk = e >> 5, c4 = e & 31;
      reinterpret_cast<f32x4*>(sB)[e] = *reinterpret_cast<const f32x4*>(md + (size_t)(k0 + kk) * ld + s1.dense_off + j0 + 4 * c4);
    }
    __syncthreads();
#pragma unroll
    for (int kp = 0; kp < 16; kp++) {
      const float av = sA[(2 * kp + h) * 64 + wi * 32 + j];
      const float b0 = sB[(2 * kp + h) * 128 + wj * 64 + j], b1 = sB[(2 * kp + h) * 128 + wj * 64 + 32 + j];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc[1], 0, 0, 0);
    }
  }
  float* sp = sim + (size_t)pair * sim_pair_stride;
#pragma unroll
  for (int jb = 0; jb < 2; jb++) {
    const int jj = j0 + wj * 64 + jb * 32 + j;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int ii = i0 + wi * 32 + acc_row(r, h);
      if (ii < np0 && jj < np1) sp[(size_t)ii * sim_ld + jj] = acc[jb][r];
    }
  }
}

void launch_sim(const float* md, int ld, const LgSet* sets, int npairs, int max_np, float* sim, int sim_ld,
                size_t sim_pair_stride, hipStream_t s) {
  hipLaunchKernelGGL(sim_kernel, dim3((max_np + 127) / 128, (max_np + 63) / 64, npairs), dim3(256), 0, s, md, ld, sets,
                     sim, sim_ld, sim_pair_stride);
}

// row statistics: lse_row[i] = max_j + log sum_j exp(sim - max) over valid j.  One wave per row.
__global__ __launch_bounds__(64) void row_lse_kernel(const float* sim, int sim_ld, size_t pstride, const LgSet* sets,
                                                     float* lse_row, int ld) {
  const int pair = blockIdx.y, i = blockIdx.x;
  const LgSet s0 = sets[2 * pair], s1 = sets[2 * pair + 1];
  if (i >= s0.n) return;
  const float* row = sim + (size_t)pair * pstride + (size_t)i * sim_ld;
  float mx = -1e30f;
  for (int jj = threadIdx.x; jj < s1.n; jj += 64) mx = fmaxf(mx, row[jj]);
  mx = wave_max(mx);
  float sm = 0.0f;
  for (int jj = threadIdx.x; jj < s1.n; jj += 64) sm += __expf(row[jj] - mx);
  sm = wave_sumf(sm);
  if (threadIdx.x == 0) lse_row[s0.dense_off + i] = mx + __logf(sm);
}

// column statistics + column argmax of (2*sim - lse_row[i] + ls0[i]): thread per column, SPLIT-way row split per
// block (64 columns x 16 row groups = 1024 threads: the 4-way version left 136 workgroups of 4 waves for the chip)
// COLS x SPLIT = 1024 threads: 64 columns x 16 row groups for batches (the 4-way version left 136 workgroups of 4 waves for
// the chip), 16 columns x 64 row groups for one or two pairs (17 workgroups of the first form walk 64 rows each in a serial
// online-logsumexp chain: 26 us; 65 workgroups walking 16 rows each: the launch is four times shorter).
template <int COLS, int SPLIT>
__global__ __launch_bounds__(COLS * SPLIT) void col_stats_kernel(const float* sim, int sim_ld, size_t pstride, const LgSet* sets,
                                                        const float* lse_row, const float* logsig, float* lse_col,
                                                        int* m1, int ld) {
  __shared__ float rmx[SPLIT][COLS], rsm[SPLIT][COLS], rbv[SPLIT][COLS];
  __shared__ int rbi[SPLIT][COLS];
  const int pair = blockIdx.y;
  const LgSet s0 = sets[2 * pair], s1 = sets[2 * pair + 1];
  const int lj = threadIdx.x % COLS, q = threadIdx.x / COLS;
  const int jj = blockIdx.x * COLS + lj;
  const bool ok = jj < s1.n;
  const float* sp = sim + (size_t)pair * pstride + (ok ? jj : 0);
  // pass 1: max over rows (online logsumexp) and best score
  float mx = -1e30f, sm = 0.0f, bv = -3e38f;
  int bi = 0;
  for (int i = q; i < s0.n; i += SPLIT) {
    const float v = sp[(size_t)i * sim_ld];
    const float nm = fmaxf(mx, v);
    sm = sm * __expf(mx - nm) + __expf(v - nm);
    mx = nm;
    const float sc = 2.0f * v - lse_row[s0.dense_off + i] + logsig[s0.dense_off + i];
    if (sc > bv) { bv = sc; bi = i; }
  }
  rmx[q][lj] = mx; rsm[q][lj] = sm; rbv[q][lj] = bv; rbi[q][lj] = bi;
  __syncthreads();
  if (q == 0 && ok) {
    float M = rmx[0][lj];
    for (int k = 1; k < SPLIT; k++) M = fmaxf(M, rmx[k][lj]);
    float S = 0.0f;
    for (int k = 0; k < SPLIT; k++) S += rsm[k][lj] * __expf(rmx[k][lj] - M);
    lse_col[s1.dense_off + jj] = M + __logf(S);
    float B = rbv[0][lj]; int I = rbi[0][lj];
    for (int k = 1; k < SPLIT; k++)
      if (rbv[k][lj] > B || (rbv[k][lj] == B && rbi[k][lj] < I)) { B = rbv[k][lj]; I = rbi[k][lj]; }
    m1[s1.dense_off + jj] = I;
  }
}

// row argmax of scores and the score value itself.  One wave per row; first maximum wins ties.
__global__ __launch_bounds__(64) void row_argmax_kernel(const float* sim, int sim_ld, size_t pstride, const LgSet* sets,
                                                        const float* lse_row, const float* lse_col, const float* logsig,
                                                        int* m0, float* max0) {
  const int pair = blockIdx.y, i = blockIdx.x;
  const LgSet s0 = sets[2 * pair], s1 = sets[2 * pair + 1];
  if (i >= s0.n) return;
  const float* row = sim + (size_t)pair * pstride + (size_t)i * sim_ld;
  const float ri = lse_row[s0.dense_off + i], zi = logsig[s0.dense_off + i];
  float bv = -3e38f;
  int bj = 0x7fffffff;
  for (int jj = threadIdx.x; jj < s1.n; jj += 64) {
    const float v = row[jj];
    const float sc = (v - ri) + (v - lse_col[s1.dense_off + jj]) + (zi + logsig[s1.dense_off + jj]);
    if (sc > bv) { bv = sc; bj = jj; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(bv, off);
    const int oj = __shfl_xor(bj, off);
    if (ov > bv || (ov == bv && oj < bj)) { bv = ov; bj = oj; }
  }
  if (threadIdx.x == 0) { m0[s0.dense_off + i] = bj; max0[s0.dense_off + i] = bv; }
}

void launch_assignment_stats(const float* sim, int sim_ld, size_t pstride, const LgSet* sets, int npairs, int max_n,
                             const float* logsig, float* lse_row, float* lse_col, int* m0, float* max0, int* m1, int ld,
                             hipStream_t s) {
  hipLaunchKernelGGL(row_lse_kernel, dim3(max_n, npairs), dim3(64), 0, s, sim, sim_ld, pstride, sets, lse_row, ld);
  if (((max_n + 63) / 64) * npairs < 64)
    hipLaunchKernelGGL((col_stats_kernel<16, 64>), dim3((max_n + 15) / 16, npairs), dim3(1024), 0, s, sim, sim_ld, pstride, sets, lse_row,
                       logsig, lse_col, m1, ld);
  else if (((max_n + 63) / 64) * npairs < 256)   // 8 pairs of ~1026 keypoints: 136 workgroups of the 64-column form for 256 CUs; 32 columns x 32 row groups: 264
    hipLaunchKernelGGL((col_stats_kernel<32, 32>), dim3((max_n + 31) / 32, npairs), dim3(1024), 0, s, sim, sim_ld, pstride, sets, lse_row,
                       logsig, lse_col, m1, ld);
  else
    hipLaunchKernelGGL((col_stats_kernel<64, 16>), dim3((max_n + 63) / 64, npairs), dim3(1024), 0, s, sim, sim_ld, pstride, sets, lse_row,
                       logsig, lse_col, m1, ld);
  hipLaunchKernelGGL(row_argmax_kernel, dim3(max_n, npairs), dim3(64), 0, s, sim, sim_ld, pstride, sets, lse_row, lse_col,
                     logsig, m0, max0);
}

}  // namespace dani
