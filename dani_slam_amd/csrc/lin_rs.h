// lin_rs.h -- register-streaming form of the LightGlue feature-major linear layer (included by lg_kernels.hip).
//
//   Y[o][t] = epilogue( sum_k W[o][k] * X[k][t] + b[o] ),   X = concat(X0 rows, X1 rows)     (as linear_fm_kernel)
//
// v_mfma_f32_32x32x2_f32 holds a SIMD's matrix pipe for 64 cycles and takes one dword per lane and operand, so a wave can
// feed it straight from L1 / L2 with global loads issued a few thousand cycles ahead: no LDS tile, no ds_write pass, no
// workgroup barrier, no lock-step between the waves of a CU.  linear_fm_kernel spends ~20 % of a K = 256 layer outside its
// MFMA loop (two barriers per 64-row chunk, staging, and every workgroup of a one-round launch in its prologue at once);
// this form measured 18-27 % less time per launch on the LightGlue shapes at 16 k tokens (tools/probes/lin_bench.hip,
// profiles/r03_lin_rs_probe.txt).
//
//   * B operand, 16 bytes per lane: lane (j, h) loads tokens 4j .. 4j+3 of row 2p + h.  Element m of that load is the B
//     operand of token block m, where block m holds the tokens 4j + m of a 128-token tile (four blocks interleaved by
//     token, not contiguous): one load feeds the four accumulators of a k-pair, a half-wave reads 512 contiguous bytes,
//     and in the epilogue the four accumulators give a lane four consecutive tokens of a row: 16-byte stores.
//   * A operand, 16 bytes per lane: weights packed as [out block of 32][k-quad][lane][4] (pack_linear_rs): element i is
//     the lane's operand of k-pair 4q + i.
//   * a wave owns NCB x 4 accumulators = NCB * 32 outputs x 128 tokens; the four waves of a workgroup take four out
//     groups of ONE token tile (the activation rows reach the CU's L1 once).  NCB is chosen per launch so that the grid is
//     a whole number of waves per SIMD where possible (launch_linear_rs).
//   * operand registers form a ring of UQ k-quads; a slot's registers are reloaded right behind the MFMAs that consumed
//     them.  hipcc cannot be made to keep such a ring of plain loads in flight (it rotates the loop into load / wait / use,
//     or drains vmcnt(0) at the loop head), so the loads are inline asm and the waits are counted by hand: before slot s is
//     used, the (UQ - 1) * (NCB + 4) loads of the other slots are the only younger ones.  Loads the compiler issues on its
//     own (bias, epilogue operands) only make these waits more conservative: vmcnt retires in issue order.
//   * the last T % 128 tokens (the shared tail region of a batch of ~1025-keypoint sets: 32 tokens) are not worth a round
//     of 128-token tiles; they go to separate small workgroups dispatched FIRST in the same launch (role `tail`): one
//     32 x 32 output block each, K split over the four waves, partial sums folded in wave order through LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "lin_epilogue.h"

namespace dani {

struct LinRsArgs {
  const float* x0; const float* x1; int K0, K1;
  const float* wq; const float* bias;      // wq: pack_linear_rs layout; bias: [OUT]
  float* y; const float* res; const float* cosb; const float* sinb;
  int OUT, T, ld, rot_rows;
  float alpha;
  int n_tail_wg;                            // workgroups [0, n_tail_wg) serve the tokens from tail_tok0 on in 32 x 32 blocks (multiple of 8)
  int tail_tok0;                            // (T / 128) * 128, or 0: every token goes through the tail role (small launches)
};

// [out][in] row-major -> [out block of 32][k-quad][lane = h * 32 + row][4 k-pairs]
inline void pack_linear_rs(const float* w_row_major, int out, int in, float* dst) {
  for (int o = 0; o < out; o++)
    for (int k = 0; k < in; k++) {
      const int o32 = o >> 5, jj = o & 31, p = k >> 1, hh = k & 1, q = p >> 2, i = p & 3;
      dst[(((size_t)o32 * (in / 8) + q) * 64 + hh * 32 + jj) * 4 + i] = w_row_major[(size_t)o * in + k];
    }
}

typedef float rs_f32x16 __attribute__((ext_vector_type(16)));
typedef float rs_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int rs_acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---- tail role: one 32-output x 32-token block per workgroup, K split over the four waves ------------------------------
// NQW = k-quads per wave (K / 32): ALL of a wave's operand loads are issued up front (straight-line code: hipcc counts the
// waits itself), so a block costs one memory round trip plus its K / 8 MFMAs -- these workgroups exist for latency
// (the tail tokens of a large launch, every token of a small one), not for throughput.
template <int NQW>
__device__ __forceinline__ void linear_rs_tail(const LinRsArgs& a, int unit, float* lds) {
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int oblocks = a.OUT / 32;
  const int tb = unit / oblocks, ob = unit % oblocks;
  const int tok = a.tail_tok0 + tb * 32 + j;
  constexpr int nq = NQW * 4;
  const float* wb = a.wq + ((size_t)ob * nq + wave * NQW) * 256 + lane * 4;
  const unsigned boff = (unsigned)h * (unsigned)a.ld + (unsigned)min(tok, a.T - 1);
  const float* xs1 = a.x1 - (size_t)a.K0 * a.ld;
  rs_f32x4 av[NQW]; float bv[NQW][4];
#pragma unroll
  for (int u = 0; u < NQW; u++) {
    const int q = wave * NQW + u;
    av[u] = *reinterpret_cast<const rs_f32x4*>(wb + (size_t)u * 256);
    const float* rb = ((8 * q < a.K0) ? a.x0 : xs1) + (size_t)(8 * q) * a.ld;
#pragma unroll
    for (int i = 0; i < 4; i++) bv[u][i] = rb[(size_t)(2 * i) * a.ld + boff];
  }
  rs_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] = wave ? 0.0f : a.bias[ob * 32 + rs_acc_row(r, h)];
#pragma unroll
  for (int u = 0; u < NQW; u++)
#pragma unroll
    for (int i = 0; i < 4; i++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][i], bv[u][i], acc, 0, 0, 0);
  float* xch = lds + ((wave - 1) * 64 + lane) * 17;
  if (wave) {
#pragma unroll
    for (int r = 0; r < 16; r++) xch[r] = acc[r];
  }
  __syncthreads();
  if (wave) return;
#pragma unroll
  for (int w = 0; w < 3; w++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] += lds[(w * 64 + lane) * 17 + r];
  LinEpi ep;
  ep.y = a.y; ep.res = a.res; ep.cosb = a.cosb; ep.sinb = a.sinb; ep.bias = nullptr; ep.ld = a.ld; ep.rot_rows = a.rot_rows; ep.alpha = a.alpha;
  lin_epilogue<false>(acc, ob * 32, h, tok, tok < a.T, ep);
}

template <int NCB, int UQ>
__global__ __launch_bounds__(256, NCB == 1 ? 3 : (NCB == 2 ? 2 : 1)) void linear_rs_kernel(LinRsArgs a) {
  __shared__ float tail_lds[3 * 64 * 17];
  if ((int)blockIdx.x < a.n_tail_wg) {
    const int units = ((a.T - a.tail_tok0 + 31) / 32) * (a.OUT / 32);
    if ((int)blockIdx.x < units) {
      if (a.K0 + a.K1 == 256) linear_rs_tail<8>(a, blockIdx.x, tail_lds);
      else linear_rs_tail<16>(a, blockIdx.x, tail_lds);   // K = 512 (launch_linear_rs admits 256 and 512 only)
    }
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int GO = 4 * NCB * 32;                  // outputs per workgroup
  const int otiles = a.OUT / GO, ntiles = a.tail_tok0 / 128;   // full token tiles only
  // 1-D grid, XCD-aware as linear_fm_kernel: all out groups of a token tile on one XCD, back to back (n_tail_wg % 8 == 0)
  const int id = blockIdx.x - a.n_tail_wg;
  const int xcd = id & 7, slot = id >> 3;
  const int tile = (slot / otiles) * 8 + xcd, ot = slot % otiles;
  if (tile >= ntiles) return;
  const int out0 = ot * GO + wave * NCB * 32;
  const int t4 = tile * 128 + 4 * j;
  const int K = a.K0 + a.K1, nq = K / 8;

  rs_f32x16 acc[NCB][4];
  const unsigned aoff = (unsigned)lane * 16u;
  const unsigned boff = ((unsigned)h * (unsigned)a.ld + (unsigned)t4) * 4u;
  const float* xs1 = a.x1 - (size_t)a.K0 * a.ld;   // row k >= K0 of the concatenation is xs1 + k * ld

  rs_f32x4 av[UQ][NCB], bv[UQ][4];
  auto load_b = [&](int s, int q, int i) {
    const float* ri = ((8 * q < a.K0) ? a.x0 : xs1) + (size_t)(8 * q + 2 * i) * a.ld;   // uniform: scalar unit
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(bv[s][i]) : "v"(boff), "s"(ri));
  };
  auto load_a = [&](int s, int q) {
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
      const float* ab = a.wq + ((size_t)(out0 / 32 + cb) * nq + q) * 256;   // uniform
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(av[s][cb]) : "v"(aoff), "s"(ab));
    }
  };
  auto group = [&](int s, int i) {   // k-pair i of the slot's quad: NCB x 4 MFMAs
#pragma unroll
    for (int cb = 0; cb < NCB; cb++)
#pragma unroll
      for (int m = 0; m < 4; m++)
        acc[cb][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][cb][i], bv[s][i][m], acc[cb][m], 0, 0, 0);
  };
  constexpr int LPQ = NCB + 4;   // loads per k-quad
  // wait until at most N_ of this wave's vector-memory operations are outstanding; naming the slot's registers as in/out
  // operands keeps every use of them (the MFMAs) behind the wait
#define RS_WAIT(N_, s_)                                                                                                  \
  {                                                                                                                      \
    if constexpr (NCB == 1)                                                                                              \
      asm volatile("s_waitcnt vmcnt(%5)" : "+v"(av[s_][0]), "+v"(bv[s_][0]), "+v"(bv[s_][1]), "+v"(bv[s_][2]), "+v"(bv[s_][3]) : "n"(N_)); \
    else if constexpr (NCB == 2)                                                                                         \
      asm volatile("s_waitcnt vmcnt(%6)" : "+v"(av[s_][0]), "+v"(av[s_][1]), "+v"(bv[s_][0]), "+v"(bv[s_][1]), "+v"(bv[s_][2]), "+v"(bv[s_][3]) : "n"(N_)); \
    else                                                                                                                 \
      asm volatile("s_waitcnt vmcnt(%7)" : "+v"(av[s_][0]), "+v"(av[s_][1]), "+v"(av[s_][NCB - 1]), "+v"(bv[s_][0]), "+v"(bv[s_][1]), "+v"(bv[s_][2]), "+v"(bv[s_][3]) : "n"(N_)); \
  }
#pragma unroll
  for (int s = 0; s < UQ; s++) {
#pragma unroll
    for (int i = 0; i < 4; i++) load_b(s, s, i);
    load_a(s, s);
  }
  // the accumulators start from the bias; its loads go out BEHIND the ring's first operands, so that the wave waits for one
  // memory round trip before its first MFMA, not for two in a row
#pragma unroll
  for (int cb = 0; cb < NCB; cb++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const float bv = a.bias[out0 + cb * 32 + rs_acc_row(r, h)];
#pragma unroll
      for (int m = 0; m < 4; m++) acc[cb][m][r] = bv;
    }
  int q0 = 0;
  for (; q0 + UQ < nq; q0 += UQ) {
#pragma unroll
    for (int s = 0; s < UQ; s++) {
      RS_WAIT((UQ - 1) * LPQ, s);
      int qn = min(q0 + s + UQ, nq - 1);   // nq is a multiple of UQ (launcher), so the clamp never fires; it keeps the address in range for the analyser
      asm volatile("" : "+s"(qn));
#pragma unroll
      for (int i = 0; i < 4; i++) {
        group(s, i);
        load_b(s, qn, i);                   // row i of the slot is free again: reload it under the next group's MFMAs
        __builtin_amdgcn_sched_barrier(0);
      }
      load_a(s, qn);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // last UQ quads: nothing is reloaded, so slot s waits for (UQ - 1 - s) younger slots (the count must be a literal)
  auto last = [&](auto sc) {
    constexpr int s = decltype(sc)::value;
    if constexpr (s < UQ) {
      RS_WAIT((UQ - 1 - s) * LPQ, s);
#pragma unroll
      for (int i = 0; i < 4; i++) group(s, i);
    }
  };
  last(std::integral_constant<int, 0>()); last(std::integral_constant<int, 1>());
  last(std::integral_constant<int, 2>()); last(std::integral_constant<int, 3>());
  static_assert(UQ <= 4, "extend the calls above");
#undef RS_WAIT

  // epilogue: lane (j, h) holds tokens t4 .. t4 + 3 (accumulators m = 0 .. 3) of the rows acc_row(r, h): 16-byte accesses.
  // Arithmetic per element exactly as lin_epilogue: rotary pair (v0, v1) -> (v0 c - v1 s, v1 c + v0 s); * alpha; + res.
#pragma unroll
  for (int cb = 0; cb < NCB; cb++) {
    const int row0 = out0 + cb * 32;
    const bool rot = row0 < a.rot_rows;
#pragma unroll
    for (int half = 0; half < 2; half++) {   // 8 rows at a time: the operand loads of a half are issued together, ahead of its arithmetic
      rs_f32x4 cs[4], sn[4], rv[8];
      if (rot) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int rr = row0 + rs_acc_row(8 * half + 2 * i, h);
          const size_t ip = (size_t)((rr & 63) >> 1) * a.ld + t4;
          cs[i] = *reinterpret_cast<const rs_f32x4*>(a.cosb + ip); sn[i] = *reinterpret_cast<const rs_f32x4*>(a.sinb + ip);
        }
      }
      if (a.res) {
#pragma unroll
        for (int i = 0; i < 8; i++) rv[i] = *reinterpret_cast<const rs_f32x4*>(a.res + (size_t)(row0 + rs_acc_row(8 * half + i, h)) * a.ld + t4);
      }
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        const int r = 8 * half + i;
        const int rr = row0 + rs_acc_row(r, h);   // this lane's rows rr, rr + 1 (a rotary pair)
        rs_f32x4 v0, v1;
#pragma unroll
        for (int m = 0; m < 4; m++) { v0[m] = acc[cb][m][r]; v1[m] = acc[cb][m][r + 1]; }
        if (rot) {
          const rs_f32x4 c = cs[i >> 1], sv = sn[i >> 1];
          const rs_f32x4 r0 = v0 * c - v1 * sv, r1 = v1 * c + v0 * sv;
          v0 = r0; v1 = r1;
        }
        v0 *= a.alpha; v1 *= a.alpha;
        if (a.res) { v0 += rv[i]; v1 += rv[i + 1]; }
        float* yp = a.y + (size_t)rr * a.ld + t4;
#ifdef RS_SKIP_STORES   // probe builds only (tools/probes/lin_bench): the launch without its store tail
        asm volatile("" :: "v"(v0), "v"(v1), "v"(yp));
#else
        *reinterpret_cast<rs_f32x4*>(yp) = v0; *reinterpret_cast<rs_f32x4*>(yp + a.ld) = v1;
#endif
      }
    }
  }
}

// Which form serves (OUT, K, T) best?  Cost in units of one wave's K / 2 MFMAs on a SIMD, rounds of 256 workgroups:
// this kernel runs NCB * 4 such units per round at ~0.9 of the pipe, linear_fm_kernel one unit per round of its 64 x 64
// workgroups at ~0.75.  Returns NCB (1, 2, 3) or 0 for linear_fm_kernel.
inline int linear_rs_choose(int OUT, int K, int T, int ld) {
  if (OUT % 128 || K % 32 || T % 32 || ld % 4 || T < 128) return 0;
  const int nt = T / 128;
  int best = 0; double best_cost = 0.0;
  for (int ncb = 1; ncb <= 3; ncb++) {
    if (OUT % (128 * ncb)) continue;
    const int nwg = nt * (OUT / (128 * ncb));
    const double cost = (double)((nwg + 255) / 256) * ncb * 4 / 0.9;
    if (!best || cost <= best_cost) { best = ncb; best_cost = cost; }   // ties: the larger tile (fewer operand loads per MFMA)
  }
  const int nwg64 = ((T + 63) / 64) * (OUT / 64);
  const double cost_fm = (double)((nwg64 + 255) / 256) / 0.75;
  return best_cost <= cost_fm ? best : 0;
}

inline bool launch_linear_rs(int ncb, const LinRsArgs& a0, hipStream_t s) {   // ncb: 1..3, or 4 = all-tail form
  LinRsArgs a = a0;
  const int K = a.K0 + a.K1;
  if (a.OUT % (128 * (ncb == 4 ? 1 : ncb)) || (K != 256 && K != 512) || a.K0 % 8 || a.ld % 4 || a.rot_rows % 32 || (size_t)a.OUT * a.ld >= ((size_t)1 << 30)) return false;
  const bool all_tail = ncb == 4;   // code 4: no 128-token tiles at all, every token in 32 x 32 K-split blocks
  if (all_tail) ncb = 1;
  a.tail_tok0 = all_tail ? 0 : (a.T / 128) * 128;
  const int units = ((a.T - a.tail_tok0 + 31) / 32) * (a.OUT / 32);
  a.n_tail_wg = (units + 7) / 8 * 8;
  const unsigned g = (unsigned)(a.n_tail_wg + ((a.tail_tok0 / 128 + 7) / 8 * 8) * (a.OUT / (128 * ncb)));
  if (g == 0) return true;
  if (ncb == 1) hipLaunchKernelGGL((linear_rs_kernel<1, 4>), dim3(g), dim3(256), 0, s, a);
  else if (ncb == 2) hipLaunchKernelGGL((linear_rs_kernel<2, 4>), dim3(g), dim3(256), 0, s, a);
  else if (ncb == 3) {
    static const int uq3 = getenv("DANI_LIN_RS_UQ3") ? atoi(getenv("DANI_LIN_RS_UQ3")) : 2;   // probe knob: ring depth of the 96-output tile
    if (uq3 == 4) hipLaunchKernelGGL((linear_rs_kernel<3, 4>), dim3(g), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((linear_rs_kernel<3, 2>), dim3(g), dim3(256), 0, s, a);
  }
  else return false;
  return true;
}

}  // namespace dani
