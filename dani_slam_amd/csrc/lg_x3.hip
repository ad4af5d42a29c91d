// lg_x3.hip -- LightGlue linear layers on the bf16 matrix cores at f32 accuracy ("bf16x3" split).
//
// The parity path (lg_kernels.hip, linear_fm_kernel) contracts in f32 on v_mfma_f32_32x32x2_f32, 157 TFLOP/s peak.
// gfx950's bf16 MFMA (v_mfma_f32_32x32x16_bf16) runs 16x the MACs per cycle.  Here every f32 operand is split into
// three bf16 terms  x = hi + mid + lo  (each the round-to-nearest bf16 of what the previous ones left, so the three
// together carry 24 significand bits, i.e. the f32 value) and the six products of weight 2^0 .. 2^-16 are formed:
//     hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid        (dropped: mid*lo, lo*mid, lo*lo  <= 2^-24 of the product)
// all accumulated in one f32 accumulator.  bf16 x bf16 products are exact in f32, so what is lost against the f32 FMA
// chain is of the order of the f32 rounding itself; the ORDER of the additions differs from the canonical order of
// the oracle, which is why this is an option of the LightGlue stage (tolerance 1e-3 on the scores) and never of
// SuperPoint / YOLO (bit-exact).  Six bf16 MFMAs per 16 k replace eight f32 MFMAs per 16 k at a quarter of the
// cycles each: the layer stops being MFMA-bound; measured 1.3-1.4x per layer (operand split on the VALU and fragment
// traffic take over).
//
// Same interface, token layout, epilogues (rotary, alpha, residual) and 1-D XCD-aware grid as linear_fm_kernel<64>.
// Workgroup = 64 outputs x 64 tokens, waves 2x2 of 32x32.  X is read as f32, split on the VALU while staging and kept
// in LDS as three [token][k] bf16 planes (128-B rows, 16-B slots XOR-swizzled by token pair: ds_read_b128 fragments
// without bank conflicts).  W is split once on the host into MFMA A-fragment order, so a wave reads its fragments
// with one 16-B load per lane straight from L2 (393 KB per 256x256 layer), two k-steps ahead of their use.
#include "common.h"
#include "lin_epilogue.h"
#include "x3_common.h"
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace dani {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct LinX3Args {
  const float* x0; const float* x1; int K0, K1;
  const u32x4* w;   // [otile][K/16][half][plane][lane] fragments of 8 bf16, see pack_linear_x3 below
  const float* bias;
  float* y; const float* res; const float* cosb; const float* sinb;
  int OUT, T, ld, rot_rows;
  float alpha;
};

namespace {
constexpr int X3_TT = 64, X3_KC = 64, X3_PLANE = X3_TT * X3_KC * 2;   // bytes per LDS plane

}  // namespace

// NCB = 32-row output blocks per wave: 1 -> workgroup = 64 outputs x 64 tokens (shipped), 2 -> 128 outputs x 64 tokens
// (the split of an X tile, its LDS traffic and its barriers are shared by twice the outputs; measured slower, kept
// behind the probe's tuning hook g_x3_force_ncb).
template <int NCB>
__global__ __launch_bounds__(256, NCB == 1 ? 4 : 3) void linear_x3_kernel(LinX3Args a) {
  constexpr int WG_OUT = 64 * NCB;
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * X3_PLANE];
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int otiles = (a.OUT + WG_OUT - 1) / WG_OUT, ntiles = (a.T + X3_TT - 1) / X3_TT;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tile = (slot / otiles) * 8 + xcd, ot = slot % otiles;
  if (tile >= ntiles) return;
  const int tok0 = tile * X3_TT;
  const int K = a.K0 + a.K1, nchunks = K / X3_KC, nsteps = K / 16;
  const int wtok = (wave & 1) * 32, wout = (wave >> 1) * 32 * NCB;

  // bias joins in the epilogue: loads issued here would sit between the prologue's prefetches and the loop and
  // make the loop-top wait vmcnt(0), i.e. for the A fragments of the next two steps as well
  f32x16 acc[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; cb++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[cb][r] = 0.0f;

  // staging role: 4 consecutive k rows x 4 consecutive tokens per thread
  // (a 16-lane group of ds_write_b64 = 4 token groups x 4 k groups: with the slot swizzle below its 16 stores fall on
  // 16 different 8-B columns of the 128-B bank row; 16 token groups x 1 k group would be 4-way conflicts)
  const int tg = ((tid >> 4) & 3) * 4 + (tid & 3), kr = (tid >> 6) * 4 + ((tid >> 2) & 3);
  f32x4 px[4];
  auto prefetch_x = [&](int ch) {
    const int k0 = ch * X3_KC + 4 * kr;
    const float* src = (k0 < a.K0) ? a.x0 + (size_t)k0 * a.ld : a.x1 + (size_t)(k0 - a.K0) * a.ld;
#pragma unroll
    for (int u = 0; u < 4; u++) px[u] = *reinterpret_cast<const f32x4*>(src + (size_t)u * a.ld + tok0 + 4 * tg);
  };
  // A fragments: ring of two k-steps
  // packed order is [64-row tile][step][32-row half][plane][lane]: block cb of this wave is
  //   NCB == 1: tile ot, half wave>>1;   NCB == 2: tile 2 ot + (wave>>1), half cb
  const u32x4* wbase = (NCB == 1) ? a.w + ((size_t)ot * nsteps * 2 + (wave >> 1)) * 3 * 64 + lane
                                  : a.w + ((size_t)(2 * ot + (wave >> 1)) * nsteps * 2) * 3 * 64 + lane;
  u32x4 pa[2][NCB][3];
  auto prefetch_a = [&](int gs, int slot_) {
    const u32x4* p = wbase + (size_t)gs * 2 * 3 * 64;
#pragma unroll
    for (int cb = 0; cb < NCB; cb++)
#pragma unroll
      for (int pl = 0; pl < 3; pl++) pa[slot_][cb][pl] = p[(cb * 3 + pl) * 64];
  };
  // B fragment read address of this lane: token row, swizzled slot per step
  const int tb = wtok + j, sw = (tb >> 1) & 7;
  const unsigned char* brow = lds + tb * 128;

  prefetch_x(0);
  __builtin_amdgcn_sched_barrier(0);   // same issue order as in the loop (X, then A): the loop-top wait stays vmcnt(6)
  prefetch_a(0, 0);
  prefetch_a(1, 1);
  __builtin_amdgcn_sched_barrier(0);
  for (int ch = 0; ch < nchunks; ch++) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; c++) {
      unsigned h0, m0, l0, h1, m1, l1;
      x3_split2(px[0][c], px[1][c], h0, m0, l0);
      x3_split2(px[2][c], px[3][c], h1, m1, l1);
      const int t = 4 * tg + c;
      unsigned char* dst = lds + t * 128 + ((((kr >> 1) ^ ((t >> 1) & 7)) << 4) | ((kr & 1) << 3));
      *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(dst + X3_PLANE) = u32x2{m0, m1};
      *reinterpret_cast<u32x2*>(dst + 2 * X3_PLANE) = u32x2{l0, l1};
    }
    __syncthreads();
    // Prefetches are unconditional (indices clamped, the last ones re-read what is already in flight): a branch
    // around a load makes the compiler's s_waitcnt pass assume the shortest path and wait for EVERY load in flight
    // before the first MFMA of the chunk, which exposes the whole L2 latency once per chunk.
    prefetch_x(min(ch + 1, nchunks - 1));
    __builtin_amdgcn_sched_barrier(0);   // left alone, hipcc sinks the prefetch loads down to their first use
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const unsigned char* bp = brow + (((2 * s + h) ^ sw) << 4);
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(bp);
      const bf16x8 bm = *reinterpret_cast<const bf16x8*>(bp + X3_PLANE);
      const bf16x8 bl = *reinterpret_cast<const bf16x8*>(bp + 2 * X3_PLANE);
#pragma unroll
      for (int cb = 0; cb < NCB; cb++) {
        const bf16x8 ah = __builtin_bit_cast(bf16x8, pa[s & 1][cb][0]);
        const bf16x8 am = __builtin_bit_cast(bf16x8, pa[s & 1][cb][1]);
        const bf16x8 al = __builtin_bit_cast(bf16x8, pa[s & 1][cb][2]);
        x3_mfma6(acc[cb], ah, am, al, bh, bm, bl);
      }
      prefetch_a(min(ch * 4 + s + 2, nsteps - 1), s & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  LinEpi ep;
  ep.y = a.y; ep.res = a.res; ep.cosb = a.cosb; ep.sinb = a.sinb; ep.bias = a.bias; ep.ld = a.ld; ep.rot_rows = a.rot_rows; ep.alpha = a.alpha;
  const int tok = tok0 + wtok + j;
#pragma unroll
  for (int cb = 0; cb < NCB; cb++) lin_epilogue<true>(acc[cb], ot * WG_OUT + wout + cb * 32, h, tok, tok < a.T, ep);
}

// Host: split a row-major Linear weight [out][in] into the fragment order the kernel reads:
//   dst[(((ot * in/16 + step) * 2 + half) * 3 + plane) * 64 + lane][8 bf16],  row = ot*64 + half*32 + lane%32,
//   k = 16 step + 8 (lane/32) + 0..7;   rows >= out are zero.  dst holds x3_packed_u16(out, in) 16-bit words.
size_t x3_packed_u16(int out, int in) { return (size_t)((out + 63) / 64) * (in / 16) * 2 * 3 * 64 * 8; }

static inline uint16_t x3_bf16_rne(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float x3_bf16_f32(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

void pack_linear_x3(const float* w, int out, int in, uint16_t* dst) {
  const int otiles = (out + 63) / 64, nsteps = in / 16;
  for (int ot = 0; ot < otiles; ot++)
    for (int st = 0; st < nsteps; st++)
      for (int half = 0; half < 2; half++)
        for (int lane = 0; lane < 64; lane++)
          for (int e = 0; e < 8; e++) {
            const int row = ot * 64 + half * 32 + (lane & 31), k = st * 16 + 8 * (lane >> 5) + e;
            const float v = row < out ? w[(size_t)row * in + k] : 0.0f;
            const uint16_t hi = x3_bf16_rne(v);
            const float r1 = v - x3_bf16_f32(hi);
            const uint16_t mid = x3_bf16_rne(r1);
            const float r2 = r1 - x3_bf16_f32(mid);
            const uint16_t lo = x3_bf16_rne(r2);
            const size_t base = ((((size_t)ot * nsteps + st) * 2 + half) * 3) * 64 * 8 + (size_t)lane * 8 + e;
            dst[base] = hi; dst[base + 64 * 8] = mid; dst[base + 2 * 64 * 8] = lo;
          }
}

int g_x3_force_ncb = 0;   // tuning hook of tools/probes/lin_x3_probe.hip

bool launch_linear_x3(const float* x0, int K0, const float* x1, int K1, const void* w_x3, const float* bias_packed,
                      float* y, int OUT, int T, int ld, const float* res, const float* cosb, const float* sinb,
                      int rot_rows, float alpha, hipStream_t s) {
  if (OUT % 64 || rot_rows % 64 || K0 % 64 || K1 % 64 || (size_t)OUT * ld >= ((size_t)1 << 30)) return false;   // lin_epilogue.h; 64-row chunks: refused
  LinX3Args a;
  a.x0 = x0; a.x1 = x1 ? x1 : x0; a.K0 = K0; a.K1 = K1; a.w = static_cast<const u32x4*>(w_x3); a.bias = bias_packed;
  a.y = y; a.res = res; a.cosb = cosb; a.sinb = sinb; a.OUT = OUT; a.T = T; a.ld = ld; a.rot_rows = rot_rows; a.alpha = alpha;
  const int ntiles8 = ((T + X3_TT - 1) / X3_TT + 7) / 8 * 8;   // see the kernel: 8 XCDs
  // the 64-output tile wins on every LightGlue shape (54 vs 57 us on Wqkv: the 128-output one runs at occupancy 3)
  if (OUT % 128 == 0 && g_x3_force_ncb == 2)
    hipLaunchKernelGGL(linear_x3_kernel<2>, dim3((unsigned)(ntiles8 * (OUT / 128))), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(linear_x3_kernel<1>, dim3((unsigned)(ntiles8 * ((OUT + 63) / 64))), dim3(256), 0, s, a);
  return true;
}

}  // namespace dani
