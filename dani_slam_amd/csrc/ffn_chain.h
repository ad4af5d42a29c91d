// ffn_chain.h -- the token-local half of a LightGlue block as ONE kernel with the activations resident in registers
// (included by lg_kernels.hip).
//
// Between two attention launches a token's path is token-local:
//     h  = W0 [x ; ctx] + b0          (ffn.0 with out_proj / to_out folded in, 512 x 512)
//     g  = GELU(LayerNorm(h))         (512 features of that token)
//     x' = x + W3 g + b3              (ffn.3, 256 x 512, residual)
//     y  = epilogue(Wd x' + bd)       (the NEXT block's projection: to_qk|to_v 512 x 256, Wqkv 768 x 256 with rotary, final_proj)
// Rounds 1-3 ran this as four launches (linear, ln_gelu, linear, linear), each paying its launch boundary, its first operand
// round trip and its store tail (~10 us of a 40-65 us launch), with h / g making three trips through HBM.
//
// v_mfma_f32_16x16x4_f32 needs ONE dword per lane and operand and holds the matrix pipe for 32 cycles, so a single wavefront
// can carry 16 tokens through the whole chain:
//   * its 16 x 512 hidden values are the 32 accumulators of stage A (128 registers) -- LayerNorm statistics are an in-lane sum
//     plus two shuffles, GELU happens in place;
//   * an accumulator IS a B operand: lane (n, g) holds rows 4g + v of a 16-row block for token n, and as the B operand of the next
//     stage it supplies k = g -- so the k-quad "(block b, register v)" contracts over the rows {16 b + 4 g + v}.  The weights of
//     ffn.3 and of the projection are packed in that k order once, on the host; no LDS, no transpose, no barrier anywhere;
//   * the weights stream from L2 through a ring of 16-byte loads (inline asm, hand-counted waits, as lin_rs.h / attn_rs.h);
//   * 16 384 main tokens of the bench's 8 pairs = 1024 blocks = one wave on every SIMD of the chip.
// Tokens beyond the 16-token blocks given to this kernel (the shared tail region) keep the four-launch path.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

namespace dani {

typedef float fc_f4 __attribute__((ext_vector_type(4)));

struct FfnChainArgs {
  float* x; const float* ctx;                  // [256][ld] each; x is updated in place (x' = x + ffn(x, ctx))
  const float* p0; const float* b0;            // ffn.0 (folded), pack_chain_a(512, 512, natural k); bias [512]
  const float* gamma; const float* beta;       // LayerNorm affine [512]
  const float* p3; const float* b3;            // ffn.3, pack_chain_a(256, 512, accumulator k order); bias [256]
  const float* pd; const float* bd;            // next projection, pack_chain_a(OUTD, 256, accumulator k order); bias [OUTD]; null: no stage D
  float* y; const float* cosb; const float* sinb;   // projection output [OUTD][ld]; rotary tables [32][ld]
  int OUTD, rot_rows; float alpha;
  int ld, nblocks;                             // nblocks 16-token blocks, tokens [0, 16 nblocks)
};

// Host: [out][in] row-major -> [k-quad][out group of 64][lane][4]: element i of a lane's 16 bytes is its A operand for out block
// 4 og + i, i.e. W[16 (4 og + i) + (lane & 15)][k(kq, lane >> 4)].  acc_order = false: k = 4 kq + g (B operand loaded from memory,
// rows in natural order); true: kq = 4 b + v contracts over the rows 16 b + 4 g + v (B operand = accumulator register v of block b).
inline void pack_chain_a(const float* w, int out, int in, bool acc_order, float* dst) {
  const int nog = out / 64, nkq = in / 4;
  for (int kq = 0; kq < nkq; kq++)
    for (int og = 0; og < nog; og++)
      for (int lane = 0; lane < 64; lane++)
        for (int i = 0; i < 4; i++) {
          const int m = lane & 15, g = lane >> 4;
          const int k = acc_order ? 16 * (kq >> 2) + 4 * g + (kq & 3) : 4 * kq + g;
          dst[(((size_t)kq * nog + og) * 64 + lane) * 4 + i] = w[(size_t)(16 * (4 * og + i) + m) * in + k];
        }
}

#define FC_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x4f32((a_), (b_), (c_), 0, 0, 0)

// f(integral_constant<int, LO>) ... f(integral_constant<int, HI - 1>), in order, by halving (template depth log2)
template <int LO, int HI, class F>
__device__ __forceinline__ void fc_for(F& f) {
  if constexpr (HI - LO == 1) f(std::integral_constant<int, LO>());
  else if constexpr (HI - LO > 1) { fc_for<LO, (LO + HI) / 2>(f); fc_for<(LO + HI) / 2, HI>(f); }
}

constexpr int FC_SLACK_FLOATS = LG_CHAIN_SLACK_FLOATS;   // a pass's last 16 refills read this far past its stream (the packed buffers carry the slack)

// One GEMM pass over 16 out blocks (256 outputs): acc[ob] += sum over NKQ k-quads of A x B.
//   B operand of k-quad q: sB[q * 64 + lane] -- this WAVE's own slice of LDS, written by the same wave (no barrier), in the order
//     [block b][register v][lane]: q = 4 b + v is the accumulator register v of block b of the previous stage;
//   A operands: a stream of 16-byte loads, 4 per k-quad (one per out group of 64), through a ring of 16 (inline asm; the waits are
//     counted by hand: vmcnt retires in issue order and nothing else loads from global memory inside the loop).
// The loop is ROLLED (4 k-quads per iteration): unrolled over all k-quads with the B operands in registers the pass ran at 50
// cycles per MFMA instead of 32 -- 2048 MFMAs of straight-line code per pass outrun the instruction fetch (tools/probes/
// ffn_chain_probe.hip; profiles/r04_ffn_chain_probe.txt).
template <int NKQ>
__device__ __forceinline__ void chain_pass16(fc_f4 (&acc)[16], const float* pa, unsigned aoff, const float* sB) {
  static_assert(NKQ % 4 == 0, "four k-quads per iteration");
  fc_f4 rg[16];
  const float* lp = pa;                      // next address of the load stream (scalar)
  auto issue = [&](auto slot_c) {
    constexpr int slot = decltype(slot_c)::value;
    const float*& p = lp; fc_f4& dst = rg[slot]; const unsigned aoff_ = aoff;
#ifdef FC_NO_LOADS   // probe builds only: no operand loads (the registers keep what they hold; values meaningless)
    asm volatile("" : "+v"(dst) : "v"(aoff_), "s"(p));
#else
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(aoff_), "s"(p));
#endif
    p += 256;
    asm volatile("" : "+s"(p));
  };
  fc_for<0, 16>(issue);                      // prologue: k-quads 0 .. 3
  float bn[4];
#pragma unroll
  for (int u = 0; u < 4; u++) bn[u] = sB[u * 64];
  for (int q0 = 0; q0 < NKQ; q0 += 4) {
    float bc[4];
#pragma unroll
    for (int u = 0; u < 4; u++) bc[u] = bn[u];
    const float* nb = sB + (size_t)min(q0 + 4, NKQ - 4) * 64;   // next iteration's B operands (the last one re-reads its own)
#pragma unroll
    for (int u = 0; u < 4; u++) bn[u] = nb[u * 64];
    auto step = [&](auto t_c) {
      constexpr int t = decltype(t_c)::value, u = t / 4, og = t % 4;
      fc_f4& cur = rg[t]; fc_f4 (&acc_)[16] = acc; float (&bc_)[4] = bc;
      asm volatile("s_waitcnt vmcnt(15)" : "+v"(cur));
#pragma unroll
      for (int i = 0; i < 4; i++) acc_[4 * og + i] = FC_MFMA(cur[i], bc_[u], acc_[4 * og + i]);
      __builtin_amdgcn_sched_barrier(0);       // the refill reuses the registers these MFMAs have just read
      issue(t_c);
      __builtin_amdgcn_sched_barrier(0);
    };
    fc_for<0, 16>(step);
  }
  // 16 refills (past the stream's end: the next pass's weights or the buffer's slack) are still in flight: nothing may reuse their
  // registers before they have landed (every ring register is named AFTER the wait, which keeps it allocated until then)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < 16; i++) asm volatile("" : "+v"(rg[i]));
}

// pin freshly loaded values: hipcc then waits for ITS loads here, not inside the ring loops (where its conservative
// vmcnt(0) at every use would drain the ring in every iteration -- seen in the first build's stage A)
#define FC_PIN4(x_) asm volatile("" : "+v"((x_)))

template <int DUMMY>
__global__ __launch_bounds__(256, 1) void ffn_chain_kernel(FfnChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) float fc_lds[];   // [4 waves][128 k-quads][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int blk = blockIdx.x * 4 + wave;
  if (blk >= a.nblocks) return;
  const int tok = blk * 16 + n;
  const unsigned aoff = (unsigned)lane * 16u;
  float* sB = fc_lds + (size_t)wave * (128 * 64) + lane;

  // ---------------- stage A: h = W0 [x ; ctx] + b0, 32 accumulators; B operands (rows 4 kq + g of the token) from memory ----------
  fc_f4 h[32];
#pragma unroll
  for (int ob = 0; ob < 32; ob++) h[ob] = *reinterpret_cast<const fc_f4*>(a.b0 + 16 * ob + 4 * g);
#pragma unroll
  for (int ob = 0; ob < 32; ob++) FC_PIN4(h[ob]);
  {
    // stream per k-quad: 1 B load (dword) then 8 A loads; ring = 2 k-quads = 18 loads
    fc_f4 ra[2][8]; float rb[2];
    const unsigned boff = ((unsigned)g * (unsigned)a.ld + (unsigned)tok) * 4u;
    const float* xs1 = a.ctx - (size_t)256 * a.ld;            // row k >= 256 of the concatenation is xs1 + k * ld
    const float* ap = a.p0;
    auto issue_q = [&](auto par_c, int kq) {                   // all 9 loads of k-quad kq into ring half `par`
      constexpr int par = decltype(par_c)::value;
      // (named here: clang does not capture a variable of the enclosing function that a generic lambda uses only as an asm operand)
      fc_f4 (&ra_)[2][8] = ra; float (&rb_)[2] = rb; const unsigned aoff_ = aoff, boff_ = boff;
      const float* rowp = ((kq < 64) ? (const float*)a.x : xs1) + (size_t)(4 * kq) * a.ld;   // uniform
#ifdef FC_NO_LOADS
      asm volatile("" : "+v"(rb_[par]) : "v"(boff_), "s"(rowp));
#else
      asm volatile("global_load_dword %0, %1, %2" : "=v"(rb_[par]) : "v"(boff_), "s"(rowp));
#endif
      const float* p = ap + (size_t)kq * (8 * 256);
#pragma unroll
      for (int og = 0; og < 8; og++) {
        const float* pp = p + og * 256;
#ifdef FC_NO_LOADS
        asm volatile("" : "+v"(ra_[par][og]) : "v"(aoff_), "s"(pp));
#else
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ra_[par][og]) : "v"(aoff_), "s"(pp));
#endif
      }
    };
    issue_q(std::integral_constant<int, 0>(), 0);
    issue_q(std::integral_constant<int, 1>(), 1);
    auto quad = [&](auto par_c, int kq_next) {
      constexpr int par = decltype(par_c)::value;
      fc_f4 (&ra_)[2][8] = ra; float (&rb_)[2] = rb; fc_f4 (&h_)[32] = h;
      // the 9 loads of this half are the oldest 9 of 18 in flight: B first
      asm volatile("s_waitcnt vmcnt(17)" : "+v"(rb_[par]));
      auto use = [&](auto og_c) {
        constexpr int og = decltype(og_c)::value;
        fc_f4 (&ra2)[2][8] = ra_; float (&rb2)[2] = rb_; fc_f4 (&h2)[32] = h_;
        // before A load og of this half is used, the younger ones are (7 - og) of this half + the 9 of the other half
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(ra2[par][og]) : "n"(16 - og));
#pragma unroll
        for (int i = 0; i < 4; i++) h2[4 * og + i] = FC_MFMA(ra2[par][og][i], rb2[par], h2[4 * og + i]);
        __builtin_amdgcn_sched_barrier(0);     // keep every wait in front of ITS MFMAs (hipcc hoisted all eight to the top)
      };
      fc_for<0, 8>(use);
      issue_q(par_c, kq_next);
      __builtin_amdgcn_sched_barrier(0);
    };
    for (int kq = 0; kq < 128; kq += 2) {
      int k2 = min(kq + 2, 127), k3 = min(kq + 3, 127);       // past the end: re-read the last k-quad (loaded, never used)
      asm volatile("" : "+s"(k2), "+s"(k3));
      quad(std::integral_constant<int, 0>(), k2);
      quad(std::integral_constant<int, 1>(), k3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(rb[0]), "+v"(rb[1]));
#pragma unroll
    for (int og = 0; og < 8; og++) asm volatile("" : "+v"(ra[0][og]), "+v"(ra[1][og]));
  }
#ifdef FC_STOP_AFTER_A   // probe builds only: stage A alone
  { float sacc = 0.0f;
#pragma unroll
    for (int ob = 0; ob < 32; ob++) sacc += h[ob][0] + h[ob][1] + h[ob][2] + h[ob][3];
    a.x[(size_t)(4 * g) * a.ld + tok] = sacc; return; }
#endif

  // ---------------- LayerNorm(512) + GELU: a token's values live in lanes (n, 0..3); the result goes to this wave's LDS slice in
  // B-operand order (k-quad 4 b + v = register v of block b) ----------------
  {
    float s = 0.0f;
#pragma unroll
    for (int ob = 0; ob < 32; ob++) s += (h[ob][0] + h[ob][1]) + (h[ob][2] + h[ob][3]);
    s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
    const float mean = s * (1.0f / 512.0f);
    float s2 = 0.0f;
#pragma unroll
    for (int ob = 0; ob < 32; ob++)
#pragma unroll
      for (int v = 0; v < 4; v++) { const float d = h[ob][v] - mean; s2 += d * d; }
    s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
    const float rstd = 1.0f / sqrtf(s2 * (1.0f / 512.0f) + 1e-5f);
#pragma unroll
    for (int ob = 0; ob < 32; ob++) {
      const fc_f4 gm = *reinterpret_cast<const fc_f4*>(a.gamma + 16 * ob + 4 * g), bt = *reinterpret_cast<const fc_f4*>(a.beta + 16 * ob + 4 * g);
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const float yv = (h[ob][v] - mean) * rstd * gm[v] + bt[v];
#ifdef FC_SKIP_GELU   // probe builds only
        sB[(4 * ob + v) * 64] = yv;
#else
        sB[(4 * ob + v) * 64] = 0.5f * yv * (1.0f + lg_erff(yv * 0.70710678118654752440f));
#endif
      }
    }
  }

  // ---------------- stage C: x' = x + W3 g + b3 ----------------
  fc_f4 xn[16];
  {
    const float* xp = a.x + (size_t)(4 * g) * a.ld + tok;
#pragma unroll
    for (int ob = 0; ob < 16; ob++) {
      const fc_f4 bv = *reinterpret_cast<const fc_f4*>(a.b3 + 16 * ob + 4 * g);
#pragma unroll
      for (int v = 0; v < 4; v++) xn[ob][v] = bv[v] + xp[(size_t)(16 * ob + v) * a.ld];   // bias + residual open the accumulation
    }
#pragma unroll
    for (int ob = 0; ob < 16; ob++) FC_PIN4(xn[ob]);
  }
  chain_pass16<128>(xn, a.p3, aoff, sB);
  {
    float* xp = a.x + (size_t)(4 * g) * a.ld + tok;
#pragma unroll
    for (int ob = 0; ob < 16; ob++)
#pragma unroll
      for (int v = 0; v < 4; v++) { xp[(size_t)(16 * ob + v) * a.ld] = xn[ob][v]; sB[(4 * ob + v) * 64] = xn[ob][v]; }   // x' to HBM, and to LDS as stage D's B operands
  }
  if (!a.pd) return;

  // ---------------- stage D: the next block's projection, 256 outputs per pass ------------
  for (int pass = 0; pass < a.OUTD / 256; pass++) {
    fc_f4 acc[16];
    const int row0 = pass * 256;
#pragma unroll
    for (int ob = 0; ob < 16; ob++) acc[ob] = *reinterpret_cast<const fc_f4*>(a.bd + row0 + 16 * ob + 4 * g);
#pragma unroll
    for (int ob = 0; ob < 16; ob++) FC_PIN4(acc[ob]);
    // the projection is packed per pass (pack_chain_proj): [pass][k-quad][4 groups][lane][4]
    chain_pass16<64>(acc, a.pd + (size_t)pass * 256 * 256, aoff, sB);
    const bool rot = row0 < a.rot_rows;                        // rot_rows is a multiple of 256 (launcher)
    float* yp = a.y + (size_t)(row0 + 4 * g) * a.ld + tok;
    if (rot) {
      // rows (rr, rr + 1) are a rotary pair with table row (rr & 63) >> 1; this lane's pairs: (4 g + 0, + 1), (4 g + 2, + 3) of a block
      const float* cp = a.cosb + (size_t)(2 * g) * a.ld + tok; const float* sp = a.sinb + (size_t)(2 * g) * a.ld + tok;
      float cs[4][2], sn[4][2];                                // the table row depends on the block only through (16 ob) & 63
#pragma unroll
      for (int o4 = 0; o4 < 4; o4++)
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const size_t ip = (size_t)(8 * o4 + u) * a.ld;
          cs[o4][u] = cp[ip]; sn[o4][u] = sp[ip];
        }
#pragma unroll
      for (int ob = 0; ob < 16; ob++)
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const float v0 = acc[ob][2 * u], v1 = acc[ob][2 * u + 1];
          acc[ob][2 * u] = v0 * cs[ob & 3][u] - v1 * sn[ob & 3][u]; acc[ob][2 * u + 1] = v1 * cs[ob & 3][u] + v0 * sn[ob & 3][u];
        }
    }
#pragma unroll
    for (int ob = 0; ob < 16; ob++)
#pragma unroll
      for (int v = 0; v < 4; v++) yp[(size_t)(16 * ob + v) * a.ld] = acc[ob][v] * a.alpha;
  }
}

// The packed projection of stage D is consumed 4 out groups (256 outputs) per pass: pack every 256-row slice on its own so that a
// pass's stream is contiguous ([pass][k-quad][4 groups][lane][4]).
inline void pack_chain_proj(const float* w, int out, int in, float* dst) {
  for (int pass = 0; pass < out / 256; pass++) pack_chain_a(w + (size_t)pass * 256 * in, 256, in, true, dst + (size_t)pass * 256 * in);
}

constexpr int FC_LDS_BYTES = 4 * 128 * 64 * 4;   // 128 KB: one workgroup (four independent waves) per CU

inline bool launch_ffn_chain(const FfnChainArgs& a, hipStream_t s) {
  if (a.nblocks <= 0) return true;
  if (a.ld % 4 || (a.pd && (a.OUTD % 256 || a.rot_rows % 256)) || (size_t)768 * a.ld >= ((size_t)1 << 30)) return false;
  static bool attr_set_dev[64] = {};   // per device; idempotent
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  if (!attr_set_dev[dev_ & 63]) {
    if (hipFuncSetAttribute((const void*)ffn_chain_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, FC_LDS_BYTES) != hipSuccess) return false;
    attr_set_dev[dev_ & 63] = true;
  }
  hipLaunchKernelGGL(ffn_chain_kernel<0>, dim3((unsigned)((a.nblocks + 3) / 4)), dim3(256), (size_t)FC_LDS_BYTES, s, a);
  return true;
}

}  // namespace dani
