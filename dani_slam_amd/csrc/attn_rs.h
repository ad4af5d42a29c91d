// attn_rs.h -- register-streaming flash attention in f32 (included by lg_kernels.hip after the attention helpers).
//
// attn_kernel<4> stages every 32-key K / V tile through LDS (global -> registers -> ds_write -> two barriers -> one
// ds_read_b32 per MFMA operand): 26 of its 150 us per launch at 16 x 1024 keypoints are staging and barriers
// (tools/probes/attn_probe.hip).  v_mfma_f32_32x32x2_f32 takes ONE dword per lane for its A operand and holds the matrix
// pipe for 64 cycles, so a wave can fetch the key / value operands itself, 16 bytes per lane, straight from L1 / L2 into
// a small register ring that runs a few thousand cycles ahead -- no LDS tile, no barrier in the key loop, waves of a
// workgroup free to drift.  Same function, same online softmax in base 2, same tail-key / tail-query handling.
//
//   * keys are processed in super-tiles of 128, as FOUR 32-key blocks interleaved by key: block m holds the keys
//     4 i + m.  S^T = K Q^T: the A operand of d-pair p for all four blocks is ONE 16-byte load (lane (i, h): keys
//     4i .. 4i+3 of row 2p + h; a half-wave reads 512 contiguous bytes).
//   * O^T = V^T P^T: the probabilities in the four S^T accumulators are the B operands register for register (as in
//     attn_kernel); k-pair (m, r) pairs the keys 4 acc_row(r, 0) + m and 4 acc_row(r, 1) + m, so the A operand of
//     (m = 0..3, r) is ONE 16-byte load of lane (d, h) at key offset 4 acc_row(r, h); four consecutive r consume one
//     128-byte line of each of the 32 d rows completely.
//   * ring of RD x 16 bytes per lane (RD = 8 as shipped, 2 waves per SIMD; RD = 4 fits 3 waves per SIMD and is no faster),
//     refilled right behind the MFMAs that consumed a slot (inline asm loads, waits counted by hand: hipcc would rotate
//     a ring of plain loads into load / wait / use); 64 loads and 256 MFMAs per super-tile and wave.
//   * wave = 32 queries (2 waves per SIMD: one wave's softmax runs under the other's MFMAs); the four waves of a workgroup
//     are four query blocks of ONE (set, head), so K / V lines are shared in the CU's L1 while the waves stay close.
#pragma once

namespace dani {
#ifdef ARS_CLOCK
__device__ unsigned long long g_ars_clock[2];
#endif

template <int RD>   // ring depth: 16-byte operand loads in flight per wave
__global__ __launch_bounds__(256, RD <= 4 ? 3 : 2) void attn_rs_kernel(const float* Q, const float* K, const float* V, float* O, int ld,
                                                         const LgSet* sets, int tail_slots, int cross, float scale) {
  __shared__ float sTK[LG_TAIL_MAX * 128];
  __shared__ float sXch[4 * 66];
  int bx, head, bz;
  const int tail_q = attn_role(tail_slots, bx, head, bz);
  const LgSet st = sets[bz];
#ifdef ARS_SKIP_TAILQ   // probe builds only (tools/probes/attn_bench): what the launch costs without its tail-query workgroups
  if (tail_q >= 0) return;
#endif
  if (tail_q >= 0) { attn_tail_query(Q, K, V, O, ld, st, tail_q, head, cross, scale, sXch); return; }
  const int mp = (st.nmain + 31) & ~31;
  if (bx * 128 >= mp) return;
#ifdef ARS_CLOCK   // probe: core clock while the launch runs (s_memtime counts core cycles, s_memrealtime 100 MHz)
  const unsigned long long ck0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const AttnKeys ks = attn_keys(st, cross);
  const int k_n = ks.nmain;                                  // the MFMA loop covers the key set's main region
  const float* Qh = Q + (size_t)head * 64 * ld;
  const float* Kh = K + (size_t)head * 64 * ld + ks.off;
  const float* Vh = V + (size_t)head * 64 * ld + ks.off;
  const int ntail = ks.n - ks.nmain;
  // tail keys go to LDS now and are read after the key loop, behind the kernel's only barrier: no wave waits for them here
  if (ntail > 0) attn_stage_tail_keys(K + (size_t)head * 64 * ld, V + (size_t)head * 64 * ld, ld, ks.tail_off, ntail, sTK);
#ifdef ARS_STAGGER   // probe: the second workgroup of a CU starts ARS_STAGGER * 64 cycles late, so that the two waves of a SIMD
  // do not reach their softmax (a serial vector chain, no MFMAs) at the same time
  if (((blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) >> 8) & 1) __builtin_amdgcn_s_sleep(ARS_STAGGER);
#endif
#ifndef ARS_NO_PRIO
  // main waves outrank the tail-query workgroups that share their SIMDs (priority, then age, decides every issue slot): the
  // tail queries' vector instructions then take the slots the MFMA stream leaves free instead of delaying it
  __builtin_amdgcn_s_setprio(2);
#endif
  const int q0 = bx * 128 + wave * 32;
  const bool active = q0 < mp;                               // wave-uniform; an idle wave still joins the barrier below
  const int qi = q0 + j;                                     // < mp: query blocks are whole (mp is a multiple of 32)
  const int qtok = st.tok_off + (active ? qi : 0);

  float qreg[32];
  f32x16 o[2], s[4];
#pragma unroll
  for (int r = 0; r < 16; r++) { o[0][r] = 0.0f; o[1][r] = 0.0f; }
  float m = -1e30f, l = 0.0f;

  const int nst = (k_n + 127) / 128;
  // per-lane byte offsets: K loads lane (i = j, h): row 2p + h, keys 4j..; V loads lane (d = j, h): row d, keys 16 h + ...
  const unsigned koff = ((unsigned)h * (unsigned)ld + 4u * (unsigned)j) * 4u;
  const unsigned voff = ((unsigned)j * (unsigned)ld + 16u * (unsigned)h) * 4u;
  f32x4 rg[RD];
  // Issue stream of a super-tile: step t < 32 = K operand of d-pair t (rows 2t, 2t + 1), step t >= 32 = V operand of
  // (r = (t - 32) / 2, d block (t - 32) % 2).  Addresses are kept as RUNNING scalar pointers made opaque after every update:
  // left to itself hipcc hoists all 64 row bases out of the key loop (128 SGPRs, spilled into vector registers).
  const float* kptr = Kh;                 // next K row pair of the issue stream
  const float* vptr0 = Vh;                // V rows of d block 0 / 1 at the current super-tile's first key
  const float* vptr1 = Vh + (size_t)32 * ld;
  auto issue = [&](auto slot_c, auto t_c) {
    constexpr int slot = decltype(slot_c)::value, t = decltype(t_c)::value;
    // (named here, outside the `if constexpr` branches: clang does not capture variables that only a discarded branch uses)
    const float*& kp = kptr; const float* const& vptr0_ = vptr0; const float* const& vptr1_ = vptr1; const unsigned koff_ = koff, voff_ = voff; f32x4& dst = rg[slot];
#ifdef ARS_NO_LOADS   // probe builds: no operand loads at all (the registers keep what they hold; values meaningless)
    asm volatile("" : "+v"(dst));
    return;
#endif
    if constexpr (t < 32) {
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(koff_), "s"(kp));
      kp += 2 * (size_t)ld;
      asm volatile("" : "+s"(kp));
    } else {
      constexpr int r = (t - 32) >> 1, db = (t - 32) & 1;
      constexpr int imm = (4 * (r & 3) + 32 * (r >> 2)) * 4;   // byte offset of the k-pair's keys: fits the instruction's 13-bit field
      if constexpr (db == 0) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff_), "s"(vptr0_), "n"(imm));
      else asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff_), "s"(vptr1_), "n"(imm));
    }
  };
  auto softmax = [&](int kbase, auto ragged_c) {
    if constexpr (decltype(ragged_c)::value) {   // last super-tile of a key set whose main region is not a multiple of 128: mask keys >= k_n
      const int lim = k_n - kbase - 16 * h;      // key 4 acc_row(r, h) + mm of the tile is masked when 4 acc_row(r, 0) + mm >= lim
#pragma unroll
      for (int mm = 0; mm < 4; mm++)
#pragma unroll
        for (int r = 0; r < 16; r++)
          if (4 * acc_row(r, 0) + mm >= lim) s[mm][r] = -1e30f;
    }
    float mx = -1e30f;
#pragma unroll
    for (int mm = 0; mm < 4; mm++)
#pragma unroll
      for (int r = 0; r < 16; r++) mx = fmaxf(mx, s[mm][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m, mx);
    if (__ballot(m_new > m)) {
      const float alpha = __builtin_amdgcn_exp2f(m - m_new);
      l *= alpha;
#pragma unroll
      for (int r = 0; r < 16; r++) { o[0][r] *= alpha; o[1][r] *= alpha; }
    }
    float ls = 0.0f;
#pragma unroll
    for (int mm = 0; mm < 4; mm++)
#pragma unroll
      for (int r = 0; r < 16; r++) { s[mm][r] = __builtin_amdgcn_exp2f(s[mm][r] - m_new); ls += s[mm][r]; }
    l += ls;
    m = m_new;
  };
  // one step: wait for the slot (RD - 1 younger loads may stay in flight), its 4 MFMAs, then refill the slot with step t + RD
  auto step = [&](auto t_c, int stile_next, auto ragged_c) {
    constexpr int t = decltype(t_c)::value, slot = t % RD;
    const float*& kp = kptr; const float (&qr)[32] = qreg; f32x16 (&sa)[4] = s; f32x16 (&oa)[2] = o;
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(rg[slot]) : "n"(RD - 1));
    if constexpr (decltype(ragged_c)::value && t >= 32) {
      // Ragged last super-tile: the V operands of masked keys (>= k_n: the next set's columns, the tail region or the slack
      // behind T) are multiplied by P = 0 in the MFMA -- harmless only while they are finite.  A call that produced Inf / NaN
      // (NaN descriptors, say) leaves them in such columns for every later, smaller batch, and 0 * NaN would then poison the
      // last set's messages for good.  Zero them in the register (once per wave and launch: 128 selects).
      const int k_n_ = k_n; const int h_ = h;
      const int lim = (k_n_ & 127) - 16 * h_;
      constexpr int r = (t - 32) >> 1;
#pragma unroll
      for (int mm = 0; mm < 4; mm++)
        if (4 * acc_row(r, 0) + mm >= lim) rg[slot][mm] = 0.0f;
    }
    if constexpr (t < 32) {
#pragma unroll
      for (int mm = 0; mm < 4; mm++) sa[mm] = __builtin_amdgcn_mfma_f32_32x32x2f32(rg[slot][mm], qr[t], sa[mm], 0, 0, 0);
    } else {
      constexpr int r = (t - 32) >> 1, db = (t - 32) & 1;
#pragma unroll
      for (int mm = 0; mm < 4; mm++) oa[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(rg[slot][mm], sa[mm][r], oa[db], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);   // the refill below reuses the registers these MFMAs have just read
    if constexpr (t + RD == 64) {   // the issue stream enters the next super-tile (the last one re-reads its own keys: loaded, never used)
      kp = Kh + stile_next * 128;
      asm volatile("" : "+s"(kp));
    }
    issue(std::integral_constant<int, slot>(), std::integral_constant<int, (t + RD) % 64>());
    __builtin_amdgcn_sched_barrier(0);
  };
#define ARS_STEPS8(B_, sn_)                                                                                           \
  step(std::integral_constant<int, B_ + 0>(), sn_, ragged_c); step(std::integral_constant<int, B_ + 1>(), sn_, ragged_c); \
  step(std::integral_constant<int, B_ + 2>(), sn_, ragged_c); step(std::integral_constant<int, B_ + 3>(), sn_, ragged_c); \
  step(std::integral_constant<int, B_ + 4>(), sn_, ragged_c); step(std::integral_constant<int, B_ + 5>(), sn_, ragged_c); \
  step(std::integral_constant<int, B_ + 6>(), sn_, ragged_c); step(std::integral_constant<int, B_ + 7>(), sn_, ragged_c);
  if (nst > 0 && active) {
    auto first = [&](auto t_c) { if constexpr (decltype(t_c)::value < RD) issue(t_c, t_c); };
    first(std::integral_constant<int, 0>()); first(std::integral_constant<int, 1>()); first(std::integral_constant<int, 2>()); first(std::integral_constant<int, 3>());
    first(std::integral_constant<int, 4>()); first(std::integral_constant<int, 5>()); first(std::integral_constant<int, 6>()); first(std::integral_constant<int, 7>());
    first(std::integral_constant<int, 8>()); first(std::integral_constant<int, 9>()); first(std::integral_constant<int, 10>()); first(std::integral_constant<int, 11>());
    first(std::integral_constant<int, 12>()); first(std::integral_constant<int, 13>()); first(std::integral_constant<int, 14>()); first(std::integral_constant<int, 15>());
  }
  // the queries are loaded BEHIND the ring's first key operands: one memory round trip in front of the first MFMA, not two
#pragma unroll
  for (int t = 0; t < 32; t++) qreg[t] = Qh[(size_t)(2 * t + h) * ld + qtok] * (scale * 1.44269504088896341f);
  if (nst > 0 && active) {
    // one super-tile: 32 K steps, softmax, 32 V steps.  Instantiated twice: the ragged last tile of a key set carries the
    // key mask, the full tiles carry no trace of it (as one body hipcc turns the mask into 64 selects on 64 precomputed lane
    // masks -- 128 scalar registers, spilled and reloaded in every tile)
    auto tile = [&](auto ragged_c, int stile, int sn) {
      vptr0 = Vh + stile * 128; vptr1 = vptr0 + (size_t)32 * ld;
      asm volatile("" : "+s"(vptr0), "+s"(vptr1));
#pragma unroll
      for (int mm = 0; mm < 4; mm++)
#pragma unroll
        for (int r = 0; r < 16; r++) s[mm][r] = 0.0f;
      ARS_STEPS8(0, sn) ARS_STEPS8(8, sn) ARS_STEPS8(16, sn) ARS_STEPS8(24, sn)
#ifndef ARS_SKIP_SOFTMAX   // probe builds: the launch without its softmax (values meaningless)
#ifdef ARS_SOFTMAX_LOW_PRIO
      // the issue arbiter prefers the older wave of a SIMD: while THAT wave is in its softmax (~1800 cycles of back-to-back
      // vector instructions, no MFMA), the younger wave's MFMAs would wait for a gap that never comes.  At priority 0 the softmax
      // yields to them (an MFMA takes the port for 4 of its 64 cycles)
      __builtin_amdgcn_s_setprio(0);
#endif
      softmax(stile * 128, ragged_c);
#ifdef ARS_SOFTMAX_LOW_PRIO
      __builtin_amdgcn_s_setprio(2);
#endif
#endif
      ARS_STEPS8(32, sn) ARS_STEPS8(40, sn) ARS_STEPS8(48, sn) ARS_STEPS8(56, sn)
    };
    const int nfull = k_n / 128;
    for (int stile = 0; stile < nfull; stile++) {
#ifdef ARS_PRIO_PROGRESS   // probe: issue priority falls with a wave's progress, so that the younger wave of a SIMD catches up
      switch ((stile * 4) / nst) {
        case 0: __builtin_amdgcn_s_setprio(3); break;
        case 1: __builtin_amdgcn_s_setprio(2); break;
        case 2: __builtin_amdgcn_s_setprio(1); break;
        default: __builtin_amdgcn_s_setprio(0); break;
      }
#endif
      int sn = min(stile + 1, nst - 1);   // the last super-tile refills with its own addresses again (loaded, never used)
      asm volatile("" : "+s"(sn));
      tile(std::false_type(), stile, sn);
    }
    if (nfull < nst) tile(std::true_type(), nfull, nfull);
    // the refills of the last super-tile are still in flight: nothing may reuse their registers before they have landed
    if constexpr (RD == 16) {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(rg[0]), "+v"(rg[1]), "+v"(rg[2]), "+v"(rg[3]), "+v"(rg[4]), "+v"(rg[5]), "+v"(rg[6]), "+v"(rg[7]));
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(rg[8]), "+v"(rg[9]), "+v"(rg[10]), "+v"(rg[11]), "+v"(rg[12]), "+v"(rg[13]), "+v"(rg[14]), "+v"(rg[RD - 1]));
    }
    else if constexpr (RD == 8) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rg[0]), "+v"(rg[1]), "+v"(rg[2]), "+v"(rg[3]), "+v"(rg[4]), "+v"(rg[5]), "+v"(rg[6]), "+v"(rg[RD - 1]));
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(rg[0]), "+v"(rg[1]), "+v"(rg[2]), "+v"(rg[RD - 1]));
    static_assert(RD == 4 || RD == 8 || RD == 16, "ring depth");
  }
#undef ARS_STEPS8
  // everything below is re-derived from the set record instead of being kept live across the key loop (registers)
  const LgSet st2 = sets[bz];
  const AttnKeys ks2 = attn_keys(st2, cross);
  const int ntail2 = ks2.n - ks2.nmain;
  if (ntail2 > 0) {                                          // uniform per workgroup
    __syncthreads();
    if (active) attn_tail_keys(sTK, ntail2, qreg, o, m, l, h);
  }
  if (!active) return;
  const float lt = l + __shfl_xor(l, 32);
  const float inv = lt > 0.0f ? 1.0f / lt : 0.0f;            // empty key set -> zero message
  // padding queries (nmain <= qi < mp) get zeros so that every column of the set stays finite
  const int qi2 = bx * 128 + wave * 32 + j;
  const float keep = qi2 < st2.nmain ? inv : 0.0f;
  float* Oh = O + (size_t)head * 64 * ld + st2.tok_off + qi2;
#pragma unroll
  for (int db = 0; db < 2; db++)
#pragma unroll
    for (int r = 0; r < 16; r++) Oh[(size_t)(db * 32 + acc_row(r, h)) * ld] = qi2 < st2.nmain ? o[db][r] * keep : 0.0f;
#ifdef ARS_CLOCK
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
    g_ars_clock[0] = __builtin_amdgcn_s_memtime() - ck0; g_ars_clock[1] = __builtin_amdgcn_s_memrealtime() - rt0;
  }
#endif
}

}  // namespace dani
