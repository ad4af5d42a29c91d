// lg_engine.cpp -- LightGlue part of the C ABI (weights packing/folding, workspaces, forward pass, match filter).
#include <atomic>
#include "engine_internal.h"

using namespace dani;
using namespace dani::eng;

// ================================================================================================
// LightGlue
// ================================================================================================
namespace dani { extern std::atomic<int> g_lin_force_rs, g_attn_force, g_ffn_chain; }
namespace dani {
namespace eng {

constexpr int LG_LAYERS = 9;

struct LgLinear { float* w = nullptr; float* b = nullptr; void* w3 = nullptr; float* wrs = nullptr; float* pc = nullptr; float* braw = nullptr; int out = 0, in = 0; };   // w3: bf16x3 split of w (lg_x3.hip); wrs: lin_rs.h layout; pc / braw: ffn_chain.h packing and the unpadded bias
struct LgBlockW { LgLinear a, b, c, ffn0, ffn3; float* ln_g = nullptr; float* ln_b = nullptr; };  // self: a=Wqkv,b=out_proj; cross: a=to_qk,b=to_v,c=to_out
struct LgWeights {
  bool ready = false;
  float* wr = nullptr;
  LgBlockW self_[LG_LAYERS], cross_[LG_LAYERS];
  LgLinear final_proj;
  float* match_w = nullptr;
  float match_b = 0.0f;
};

constexpr int LG_LINEARS = 6 * LG_LAYERS + 1;   // per layer: Wqkv, ffn.0, ffn.3 (self) + qk|v, ffn.0, ffn.3 (cross); final_proj
struct LgContext {
  int pairs = 0, kmax = 0, ld = 0, sim_ld = 0;
  float *kpts = nullptr, *desc_tm = nullptr;  // token-major staging [ld][2], [ld][256]
  LgSet* d_sets = nullptr;
  float *x = nullptr, *qkv = nullptr, *ctx = nullptr, *msg = nullptr, *hbuf = nullptr, *qk = nullptr, *v = nullptr;
  float *cosb = nullptr, *sinb = nullptr, *md = nullptr, *logsig = nullptr, *lse_row = nullptr, *lse_col = nullptr, *max0 = nullptr;
  float *md_dense = nullptr, *logsig_dense = nullptr;   // assignment stage: sets contiguous (LgSet::dense_off)
  int *m0 = nullptr, *m1 = nullptr;
  float* sim = nullptr;
  // pinned host
  LgSet* h_sets = nullptr;
  float* h_stage = nullptr;   // kpts + desc staging
  int* h_m0 = nullptr; int* h_m1 = nullptr; float* h_max0 = nullptr;
  std::vector<void*> dev_allocs, host_allocs;
  int last_nsets = 0;
  std::vector<LgSet> last_sets;
  // profiling of the attention kernel (dominant kernel of the full front-end): events around its 18 launches
  hipEvent_t aev[36] = {};
  hipEvent_t lnev[2 * LG_LINEARS] = {};   // around every linear_fm launch of a forward pass
  bool aev_ready = false;
  float attn_ms = 0.0f, lin_ms = 0.0f;
  int lin_launches = 0;                   // linear-layer launches (or FFN-chain launches) of the last profiled forward pass
  double attn_flops = 0.0, attn_flops_alg = 0.0, lin_flops = 0.0;
};


LgWeights* lg_weights(dani_engine* e) {
  std::lock_guard<std::mutex> lk(e->mu);
  if (!e->lgw) e->lgw = new LgWeights();
  return e->lgw;
}

void lg_release(dani_engine* e) {
  std::lock_guard<std::mutex> lk(e->mu);
  for (Context* c : e->ctxs) {
    LgContext* L = c->lg;
    if (!L) continue;
    for (void* p : L->dev_allocs) (void)hipFree(p);
    for (void* p : L->host_allocs) (void)hipHostFree(p);
    if (L->aev_ready) { for (auto& ev : L->aev) (void)hipEventDestroy(ev); for (auto& ev : L->lnev) (void)hipEventDestroy(ev); }
    delete L;
    c->lg = nullptr;
  }
  delete e->lgw;   // device weights are in e->weight_allocs
  e->lgw = nullptr;
}

int lg_upload(dani_engine* e, const std::vector<float>& v, float** out) {
  HIP_TRY(e, hipMalloc((void**)out, v.size() * 4));
  e->weight_allocs.push_back(*out);
  HIP_TRY(e, hipMemcpy(*out, v.data(), v.size() * 4, hipMemcpyHostToDevice));
  return 0;
}

// pack a Linear ([out, in] row-major + bias) for linear_fm_kernel
// chain_role: how ffn_chain_kernel consumes this matrix (ffn_chain.h) -- 1: stage A, B operand from memory (natural k order);
// 2: stage C, B operand = the previous stage's accumulators; 3: stage D, the same per 256-output pass
int lg_pack_raw(dani_engine* e, const std::vector<float>& w, const std::vector<float>& b, int out, int in, LgLinear* L, int chain_role) {
  const int tiles = (out + 63) / 64;
  std::vector<float> wp((size_t)tiles * 64 * in), bp((size_t)tiles * 64);
  pack_conv_weights(w.data(), b.data(), in, out, 1, 32, 64, wp.data(), bp.data());
  L->out = out; L->in = in;
  int rc;
  if ((rc = lg_upload(e, wp, &L->w))) return rc;
  if ((rc = lg_upload(e, bp, &L->b))) return rc;
  if (out % 128 == 0 && in % 32 == 0) {   // the register-streaming form's operand layout (linear_rs_kernel)
    std::vector<float> wr((size_t)out * in);
    pack_linear_rs_host(w.data(), out, in, wr.data());
    if ((rc = lg_upload(e, wr, &L->wrs))) return rc;
  }
  if (chain_role && out % 256 == 0 && in % 16 == 0) {   // operand layout of the register-resident FFN chain
    std::vector<float> pc((size_t)out * in + LG_CHAIN_SLACK_FLOATS, 0.0f);   // the kernel's last refills read into the slack
    pack_chain_weights_host(w.data(), out, in, chain_role, pc.data());
    if ((rc = lg_upload(e, pc, &L->pc))) return rc;
    if ((rc = lg_upload(e, b, &L->braw))) return rc;
  }
  // the same weight split into three bf16 terms, for the DANI_LG_GEMM_BF16X3 option
  std::vector<uint16_t> w3(x3_packed_u16(out, in));
  pack_linear_x3(w.data(), out, in, w3.data());
  HIP_TRY(e, hipMalloc(&L->w3, w3.size() * 2));
  e->weight_allocs.push_back(L->w3);
  HIP_TRY(e, hipMemcpy(L->w3, w3.data(), w3.size() * 2, hipMemcpyHostToDevice));
  return 0;
}

int lg_get_linear(dani_engine* e, const std::string& name, int out, int in, std::vector<float>& w, std::vector<float>& b) {
  auto wi = e->tensors.find(name + ".weight"), bi = e->tensors.find(name + ".bias");
  if (wi == e->tensors.end() || bi == e->tensors.end()) return fail(e, DANI_ERR_WEIGHTS, "missing tensor %s.weight/.bias", name.c_str());
  if (wi->second.size() != (size_t)out * in || bi->second.size() != (size_t)out)
    return fail(e, DANI_ERR_WEIGHTS, "tensor %s.weight has %zu floats, expected [%d,%d]", name.c_str(), wi->second.size(), out, in);
  w = wi->second; b = bi->second;
  return 0;
}

int lg_pack_linear(dani_engine* e, const std::string& name, int out, int in, LgLinear* L, int chain_role, const std::vector<int>* perm = nullptr) {
  std::vector<float> w, b;
  int rc = lg_get_linear(e, name, out, in, w, b);
  if (rc) return rc;
  if (perm) {
    std::vector<float> w2(w.size()), b2(b.size());
    for (int r = 0; r < out; r++) {
      memcpy(&w2[(size_t)r * in], &w[(size_t)(*perm)[r] * in], (size_t)in * 4);
      b2[r] = b[(*perm)[r]];
    }
    w.swap(w2); b.swap(b2);
  }
  return lg_pack_raw(e, w, b, out, in, L, chain_role);
}

// ffn.0 applied to cat[x, P(ctx)] with P an affine map (out_proj / to_out) folded in:
//   W0 [x ; Wp ctx + bp] + b0 = [W0a | W0b Wp] [x ; ctx] + (b0 + W0b bp)        (products accumulated in double)
// This removes one 256x256 GEMM launch per block (18 per forward) without changing the function computed.
int lg_pack_ffn0_folded(dani_engine* e, const std::string& ffn0, const std::string& proj, LgLinear* L) {
  std::vector<float> w0, b0, wp, bp;
  int rc;
  if ((rc = lg_get_linear(e, ffn0, 512, 512, w0, b0))) return rc;
  if ((rc = lg_get_linear(e, proj, 256, 256, wp, bp))) return rc;
  std::vector<float> w(w0), b(b0);
  for (int o = 0; o < 512; o++) {
    const float* w0b = &w0[(size_t)o * 512 + 256];
    for (int k = 0; k < 256; k++) {
      double acc = 0.0;
      for (int m = 0; m < 256; m++) acc += (double)w0b[m] * (double)wp[(size_t)m * 256 + k];
      w[(size_t)o * 512 + 256 + k] = (float)acc;
    }
    double bb = (double)b0[o];
    for (int m = 0; m < 256; m++) bb += (double)w0b[m] * (double)bp[m];
    b[o] = (float)bb;
  }
  return lg_pack_raw(e, w, b, 512, 512, L, 1);
}

// to_qk (pre-scaled by 64^-1/4, the factor the public CrossBlock applies to both sides) and to_v as one 512-row GEMM
int lg_pack_qkv_cross(dani_engine* e, const std::string& pc, LgLinear* L) {
  std::vector<float> wq, bq, wv, bv;
  int rc;
  if ((rc = lg_get_linear(e, pc + ".to_qk", 256, 256, wq, bq))) return rc;
  if ((rc = lg_get_linear(e, pc + ".to_v", 256, 256, wv, bv))) return rc;
  const float sc = 0.35355339059327373f;
  std::vector<float> w((size_t)512 * 256), b(512);
  for (size_t i = 0; i < wq.size(); i++) w[i] = wq[i] * sc;
  for (int i = 0; i < 256; i++) b[i] = bq[i] * sc;
  memcpy(&w[(size_t)256 * 256], wv.data(), wv.size() * 4);
  memcpy(&b[256], bv.data(), 256 * 4);
  return lg_pack_raw(e, w, b, 512, 256, L, 3);
}

int lg_vec(dani_engine* e, const std::string& name, size_t n, float** out) {
  auto it = e->tensors.find(name);
  if (it == e->tensors.end() || it->second.size() != n) return fail(e, DANI_ERR_WEIGHTS, "missing or mis-shaped tensor %s", name.c_str());
  return lg_upload(e, it->second, out);
}

int lg_finalize(dani_engine* e) {
  LgWeights* W = lg_weights(e);
  W->ready = false;
  if (e->tensors.find("posenc.Wr.weight") == e->tensors.end()) return 0;  // LightGlue weights not supplied: SP-only engine
  int rc;
  if ((rc = lg_vec(e, "posenc.Wr.weight", 64, &W->wr))) return rc;
  // Wqkv rows: reference order h*192 + d*3 + {q,k,v}  ->  kernel order {q,k,v}*256 + h*64 + d
  std::vector<int> perm(768);
  for (int which = 0; which < 3; which++)
    for (int h = 0; h < 4; h++)
      for (int d = 0; d < 64; d++) perm[which * 256 + h * 64 + d] = h * 192 + d * 3 + which;
  for (int i = 0; i < LG_LAYERS; i++) {
    const std::string ps = "transformers." + std::to_string(i) + ".self_attn", pc = "transformers." + std::to_string(i) + ".cross_attn";
    LgBlockW& S = W->self_[i]; LgBlockW& C = W->cross_[i];
    if ((rc = lg_pack_linear(e, ps + ".Wqkv", 768, 256, &S.a, 3, &perm))) return rc;
    if ((rc = lg_pack_ffn0_folded(e, ps + ".ffn.0", ps + ".out_proj", &S.ffn0))) return rc;
    if ((rc = lg_vec(e, ps + ".ffn.1.weight", 512, &S.ln_g))) return rc;
    if ((rc = lg_vec(e, ps + ".ffn.1.bias", 512, &S.ln_b))) return rc;
    if ((rc = lg_pack_linear(e, ps + ".ffn.3", 256, 512, &S.ffn3, 2))) return rc;
    if ((rc = lg_pack_qkv_cross(e, pc, &C.a))) return rc;
    if ((rc = lg_pack_ffn0_folded(e, pc + ".ffn.0", pc + ".to_out", &C.ffn0))) return rc;
    if ((rc = lg_vec(e, pc + ".ffn.1.weight", 512, &C.ln_g))) return rc;
    if ((rc = lg_vec(e, pc + ".ffn.1.bias", 512, &C.ln_b))) return rc;
    if ((rc = lg_pack_linear(e, pc + ".ffn.3", 256, 512, &C.ffn3, 2))) return rc;
  }
  const std::string pa = "log_assignment." + std::to_string(LG_LAYERS - 1);
  if ((rc = lg_pack_linear(e, pa + ".final_proj", 256, 256, &W->final_proj, 3))) return rc;
  if ((rc = lg_vec(e, pa + ".matchability.weight", 256, &W->match_w))) return rc;
  auto mb = e->tensors.find(pa + ".matchability.bias");
  if (mb == e->tensors.end() || mb->second.size() != 1) return fail(e, DANI_ERR_WEIGHTS, "missing %s.matchability.bias", pa.c_str());
  W->match_b = mb->second[0];
  W->ready = true;
  return 0;
}

int lg_dev(dani_engine* e, LgContext* L, void** p, size_t bytes) {
  HIP_TRY(e, hipMalloc(p, bytes));
  L->dev_allocs.push_back(*p);
  HIP_TRY(e, hipMemset(*p, 0, bytes));
  return 0;
}
int lg_host(dani_engine* e, LgContext* L, void** p, size_t bytes) {
  HIP_TRY(e, hipHostMalloc(p, bytes, hipHostMallocDefault));
  L->host_allocs.push_back(*p);
  return 0;
}

int lg_ensure(dani_engine* e, Context* c, LgContext** out, int pairs, int kmax) {
  LgContext* L;
  {
    std::lock_guard<std::mutex> lk(e->mu);
    if (!c->lg) c->lg = new LgContext();
    L = c->lg;
  }
  *out = L;
  if (L->pairs >= pairs && L->kmax >= kmax) return 0;
  for (void* p : L->dev_allocs) (void)hipFree(p);
  for (void* p : L->host_allocs) (void)hipHostFree(p);
  L->dev_allocs.clear(); L->host_allocs.clear();
  pairs = std::max(pairs, L->pairs); kmax = std::max(kmax, L->kmax);
  const int np = round_up(kmax, 32);
  const int ld = round_up(pairs * 2 * np + round_up(pairs * 2 * LG_TAIL_MAX, 32) + 128, 256);   // main regions + shared tail region
  if ((size_t)768 * ld >= ((size_t)1 << 30)) return fail(e, DANI_ERR_INVALID, "LightGlue batch too large: %d token columns", ld);
  int rc;
#define LGA(ptr, rows) if ((rc = lg_dev(e, L, (void**)&(ptr), (size_t)(rows) * ld * 4))) return rc;
  LGA(L->x, 256) LGA(L->qkv, 768) LGA(L->ctx, 256) LGA(L->msg, 256) LGA(L->hbuf, 512) LGA(L->qk, 256) LGA(L->v, 256)
  LGA(L->cosb, 32) LGA(L->sinb, 32) LGA(L->md, 256) LGA(L->logsig, 1) LGA(L->lse_row, 1) LGA(L->lse_col, 1) LGA(L->md_dense, 256) LGA(L->logsig_dense, 1)
  LGA(L->m0, 3) LGA(L->kpts, 2) LGA(L->desc_tm, 256)   // m0 | m1 | max0 are rows of one block: one read-back
  L->m1 = L->m0 + ld; L->max0 = reinterpret_cast<float*>(L->m0 + 2 * (size_t)ld);
#undef LGA
  L->sim_ld = round_up(np + 128, 64);
  if ((rc = lg_dev(e, L, (void**)&L->sim, (size_t)pairs * (np + 64) * L->sim_ld * 4))) return rc;
  if ((rc = lg_dev(e, L, (void**)&L->d_sets, sizeof(LgSet) * 2 * pairs))) return rc;
  if ((rc = lg_host(e, L, (void**)&L->h_sets, sizeof(LgSet) * 2 * pairs))) return rc;
  if ((rc = lg_host(e, L, (void**)&L->h_stage, (size_t)ld * 258 * 4))) return rc;
  if ((rc = lg_host(e, L, (void**)&L->h_m0, (size_t)ld * 12))) return rc;
  L->h_m1 = L->h_m0 + ld; L->h_max0 = reinterpret_cast<float*>(L->h_m0 + 2 * (size_t)ld);
  HIP_TRY(e, hipDeviceSynchronize());   // null-stream zero fills vs the engine's non-blocking streams
  L->pairs = pairs; L->kmax = kmax; L->ld = ld;
  return 0;
}

// the 9-layer forward + assignment statistics on stream s; the sets in L->d_sets point at their token-major inputs
struct LgShape { int nsets, max_n, max_mp, max_span, max_tail, T, T_main; };   // T = all main regions (T_main) + the tail region

// false: a linear layer refused its shape (nothing of that layer was launched)
bool lg_forward(const LgWeights* W, LgContext* L, const LgShape& sh, hipStream_t s, bool prof, int gemm) {
  const int ld = L->ld, nsets = sh.nsets, max_n = sh.max_n, T = sh.T, max_np = round_up(max_n, 32);
  if (prof && !L->aev_ready) {
    for (auto& ev : L->aev) (void)hipEventCreate(&ev);
    for (auto& ev : L->lnev) (void)hipEventCreate(&ev);
    L->aev_ready = true;
  }
  int ai = 0, li = 0;
  auto attn = [&](const float* Q, const float* K, const float* V, int cross, float scale) {
    if (prof) (void)hipEventRecord(L->aev[ai++], s);
    launch_attention(Q, K, V, L->ctx, ld, L->d_sets, nsets, sh.max_mp, sh.max_tail, cross, scale, s, gemm == DANI_LG_GEMM_BF16X3_ALL);
    if (prof) (void)hipEventRecord(L->aev[ai++], s);
  };
  launch_desc_to_fm(L->d_sets, nsets, sh.max_span, L->x, ld, s);
  launch_posenc(L->d_sets, nsets, sh.max_span, W->wr, L->cosb, L->sinb, ld, s);
  bool ok = true;
  auto lin = [&](const LgLinear& l, const float* x0, int K0, const float* x1, int K1, float* y, const float* res, int rot, float alpha) {
    if (prof) (void)hipEventRecord(L->lnev[li++], s);
    if (gemm != DANI_LG_GEMM_F32) ok = launch_linear_x3(x0, K0, x1, K1, l.w3, l.b, y, l.out, T, ld, res, L->cosb, L->sinb, rot, alpha, s) && ok;
    else ok = launch_linear_fm(x0, K0, x1, K1, l.w, l.b, y, l.out, T, ld, res, L->cosb, L->sinb, rot, alpha, s, l.wrs) && ok;
    if (prof) (void)hipEventRecord(L->lnev[li++], s);
  };
  // The token-local chain between two attention launches -- ffn.0 -> LayerNorm + GELU -> ffn.3 (+ residual) -> the NEXT block's
  // projection -- as ONE register-resident launch per block (ffn_chain.h) where that pays.  Measured on MI355X (tools/probes/
  // ffn_chain_probe.hip, profiles/r04_ffn_chain_probe.txt): one round of 1024 blocks (one wave per SIMD) takes ~160 us against ~181 us
  // for the four launches over the same 16 k tokens -- but the tokens behind the whole rounds (the shared tail region, 32-64 columns)
  // keep the four small launches, ~33 us in the pipeline, so ONE round (the default bench's 8 pairs) does not pay (10.99 against
  // 10.72 ms per step, profiles/r04_null_results.md) and several do: 64 frames per call = 8 rounds.
  const int nb_all = T / 16, nb_main = sh.T_main / 16;
  int chain_blocks = 0;
  {
    static const int env_chain = getenv("DANI_FFN_CHAIN") ? atoi(getenv("DANI_FFN_CHAIN")) : -1;
    const int force = g_ffn_chain.load(std::memory_order_relaxed) >= 0 ? g_ffn_chain.load(std::memory_order_relaxed) : env_chain;
    const bool have = W->self_[0].ffn0.pc && W->self_[0].ffn3.pc && W->cross_[0].a.pc && W->self_[0].a.pc && W->final_proj.pc;
    auto rounds = [](int nb) { return (nb + 1023) / 1024; };
    const bool all_in = rounds(nb_all) == rounds(nb_main);      // the tail region fits the last round: no launches of its own
    // (re-measured on the final code: 16 frames per call = 2 rounds + tail is 0.05 ms per step SLOWER with the chain, 32 frames =
    //  4 rounds + tail 0.4 ms faster, 64 frames 1.3 ms faster: the tail's price and the margin below are set to that)
    const double chain_us = rounds(all_in ? nb_all : nb_main) * 160.0 + (all_in ? 0.0 : 40.0), four_us = (double)T / 16416.0 * 181.0;
    const bool pays = nb_main >= 768 && chain_us < four_us - 10.0;
    if (gemm == DANI_LG_GEMM_F32 && have && force != 0 && (force == 1 || pays)) chain_blocks = all_in ? nb_all : nb_main;
  }
  const int T_rest = T - 16 * chain_blocks;                 // tokens of the tail region that keep the four-launch path
  // one FFN block + the projection that follows it; `next` = nullptr: none (never happens: the last block feeds final_proj)
  auto ffn_and_next = [&](const LgBlockW& Bk, const LgLinear& next, float* y, int rot, float alpha) {
    if (chain_blocks > 0) {
      if (prof) (void)hipEventRecord(L->lnev[li++], s);
      ok = launch_ffn_chain_fm(L->x, L->ctx, Bk.ffn0.pc, Bk.ffn0.braw, Bk.ln_g, Bk.ln_b, Bk.ffn3.pc, Bk.ffn3.braw, next.pc, next.braw, next.out, y,
                               L->cosb, L->sinb, rot, alpha, chain_blocks, ld, s) && ok;
      if (T_rest > 0) {
        const size_t o = (size_t)16 * chain_blocks;
        ok = launch_linear_fm(L->x + o, 256, L->ctx + o, 256, Bk.ffn0.w, Bk.ffn0.b, L->hbuf + o, 512, T_rest, ld, nullptr, nullptr, nullptr, 0, 1.0f, s, Bk.ffn0.wrs) && ok;
        launch_ln_gelu(L->hbuf + o, Bk.ln_g, Bk.ln_b, T_rest, ld, s);
        ok = launch_linear_fm(L->hbuf + o, 512, nullptr, 0, Bk.ffn3.w, Bk.ffn3.b, L->x + o, 256, T_rest, ld, L->x + o, nullptr, nullptr, 0, 1.0f, s, Bk.ffn3.wrs) && ok;
        ok = launch_linear_fm(L->x + o, 256, nullptr, 0, next.w, next.b, y + o, next.out, T_rest, ld, nullptr, L->cosb + o, L->sinb + o, rot, alpha, s, next.wrs) && ok;
      }
      if (prof) (void)hipEventRecord(L->lnev[li++], s);
      return;
    }
    lin(Bk.ffn0, L->x, 256, L->ctx, 256, L->hbuf, nullptr, 0, 1.0f);
    launch_ln_gelu(L->hbuf, Bk.ln_g, Bk.ln_b, T, ld, s);
    lin(Bk.ffn3, L->hbuf, 512, nullptr, 0, L->x, L->x, 0, 1.0f);
    lin(next, L->x, 256, nullptr, 0, y, nullptr, rot, alpha);
  };
  // self block: Wqkv (+rotary) -> attention -> [out_proj folded into] ffn.0 -> LN+GELU -> ffn.3 + residual;
  // cross block: [to_qk*s | to_v] in one GEMM -> attention over the partner set -> [to_out folded into] ffn.0 ...
  lin(W->self_[0].a, L->x, 256, nullptr, 0, L->qkv, nullptr, 512, 1.0f);
  for (int i = 0; i < LG_LAYERS; i++) {
    const LgBlockW& S = W->self_[i]; const LgBlockW& C = W->cross_[i];
    attn(L->qkv, L->qkv + (size_t)256 * ld, L->qkv + (size_t)512 * ld, 0, 0.125f);
    ffn_and_next(S, C.a, L->qkv, 0, 1.0f);
    attn(L->qkv, L->qkv, L->qkv + (size_t)256 * ld, 1, 1.0f);
    if (i + 1 < LG_LAYERS) ffn_and_next(C, W->self_[i + 1].a, L->qkv, 512, 1.0f);
    else { near_mark(s); ffn_and_next(C, W->final_proj, L->md, 0, 0.25f); }   // / d^0.25, d = 256; the caller's final wait is one FFN + the match head away
  }
  L->lin_launches = li / 2;
  const size_t pstride = (size_t)(round_up(L->kmax, 32) + 64) * L->sim_ld;
  launch_match_densify(L->x, W->match_w, W->match_b, L->md, L->d_sets, nsets, max_n, L->md_dense, L->logsig_dense, ld, s);
  launch_sim(L->md_dense, ld, L->d_sets, nsets / 2, max_np, L->sim, L->sim_ld, pstride, s);
  launch_assignment_stats(L->sim, L->sim_ld, pstride, L->d_sets, nsets / 2, max_n, L->logsig_dense, L->lse_row, L->lse_col,
                          L->m0, L->max0, L->m1, ld, s);
  return ok;
}

thread_local LgContext* t_last_lg = nullptr;
LgContext* last_lg(dani_engine* e) {
  if (!e || !t_last_lg) return nullptr;
  std::lock_guard<std::mutex> lk(e->mu);
  for (Context* c : e->ctxs) if (c->lg && c->lg == t_last_lg) return t_last_lg;
  return nullptr;
}

}  // namespace eng
}  // namespace dani

extern "C" {

int dani_lg_set_kernel_option(const char* name, int32_t value) {
  if (!name) return DANI_ERR_INVALID;
  if (!strcmp(name, "lin_rs")) { if (value < -1 || value > 4) return DANI_ERR_INVALID; dani::g_lin_force_rs = value; return DANI_OK; }
  if (!strcmp(name, "ffn_chain")) { if (value < -1 || value > 1) return DANI_ERR_INVALID; dani::g_ffn_chain = value; return DANI_OK; }
  if (!strcmp(name, "attn")) {   // the documented kernels only: 1 / 3 / 4 / 5 (attn_ks, the bf16x3 forms on the f32 path, attn_k4 at any size) stay probe-only
    if (value != 0 && value != 2 && value != 6 && value != 7 && value != 8) return DANI_ERR_INVALID;
    dani::g_attn_force = value; return DANI_OK;
  }
  return DANI_ERR_INVALID;
}

int dani_lg_set_gemm(dani_engine* e, int32_t mode) {
  if (!e) return DANI_ERR_INVALID;
  if (mode != DANI_LG_GEMM_F32 && mode != DANI_LG_GEMM_BF16X3 && mode != DANI_LG_GEMM_BF16X3_ALL) return fail(e, DANI_ERR_INVALID, "unknown LightGlue GEMM mode %d", mode);
  e->lg_gemm = mode;
  return DANI_OK;
}

}  // extern "C"

namespace dani {
namespace eng {

int lg_match_impl(dani_engine* e, const LgCall& call, int32_t P, int32_t in_cap, int32_t on_device, const float* kpts0,
                  const float* desc0, const int32_t* n0, const float* kpts1, const float* desc1, const int32_t* n1,
                  const float* size0, const float* size1, float threshold, int32_t* matches, float* scores,
                  int32_t cap, int32_t* k_out) {
  if (!e) return DANI_ERR_INVALID;
  if (P < 1 || !n0 || !n1 || !kpts0 || !kpts1 || !desc0 || !desc1 || !matches || !scores || !k_out || cap < 1 || in_cap < 1)
    return fail(e, DANI_ERR_INVALID, "bad dani_lg_match argument");
  LgWeights* W = lg_weights(e);
  if (!e->finalized || !W->ready) return fail(e, DANI_ERR_WEIGHTS, "LightGlue weights not loaded/finalized");
  int kmax = 1;
  for (int p = 0; p < P; p++) {
    if (n0[p] < 0 || n1[p] < 0 || n0[p] > in_cap || n1[p] > in_cap) return fail(e, DANI_ERR_INVALID, "pair %d: bad keypoint count", p);
    kmax = std::max(kmax, std::max(n0[p], n1[p]));
  }
  HIP_TRY(e, hipSetDevice(e->device));
  CtxLease lease(e, call.ctx);
  Context* c = lease.c;
  if (!c) return fail(e, DANI_ERR_INVALID, "all contexts busy");
  LgContext* L = nullptr;
  int rc = lg_ensure(e, c, &L, P, kmax);
  if (rc) return rc;
  t_last_lg = L;
  hipStream_t s = c->stream;
  // token layout (common.h, LgSet): set 2p = side 0 of pair p, set 2p+1 = side 1.  Main regions first, each padded to a
  // multiple of 32 tokens; then the shared tail region; the dense layout of the assignment stage pads every set to 32.
  int off = 0, doff = 0, ntail = 0, max_n = 0;
  LgShape sh{2 * P, 0, 0, 0, 0, 0, 0};
  L->last_sets.assign(2 * P, LgSet());
  auto place = [&](LgSet& a, int n) {
    const int t = n % 32;
    a.n = n; a.nmain = (n >= 32 && t > 0 && t <= LG_TAIL_MAX) ? n - t : n;
    a.tok_off = off; off += round_up(std::max(a.nmain, 1), 32);
    a.dense_off = doff; doff += round_up(std::max(n, 1), 32);
    a.tail_off = ntail; ntail += n - a.nmain;            // relative to the tail region for now
    max_n = std::max(max_n, n);
    sh.max_mp = std::max(sh.max_mp, round_up(a.nmain, 32));
    sh.max_span = std::max(sh.max_span, round_up(a.nmain, 32) + (n - a.nmain));
    sh.max_tail = std::max(sh.max_tail, n - a.nmain);
  };
  for (int p = 0; p < P; p++) { place(L->last_sets[2 * p], n0[p]); place(L->last_sets[2 * p + 1], n1[p]); }
  const int T_main = off;
  for (LgSet& a : L->last_sets) a.tail_off += T_main;
  for (int p = 0; p < P; p++) {
    LgSet& a = L->last_sets[2 * p]; LgSet& b = L->last_sets[2 * p + 1];
    a.partner_off = b.tok_off; a.partner_n = b.n; a.partner_nmain = b.nmain; a.partner_tail_off = b.tail_off;
    b.partner_off = a.tok_off; b.partner_n = a.n; b.partner_nmain = a.nmain; b.partner_tail_off = a.tail_off;
  }
  const int T = T_main + round_up(ntail, 32);
  sh.max_n = max_n; sh.T = T; sh.T_main = T_main;
  // sizes: explicit (w,h) or 1 + max - min of the keypoints (needs host keypoints)
  std::vector<float> hk;
  auto host_kpts = [&](const float* k, int p, int n) -> const float* {
    if (!on_device) return k + (size_t)p * in_cap * 2;
    hk.resize((size_t)n * 2);
    (void)hipMemcpy(hk.data(), k + (size_t)p * in_cap * 2, (size_t)n * 8, hipMemcpyDeviceToHost);
    return hk.data();
  };
  for (int p = 0; p < P; p++)
    for (int side = 0; side < 2; side++) {
      LgSet& st = L->last_sets[2 * p + side];
      const float* sz = side ? size1 : size0;
      if (sz) { st.size_w = sz[0]; st.size_h = sz[1]; }
      else {
        const float* k = host_kpts(side ? kpts1 : kpts0, p, st.n);
        float mnx = 3e38f, mny = 3e38f, mxx = -3e38f, mxy = -3e38f;
        for (int i = 0; i < st.n; i++) { mnx = std::min(mnx, k[2 * i]); mxx = std::max(mxx, k[2 * i]); mny = std::min(mny, k[2 * i + 1]); mxy = std::max(mxy, k[2 * i + 1]); }
        st.size_w = st.n ? 1.0f + (mxx - mnx) : 1.0f; st.size_h = st.n ? 1.0f + (mxy - mny) : 1.0f;
      }
    }
  // inputs: device-resident sets are read in place; host sets go through ONE pinned staging block and two copies
  for (int p = 0; p < P; p++)
    for (int side = 0; side < 2; side++) {
      LgSet& st = L->last_sets[2 * p + side];
      const float* k = (side ? kpts1 : kpts0) + (size_t)p * in_cap * 2;
      const float* d = (side ? desc1 : desc0) + (size_t)p * in_cap * 256;
      if (on_device) { st.kp = k; st.desc = d; continue; }
      st.kp = L->kpts + (size_t)st.dense_off * 2; st.desc = L->desc_tm + (size_t)st.dense_off * 256;
      memcpy(L->h_stage + (size_t)st.dense_off * 2, k, (size_t)st.n * 8);
      memcpy(L->h_stage + (size_t)L->ld * 2 + (size_t)st.dense_off * 256, d, (size_t)st.n * 1024);
    }
  if (!on_device && doff > 0) {
    HIP_TRY(e, hipMemcpyAsync(L->kpts, L->h_stage, (size_t)doff * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(e, hipMemcpyAsync(L->desc_tm, L->h_stage + (size_t)L->ld * 2, (size_t)doff * 1024, hipMemcpyHostToDevice, s));
  }
  // padding columns of the tail region: keep the residual stream finite there.  (They and the columns behind T CAN be read as
  // key / value operands by attn_rs_kernel's ragged last super-tile, which loads whole 128-key tiles: their scores are masked
  // and their V operands zeroed in the register -- attn_rs.h -- so nothing depends on what a previous, larger call left there.)
  if (T > T_main + ntail)
    HIP_TRY(e, hipMemset2DAsync(L->x + T_main + ntail, (size_t)L->ld * 4, 0, (size_t)(T - T_main - ntail) * 4, 256, s));
  memcpy(L->h_sets, L->last_sets.data(), sizeof(LgSet) * 2 * P);
  HIP_TRY(e, hipMemcpyAsync(L->d_sets, L->h_sets, sizeof(LgSet) * 2 * P, hipMemcpyHostToDevice, s));
  HIP_TRY(e, hipEventRecord(c->ev_lg0, s));
  const bool prof = e->profiling;
  if (max_n > 0 && !lg_forward(W, L, sh, s, prof, e->lg_gemm)) {
    (void)hipStreamSynchronize(s);
    return fail(e, DANI_ERR_UNSUPPORTED, "LightGlue batch of %d token columns exceeds the linear kernels' addressing range", L->ld);
  }
  HIP_TRY(e, hipEventRecord(c->ev_lg1, s));
  HIP_TRY(e, hipGetLastError());
  trace_mark("lg_enqueued");
  // m0 | m1 | max0 are three rows of one block: ONE linear copy up to the end of the third row's valid part (the strided 2-D form
  // took a slower path in the runtime and sits at the very end of the call)
  HIP_TRY(e, hipMemcpyAsync(L->h_m0, L->m0, ((size_t)2 * L->ld + (size_t)doff) * 4, hipMemcpyDeviceToHost, s));
  if (call.overlap) (*call.overlap)();   // host work of the caller that can run under the forward pass
  trace_mark("desc_moved");
  HIP_TRY(e, hipStreamSynchronize(s));
  trace_mark("lg_synced");
  (void)hipEventElapsedTime(&c->ms[0], c->ev_lg0, c->ev_lg1);
  c->ms[4] = c->ms[0];
  t_last_ctx = c;
  if (prof && max_n > 0) {
    float tot = 0.0f;
    for (int i = 0; i < 36; i += 2) { float m = 0.0f; (void)hipEventElapsedTime(&m, L->aev[i], L->aev[i + 1]); tot += m; }
    L->attn_ms = tot;
    // EXECUTED flop, valid keypoints only: per launch and set 4 heads x (QK^T + PV) = 2 * 2 * nq * nk * 256.  A cross launch forms
    // K Q^T once per DIRECTION (S and S^T of a pair are each computed), so the executed count of a pair's cross layer is 4 N M d MAC.
    // ALGORITHMIC: the cross block needs S once, 3 N M d MAC per pair = 3/4 of that (DESIGN.md section 5.2:
    // why the flash form computes it twice).  Both figures are reported; roofline fractions quote the algorithmic one.
    double fl = 0.0, fla = 0.0;
    for (const LgSet& st : L->last_sets) {
      fl += 9.0 * 4.0 * 256.0 * st.n * ((double)st.n + (double)st.partner_n);
      fla += 9.0 * 256.0 * st.n * (4.0 * (double)st.n + 3.0 * (double)st.partner_n);
    }
    L->attn_flops = fl; L->attn_flops_alg = fla;
    tot = 0.0f;
    for (int i = 0; i < 2 * L->lin_launches && i < 2 * LG_LINEARS; i += 2) { float m = 0.0f; (void)hipEventElapsedTime(&m, L->lnev[i], L->lnev[i + 1]); tot += m; }
    L->lin_ms = tot;
    double ntok = 0.0;   // algorithmic: valid tokens only; per token 2 * out * in flop per layer
    for (const LgSet& st : L->last_sets) ntok += st.n;
    L->lin_flops = ntok * 2.0 * (LG_LAYERS * (768.0 * 256 + 512.0 * 256 + 2 * (512.0 * 512 + 256.0 * 512)) + 256.0 * 256);
  }
  // mutual check + thresholds (filter_matches of the model at 0.1, then LightGlue::filterMatches, src/LightGlue.cpp:98-122)
  int status = DANI_OK;
  for (int p = 0; p < P; p++) {
    const LgSet& a = L->last_sets[2 * p]; const LgSet& b = L->last_sets[2 * p + 1];
    int k = 0;
    const size_t ostride = call.out_stride > 0 ? (size_t)call.out_stride : (size_t)cap;
    int32_t* mo = matches + (size_t)p * ostride * 2;
    float* so = scores + (size_t)p * ostride;
    if (a.n > 0 && b.n > 0)
      for (int i = 0; i < a.n; i++) {
        const int jj = L->h_m0[a.dense_off + i];
        if (jj < 0 || jj >= b.n || L->h_m1[b.dense_off + jj] != i) continue;
        const float ms = std::exp(L->h_max0[a.dense_off + i]);
        if (!(ms > 0.1f) || !(ms > threshold)) continue;
        if (k < cap) { mo[2 * k] = i; mo[2 * k + 1] = jj; so[k] = ms; }
        k++;
      }
    k_out[p] = k;
    if (k > cap) status = DANI_ERR_CAPACITY;
  }
  if (status != DANI_OK) return fail(e, status, "match capacity %d too small", cap);
  return DANI_OK;
}

}  // namespace eng
}  // namespace dani

extern "C" {

int dani_lg_match_batch(dani_engine* e, int32_t P, int32_t in_cap, int32_t on_device, const float* kpts0,
                        const float* desc0, const int32_t* n0, const float* kpts1, const float* desc1, const int32_t* n1,
                        const float* size0, const float* size1, float threshold, int32_t* matches, float* scores,
                        int32_t cap, int32_t* k_out) {
  return lg_match_impl(e, LgCall{}, P, in_cap, on_device, kpts0, desc0, n0, kpts1, desc1, n1, size0, size1, threshold, matches, scores,
                       cap, k_out);
}

int dani_lg_match(dani_engine* e, const float* kpts0, const float* desc0, int32_t n0, const float* kpts1,
                  const float* desc1, int32_t n1, const float* size0, const float* size1, float threshold,
                  int32_t* matches, float* scores, int32_t cap, int32_t* k_out) {
  if (n0 == 0 || n1 == 0) { if (k_out) *k_out = 0; return e ? DANI_OK : DANI_ERR_INVALID; }
  const int in_cap = std::max(n0, n1);
  return dani_lg_match_batch(e, 1, in_cap, 0, kpts0, desc0, &n0, kpts1, desc1, &n1, size0, size1, threshold, matches,
                             scores, cap, k_out);
}

int dani_lg_match_groups(dani_engine* e, const float* kpts0, const float* desc0, int32_t n0, const float* kpts1,
                         const float* desc1, int32_t n1, int32_t G, const int32_t* off0, const int32_t* idx0, const int32_t* off1,
                         const int32_t* idx1, const float* size0, const float* size1, float threshold, int32_t* matches,
                         float* scores, int32_t* match_offsets, int32_t cap) {
  if (!e) return DANI_ERR_INVALID;
  if (G < 0 || n0 < 0 || n1 < 0 || !match_offsets || (G && (!off0 || !off1)) || cap < 0 || (cap && (!matches || !scores)))
    return fail(e, DANI_ERR_INVALID, "bad dani_lg_match_groups argument");
  match_offsets[0] = 0;
  if (G == 0) return DANI_OK;
  if (off0[0] != 0 || off1[0] != 0) return fail(e, DANI_ERR_INVALID, "group offsets must start at 0");
  int maxn = 1;
  for (int g = 0; g < G; g++) {
    const int a = off0[g + 1] - off0[g], b = off1[g + 1] - off1[g];
    if (a < 0 || b < 0) return fail(e, DANI_ERR_INVALID, "group %d: offsets not monotone", g);
    maxn = std::max(maxn, std::max(a, b));
  }
  const int t0 = off0[G], t1 = off1[G];
  if ((t0 && (!idx0 || !kpts0 || !desc0)) || (t1 && (!idx1 || !kpts1 || !desc1))) return fail(e, DANI_ERR_INVALID, "null feature arrays");
  for (int i = 0; i < t0; i++) if (idx0[i] < 0 || idx0[i] >= n0) return fail(e, DANI_ERR_INVALID, "idx0[%d] = %d outside [0, %d)", i, idx0[i], n0);
  for (int i = 0; i < t1; i++) if (idx1[i] < 0 || idx1[i] >= n1) return fail(e, DANI_ERR_INVALID, "idx1[%d] = %d outside [0, %d)", i, idx1[i], n1);
  // Only groups with two non-empty sides reach LightGlue (the reference's `if (!kpsKF1.empty() && ... )`, :970); they become
  // the pairs of ONE batched forward pass: every set is padded to 32 tokens, all pairs share each of the ~170 launches.
  std::vector<int> live;
  for (int g = 0; g < G; g++) if (off0[g + 1] > off0[g] && off1[g + 1] > off1[g]) live.push_back(g);
  const int P = (int)live.size();
  std::vector<int32_t> kk(std::max(P, 1), 0);
  std::vector<int32_t> m; std::vector<float> sc;
  if (P) {
    try {
      // gather: what the reference does row by row with cv::Mat::push_back, into the slab layout of dani_lg_match_batch
      std::vector<float> K0((size_t)P * maxn * 2), K1((size_t)P * maxn * 2), D0((size_t)P * maxn * 256), D1((size_t)P * maxn * 256);
      std::vector<int32_t> c0(P), c1(P);
      host_pool().run(P, [&](int p) {
        const int g = live[p];
        c0[p] = off0[g + 1] - off0[g]; c1[p] = off1[g + 1] - off1[g];
        for (int i = 0; i < c0[p]; i++) {
          const int r = idx0[off0[g] + i];
          memcpy(&K0[((size_t)p * maxn + i) * 2], kpts0 + (size_t)r * 2, 8);
          memcpy(&D0[((size_t)p * maxn + i) * 256], desc0 + (size_t)r * 256, 1024);
        }
        for (int i = 0; i < c1[p]; i++) {
          const int r = idx1[off1[g] + i];
          memcpy(&K1[((size_t)p * maxn + i) * 2], kpts1 + (size_t)r * 2, 8);
          memcpy(&D1[((size_t)p * maxn + i) * 256], desc1 + (size_t)r * 256, 1024);
        }
      });
      m.resize((size_t)P * maxn * 2); sc.resize((size_t)P * maxn);
      const int rc = lg_match_impl(e, LgCall{}, P, maxn, 0, K0.data(), D0.data(), c0.data(), K1.data(), D1.data(), c1.data(), size0, size1,
                                   threshold, m.data(), sc.data(), maxn, kk.data());
      if (rc) return rc;
    } catch (const std::exception&) {
      return fail(e, DANI_ERR_INVALID, "out of host memory for %d groups of up to %d keypoints", P, maxn);
    }
  }
  int total = 0, p = 0;
  for (int g = 0; g < G; g++) {
    const bool is_live = p < P && live[p] == g;
    const int k = is_live ? kk[p] : 0;
    if (total + k <= cap && k) {
      memcpy(matches + (size_t)total * 2, &m[(size_t)p * maxn * 2], (size_t)k * 8);
      memcpy(scores + total, &sc[(size_t)p * maxn], (size_t)k * 4);
    }
    total += k;
    match_offsets[g + 1] = total;
    if (is_live) p++;
  }
  if (total > cap) return fail(e, DANI_ERR_CAPACITY, "match capacity %d too small for %d matches", cap, total);
  return DANI_OK;
}

int dani_lg_attention_profile(dani_engine* e, float* total_ms, double* total_flop, int32_t* launches) {
  LgContext* L = last_lg(e);
  if (!e || !L) return DANI_ERR_INVALID;
  if (total_ms) *total_ms = L->attn_ms;
  if (total_flop) *total_flop = L->attn_flops;
  if (launches) *launches = 18;
  return DANI_OK;
}

int dani_lg_attention_profile_ex(dani_engine* e, float* total_ms, double* executed_flop, double* algorithmic_flop, int32_t* launches) {
  LgContext* L = last_lg(e);
  if (!e || !L) return DANI_ERR_INVALID;
  if (total_ms) *total_ms = L->attn_ms;
  if (executed_flop) *executed_flop = L->attn_flops;
  if (algorithmic_flop) *algorithmic_flop = L->attn_flops_alg;
  if (launches) *launches = 18;
  return DANI_OK;
}

int dani_lg_linear_profile(dani_engine* e, float* total_ms, double* total_flop, int32_t* launches) {
  LgContext* L = last_lg(e);
  if (!e || !L) return DANI_ERR_INVALID;
  if (total_ms) *total_ms = L->lin_ms;
  if (total_flop) *total_flop = L->lin_flops;
  if (launches) *launches = L->lin_launches;   // 55 linear launches, or 1 + 18 with the FFN chain (each chain counted once, its tail-region launches included)
  return DANI_OK;
}

int dani_lg_debug_fetch(dani_engine* e, int32_t set, float* desc_out, int32_t* m0, float* max0, int32_t cap) {
  LgContext* L = last_lg(e);
  if (!e || !L || set < 0 || set >= (int)L->last_sets.size()) return fail(e, DANI_ERR_INVALID, "no previous LightGlue call on this thread");
  const LgSet& st = L->last_sets[set];
  if (st.n > cap) return fail(e, DANI_ERR_CAPACITY, "need %d", st.n);
  HIP_TRY(e, hipSetDevice(e->device));
  if (desc_out) {
    std::vector<float> tmp((size_t)256 * st.n);
    for (int f = 0; f < 256; f++) {
      HIP_TRY(e, hipMemcpy(&tmp[(size_t)f * st.n], L->x + (size_t)f * L->ld + st.tok_off, (size_t)st.nmain * 4, hipMemcpyDeviceToHost));
      if (st.n > st.nmain)
        HIP_TRY(e, hipMemcpy(&tmp[(size_t)f * st.n + st.nmain], L->x + (size_t)f * L->ld + st.tail_off, (size_t)(st.n - st.nmain) * 4, hipMemcpyDeviceToHost));
    }
    for (int i = 0; i < st.n; i++)
      for (int f = 0; f < 256; f++) desc_out[(size_t)i * 256 + f] = tmp[(size_t)f * st.n + i];
  }
  if (m0) memcpy(m0, L->h_m0 + st.dense_off, (size_t)st.n * 4);
  if (max0) memcpy(max0, L->h_max0 + st.dense_off, (size_t)st.n * 4);
  return DANI_OK;
}

}  // extern "C"

