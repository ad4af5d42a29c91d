// yolo_engine.cpp -- YOLOv5s part of the C ABI (weights packing, workspaces, forward pass, read-back, host NMS).
#include "engine_internal.h"

using namespace dani;
using namespace dani::eng;

// ================================================================================================
// YOLOv5s dynamic-object detector
// ================================================================================================
namespace dani {
namespace eng {

constexpr int YOLO_W = 640, YOLO_H = 380;   // cv::resize(mRGB, img, cv::Size(640, 380)), src/YoloDetect.cpp:39
constexpr int YOLO_CAND_CAP = 15120;   // = 3 x (48*80 + 24*40 + 12*20): EVERY prediction of the 640x380 input; the decode appends each
                                       // prediction at most once, so the candidate buffer cannot overflow at any score threshold
constexpr int YOLO_CAND_FAST = 512;   // records per frame fetched together with the counts

struct YConv { ConvLayerPlan plan; bool ready = false; };
struct YoloWeights {
  bool ready = false;
  std::map<std::string, YConv> convs;   // by public tensor prefix, e.g. "model.2.cv1.conv"
};

YoloWeights* yolo_weights(dani_engine* e) {
  std::lock_guard<std::mutex> lk(e->mu);
  if (!e->yw) e->yw = new YoloWeights();
  return e->yw;
}

struct YoloContext {
  int B = 0;
  std::map<std::string, ActDesc> buf;
  uint8_t* d_img = nullptr; size_t d_img_bytes = 0;
  float* cand = nullptr; int* cand_count = nullptr;
  float* h_cand = nullptr; int* h_count = nullptr;
  // counts + the first YOLO_CAND_FAST records of every frame as ONE contiguous block (device and pinned host): one linear copy.
  // (hipMemcpy2DAsync of the strided [frame][15120] layout took a path whose completion reached the host ~0.1 ms after the GPU was
  //  done -- with the GPU idle behind it.)  fast_block = [64 ints: counts][B][YOLO_CAND_FAST][8]; cand_count / h_count point at its head.
  char* fast_block = nullptr; char* h_fast_block = nullptr;
  float* cand_fast = nullptr; float* h_cand_fast = nullptr;
  std::vector<void*> dev_allocs, host_allocs;
  // the Detect heads of the 48x80 and 24x40 levels run on a side stream as soon as their inputs exist (model.17 / model.20),
  // under the rest of the chain, instead of at its end
  hipStream_t side = nullptr;
  hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
  bool side_tried = false;   // creation of the side stream is attempted once per context
};

void yolo_release(dani_engine* e) {
  std::lock_guard<std::mutex> lk(e->mu);
  for (Context* c : e->ctxs) {
    YoloContext* Y = c->yolo;
    if (!Y) continue;
    for (void* p : Y->dev_allocs) (void)hipFree(p);
    for (void* p : Y->host_allocs) (void)hipHostFree(p);
    if (Y->side) (void)hipStreamDestroy(Y->side);
    for (int i = 0; i < 2; i++) { if (Y->ev_in[i]) (void)hipEventDestroy(Y->ev_in[i]); if (Y->ev_out[i]) (void)hipEventDestroy(Y->ev_out[i]); }
    delete Y;
    c->yolo = nullptr;
  }
  delete e->yw;
  e->yw = nullptr;
}

void yolo_shape(ActDesc& a, int C, int H, int W, int pady) {
  a.C = C; a.H = H; a.W = W; a.pady = pady;
  a.Hp = pady + H + 40;
  a.Wp = PADX + round_up(W, 32) + 72;
}

struct YLayer { const char* name; int cin, cout, k, s; int scale; bool act; };  // scale = output pyramid level 0..4

// output dims of the five pyramid levels for the fixed 380x640 network input
constexpr int kYH[5] = {190, 95, 48, 24, 12}, kYW[5] = {320, 160, 80, 40, 20};
static_assert(kYH[4] * kYW[4] <= 1024, "sppf_pool_kernel keeps one stride-32 plane in LDS");
static_assert(3 * (kYH[2] * kYW[2] + kYH[3] * kYW[3] + kYH[4] * kYW[4]) == YOLO_CAND_CAP, "one candidate slot per prediction");

ConvVariant yolo_variant(int k, int s, int scale, bool act) {
  const bool row32 = scale <= 1;   // 190x320 and 95x160 use 8x32 tiles, the rest 16x16
  if (k == 6) return CONV6S2_ROW32_SILU;
  if (k == 3 && s == 2) return row32 ? CONV3S2_ROW32_SILU : CONV3S2_BLK16_SILU;
  if (k == 3) return row32 ? CONV3_ROW32_SILU : CONV3_BLK16_SILU;
  if (!act) return CONV1_BLK16_LINEAR;
  return row32 ? CONV1_ROW32_SILU : CONV1_BLK16_SILU;
}

int yolo_get(dani_engine* e, const std::string& name, int cin, int cout, int k, const std::vector<float>** w, const std::vector<float>** b) {
  const std::string wn = name + ".weight", bn = name + ".bias";
  auto wi = e->tensors.find(wn), bi = e->tensors.find(bn);
  if (wi == e->tensors.end() || bi == e->tensors.end()) return fail(e, DANI_ERR_WEIGHTS, "missing tensor %s / %s", wn.c_str(), bn.c_str());
  if (wi->second.size() != (size_t)cout * cin * k * k || bi->second.size() != (size_t)cout)
    return fail(e, DANI_ERR_WEIGHTS, "tensor %s has %zu floats, expected [%d,%d,%d,%d]", wn.c_str(), wi->second.size(), cout, cin, k, k);
  *w = &wi->second; *b = &bi->second;
  return 0;
}

int yolo_pack_raw(dani_engine* e, YoloWeights* W, const std::string& name, const std::vector<float>& wsrc, const std::vector<float>& bsrc,
                  int cin, int cout, int k, int s, int scale, bool act);

int yolo_pack(dani_engine* e, YoloWeights* W, const std::string& name, int cin, int cout, int k, int s, int scale, bool act,
              bool detect) {
  (void)detect;
  const std::vector<float>* w = nullptr; const std::vector<float>* b = nullptr;
  int rc = yolo_get(e, name, cin, cout, k, &w, &b);
  if (rc) return rc;
  return yolo_pack_raw(e, W, name, *w, *b, cin, cout, k, s, scale, act);
}

int yolo_pack_raw(dani_engine* e, YoloWeights* W, const std::string& name, const std::vector<float>& wsrc, const std::vector<float>& bsrc,
                  int cin, int cout, int k, int s, int scale, bool act) {
  YConv& yc = W->convs[name];
  ConvLayerPlan& L = yc.plan;
  L.variant = yolo_variant(k, s, scale, act);
  if (cout <= 32) {   // narrow twins: do not compute 32 zero-padded output channels
    if (L.variant == CONV6S2_ROW32_SILU) L.variant = CONV6S2_ROW32N_SILU;
    else if (L.variant == CONV3_ROW32_SILU) L.variant = CONV3_ROW32N_SILU;
    else if (L.variant == CONV1_ROW32_SILU) L.variant = CONV1_ROW32N_SILU;
  }
  if (scale >= 2) {   // 16x16-tile layers: small-tile twin when the grid would not fill the chip
    const int tiles = ((kYH[scale] + 15) / 16) * ((kYW[scale] + 15) / 16);
    if (e->cfg.max_batch * tiles * ((cout + 63) / 64) < 768) L.variant = conv_small_variant(L.variant);
  }
  if (scale == 4) {   // 12 x 20: 4 x 32-pixel tiles waste less of every MFMA on padding pixels
    if (k == 3 && s == 2) L.variant = CONV3S2_R4_SILU;
    else if (k == 3) L.variant = CONV3_R4_SILU;
    else if (k == 1) L.variant = act ? CONV1_R4_SILU : CONV1_R4_LINEAR;
  }
  if (scale <= 1 && k != 6) {   // 190x320 / 95x160 at small batch: 4x32-pixel x 32-cout tiles, 4x the workgroups (same k order)
    const int tiles = ((kYH[scale] + 7) / 8) * ((kYW[scale] + 31) / 32);
    static const int small_wgs = getenv("DANI_YOLO_SMALL_WGS") ? atoi(getenv("DANI_YOLO_SMALL_WGS")) : 256;
    if (e->cfg.max_batch * tiles * ((cout + 63) / 64) < small_wgs) {
      if (k == 3 && s == 2) L.variant = CONV3S2_R4_SILU;
      else if (k == 3) L.variant = CONV3_R4_SILU;
      else if (k == 1) L.variant = act ? CONV1_R4_SILU : CONV1_R4_LINEAR;
    }
  }
  // Long contractions: the canonical order is a sum of K segments (conv_kseg: a function of the layer's shape, the oracle
  // follows the same rule), each segment on its own wave(s) of a workgroup.  Such layers always take a small-tile variant.
  const int kseg = conv_kseg(cin, k, s);
  if (kseg > 1) {
    L.variant = conv_small_variant(L.variant);
    switch (L.variant) {
      case CONV3_BLK16S_SILU: L.variant = kseg == 2 ? CONV3_BLK16S_SILU_K2 : L.variant; break;
      case CONV3S2_BLK16S_SILU: L.variant = kseg == 2 ? CONV3S2_BLK16S_SILU_K2 : L.variant; break;
      case CONV3_R4_SILU: L.variant = kseg == 4 ? CONV3_R4_SILU_K4 : L.variant; break;
      case CONV3S2_R4_SILU: L.variant = kseg == 2 ? CONV3S2_R4_SILU_K2 : L.variant; break;
      case CONV1_BLK16S_SILU: L.variant = kseg == 2 ? CONV1_BLK16S_SILU_K2 : L.variant; break;
      case CONV1_R4_SILU: L.variant = kseg == 2 ? CONV1_R4_SILU_K2 : L.variant; break;
      case CONV1_R4_LINEAR: L.variant = kseg == 2 ? CONV1_R4_LINEAR_K2 : L.variant; break;
      default: break;
    }
    if (conv_variant_kseg(L.variant) != kseg)
      return fail(e, DANI_ERR_UNSUPPORTED, "layer %s (%d -> %d, %dx%d at level %d): no kernel for its %d-segment contraction", name.c_str(), cin, cout, k, k, scale, kseg);
  }
  // The two deepest levels (24 x 40, 12 x 20) take flat pixel tiles: 32 consecutive pixels of a frame per wave instead of a
  // rectangle that the level's shape fills at 83 % / 62 % (conv_mfma.hip, Cfg::FW).  Same k order and K segments: bit-identical.
  // DANI_YOLO_FLAT=0 keeps the rectangular tiles of rounds 2-3 (A/B measurements).
  static const bool flat_off = getenv("DANI_YOLO_FLAT") && atoi(getenv("DANI_YOLO_FLAT")) == 0;
  if (!flat_off && scale >= 3 && kYW[3] == 40 && kYW[4] == 20) {
    ConvVariant f = L.variant;
    if (scale == 3) {
      if (k == 3 && s == 1 && kseg == 2) f = CONV3_F40_SILU_K2;
      else if (k == 3 && s == 2 && kseg == 2) f = CONV3S2_F40_SILU_K2;
      else if (k == 1 && act && kseg == 1) f = CONV1_F40_SILU;
      else if (k == 1 && !act && kseg == 1) f = CONV1_F40_LINEAR;
      // (1x1 with two K segments -- model.13's cv1|cv2 -- keeps the 8x16 tiles: 35.7 us against 38.5 flat; a 1x1 layer gains
      // nothing from full-width rows, it only stages more of them.  profiles/r04_yolo_chain_alone_flat*.txt)
    } else {
      if (k == 3 && s == 1 && kseg == 4) f = CONV3_F20_SILU_K4;
      else if (k == 3 && s == 2 && kseg == 2) f = CONV3S2_F20_SILU_K2;
      // 1x1 layers at 12 x 20: flat tiles only where they still launch two workgroups per CU (512 outputs); the 256-output
      // ones would drop to 128-256 workgroups of 64-128 pixels (15.5 us against 13.1, 18.8 against 17.9)
      else if (k == 1 && act && kseg == 2 && cout >= 512) f = CONV1_F20_SILU_K2;
    }
    L.variant = f;
  }
  if (conv_variant_kseg(L.variant) != kseg)
    return fail(e, DANI_ERR_UNSUPPORTED, "layer %s: kernel variant %d does not implement its %d-segment contraction", name.c_str(), (int)L.variant, kseg);
  L.Cin = cin; L.Cout = cout; L.ksize = k;
  L.CC = conv_variant_cc(L.variant);
  L.CinPad = round_up(cin, L.CC);
  const int nco = conv_variant_nco(L.variant);
  L.cout_tiles = (cout + nco - 1) / nco;
  std::vector<float> wp((size_t)L.cout_tiles * nco * L.CinPad * k * k), bp((size_t)L.cout_tiles * nco);
  pack_conv_weights(wsrc.data(), bsrc.data(), cin, cout, k, L.CC, nco, wp.data(), bp.data());
  HIP_TRY(e, hipMalloc((void**)&L.w_packed, wp.size() * 4)); e->weight_allocs.push_back(L.w_packed);
  HIP_TRY(e, hipMalloc((void**)&L.bias_packed, bp.size() * 4)); e->weight_allocs.push_back(L.bias_packed);
  HIP_TRY(e, hipMemcpy(L.w_packed, wp.data(), wp.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(e, hipMemcpy(L.bias_packed, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
  yc.ready = true;
  return 0;
}

struct C3Def { const char* name; int c1, c2, n, scale; };
const C3Def kC3[8] = {{"model.2", 64, 64, 1, 1},   {"model.4", 128, 128, 2, 2}, {"model.6", 256, 256, 3, 3},
                      {"model.8", 512, 512, 1, 4}, {"model.13", 512, 256, 1, 3}, {"model.17", 256, 128, 1, 2},
                      {"model.20", 256, 256, 1, 3}, {"model.23", 512, 512, 1, 4}};

int yolo_finalize(dani_engine* e) {
  YoloWeights* W = yolo_weights(e);
  W->ready = false;
  if (e->tensors.find("model.0.conv.weight") == e->tensors.end()) return 0;   // YOLO weights not supplied
  int rc;
#define YP(name, cin, cout, k, s, scale) if ((rc = yolo_pack(e, W, name, cin, cout, k, s, scale, true, false))) return rc;
  YP("model.0.conv", 3, 32, 6, 2, 0) YP("model.1.conv", 32, 64, 3, 2, 1) YP("model.3.conv", 64, 128, 3, 2, 2)
  YP("model.5.conv", 128, 256, 3, 2, 3) YP("model.7.conv", 256, 512, 3, 2, 4)
  YP("model.9.cv1.conv", 512, 256, 1, 1, 4) YP("model.9.cv2.conv", 1024, 512, 1, 1, 4)
  YP("model.10.conv", 512, 256, 1, 1, 4) YP("model.14.conv", 256, 128, 1, 1, 3)
  YP("model.18.conv", 128, 128, 3, 2, 3) YP("model.21.conv", 256, 256, 3, 2, 4)
  for (const C3Def& c : kC3) {
    const std::string p = c.name;
    const int c_ = c.c2 / 2;
    {   // cv1 and cv2 read the same input and write adjacent channel slices of the concat buffer: one 2c_-row conv
      const std::vector<float>*w1, *b1, *w2, *b2;
      if ((rc = yolo_get(e, p + ".cv1.conv", c.c1, c_, 1, &w1, &b1))) return rc;
      if ((rc = yolo_get(e, p + ".cv2.conv", c.c1, c_, 1, &w2, &b2))) return rc;
      std::vector<float> w12(*w1), b12(*b1);
      w12.insert(w12.end(), w2->begin(), w2->end()); b12.insert(b12.end(), b2->begin(), b2->end());
      if ((rc = yolo_pack_raw(e, W, p + ".cv12", w12, b12, c.c1, 2 * c_, 1, 1, c.scale, true))) return rc;
    }
    YP(p + ".cv3.conv", 2 * c_, c.c2, 1, 1, c.scale)
    for (int j = 0; j < c.n; j++) {
      YP(p + ".m." + std::to_string(j) + ".cv1.conv", c_, c_, 1, 1, c.scale)
      YP(p + ".m." + std::to_string(j) + ".cv2.conv", c_, c_, 3, 1, c.scale)
    }
  }
#undef YP
  const int dch[3] = {128, 256, 512};
  for (int i = 0; i < 3; i++)
    if ((rc = yolo_pack(e, W, "model.24.m." + std::to_string(i), dch[i], 255, 1, 1, 2 + i, false, true))) return rc;
  W->ready = true;
  return 0;
}

int yolo_ensure(dani_engine* e, Context* c, YoloContext** out, int B, size_t img_bytes) {
  YoloContext* Y;
  {
    std::lock_guard<std::mutex> lk(e->mu);
    if (!c->yolo) c->yolo = new YoloContext();
    Y = c->yolo;
  }
  *out = Y;
  // side stream of the two early Detect heads: it and its four events exist completely or not at all (then the heads
  // simply stay on the chain's stream)
  static const bool side_heads = !getenv("DANI_YOLO_NO_SIDE_HEADS");
  if (side_heads && !Y->side && !Y->side_tried) {
    Y->side_tried = true;
    bool ok = create_priority_stream(&Y->side) == hipSuccess;
    for (int i = 0; i < 2 && ok; i++)
      ok = hipEventCreateWithFlags(&Y->ev_in[i], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&Y->ev_out[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
      for (int i = 0; i < 2; i++) {
        if (Y->ev_in[i]) { (void)hipEventDestroy(Y->ev_in[i]); Y->ev_in[i] = nullptr; }
        if (Y->ev_out[i]) { (void)hipEventDestroy(Y->ev_out[i]); Y->ev_out[i] = nullptr; }
      }
      if (Y->side) { (void)hipStreamDestroy(Y->side); Y->side = nullptr; }
      (void)hipGetLastError();
    }
  }
  if (Y->B >= B && Y->d_img_bytes >= img_bytes) return 0;
  for (void* p : Y->dev_allocs) (void)hipFree(p);
  for (void* p : Y->host_allocs) (void)hipHostFree(p);
  Y->dev_allocs.clear(); Y->host_allocs.clear(); Y->buf.clear();
  B = std::max(B, Y->B);
  auto mk = [&](const char* name, int C, int scale) -> int {
    ActDesc a;
    yolo_shape(a, C, kYH[scale], kYW[scale], PADY);
    HIP_TRY(e, hipMalloc((void**)&a.p, a.bytes(B)));
    Y->dev_allocs.push_back(a.p);
    HIP_TRY(e, hipMemset(a.p, 0, a.bytes(B)));   // zero borders once
    Y->buf[name] = a;
    return 0;
  };
  int rc;
  {
    ActDesc a;
    yolo_shape(a, 4, YOLO_H, YOLO_W, 2);
    HIP_TRY(e, hipMalloc((void**)&a.p, a.bytes(B))); Y->dev_allocs.push_back(a.p);
    HIP_TRY(e, hipMemset(a.p, 0, a.bytes(B)));
    Y->buf["in"] = a;
  }
#define MK(n, C, sc) if ((rc = mk(n, C, sc))) return rc;
  MK("b0", 32, 0) MK("b1", 64, 1) MK("b2", 64, 1) MK("b3", 128, 2) MK("b5", 256, 3) MK("b7", 512, 4) MK("b8", 512, 4)
  MK("b9", 512, 4) MK("b13", 256, 3) MK("b17", 128, 2) MK("b20", 256, 3) MK("b23", 512, 4)
  MK("cat9", 1024, 4) MK("cat12", 512, 3) MK("cat16", 256, 2) MK("cat19", 256, 3) MK("cat22", 512, 4)
  MK("raw0", 255, 2) MK("raw1", 255, 3) MK("raw2", 255, 4)
  for (const C3Def& c3 : kC3) {
    MK((std::string(c3.name) + ".cat").c_str(), c3.c2, c3.scale)
    MK((std::string(c3.name) + ".tmp").c_str(), c3.c2 / 2, c3.scale)
  }
#undef MK
  HIP_TRY(e, hipMalloc((void**)&Y->cand, (size_t)B * YOLO_CAND_CAP * 8 * 4)); Y->dev_allocs.push_back(Y->cand);
  const size_t count_bytes = (size_t)round_up(B, 64) * 4, fast_bytes = count_bytes + (size_t)B * YOLO_CAND_FAST * 32;
  HIP_TRY(e, hipMalloc((void**)&Y->fast_block, fast_bytes)); Y->dev_allocs.push_back(Y->fast_block);
  HIP_TRY(e, hipMemset(Y->fast_block, 0, fast_bytes));
  Y->cand_count = reinterpret_cast<int*>(Y->fast_block); Y->cand_fast = reinterpret_cast<float*>(Y->fast_block + count_bytes);
  HIP_TRY(e, hipHostMalloc((void**)&Y->h_cand, (size_t)B * YOLO_CAND_CAP * 8 * 4, hipHostMallocDefault)); Y->host_allocs.push_back(Y->h_cand);
  HIP_TRY(e, hipHostMalloc((void**)&Y->h_fast_block, fast_bytes, hipHostMallocDefault)); Y->host_allocs.push_back(Y->h_fast_block);
  Y->h_count = reinterpret_cast<int*>(Y->h_fast_block); Y->h_cand_fast = reinterpret_cast<float*>(Y->h_fast_block + count_bytes);
  Y->d_img_bytes = std::max(img_bytes, Y->d_img_bytes);
  HIP_TRY(e, hipMalloc((void**)&Y->d_img, Y->d_img_bytes)); Y->dev_allocs.push_back(Y->d_img);
  // the zero fills above ran on the null stream, which the engine's non-blocking streams do not wait for
  HIP_TRY(e, hipDeviceSynchronize());
  Y->B = B;
  return 0;
}

// a view of channels [c0, c0+C) of buffer `a`
struct YView { const ActDesc* a; int c0, C; };

thread_local bool t_yolo_launch_ok = true;   // cleared when a conv launch refuses its shape (cannot happen at the fixed 640x380 input)
void yolo_conv(const YoloWeights* W, const std::string& name, YView in, YView out, const YView* res, int B, hipStream_t s) {
  const ConvLayerPlan& L = W->convs.at(name).plan;
  ConvIO io;
  const size_t iplane = (size_t)in.a->Hp * in.a->Wp, oplane = (size_t)out.a->Hp * out.a->Wp;
  io.in = in.a->p + (size_t)in.c0 * iplane; io.out = out.a->p + (size_t)out.c0 * oplane;
  io.res = res ? res->a->p + (size_t)res->c0 * ((size_t)res->a->Hp * res->a->Wp) : nullptr;
  io.H = out.a->H; io.W = out.a->W;
  io.iHp = in.a->Hp; io.iWp = in.a->Wp; io.ipady = in.a->pady; io.oHp = out.a->Hp; io.oWp = out.a->Wp; io.opady = out.a->pady;
  io.in_fs = in.a->frame_stride(); io.out_fs = out.a->frame_stride(); io.res_fs = res ? res->a->frame_stride() : 0;
  if (!launch_conv_io(L, io, B, s)) t_yolo_launch_ok = false;
}

void yolo_c3(const YoloWeights* W, YoloContext* Y, const C3Def& c3, YView x, YView out, int B, hipStream_t s) {
  const std::string p = c3.name;
  const int c_ = c3.c2 / 2;
  const bool shortcut = std::string(c3.name) == "model.2" || std::string(c3.name) == "model.4" ||
                        std::string(c3.name) == "model.6" || std::string(c3.name) == "model.8";   // backbone C3s
  const ActDesc& cat = Y->buf.at(p + ".cat"); const ActDesc& tmp = Y->buf.at(p + ".tmp");
  YView y{&cat, 0, c_}, y2{&cat, c_, c_}, t{&tmp, 0, c_};
  (void)y2;
  yolo_conv(W, p + ".cv12", x, YView{&cat, 0, 2 * c_}, nullptr, B, s);   // [cv1(x) | cv2(x)] in one launch
  for (int j = 0; j < c3.n; j++) {
    yolo_conv(W, p + ".m." + std::to_string(j) + ".cv1.conv", y, t, nullptr, B, s);
    yolo_conv(W, p + ".m." + std::to_string(j) + ".cv2.conv", t, y, shortcut ? &y : nullptr, B, s);   // y = y + cv2(cv1(y)) in place
  }
  yolo_conv(W, p + ".cv3.conv", YView{&cat, 0, 2 * c_}, out, nullptr, B, s);
}

void yolo_forward(const YoloWeights* W, YoloContext* Y, int B, float score_thresh, hipStream_t s, const YoloCall* call = nullptr) {
  auto& b = Y->buf;
  int stage = 0;
  auto mark = [&] { if (call && call->mid_chain && ++stage == call->mid_after) (*call->mid_chain)(); };
  auto V = [&](const char* n) { const ActDesc& a = b.at(n); return YView{&a, 0, a.C}; };
  auto S = [&](const char* n, int c0, int C) { return YView{&b.at(n), c0, C}; };
  (void)hipMemsetAsync(Y->cand_count, 0, (size_t)B * 4, s);   // candidate counters of the decode at the end
  yolo_conv(W, "model.0.conv", YView{&b.at("in"), 0, 4}, V("b0"), nullptr, B, s); mark();
  yolo_conv(W, "model.1.conv", V("b0"), V("b1"), nullptr, B, s); mark();
  yolo_c3(W, Y, kC3[0], V("b1"), V("b2"), B, s); mark();
  yolo_conv(W, "model.3.conv", V("b2"), V("b3"), nullptr, B, s); mark();
  yolo_c3(W, Y, kC3[1], V("b3"), S("cat16", 128, 128), B, s); mark();            // x4 -> second half of cat16
  yolo_conv(W, "model.5.conv", S("cat16", 128, 128), V("b5"), nullptr, B, s); mark();
  yolo_c3(W, Y, kC3[2], V("b5"), S("cat12", 256, 256), B, s); mark();            // x6 -> second half of cat12
  yolo_conv(W, "model.7.conv", S("cat12", 256, 256), V("b7"), nullptr, B, s);
  yolo_c3(W, Y, kC3[3], V("b7"), V("b8"), B, s);
  // SPPF
  yolo_conv(W, "model.9.cv1.conv", V("b8"), S("cat9", 0, 256), nullptr, B, s);
  {
    const ActDesc& c9 = b.at("cat9");
    launch_sppf_pool(c9.p, 256, c9.H, c9.W, c9.Hp, c9.Wp, c9.frame_stride(), B, s);
  }
  yolo_conv(W, "model.9.cv2.conv", V("cat9"), V("b9"), nullptr, B, s);
  yolo_conv(W, "model.10.conv", V("b9"), S("cat22", 256, 256), nullptr, B, s);   // x10 -> second half of cat22
  auto up = [&](const char* src, int sc0, int C, const char* dst) {
    const ActDesc& sa = b.at(src); const ActDesc& da = b.at(dst);
    launch_upsample2(sa.p + (size_t)sc0 * sa.Hp * sa.Wp, sa.Hp, sa.Wp, sa.frame_stride(), da.p, C, da.H, da.W, da.Hp, da.Wp,
                     da.frame_stride(), B, s);
  };
  up("cat22", 256, 256, "cat12");
  yolo_c3(W, Y, kC3[4], V("cat12"), V("b13"), B, s);
  yolo_conv(W, "model.14.conv", V("b13"), S("cat19", 128, 128), nullptr, B, s);  // x14 -> second half of cat19
  up("cat19", 128, 128, "cat16");
  // head i on the side stream once `in` is complete (stream and events come from yolo_ensure).  A failed record / wait
  // before the launch leaves the head on the chain's stream; after the launch the chain waits for the side stream instead.
  bool side_sync = false;
  auto head_aside = [&](int i, const char* name, const char* in, const char* out) {
    if (!Y->side) return false;
    if (hipEventRecord(Y->ev_in[i], s) != hipSuccess || hipStreamWaitEvent(Y->side, Y->ev_in[i], 0) != hipSuccess) { (void)hipGetLastError(); return false; }
    yolo_conv(W, name, V(in), V(out), nullptr, B, Y->side);
    if (hipEventRecord(Y->ev_out[i], Y->side) != hipSuccess) { (void)hipGetLastError(); side_sync = true; }
    return true;
  };
  yolo_c3(W, Y, kC3[5], V("cat16"), V("b17"), B, s);
  const bool h0 = head_aside(0, "model.24.m.0", "b17", "raw0");
  yolo_conv(W, "model.18.conv", V("b17"), S("cat19", 0, 128), nullptr, B, s);
  yolo_c3(W, Y, kC3[6], V("cat19"), V("b20"), B, s);
  const bool h1 = head_aside(1, "model.24.m.1", "b20", "raw1");
  yolo_conv(W, "model.21.conv", V("b20"), S("cat22", 0, 256), nullptr, B, s);
  yolo_c3(W, Y, kC3[7], V("cat22"), V("b23"), B, s);
  if (!h0) yolo_conv(W, "model.24.m.0", V("b17"), V("raw0"), nullptr, B, s);
  if (!h1) yolo_conv(W, "model.24.m.1", V("b20"), V("raw1"), nullptr, B, s);
  yolo_conv(W, "model.24.m.2", V("b23"), V("raw2"), nullptr, B, s);
  if (h0 && hipStreamWaitEvent(s, Y->ev_out[0], 0) != hipSuccess) side_sync = true;
  if (h1 && hipStreamWaitEvent(s, Y->ev_out[1], 0) != hipSuccess) side_sync = true;
  if (side_sync) { (void)hipGetLastError(); (void)hipStreamSynchronize(Y->side); }   // ordering by brute force, never a silent race
  const ActDesc* raws[3] = {&b.at("raw0"), &b.at("raw1"), &b.at("raw2")};
  launch_yolo_decode(raws, score_thresh, Y->cand, Y->cand_fast, YOLO_CAND_FAST, Y->cand_count, YOLO_CAND_CAP, B, s);
}

thread_local YoloContext* t_last_yolo = nullptr;
YoloContext* last_yolo(dani_engine* e) {
  if (!e || !t_last_yolo) return nullptr;
  std::lock_guard<std::mutex> lk(e->mu);
  for (Context* c : e->ctxs) if (c->yolo && c->yolo == t_last_yolo) return t_last_yolo;
  return nullptr;
}

}  // namespace eng
}  // namespace dani

extern "C" {

}  // extern "C"

namespace dani {
namespace eng {

int yolo_detect_impl(dani_engine* e, const YoloCall& call, const uint8_t* bgr, int32_t on_device, int32_t B, int32_t W, int32_t H,
                     int32_t stride, float score_thresh, float iou_thresh, float* dets6, int32_t cap, int32_t* n_out) {
  if (!e) return DANI_ERR_INVALID;
  if (!bgr || W <= 0 || H <= 0) return fail(e, DANI_ERR_EMPTY, "Read RGB failed!");   // src/YoloDetect.cpp:32-36
  if (B < 1 || B > e->cfg.max_batch || stride < 3 * W || !dets6 || !n_out || cap < 1) return fail(e, DANI_ERR_INVALID, "bad dani_yolo_detect argument");
  YoloWeights* Wt = yolo_weights(e);
  if (!e->finalized || !Wt->ready) return fail(e, DANI_ERR_WEIGHTS, "YOLO weights not loaded/finalized");
  HIP_TRY(e, hipSetDevice(e->device));
  CtxLease lease(e, call.ctx);
  Context* c = lease.c;
  if (!c) return fail(e, DANI_ERR_INVALID, "all contexts busy");
  YoloContext* Y = nullptr;
  const size_t fbytes = (size_t)H * stride;
  int rc = yolo_ensure(e, c, &Y, e->cfg.max_batch, fbytes * e->cfg.max_batch);
  if (rc) return rc;
  t_last_yolo = Y; t_last_ctx = c;
  hipStream_t s = call.alt_stream ? c->stream2 : c->stream;
  const uint8_t* dimg = bgr;
  if (!on_device) {
    HIP_TRY(e, hipMemcpyAsync(Y->d_img, bgr, fbytes * B, hipMemcpyHostToDevice, s));
    dimg = Y->d_img;
  }
  HIP_TRY(e, hipEventRecord(c->ev_y0, s));
  trace_mark("yolo_first_launch");
  launch_yolo_preproc(dimg, B, W, H, stride, fbytes, Y->buf.at("in"), s);
  t_yolo_launch_ok = true;
  yolo_forward(Wt, Y, B, score_thresh, s, &call);
  if (call.mid_chain && (call.mid_after < 1 || call.mid_after > 7)) (*call.mid_chain)();   // never skip the caller's work
  if (!t_yolo_launch_ok) { (void)hipStreamSynchronize(s); return fail(e, DANI_ERR_UNSUPPORTED, "YOLO conv launch refused its tensor shape"); }
  HIP_TRY(e, hipEventRecord(c->ev_y1, s));
  HIP_TRY(e, hipGetLastError());
  // read-back: the counts and, optimistically, the first YOLO_CAND_FAST records of every frame in one round trip;
  // a frame with more candidates than that costs a second copy
  HIP_TRY(e, hipMemcpyAsync(Y->h_fast_block, Y->fast_block, (size_t)round_up(e->cfg.max_batch, 64) * 4 + (size_t)B * YOLO_CAND_FAST * 32, hipMemcpyDeviceToHost, s));
  trace_mark("yolo_enqueued");
  if (call.before_sync) (*call.before_sync)();   // the caller's host work that does not need the detections
  trace_mark("speculation_done");
  HIP_TRY(e, hipStreamSynchronize(s));
  trace_mark("yolo_synced");
  if (t_trace.on && t_trace.n < 40) {   // device-side duration of the chain (ev_y0 .. ev_y1, in front of the read-back), as a pseudo mark
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, c->ev_y0, c->ev_y1);
    t_trace.label[t_trace.n] = "zz_gpu_yolo_chain_duration"; t_trace.us[t_trace.n++] = ms * 1000.0f;
  }
  bool more = false;
  for (int b = 0; b < B; b++) {
    if (Y->h_count[b] < 0 || Y->h_count[b] > YOLO_CAND_CAP)   // more records than predictions: memory corruption, not a capacity case
      return fail(e, DANI_ERR_HIP, "frame %d: candidate counter %d outside [0, %d]", b, Y->h_count[b], YOLO_CAND_CAP);
    if (Y->h_count[b] > YOLO_CAND_FAST) {
      more = true;
      memcpy(Y->h_cand + (size_t)b * YOLO_CAND_CAP * 8, Y->h_cand_fast + (size_t)b * YOLO_CAND_FAST * 8, (size_t)YOLO_CAND_FAST * 32);   // the frame's list, contiguous
      HIP_TRY(e, hipMemcpyAsync(Y->h_cand + ((size_t)b * YOLO_CAND_CAP + YOLO_CAND_FAST) * 8, Y->cand + ((size_t)b * YOLO_CAND_CAP + YOLO_CAND_FAST) * 8,
                                (size_t)(Y->h_count[b] - YOLO_CAND_FAST) * 32, hipMemcpyDeviceToHost, s));
    }
  }
  if (more) HIP_TRY(e, hipStreamSynchronize(s));
  if (!call.alt_stream) { (void)hipEventElapsedTime(&c->ms[0], c->ev_y0, c->ev_y1); c->ms[4] = c->ms[0]; }
  int status = DANI_OK;
  // greedy NMS is O(k n) per frame and independent across frames.  A handful of records per frame (the usual case) is done by the
  // calling thread at once: posting eight microsecond tasks to sleeping workers cost 24 us with the GPU idle (tools/debug/host_trace.py)
  auto nms_task = [&](int b) {
    float* recs = Y->h_count[b] > YOLO_CAND_FAST ? Y->h_cand + (size_t)b * YOLO_CAND_CAP * 8 : Y->h_cand_fast + (size_t)b * YOLO_CAND_FAST * 8;
    n_out[b] = yolo_nms_host(recs, Y->h_count[b], iou_thresh, dets6 + (size_t)b * cap * 6, cap);
    if (call.post && n_out[b] <= cap) (*call.post)(b, dets6 + (size_t)b * cap * 6, n_out[b]);
  };
  int total_records = 0;
  for (int b = 0; b < B; b++) total_records += Y->h_count[b];
  static const bool old_order = getenv("DANI_FE_OLD_ORDER") && atoi(getenv("DANI_FE_OLD_ORDER")) != 0;
  if (total_records <= 256 && !old_order) { for (int b = 0; b < B; b++) nms_task(b); }
  else host_pool().run(B, nms_task);
  trace_mark("nms_done");
  for (int b = 0; b < B; b++)
    if (n_out[b] > cap) status = DANI_ERR_CAPACITY;
  if (status != DANI_OK) return fail(e, status, "detection capacity %d too small", cap);
  return DANI_OK;
}

}  // namespace eng
}  // namespace dani

extern "C" {

int dani_yolo_detect_batch(dani_engine* e, const uint8_t* bgr, int32_t on_device, int32_t B, int32_t W, int32_t H,
                           int32_t stride, float score_thresh, float iou_thresh, float* dets6, int32_t cap, int32_t* n_out) {
  return yolo_detect_impl(e, YoloCall{}, bgr, on_device, B, W, H, stride, score_thresh, iou_thresh, dets6, cap, n_out);
}

int dani_yolo_detect(dani_engine* e, const uint8_t* bgr, int32_t W, int32_t H, int32_t stride, float score_thresh,
                     float iou_thresh, float* dets6, int32_t cap, int32_t* n_out) {
  return dani_yolo_detect_batch(e, bgr, 0, 1, W, H, stride, score_thresh, iou_thresh, dets6, cap, n_out);
}

int dani_host_dynamic_areas(const float* dets6, int32_t n, int32_t cols, int32_t rows, const char* const* class_names,
                            int32_t n_classes, const char* const* dynamic_names, int32_t n_dynamic, int32_t* det_rects,
                            int32_t* dyn_boxes, int32_t cap) {
  if (n < 0 || (n && !dets6) || !class_names || !dynamic_names || !dyn_boxes) return DANI_ERR_INVALID;
  const int k = yolo_dynamic_areas(dets6, n, cols, rows, class_names, n_classes, dynamic_names, n_dynamic, det_rects, dyn_boxes, cap);
  return k > cap ? DANI_ERR_CAPACITY : k;
}

int dani_yolo_debug_raw(dani_engine* e, int32_t b, int32_t level, float* out, int64_t count) {
  YoloContext* Y = last_yolo(e);
  Context* c = last_ctx(e);
  if (!e || !Y || !c || level < 0 || level > 2 || b < 0 || b >= Y->B || !out) return fail(e, DANI_ERR_INVALID, "bad argument");
  const char* raws[3] = {"raw0", "raw1", "raw2"};
  const ActDesc& a = Y->buf.at(raws[level]);
  const size_t need = (size_t)a.C * a.H * a.W;
  if ((size_t)count < need) return fail(e, DANI_ERR_CAPACITY, "need %zu floats", need);
  HIP_TRY(e, hipSetDevice(e->device));
  float* tmp = nullptr;
  HIP_TRY(e, hipMalloc((void**)&tmp, need * 4));
  launch_copy_layer_dense(a, b, tmp, c->stream);
  hipError_t st = hipMemcpyAsync(out, tmp, need * 4, hipMemcpyDeviceToHost, c->stream);
  if (st == hipSuccess) st = hipStreamSynchronize(c->stream);
  (void)hipFree(tmp);
  if (st != hipSuccess) return fail(e, DANI_ERR_HIP, "debug copy failed");
  return DANI_OK;
}

}  // extern "C"

