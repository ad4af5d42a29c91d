// sp_engine.cpp -- SuperPoint part of the C ABI.
//
// Host side of the reference calls replaced here:
//   SPextractor::SPextractor / operator() / ComputeKeyPointsOctTree   /root/reference/src/SPextractor.cpp:80-386
//   SPDetector::detect / getKeyPoints / computeDescriptors             src/SuperPoint.cpp:186-274
// Every stage of the network runs in the HIP kernels of conv_mfma.hip / sp_post.hip; only the order-defining quadtree
// (host_logic.cpp) runs on the host, as it does in the reference.
#include "engine_internal.h"

#include <cmath>

using namespace dani;
using namespace dani::eng;

namespace dani {
namespace eng {

struct SpLayerDef { const char* name; int cin, cout, k; };
const SpLayerDef kSp[12] = {
    {"conv1a", 1, 64, 3},    {"conv1b", 64, 64, 3},   {"conv2a", 64, 64, 3},    {"conv2b", 64, 64, 3},
    {"conv3a", 64, 128, 3},  {"conv3b", 128, 128, 3}, {"conv4a", 128, 128, 3},  {"conv4b", 128, 128, 3},
    {"convPa", 128, 256, 3}, {"convPb", 256, 65, 1},  {"convDa", 128, 256, 3},  {"convDb", 256, 256, 1}};
const ConvVariant kSpVariant[12] = {
    CONV3_ROW32_RELU /*unused for 1a*/, CONV3_ROW32_RELU_POOL, CONV3_ROW32_RELU, CONV3_ROW32_RELU_POOL,
    CONV3_ROW32_RELU, CONV3_ROW32_RELU_POOL, CONV3_BLK16_RELU, CONV3_BLK16_RELU,
    CONV3_BLK16_RELU, CONV1_BLK16_LINEAR, CONV3_BLK16_RELU, CONV1_BLK16_LINEAR};

// pyramid level sizes: cvRound(dim * mvInvScaleFactor[level]) (src/SPextractor.cpp:635-636)
void level_dims(int W, int H, float scaleFactor, int nlevels, std::vector<int>& lw, std::vector<int>& lh, std::vector<float>& scales) {
  std::vector<int> per;
  features_per_level(1000, scaleFactor, nlevels, per, scales);
  lw.assign(nlevels, W); lh.assign(nlevels, H);
  for (int l = 1; l < nlevels; l++) {
    const float inv = 1.0f / scales[l];
    lw[l] = (int)lrintf((float)W * inv);
    lh[l] = (int)lrintf((float)H * inv);
  }
}

// (re)build the per-context buffers for image size W x H, batch B and the pyramid (nlevels, scaleFactor)
int ensure_shapes(dani_engine* e, Context* c, int W, int H, int B, int kcap, int nlevels, float scaleFactor) {
  if (c->W == W && c->H == H && c->B >= B && c->kcap >= kcap && c->nlevels == nlevels && c->scale_factor == scaleFactor) return 0;
  HIP_TRY(e, hipStreamSynchronize(c->stream));
  free_buffers(c);
  std::vector<int> lw, lh; std::vector<float> scales;
  level_dims(W, H, scaleFactor, nlevels, lw, lh, scales);
  c->lv.resize(nlevels);
  int rc;
  for (int l = 0; l < nlevels; l++) {
    Context::Level& L = c->lv[l];
    L.W = lw[l]; L.H = lh[l];
    const int h = L.H, w = L.W;
    if (w < 80 || h < 80) return fail(e, DANI_ERR_UNSUPPORTED, "pyramid level %d is %dx%d: smaller than 80x80", l, w, h);
    // DistributeOctTree starts from nIni = round(width / height) root nodes over the bordered region (src/SPextractor.cpp:410);
    // for a portrait level nIni = 0 and the reference indexes an empty vector (:433): no defined result to reproduce
    if ((int)std::round((float)(w - 32) / (float)(h - 32)) < 1)
      return fail(e, DANI_ERR_UNSUPPORTED, "pyramid level %d is %dx%d: width / height below 0.5 (DistributeOctTree has no root node)", l, w, h);
    const int dims[12][2] = {{h, w}, {h / 2, w / 2}, {h / 2, w / 2}, {h / 4, w / 4}, {h / 4, w / 4}, {h / 8, w / 8},
                             {h / 8, w / 8}, {h / 8, w / 8}, {h / 8, w / 8}, {h / 8, w / 8}, {h / 8, w / 8}, {h / 8, w / 8}};
    for (int i = 0; i < 12; i++) {
      act_shape(L.act[i], kSp[i].cout, dims[i][0], dims[i][1]);
      if ((rc = dev_alloc(e, c, (void**)&L.act[i].p, L.act[i].bytes(B), true))) return rc;  // zero borders, once
    }
    const int Hc = h / 8, Wc = w / 8, PH = Hc * 8, PW = Wc * 8;
    if ((rc = dev_alloc(e, c, (void**)&L.prob, (size_t)B * PH * PW * 4, false))) return rc;
    if ((rc = dev_alloc(e, c, (void**)&L.desc_nhwc, (size_t)B * Hc * Wc * 256 * 4, false))) return rc;
    if ((rc = dev_alloc(e, c, (void**)&L.desc_nchw, (size_t)B * Hc * Wc * 256 * 4, false))) return rc;
    cell_grid(w, h, L.cells);
    int max_area = 1;
    for (const CellRect& r : L.cells) {
      const int cw = r.maxX - r.iniX, chh = r.maxY - r.iniY;
      if (cw > CELL_MAX_DIM || chh > CELL_MAX_DIM || cw <= 0 || chh <= 0)
        return fail(e, DANI_ERR_UNSUPPORTED, "cell %dx%d exceeds the %d-pixel cell kernel limit", cw, chh, CELL_MAX_DIM);
      max_area = std::max(max_area, cw * chh);
    }
    L.slot_stride = max_area;
    // Candidate capacity = the sum of the cell areas: EVERY pixel of every cell (the 6-px overlap strips count twice, quirk Q5)
    // can be a candidate, and the reference processes any count (vToDistributeKeys grows, src/SPextractor.cpp:248-312).  At
    // 640x480 that is 204 cells x <= 1848 pixels = 4.4 MB per frame -- nothing against 288 GB -- so a dense frame (flat-high
    // probability map) cannot overflow it; rounds 1-3 kept 32768 and answered DANI_ERR_CAPACITY beyond.
    size_t area_sum = 0;
    for (const CellRect& r : L.cells) area_sum += (size_t)(r.maxX - r.iniX) * (r.maxY - r.iniY);
    L.cand_cap = (int)std::max<size_t>(area_sum, CAND_FAST);
    const int nc = std::max<int>(1, (int)L.cells.size());
    if ((rc = dev_alloc(e, c, (void**)&L.d_cells, sizeof(CellRect) * nc, false))) return rc;
    if (!L.cells.empty())
      HIP_TRY(e, hipMemcpyAsync(L.d_cells, L.cells.data(), sizeof(CellRect) * L.cells.size(), hipMemcpyHostToDevice, c->stream));
    if ((rc = dev_alloc(e, c, (void**)&L.slot_xyr, (size_t)B * nc * L.slot_stride * 3 * 4, false))) return rc;
    if ((rc = dev_alloc(e, c, (void**)&L.slot_count, (size_t)B * nc * 4, true))) return rc;
    if ((rc = dev_alloc(e, c, (void**)&L.cand, (size_t)B * L.cand_cap * 3 * 4, false))) return rc;
    if ((rc = dev_alloc(e, c, (void**)&L.cand_count, (size_t)B * 4, true))) return rc;
    if ((rc = host_alloc(e, c, (void**)&L.h_cand, (size_t)B * L.cand_cap * 3 * 4))) return rc;
    if ((rc = host_alloc(e, c, (void**)&L.h_cand_count, (size_t)B * 4))) return rc;
    if (l > 0 && (rc = dev_alloc(e, c, (void**)&L.d_level_img, (size_t)B * h * w, false))) return rc;
  }
  c->box_cap = 1024;
  if ((rc = dev_alloc(e, c, (void**)&c->d_boxes, (size_t)(c->box_cap * 4 + B + 1) * 4, true))) return rc;
  c->d_box_offsets = c->d_boxes + c->box_cap * 4;
  c->d_img_bytes = (size_t)B * H * round_up(W, 64) * 2;
  if ((rc = dev_alloc(e, c, (void**)&c->d_img, c->d_img_bytes, false))) return rc;
  c->kcap = kcap;
  const size_t nints = (size_t)nlevels * 2 * B;
  // feature slabs: slot 0 holds the last frame of the previous dani_frontend_batch call (sequence matching),
  // slots 1..B the current frames; d_kxy / d_descout point at slot 1; the per-level (offset, count) ints follow
  if ((rc = dev_alloc(e, c, (void**)&c->d_kxy_all, ((size_t)(B + 1) * kcap * 2 + nints) * 4, true))) return rc;
  c->d_kxy = c->d_kxy_all + (size_t)kcap * 2;
  c->d_nk = reinterpret_cast<int32_t*>(c->d_kxy + (size_t)B * kcap * 2);
  if ((rc = dev_alloc(e, c, (void**)&c->d_sampled, ((size_t)B * kcap * 256 + B * 256) * 4, false))) return rc;
  if ((rc = dev_alloc(e, c, (void**)&c->d_desc_all, (size_t)(B + 1) * kcap * 256 * 4, true))) return rc;
  c->d_descout = c->d_desc_all + (size_t)kcap * 256;
  // keypoint->pt as operator() returns it (level scale applied, mono / stereo partition order): LightGlue's keypoint input
  if ((rc = dev_alloc(e, c, (void**)&c->d_kimg_all, (size_t)(B + 1) * kcap * 2 * 4, true))) return rc;
  c->d_kimg = c->d_kimg_all + (size_t)kcap * 2;
  if ((rc = host_alloc(e, c, (void**)&c->h_kimg, (size_t)B * kcap * 2 * 4))) return rc;
  c->prev_n = 0;
  if ((rc = host_alloc(e, c, (void**)&c->h_kxy, ((size_t)B * kcap * 2 + nints) * 4))) return rc;
  if ((rc = host_alloc(e, c, (void**)&c->h_boxes, (size_t)(c->box_cap * 4 + B + 1) * 4))) return rc;
  if ((rc = host_alloc(e, c, (void**)&c->h_desc, (size_t)B * kcap * 256 * 4))) return rc;
  HIP_TRY(e, hipStreamSynchronize(c->stream));
  c->W = W; c->H = H; c->B = B; c->nlevels = nlevels; c->scale_factor = scaleFactor;
  return 0;
}

int check_params(dani_engine* e, int W, int H, const dani_sp_params* p) {
  if (!p) return fail(e, DANI_ERR_INVALID, "null params");
  if (p->nlevels < 1 || p->nlevels > 16) return fail(e, DANI_ERR_INVALID, "nlevels=%d outside [1, 16]", p->nlevels);
  if (p->nlevels > 1 && !(p->scale_factor > 1.0f)) return fail(e, DANI_ERR_INVALID, "scaleFactor must be > 1 for a pyramid");
  if (W < 80 || H < 80)
    return fail(e, DANI_ERR_UNSUPPORTED, "image %dx%d: width and height must be >= 80", W, H);
  if (W > e->cfg.max_width || H > e->cfg.max_height)
    return fail(e, DANI_ERR_INVALID, "image %dx%d exceeds the engine's configured maximum %dx%d", W, H, e->cfg.max_width, e->cfg.max_height);
  if (p->nfeatures <= 0 || !(p->ini_th >= 0.0f) || !(p->min_th >= 0.0f)) return fail(e, DANI_ERR_INVALID, "bad extractor params");
  {   // the conv epilogue addresses one frame's activation tensor with 32-bit byte offsets (conv_mfma.hip, launch_cfg)
    ActDesc a;
    act_shape(a, 64, H, W);
    if ((size_t)(64 + 8) * a.Hp * a.Wp >= ((size_t)1 << 30))
      return fail(e, DANI_ERR_UNSUPPORTED, "image %dx%d: a 64-channel full-resolution activation exceeds the kernels' 4 GiB per-frame addressing range", W, H);
  }
  return 0;
}

void sp_read_timers(Context* c, bool prof) {
  (void)hipEventElapsedTime(&c->ms[0], c->ev[0], c->ev[1]);
  (void)hipEventElapsedTime(&c->ms[1], c->ev[1], c->ev[2]);
  (void)hipEventElapsedTime(&c->ms[3], c->ev[3], c->ev[4]);
  (void)hipEventElapsedTime(&c->ms[4], c->ev[0], c->ev[4]);
  if (prof)
    for (int i = 0; i < 12; i++) (void)hipEventElapsedTime(&c->layer_ms[i], c->lev[i], c->lev[i + 1]);
}

}  // namespace eng
}  // namespace dani

extern "C" {

int dani_finalize_weights(dani_engine* e) {
  if (!e) return DANI_ERR_INVALID;
  {   // every name and shape first, on the host (weights_spec.cpp): the error names the offending tensor
    std::map<std::string, size_t> have;
    for (const auto& kv : e->tensors) have[kv.first] = kv.second.size();
    std::string msg;
    if (check_tensor_counts(have, msg, nullptr, nullptr) != DANI_OK) return fail(e, DANI_ERR_WEIGHTS, "%s", msg.c_str());
  }
  HIP_TRY(e, hipSetDevice(e->device));
  for (void* p : e->weight_allocs) (void)hipFree(p);
  e->weight_allocs.clear();
  for (int i = 0; i < 12; i++) {
    const SpLayerDef& d = kSp[i];
    const std::string wn = std::string(d.name) + ".weight", bn = std::string(d.name) + ".bias";
    auto wi = e->tensors.find(wn), bi = e->tensors.find(bn);
    if (wi == e->tensors.end() || bi == e->tensors.end())
      return fail(e, DANI_ERR_WEIGHTS, "missing tensor %s / %s", wn.c_str(), bn.c_str());
    const size_t wexp = (size_t)d.cout * d.cin * d.k * d.k;
    if (wi->second.size() != wexp || bi->second.size() != (size_t)d.cout)
      return fail(e, DANI_ERR_WEIGHTS, "tensor %s has %zu floats, expected %zu ([%d,%d,%d,%d])", wn.c_str(), wi->second.size(), wexp, d.cout, d.cin, d.k, d.k);
    if (i == 0) {
      HIP_TRY(e, hipMalloc((void**)&e->w1a, wexp * 4)); e->weight_allocs.push_back(e->w1a);
      HIP_TRY(e, hipMalloc((void**)&e->b1a, d.cout * 4)); e->weight_allocs.push_back(e->b1a);
      HIP_TRY(e, hipMemcpy(e->w1a, wi->second.data(), wexp * 4, hipMemcpyHostToDevice));
      HIP_TRY(e, hipMemcpy(e->b1a, bi->second.data(), d.cout * 4, hipMemcpyHostToDevice));
      continue;
    }
    ConvLayerPlan& L = e->plan[i];
    L.variant = kSpVariant[i];
    if (i >= 6) {   // 1/8-resolution layers: switch to the small-tile twin when the grid would not fill the chip
      const int tiles = ((e->cfg.max_height / 8 + 15) / 16) * ((e->cfg.max_width / 8 + 15) / 16);
      if (e->cfg.max_batch * tiles * ((d.cout + 63) / 64) < 768) L.variant = conv_small_variant(L.variant);
    }
    if (i >= 2 && i <= 5) {   // conv2a .. conv3b at small batch: more, smaller workgroups (same k order: bit-identical)
      const int lvl = i <= 3 ? 2 : 4;                                   // resolution divisor of the layer's output (before pooling)
      const int tiles = ((e->cfg.max_height / lvl + 7) / 8) * ((e->cfg.max_width / lvl + 31) / 32);
      const int wgs = e->cfg.max_batch * tiles * ((d.cout + 63) / 64);
      static const int small_wgs = getenv("DANI_SP_SMALL_WGS") ? atoi(getenv("DANI_SP_SMALL_WGS")) : 512;
      const bool pooled = L.variant == CONV3_ROW32_RELU_POOL;
      if (wgs < small_wgs) L.variant = pooled ? CONV3_ROW32N_RELU_POOL : (2 * wgs < small_wgs ? CONV3_R4_RELU : CONV3_ROW32N_RELU);
    }
    if (i == 1) {   // conv1b computes conv1a itself from the u8 image (conv_mfma.hip, Cfg::F1A); DANI_SP_FUSE_1A=0: the two launches of rounds 1-3
      static const bool fuse_off = getenv("DANI_SP_FUSE_1A") && atoi(getenv("DANI_SP_FUSE_1A")) == 0;
      if (!fuse_off) L.variant = L.variant == CONV3_ROW32_RELU_POOL ? CONV3_ROW32_RELU_POOL_F1A : (L.variant == CONV3_ROW32N_RELU_POOL ? CONV3_ROW32N_RELU_POOL_F1A : L.variant);
    }
    auto pack_plan = [&](ConvLayerPlan& P) -> int {
      P.Cin = d.cin; P.Cout = d.cout; P.ksize = d.k; P.CinPad = d.cin;
      P.CC = conv_variant_cc(P.variant);
      const int nco = conv_variant_nco(P.variant);
      P.cout_tiles = (d.cout + nco - 1) / nco;
      P.w_packed_floats = (size_t)P.cout_tiles * nco * d.cin * d.k * d.k;
      std::vector<float> wp(P.w_packed_floats), bp((size_t)P.cout_tiles * nco);
      pack_conv_weights(wi->second.data(), bi->second.data(), d.cin, d.cout, d.k, P.CC, nco, wp.data(), bp.data());
      HIP_TRY(e, hipMalloc((void**)&P.w_packed, wp.size() * 4)); e->weight_allocs.push_back(P.w_packed);
      HIP_TRY(e, hipMalloc((void**)&P.bias_packed, bp.size() * 4)); e->weight_allocs.push_back(P.bias_packed);
      HIP_TRY(e, hipMemcpy(P.w_packed, wp.data(), wp.size() * 4, hipMemcpyHostToDevice));
      HIP_TRY(e, hipMemcpy(P.bias_packed, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
      return 0;
    };
    if (i == 9 || i == 11) {
      // The heads' last layers finish the map in their epilogue (conv_mfma.hip, Cfg::EPI: softmax + depth-to-space / L2 norm + NHWC)
      // when the batch fills the chip with 128-cell workgroups that hold every output channel; one or two frames keep the plain
      // layers (32 couts per workgroup, 8x the workgroups) + softmax_d2s_kernel / desc_norm_kernel.  DANI_SP_FUSE_HEADS=0/1 forces.
      const int tiles = ((e->cfg.max_height / 8 + 7) / 8) * ((e->cfg.max_width / 8 + 15) / 16);
      const char* fh = getenv("DANI_SP_FUSE_HEADS");
      e->fused_heads = fh ? atoi(fh) != 0 : e->cfg.max_batch * tiles >= 192;
      if (e->fused_heads) {
        ConvLayerPlan& P = i == 9 ? e->plan_plain_pb : e->plan_plain_db;
        P.variant = L.variant;
        { int rc = pack_plan(P); if (rc) return rc; }
        L.variant = i == 9 ? CONV1_BLK16S_SOFTMAX_D2S : CONV1_BLK16S_L2NORM_NHWC;
      }
    }
    { int rc = pack_plan(L); if (rc) return rc; }
  }
  { int rc = lg_finalize(e); if (rc) return rc; }
  { int rc = yolo_finalize(e); if (rc) return rc; }
  e->finalized = true;
  return DANI_OK;
}

int dani_set_profiling(dani_engine* e, int on) {
  if (!e) return DANI_ERR_INVALID;
  e->profiling = on != 0;
  return DANI_OK;
}

}  // extern "C"

namespace dani {
namespace eng {

int sp_extract_impl(dani_engine* e, const SpCall& call, const uint8_t* imgs, int32_t imgs_on_device, int32_t B, int32_t W, int32_t H,
                    int32_t stride, const dani_sp_params* p, const int32_t* boxes, const int32_t* box_offsets,
                    const int32_t lapping[2], float* kpts_xy, float* resp, int32_t* octave, float* size,
                    float* desc, int32_t cap, int32_t* n_out, int32_t* mono_index) {
  if (!e) return DANI_ERR_INVALID;
  if (!imgs || W <= 0 || H <= 0) return fail(e, DANI_ERR_EMPTY, "empty image");
  if (!e->finalized) return fail(e, DANI_ERR_WEIGHTS, "weights not finalized (call dani_finalize_weights)");
  if (B < 1 || B > e->cfg.max_batch) return fail(e, DANI_ERR_INVALID, "batch %d outside [1, %d]", B, e->cfg.max_batch);
  if (stride < W || cap < 1 || !n_out || !kpts_xy || !resp || (!desc && !call.keep_desc_on_device)) return fail(e, DANI_ERR_INVALID, "bad argument");
  int rc = check_params(e, W, H, p);
  if (rc) return rc;
  HIP_TRY(e, hipSetDevice(e->device));
  CtxLease lease(e, call.ctx);
  Context* c = lease.c;
  if (!c) return fail(e, DANI_ERR_INVALID, "all %d contexts busy (raise dani_config.n_contexts)", e->cfg.n_contexts);
  t_last_ctx = c;
  std::vector<int> per_level; std::vector<float> scales;
  features_per_level(p->nfeatures, p->scale_factor, p->nlevels, per_level, scales);
  const int NL = p->nlevels;
  int kcap = 8;
  for (int l = 0; l < NL; l++) kcap += per_level[l] + 3;  // DistributeOctTree can exceed N by at most 3 per level
  if ((rc = ensure_shapes(e, c, W, H, e->cfg.max_batch, std::max(kcap, c->kcap), NL, p->scale_factor))) return rc;
  hipStream_t s = c->stream;
  const int MB = e->cfg.max_batch;

  const bool prof = e->profiling;
  const int phase = call.phase;
  // candidate NMS per cell + compaction + read-back (counts and, optimistically, the first CAND_FAST candidates of
  // every frame in the same stream pass: one sync instead of two; frames with more get a second copy below)
  auto enqueue_candidates = [&](const int32_t* d_boxes, const int32_t* d_box_offsets) -> int {
    for (int l = 0; l < NL; l++) {
      Context::Level& L = c->lv[l];
      const int PH = (L.H / 8) * 8, PW = (L.W / 8) * 8, ncells = (int)L.cells.size();
      if (ncells > 0) {
        launch_cell_nms(L.prob, PH, PW, B, L.d_cells, ncells, p->ini_th, p->min_th, d_boxes, d_box_offsets,
                        19 - 3, scales[l], L.slot_xyr, L.slot_count, L.slot_stride, s);
        launch_cell_compact(L.slot_xyr, L.slot_count, ncells, L.slot_stride, B, L.cand, L.cand_count, L.cand_cap, s);
        HIP_TRY(e, hipMemcpyAsync(L.h_cand_count, L.cand_count, (size_t)B * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(e, hipMemcpy2DAsync(L.h_cand, (size_t)L.cand_cap * 12, L.cand, (size_t)L.cand_cap * 12, (size_t)CAND_FAST * 12, B, hipMemcpyDeviceToHost, s));
      }
    }
    return 0;
  };
  if (phase != 2 && phase != 3) {
    c->cands_enqueued = false;
    c->spec_valid = false;
    // ---- inputs
    const uint8_t* dimg = imgs;
    const size_t fbytes = (size_t)H * stride;
    if (!imgs_on_device) {
      if (fbytes * B > c->d_img_bytes) return fail(e, DANI_ERR_INVALID, "stride %d too large for the staging buffer", stride);
      HIP_TRY(e, hipMemcpyAsync(c->d_img, imgs, fbytes * B, hipMemcpyHostToDevice, s));
      dimg = c->d_img;
    }
    // ---- pyramid + network, level by level
    HIP_TRY(e, hipEventRecord(c->ev[0], s));
    for (int l = 0; l < NL; l++) {
      Context::Level& L = c->lv[l];
      const uint8_t* limg = dimg; int lstride = stride; size_t lframe = fbytes;
      if (l > 0) {
        const Context::Level& P = c->lv[l - 1];
        const uint8_t* src = (l == 1) ? dimg : P.d_level_img;
        const int sstride = (l == 1) ? stride : P.W;
        const size_t sframe = (l == 1) ? fbytes : (size_t)P.W * P.H;
        launch_resize_u8c1(src, B, P.W, P.H, sstride, sframe, L.d_level_img, L.W, L.H, s);
        limg = L.d_level_img; lstride = L.W; lframe = (size_t)L.W * L.H;
      }
      const bool lprof = prof && l == 0;
      const bool fused1a = e->plan[1].variant == CONV3_ROW32_RELU_POOL_F1A || e->plan[1].variant == CONV3_ROW32N_RELU_POOL_F1A;
      L.last_img = limg; L.last_stride = lstride; L.last_frame = lframe; L.last_B = B; L.act0_valid = !fused1a;
      const bool fheads = e->fused_heads;
      L.heads_valid = !fheads;
      if (lprof) HIP_TRY(e, hipEventRecord(c->lev[0], s));
      if (!fused1a) launch_conv1a(limg, B, L.H, L.W, lstride, lframe, e->w1a, e->b1a, L.act[0], s);   // fused: conv1b forms it on the fly
      if (lprof) HIP_TRY(e, hipEventRecord(c->lev[1], s));
      const int src_of[12] = {-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 7, 10};
      // One or two frames: the detector head (convPa, convPb, softmax) and the descriptor head (convDa, convDb, norm) are
      // independent branches behind conv4b and each launches fewer workgroups than the chip has slots for; when the context's
      // second stream is idle (no YOLO beside this call) the descriptor branch runs there, next to the detector branch.
      static const bool fork_off = getenv("DANI_SP_NO_HEAD_FORK") != nullptr;
      const bool fork = !fork_off && B <= 2 && !lprof && (phase == 0 || call.side_free) && !call.wait_event;
      for (int i = 1; i < 12; i++) {
        if (call.wait_event && l == 0 && i == call.wait_layer) HIP_TRY(e, hipStreamWaitEvent(s, call.wait_event, 0));
        const int reserve = (call.reserve_lds_mask >> i) & 1 ? call.reserve_lds_bytes : 0;
        hipStream_t si = (fork && i >= 10) ? c->stream2 : s;
        if (fork && i == 8) {   // conv4b is enqueued: the descriptor branch may start behind it
          HIP_TRY(e, hipEventRecord(c->ev_fork0, s));
          HIP_TRY(e, hipStreamWaitEvent(c->stream2, c->ev_fork0, 0));
        }
        const int Hc8 = (L.H / 8) * 8, Wc8 = (L.W / 8) * 8;
        const bool launched = (i == 1 && fused1a)
            ? launch_conv_mfma_f1a(e->plan[1], limg, lstride, lframe, e->w1a, e->b1a, L.act[0], L.act[1], B, si, reserve)
            : (fheads && i == 9)  ? launch_conv_head(e->plan[9], L.act[8], L.prob, (size_t)Hc8 * Wc8, B, si, reserve)
            : (fheads && i == 11) ? launch_conv_head(e->plan[11], L.act[10], L.desc_nhwc, (size_t)(Hc8 / 8) * (Wc8 / 8) * 256, B, si, reserve)
            : launch_conv_mfma(e->plan[i], L.act[src_of[i]], L.act[i], B, si, reserve);
        if (!launched) {
          (void)hipStreamSynchronize(s);
          if (fork) (void)hipStreamSynchronize(c->stream2);
          return fail(e, DANI_ERR_UNSUPPORTED, "%s at %dx%d: activation tensor exceeds the conv kernel's 32-bit addressing range", kSp[i].name, L.W, L.H);
        }
        if (lprof) HIP_TRY(e, hipEventRecord(c->lev[i + 1], s));
        if (i == 9 && l == NL - 1 && si == s) near_mark(s);   // the detector head is enqueued: the candidate read-back is ~0.4 ms of GPU work away
      }
      if (l == NL - 1) HIP_TRY(e, hipEventRecord(c->ev[1], s));   // end of the main stream's convolutions, on both paths
      if (fork) {
        if (!fheads) launch_desc_norm(L.act[11], L.desc_nhwc, L.desc_nchw, B, c->stream2);
        HIP_TRY(e, hipEventRecord(c->ev_fork1, c->stream2));
        if (!fheads) launch_softmax_d2s(L.act[9], L.prob, B, s);
        HIP_TRY(e, hipStreamWaitEvent(s, c->ev_fork1, 0));   // everything later on the main stream sees both branches
      } else if (!fheads) {
        launch_softmax_d2s(L.act[9], L.prob, B, s);
        launch_desc_norm(L.act[11], L.desc_nhwc, L.desc_nchw, B, s);
      }
    }
    HIP_TRY(e, hipGetLastError());
    if (phase == 1) {   // network enqueued; the caller continues with phase 2 once the boxes are known
      if (call.host_cull && NL == 1) {   // scale 1: the cull is an exact filter, so the candidates need not wait for the boxes
        if ((rc = enqueue_candidates(nullptr, nullptr))) return rc;
        c->cands_enqueued = true;
      }
      return DANI_OK;
    }
  }
  const int minBorderX = 19 - 3, minBorderY = minBorderX;
  // wait for the candidate read-back; frames with more than CAND_FAST candidates get their remainder in a second copy
  auto fetch_candidates = [&]() -> int {
    // the quadtree tasks start as soon as this wait returns: the pool's workers are woken when its spinning stage begins
    HIP_TRY(e, stream_wait(s, [](void*) { host_pool().prewake(); }, nullptr));
    bool more = false;
    for (int l = 0; l < NL; l++) {
      Context::Level& L = c->lv[l];
      for (int b = 0; b < B; b++) {
        int nc = L.cells.empty() ? 0 : L.h_cand_count[b];
        if (nc < 0 || nc > L.cand_cap)   // more candidates than cell pixels: memory corruption, not a capacity case
          return fail(e, DANI_ERR_HIP, "frame %d level %d: candidate counter %d outside [0, %d]", b, l, nc, L.cand_cap);
        L.h_cand_count[b] = nc;
        if (nc > CAND_FAST) {
          more = true;
          HIP_TRY(e, hipMemcpyAsync(L.h_cand + ((size_t)b * L.cand_cap + CAND_FAST) * 3, L.cand + ((size_t)b * L.cand_cap + CAND_FAST) * 3, (size_t)(nc - CAND_FAST) * 12, hipMemcpyDeviceToHost, s));
        }
      }
    }
    if (more) HIP_TRY(e, hipStreamSynchronize(s));
    return 0;
  };
  if (phase == 3) {
    // Speculation (fused single-level call): the candidates are on the host before YOLO's chain has finished, and the
    // host is idle until it has.  The dynamic-box cull (src/SPextractor.cpp:314-352) only REMOVES candidates, so for a frame
    // whose boxes turn out to contain none of its candidates -- every frame without a dynamic object -- the quadtree
    // over the un-culled list is the final one.  Phase 2 checks that per frame and redoes the others.
    c->spec_valid = false;
    if (!c->cands_enqueued || NL != 1) return DANI_OK;
    if ((rc = fetch_candidates())) return rc;
    trace_mark("sp_candidates_on_host");
    if ((int)c->spec_sels.size() < B) c->spec_sels.resize(B);
    std::vector<char> okf(B, 1);
    host_pool().run(B, [&](int b) {
      const Context::Level& L = c->lv[0];
      const HostKey* keys = reinterpret_cast<const HostKey*>(L.h_cand + (size_t)b * L.cand_cap * 3);
      okf[b] = distribute_octtree(keys, L.h_cand_count[b], minBorderX, L.W - 19 + 3, minBorderY, L.H - 19 + 3, per_level[0], c->spec_sels[b]);
    });
    bool all_ok = true;
    for (int b = 0; b < B; b++) all_ok = all_ok && okf[b];
    c->spec_valid = all_ok;
    return DANI_OK;
  }
  int nboxes_total = 0;
  if (boxes && box_offsets) nboxes_total = box_offsets[B];
  if (nboxes_total > c->box_cap) return fail(e, DANI_ERR_INVALID, "too many dynamic boxes (%d)", nboxes_total);
  if (!c->cands_enqueued) {
    int32_t* hb = c->h_boxes;
    if (nboxes_total) memcpy(hb, boxes, (size_t)nboxes_total * 16);
    int32_t* ho = hb + c->box_cap * 4;
    for (int b = 0; b <= B; b++) ho[b] = (boxes && box_offsets) ? box_offsets[b] : 0;
    if (nboxes_total) HIP_TRY(e, hipMemcpyAsync(c->d_boxes, hb, (size_t)nboxes_total * 16, hipMemcpyHostToDevice, s));
    HIP_TRY(e, hipMemcpyAsync(c->d_box_offsets, ho, (size_t)(B + 1) * 4, hipMemcpyHostToDevice, s));
  }
  if (!c->cands_enqueued && (rc = enqueue_candidates(c->d_boxes, c->d_box_offsets))) return rc;
  HIP_TRY(e, hipEventRecord(c->ev[2], s));
  HIP_TRY(e, hipGetLastError());
  const bool speculated = c->spec_valid && c->cands_enqueued && NL == 1;
  c->spec_valid = false;
  if (!speculated && (rc = fetch_candidates())) return rc;

  // ---- host: quadtree per (frame, level) + keypoint assembly (SPextractor.cpp:359-376, :164-216), frames in parallel
  const auto t0 = std::chrono::steady_clock::now();
  int32_t* h_lv = reinterpret_cast<int32_t*>(c->h_kxy + (size_t)MB * c->kcap * 2);   // [level][0: offset, 1: count][MB]
  std::vector<int> frame_status(B, DANI_OK);
  const bool host_cull = c->cands_enqueued;
  // one task per frame: (cull,) quadtree of every level, then the keypoint assembly of that frame
  host_pool().run(B, [&](int b) {
    std::vector<std::vector<int>> sels(NL);
    for (int l = 0; l < NL; l++) {
      const Context::Level& L = c->lv[l];
      HostKey* keys = reinterpret_cast<HostKey*>(L.h_cand + (size_t)b * L.cand_cap * 3);
      bool culled = false;
      if (host_cull && boxes && box_offsets && box_offsets[b + 1] > box_offsets[b]) {
        // dynamic-box cull of SPextractor.cpp:314-352 at scale 1: (p + border) tested with Rect::contains (left / top
        // inclusive, right / bottom exclusive, saturate_cast<int> = round half to even); the round trip is exact, so
        // one pass over the finished list equals the reference's pass after every cell.  Order is preserved.
        int m = 0;
        for (int i = 0; i < L.h_cand_count[b]; i++) {
          const int px = (int)lrintf(keys[i].x + (float)minBorderX), py = (int)lrintf(keys[i].y + (float)minBorderY);
          bool inside = false;
          for (int bi = box_offsets[b]; bi < box_offsets[b + 1] && !inside; bi++) {
            const int32_t* r = boxes + 4 * bi;
            inside = r[0] <= px && px < r[0] + r[2] && r[1] <= py && py < r[1] + r[3];
          }
          if (!inside) keys[m++] = keys[i];
        }
        culled = m != L.h_cand_count[b];
        L.h_cand_count[b] = m;
      }
      if (speculated && !culled) { sels[l].swap(c->spec_sels[b]); continue; }   // nothing was culled: the speculative quadtree stands
      if (!distribute_octtree(keys, L.h_cand_count[b], minBorderX, L.W - 19 + 3, minBorderY, L.H - 19 + 3, per_level[l], sels[l]))
        frame_status[b] = DANI_ERR_UNSUPPORTED;   // cannot happen for levels ensure_shapes accepted and in-cell candidates
    }
    int n = 0;
    for (int l = 0; l < NL; l++) n += (int)sels[l].size();
    n_out[b] = n;
    const bool fits = n <= cap && n <= c->kcap;
    if (!fits && frame_status[b] == DANI_OK) frame_status[b] = DANI_ERR_CAPACITY;
    float* kx = c->h_kxy + (size_t)b * c->kcap * 2;
    float* ki = c->h_kimg + (size_t)b * c->kcap * 2;
    float* ok = kpts_xy + (size_t)b * cap * 2;
    float* orr = resp + (size_t)b * cap;
    // the mono / stereo partition runs over the levels in order (operator(), :175-213); descriptor rows keep the
    // pre-partition order
    int monoIndex = 0, stereoIndex = n - 1, off = 0;
    const int l0 = lapping ? lapping[0] : 0, l1 = lapping ? lapping[1] : 0;
    for (int l = 0; l < NL; l++) {
      const Context::Level& L = c->lv[l];
      const HostKey* keys = reinterpret_cast<const HostKey*>(L.h_cand + (size_t)b * L.cand_cap * 3);
      const std::vector<int>& sel = sels[l];
      const int nl = fits ? (int)sel.size() : 0;
      h_lv[(l * 2 + 0) * MB + b] = off;
      h_lv[(l * 2 + 1) * MB + b] = nl;
      const int scaledPatchSize = (int)(31 * scales[l]);
      for (int i = 0; i < nl; i++) {
        const float x = keys[sel[i]].x + minBorderX, y = keys[sel[i]].y + minBorderY;   // level coordinates (:369-370)
        kx[2 * (off + i)] = x; kx[2 * (off + i) + 1] = y;
        const float fx = l ? x * scales[l] : x, fy = l ? y * scales[l] : y;              // keypoint->pt *= scale (:190-192)
        const int dst = (fx >= (float)l0 && fx <= (float)l1) ? stereoIndex-- : monoIndex++;
        ok[2 * dst] = fx; ok[2 * dst + 1] = fy; orr[dst] = keys[sel[i]].r;
        ki[2 * dst] = fx; ki[2 * dst + 1] = fy;
        if (octave) octave[(size_t)b * cap + dst] = l;
        if (size) size[(size_t)b * cap + dst] = (float)scaledPatchSize;
      }
      off += nl;
    }
    if (mono_index) mono_index[b] = monoIndex;
  });
  trace_mark("cull_quadtree_assembly_done");
  c->cands_enqueued = false;
  int status = DANI_OK;
  for (int b = 0; b < B; b++) if (frame_status[b] != DANI_OK) status = frame_status[b];
  c->ms[2] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (status == DANI_ERR_CAPACITY) return fail(e, status, "output capacity %d too small", cap);
  if (status != DANI_OK) return fail(e, status, "DistributeOctTree: region or candidate outside what the reference defines");

  // ---- descriptors: each level samples its own map and is normalised over its own keypoints (Q7)
  HIP_TRY(e, hipMemcpyAsync(c->d_kxy, c->h_kxy, ((size_t)MB * c->kcap * 2 + (size_t)NL * 2 * MB) * 4, hipMemcpyHostToDevice, s));
  if (call.keep_desc_on_device)   // fused call: LightGlue reads the image-space keypoints on the device
    HIP_TRY(e, hipMemcpyAsync(c->d_kimg, c->h_kimg, (size_t)B * c->kcap * 2 * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(e, hipEventRecord(c->ev[3], s));
  for (int l = 0; l < NL; l++) {
    const Context::Level& L = c->lv[l];
    const int Hc = L.H / 8, Wc = L.W / 8;
    int maxn = 0;
    for (int b = 0; b < B; b++) maxn = std::max(maxn, h_lv[(l * 2 + 1) * MB + b]);
    if (maxn == 0) continue;
    launch_descriptors(L.desc_nhwc, Hc, Wc, Hc * 8, Wc * 8, c->d_kxy, c->d_nk + (size_t)(l * 2 + 0) * MB, c->d_nk + (size_t)(l * 2 + 1) * MB, B, c->kcap,
                       maxn, c->d_sampled, c->d_descout, s);
  }
  HIP_TRY(e, hipEventRecord(c->ev[4], s));
  trace_mark("desc_kernels_enqueued");
  if (desc) {   // descriptor read-back on the second stream so that it can run under the next stage's kernels
    HIP_TRY(e, hipEventRecord(c->ev_desc, s));
    HIP_TRY(e, hipStreamWaitEvent(c->stream2, c->ev_desc, 0));
    // fused call: into pinned staging (a copy into pageable memory would block this thread until it is done); the
    // caller moves it to `desc` while LightGlue runs
    if (!(call.defer_desc_sync && call.defer_desc_copies))
      for (int b = 0; b < B; b++)
        if (n_out[b])
          HIP_TRY(e, hipMemcpyAsync(call.defer_desc_sync ? c->h_desc + (size_t)b * c->kcap * 256 : desc + (size_t)b * cap * 256,
                                    c->d_descout + (size_t)b * c->kcap * 256, (size_t)n_out[b] * 1024, hipMemcpyDeviceToHost, c->stream2));
    if (!call.defer_desc_sync) HIP_TRY(e, hipStreamSynchronize(c->stream2));
  }
  HIP_TRY(e, hipGetLastError());
  if (call.defer_desc_sync) return DANI_OK;   // the fused call synchronises (and reads the timers) after LightGlue
  HIP_TRY(e, hipStreamSynchronize(s));
  sp_read_timers(c, prof);
  return DANI_OK;
}

// the fused call's descriptor read-back into the pinned staging buffer (SpCall::defer_desc_copies): stream2 already waits for ev_desc
int sp_enqueue_desc_readback(dani_engine* e, Context* c, const int32_t* n_out, int B) {
  for (int b = 0; b < B; b++)
    if (n_out[b])
      HIP_TRY(e, hipMemcpyAsync(c->h_desc + (size_t)b * c->kcap * 256, c->d_descout + (size_t)b * c->kcap * 256, (size_t)n_out[b] * 1024,
                                hipMemcpyDeviceToHost, c->stream2));
  return DANI_OK;
}

}  // namespace eng
}  // namespace dani

extern "C" {

int dani_sp_extract_batch(dani_engine* e, const uint8_t* imgs, int32_t imgs_on_device, int32_t B, int32_t W, int32_t H,
                          int32_t stride, const dani_sp_params* p, const int32_t* boxes, const int32_t* box_offsets,
                          const int32_t lapping[2], float* kpts_xy, float* resp, int32_t* octave, float* size,
                          float* desc, int32_t cap, int32_t* n_out, int32_t* mono_index) {
  return sp_extract_impl(e, SpCall{}, imgs, imgs_on_device, B, W, H, stride, p, boxes, box_offsets, lapping, kpts_xy, resp, octave,
                         size, desc, cap, n_out, mono_index);
}

int dani_sp_extract(dani_engine* e, const uint8_t* img, int32_t W, int32_t H, int32_t stride, const dani_sp_params* p,
                    const int32_t* boxes, int32_t nboxes, const int32_t lapping[2], float* kpts_xy, float* resp,
                    int32_t* octave, float* size, float* desc, int32_t cap, int32_t* n_out, int32_t* mono_index) {
  const int32_t offs[2] = {0, nboxes};
  return dani_sp_extract_batch(e, img, 0, 1, W, H, stride, p, boxes, (boxes && nboxes > 0) ? offs : nullptr, lapping,
                               kpts_xy, resp, octave, size, desc, cap, n_out, mono_index);
}

int dani_sp_debug_fetch(dani_engine* e, int32_t b, float* prob, float* desc_map, float* cand_xyr, int32_t cand_cap,
                        int32_t* n_cand) {
  Context* c = last_ctx(e);
  if (!e || !c || b < 0 || b >= c->B) return fail(e, DANI_ERR_INVALID, "no previous call on this thread");
  HIP_TRY(e, hipSetDevice(e->device));
  if (c->lv.empty()) return fail(e, DANI_ERR_INVALID, "no previous call on this thread");
  const char* lvl_env = getenv("DANI_DEBUG_LEVEL");   // tests: inspect another pyramid level
  const int dl = lvl_env ? std::min(std::max(atoi(lvl_env), 0), (int)c->lv.size() - 1) : 0;
  const Context::Level& L = c->lv[dl];
  const int Hc = L.H / 8, Wc = L.W / 8;
  if (prob) HIP_TRY(e, hipMemcpy(prob, L.prob + (size_t)b * Hc * 8 * Wc * 8, (size_t)Hc * 8 * Wc * 8 * 4, hipMemcpyDeviceToHost));
  if (desc_map && !e->fused_heads) HIP_TRY(e, hipMemcpy(desc_map, L.desc_nchw + (size_t)b * 256 * Hc * Wc, (size_t)256 * Hc * Wc * 4, hipMemcpyDeviceToHost));
  if (desc_map && e->fused_heads) {
    // fused heads: the normalised map exists as [cell][256] only (what the sampler reads); the hook's [256][cell] view is its transpose
    const size_t npix = (size_t)Hc * Wc;
    std::vector<float> t(npix * 256);
    HIP_TRY(e, hipMemcpy(t.data(), L.desc_nhwc + (size_t)b * npix * 256, npix * 256 * 4, hipMemcpyDeviceToHost));
    for (size_t p = 0; p < npix; p++)
      for (int ch = 0; ch < 256; ch++) desc_map[(size_t)ch * npix + p] = t[p * 256 + ch];
  }
  if (n_cand) *n_cand = L.h_cand_count[b];
  if (cand_xyr) {
    const int n = std::min(L.h_cand_count[b], cand_cap);
    memcpy(cand_xyr, L.h_cand + (size_t)b * L.cand_cap * 3, (size_t)n * 12);
  }
  return DANI_OK;
}

int dani_sp_debug_layer(dani_engine* e, int32_t b, int32_t layer, float* out, int64_t count) {
  Context* c = last_ctx(e);
  if (!e || !c || b < 0 || b >= c->B || layer < 0 || layer >= 12 || !out) return fail(e, DANI_ERR_INVALID, "bad argument");
  if (c->lv.empty()) return fail(e, DANI_ERR_INVALID, "no previous call on this thread");
  const ActDesc& a = c->lv[0].act[layer];
  const size_t need = (size_t)a.C * a.H * a.W;
  if ((size_t)count < need) return fail(e, DANI_ERR_CAPACITY, "layer %d needs %zu floats", layer, need);
  HIP_TRY(e, hipSetDevice(e->device));
  if (layer == 0 && !c->lv[0].act0_valid) {
    // conv1a's activation never reaches HBM when conv1b computes it on the fly: the test hook runs the stand-alone kernel on the
    // last call's image (still in the staging buffer, or at the caller's device pointer)
    Context::Level& L0 = c->lv[0];
    if (!L0.last_img || b >= L0.last_B) return fail(e, DANI_ERR_INVALID, "no previous call on this thread");
    launch_conv1a(L0.last_img, L0.last_B, L0.H, L0.W, L0.last_stride, L0.last_frame, e->w1a, e->b1a, L0.act[0], c->stream);
    HIP_TRY(e, hipGetLastError());
    L0.act0_valid = true;
  }
  if ((layer == 9 || layer == 11) && !c->lv[0].heads_valid) {
    // the heads' raw last layers never reach HBM when their epilogues finish the maps: the hook runs the plain layers on the
    // last call's convPa / convDa activations (still in place)
    Context::Level& L0 = c->lv[0];
    if (b >= L0.last_B) return fail(e, DANI_ERR_INVALID, "no previous call on this thread");
    if (!launch_conv_mfma(e->plan_plain_pb, L0.act[8], L0.act[9], L0.last_B, c->stream) ||
        !launch_conv_mfma(e->plan_plain_db, L0.act[10], L0.act[11], L0.last_B, c->stream))
      return fail(e, DANI_ERR_UNSUPPORTED, "debug layer: plain head layers refused the shape");
    HIP_TRY(e, hipGetLastError());
    L0.heads_valid = true;   // (desc_nchw stays unwritten: dani_sp_debug_fetch transposes the fused map instead)
  }
  float* tmp = nullptr;
  HIP_TRY(e, hipMalloc((void**)&tmp, need * 4));
  launch_copy_layer_dense(a, b, tmp, c->stream);
  hipError_t st = hipMemcpyAsync(out, tmp, need * 4, hipMemcpyDeviceToHost, c->stream);
  if (st == hipSuccess) st = hipStreamSynchronize(c->stream);
  (void)hipFree(tmp);
  if (st != hipSuccess) return fail(e, DANI_ERR_HIP, "debug layer copy failed: %s", hipGetErrorString(st));
  return DANI_OK;
}

int dani_sp_layer_kernel(dani_engine* e, int32_t layer, char* name, int32_t cap) {
  if (!e || !name || cap < 1) return DANI_ERR_INVALID;
  if (layer < 0 || layer >= 12) return fail(e, DANI_ERR_INVALID, "layer %d outside [0, 12)", layer);
  if (!e->finalized) return fail(e, DANI_ERR_WEIGHTS, "weights not finalized (call dani_finalize_weights)");
  const bool fused1a = e->plan[1].variant == CONV3_ROW32_RELU_POOL_F1A || e->plan[1].variant == CONV3_ROW32N_RELU_POOL_F1A;
  const char* nm = layer == 0 ? (fused1a ? "(computed inside conv1b)" : "conv1a_kernel") : conv_kernel_name(e->plan[layer].variant);
  snprintf(name, (size_t)cap, "%s", nm);
  return DANI_OK;
}

int dani_last_timings(dani_engine* e, float* ms, int32_t n) {
  Context* c = last_ctx(e);
  if (!e || !c || !ms) return DANI_ERR_INVALID;
  for (int i = 0; i < n && i < 5; i++) ms[i] = c->ms[i];
  return DANI_OK;
}

int dani_layer_timings(dani_engine* e, float* ms12) {
  Context* c = last_ctx(e);
  if (!e || !c || !ms12) return DANI_ERR_INVALID;
  for (int i = 0; i < 12; i++) ms12[i] = c->layer_ms[i];
  return DANI_OK;
}

// ---- host logic, callable without a GPU ----------------------------------------------------------
int dani_host_distribute_octtree(const float* in_xyr, int32_t n, int32_t min_x, int32_t max_x, int32_t min_y,
                                 int32_t max_y, int32_t n_features, int32_t* out_idx, int32_t cap) {
  if (n < 0 || (n && !in_xyr) || !out_idx) return DANI_ERR_INVALID;
  std::vector<int> sel;
  if (!distribute_octtree(reinterpret_cast<const HostKey*>(in_xyr), n, min_x, max_x, min_y, max_y, n_features, sel))
    return DANI_ERR_UNSUPPORTED;   // round(width / height) < 1, or a key outside [0, max_x - min_x): undefined in the reference
  if ((int)sel.size() > cap) return DANI_ERR_CAPACITY;
  for (size_t i = 0; i < sel.size(); i++) out_idx[i] = sel[i];
  return (int)sel.size();
}

int dani_host_cell_grid(int32_t width, int32_t height, int32_t* cells6, int32_t cap) {
  std::vector<CellRect> cells;
  const int n = cell_grid(width, height, cells);
  for (int i = 0; i < n && i < cap; i++) memcpy(cells6 + 6 * i, &cells[i], 24);
  return n;
}

int dani_host_assign_features_to_grid(const float* kpts_xy, int32_t n, const dani_grid* g, int32_t* cell_offsets, int32_t* cell_idx) {
  if (n < 0 || (n && !kpts_xy) || !g || g->cols < 1 || g->rows < 1 || !cell_offsets || (n && !cell_idx)) return DANI_ERR_INVALID;
  assign_features_to_grid(kpts_xy, n, g->min_x, g->min_y, g->inv_w, g->inv_h, g->cols, g->rows, cell_offsets, cell_idx);
  return cell_offsets[g->cols * g->rows];
}

int dani_host_features_in_area(const float* kpts_xy, const int32_t* octave, int32_t n, const dani_grid* g, const int32_t* cell_offsets,
                               const int32_t* cell_idx, const float* qx, const float* qy, const float* qr, const int32_t* min_level,
                               const int32_t* max_level, int32_t nq, int32_t* cand_offsets, int32_t* cand_idx, int32_t cap) {
  if (n < 0 || nq < 0 || !g || !cell_offsets || (n && (!kpts_xy || !cell_idx)) || (nq && (!qx || !qy || !qr)) || !cand_offsets || cap < 0)
    return DANI_ERR_INVALID;
  std::vector<int32_t> out;
  cand_offsets[0] = 0;
  for (int q = 0; q < nq; q++) {
    features_in_area(kpts_xy, octave, g->min_x, g->min_y, g->inv_w, g->inv_h, g->cols, g->rows, cell_offsets, cell_idx, qx[q], qy[q], qr[q],
                     min_level ? min_level[q] : -1, max_level ? max_level[q] : -1, out);
    cand_offsets[q + 1] = (int32_t)out.size();
  }
  if ((int)out.size() > cap) return DANI_ERR_CAPACITY;
  if (!out.empty()) memcpy(cand_idx, out.data(), out.size() * 4);
  return (int)out.size();
}

int dani_host_stereo_from_matches(const float* ul_x, int32_t nl, const float* ur_x, int32_t nr, const int32_t* matches,
                                  const float* scores, int32_t k, float mb, float mbf, float* u_right, float* depth) {
  if (nl < 0 || nr < 0 || k < 0 || (nl && (!ul_x || !u_right || !depth)) || (nr && !ur_x) || (k && (!matches || !scores)) || !(mb > 0.0f))
    return DANI_ERR_INVALID;
  stereo_from_matches(ul_x, nl, ur_x, nr, matches, scores, k, mb, mbf, u_right, depth);
  return DANI_OK;
}

int dani_host_undistort_keypoints(const float* kpts_xy, int32_t n, const float* K4, const float* dist, int32_t ndist, float* out_xy) {
  if (n < 0 || !K4 || (n && (!kpts_xy || !out_xy)) || ndist < 0 || ndist > 5 || (ndist && !dist) || !(K4[0] != 0.0f) || !(K4[1] != 0.0f))
    return DANI_ERR_INVALID;
  undistort_keypoints(kpts_xy, n, K4, dist, ndist, out_xy);
  return DANI_OK;
}

int dani_host_features_per_level(int32_t nfeatures, float scale_factor, int32_t nlevels, int32_t* per_level,
                                 float* scale_factors) {
  if (nlevels < 1 || !per_level) return DANI_ERR_INVALID;
  std::vector<int> f; std::vector<float> s;
  features_per_level(nfeatures, scale_factor, nlevels, f, s);
  for (int i = 0; i < nlevels; i++) { per_level[i] = f[i]; if (scale_factors) scale_factors[i] = s[i]; }
  return DANI_OK;
}

}  // extern "C"

