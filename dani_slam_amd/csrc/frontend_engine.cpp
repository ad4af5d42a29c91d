// frontend_engine.cpp -- the fused per-frame front-end (sequence mode) and the projection-search descriptor scans.
#include "engine_internal.h"

#include <cmath>

using namespace dani;
using namespace dani::eng;

// ================================================================================================
// Fused per-frame front-end (sequence mode)
// ================================================================================================
namespace dani {
namespace eng {

// finish a deferred descriptor read-back (wait for the staging copy, move it into the caller's array); S->mu is held
void finish_pending_desc(SeqState* S) {
  if (!S->pend_desc || !S->ctx) { S->pend_desc = nullptr; return; }
  Context* c = S->ctx;
  float* desc = S->pend_desc; const int cap = S->pend_cap, B = S->res_B;
  S->pend_desc = nullptr;
  if (hipStreamSynchronize(c->stream2) != hipSuccess) return;
  host_pool().run(B, [&](int b) {
    if (S->res_n[b]) memcpy(desc + (size_t)b * cap * 256, c->h_desc + (size_t)b * c->kcap * 256, (size_t)S->res_n[b] * 1024);
  });
}
// one SeqState (context + cached previous frame) per (engine, sequence id); id 0 also holds the class lists
SeqState* seq_state(dani_engine* e, int id) {
  std::lock_guard<std::mutex> lk(e->mu);
  auto it = e->seqs.find(id);
  if (it != e->seqs.end()) return it->second;
  SeqState* s = new SeqState();
  e->seqs[id] = s;
  return s;
}
thread_local SeqState* t_last_seq = nullptr;

void seq_release(dani_engine* e) {
  std::lock_guard<std::mutex> lk(e->mu);
  // dani_engine_destroy has synchronised every stream: a descriptor read-back still pending here (defer_desc) is CANCELLED,
  // never copied -- the caller's array may be gone by now (include/dani_frontend.h, dani_frontend_finish_pending)
  for (auto& kv : e->seqs) { if (t_last_seq == kv.second) t_last_seq = nullptr; kv.second->pend_desc = nullptr; delete kv.second; }
  e->seqs.clear();
}

}  // namespace eng
}  // namespace dani


namespace dani {
namespace eng {

// dani_frontend_match_resident with S->mu held (also the tail of dani_frontend_match_exchanged, comm_engine.cpp)
int match_resident_locked(dani_engine* e, SeqState* S, int32_t sequence, const float* pred, int32_t pred_on_device, int32_t pred_cap,
                          float lg_threshold, int32_t* matches, float* scores, int32_t match_cap, int32_t* k_out) {
  Context* c = S->ctx;
  const int B = S->res_B;
  if (!c || B < 1) return fail(e, DANI_ERR_INVALID, "sequence %d holds no resident frames (call dani_frontend_batch first)", sequence);
  HIP_TRY(e, hipSetDevice(e->device));
  t_last_seq = S;
  if (pred) {   // the neighbouring rank's last frame becomes slot 0 of the feature slabs
    float cnt = 0.0f;
    if (pred_on_device) {
      // ordered on the sequence's own stream: a slab that a collective enqueued on this stream is still writing
      // (dani_frontend_exchange_boundary) is complete when the count arrives; a null-stream hipMemcpy would not wait for it
      HIP_TRY(e, hipMemcpyAsync(c->h_boxes, pred, 4, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(e, hipStreamSynchronize(c->stream));
      memcpy(&cnt, c->h_boxes, 4);
    } else cnt = pred[0];
    const int pn = (int)cnt;
    if (!(cnt >= 0.0f) || pn < 0 || pn > pred_cap || pn > c->kcap) return fail(e, DANI_ERR_INVALID, "predecessor slab holds %d keypoints (cap %d, engine %d)", pn, pred_cap, c->kcap);
    const hipMemcpyKind kind = pred_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (pn) {
      HIP_TRY(e, hipMemcpyAsync(c->d_kimg_all, pred + 1, (size_t)pn * 8, kind, c->stream));
      HIP_TRY(e, hipMemcpyAsync(c->d_desc_all, pred + 1 + (size_t)pred_cap * 2, (size_t)pn * 1024, kind, c->stream));
    }
    c->prev_n = pn;
  }
  std::vector<int32_t> n0(B), n1(B);
  for (int b = 0; b < B; b++) { n0[b] = b ? S->res_n[b - 1] : c->prev_n; n1[b] = S->res_n[b]; }
  const float sz[2] = {(float)S->res_W, (float)S->res_H};
  LgCall lg_call;
  lg_call.ctx = c;
  std::function<void()> finish_desc = [&] { finish_pending_desc(S); };   // a deferred descriptor read-back ends under the forward pass
  lg_call.overlap = &finish_desc;
  int rc = lg_match_impl(e, lg_call, B, c->kcap, 1, c->d_kimg_all, c->d_desc_all, n0.data(), c->d_kimg, c->d_descout, n1.data(), sz, sz,
                         lg_threshold, matches, scores, match_cap, k_out);
  finish_pending_desc(S);
  if (rc) return rc;
  HIP_TRY(e, hipMemcpyAsync(c->d_kimg_all, c->d_kimg + (size_t)(B - 1) * c->kcap * 2, (size_t)c->kcap * 2 * 4, hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(e, hipMemcpyAsync(c->d_desc_all, c->d_descout + (size_t)(B - 1) * c->kcap * 256, (size_t)c->kcap * 256 * 4, hipMemcpyDeviceToDevice, c->stream));
  // no host wait: every later use of the cache slabs is enqueued on this same stream
  c->prev_n = S->res_n[B - 1];
  return DANI_OK;
}

}  // namespace eng
}  // namespace dani

extern "C" {

int dani_yolo_set_classes(dani_engine* e, const char* const* class_names, int32_t n_classes,
                          const char* const* dynamic_names, int32_t n_dynamic) {
  if (!e || !class_names || !dynamic_names || n_classes < 1 || n_dynamic < 0) return fail(e, DANI_ERR_INVALID, "bad class list");
  SeqState* S = seq_state(e);
  std::lock_guard<std::mutex> lk(S->mu);
  S->class_names.assign(class_names, class_names + n_classes);
  S->dynamic_names.assign(dynamic_names, dynamic_names + n_dynamic);
  return DANI_OK;
}

int dani_frontend_reset(dani_engine* e, int32_t sequence) {
  if (!e) return DANI_ERR_INVALID;
  SeqState* S = seq_state(e, sequence);
  std::lock_guard<std::mutex> lk(S->mu);
  finish_pending_desc(S);
  if (S->ctx) S->ctx->prev_n = 0;
  return DANI_OK;
}

int dani_frontend_finish_pending(dani_engine* e, int32_t sequence) {
  if (!e) return DANI_ERR_INVALID;
  SeqState* S = seq_state(e, sequence);
  std::lock_guard<std::mutex> lk(S->mu);
  if (S->pend_desc) { HIP_TRY(e, hipSetDevice(e->device)); finish_pending_desc(S); }
  return DANI_OK;
}

int dani_frontend_timings(dani_engine* e, float* ms4) {
  if (!e || !ms4 || !t_last_seq) return DANI_ERR_INVALID;
  {
    std::lock_guard<std::mutex> lk(e->mu);
    bool mine = false;
    for (auto& kv : e->seqs) mine = mine || kv.second == t_last_seq;
    if (!mine) return DANI_ERR_INVALID;
  }
  for (int i = 0; i < 4; i++) ms4[i] = t_last_seq->ms[i];
  return DANI_OK;
}

int dani_frontend_batch(dani_engine* e, int32_t sequence, const uint8_t* gray, int32_t gray_stride, const uint8_t* bgr, int32_t bgr_stride,
                        int32_t on_device, int32_t B, int32_t W, int32_t H, const dani_frontend_params* p, float* kpts_xy,
                        float* resp, float* desc, int32_t cap, int32_t* n_out, int32_t* mono_index, int32_t* dyn_boxes,
                        int32_t box_cap, int32_t* n_boxes, int32_t* matches, float* scores, int32_t match_cap,
                        int32_t* k_out) {
  if (!e || !p) return DANI_ERR_INVALID;
  if (!gray || W <= 0 || H <= 0) return fail(e, DANI_ERR_EMPTY, "empty image");
  if (B < 1 || B > e->cfg.max_batch) return fail(e, DANI_ERR_INVALID, "batch %d outside [1, %d]", B, e->cfg.max_batch);
  if (p->use_yolo && (!bgr || !dyn_boxes || !n_boxes || box_cap < 1)) return fail(e, DANI_ERR_INVALID, "use_yolo needs bgr + box outputs");
  if (p->match_previous && (!matches || !scores || !k_out || match_cap < 1)) return fail(e, DANI_ERR_INVALID, "match_previous needs match outputs");
  if (p->match_previous && p->pair_mode != 0 && (p->pair_mode != 1 || (B & 1))) return fail(e, DANI_ERR_INVALID, "pair_mode 1 (stereo) needs an even number of frames; got pair_mode %d, B %d", p->pair_mode, B);
  SeqState* S0 = seq_state(e, 0);
  SeqState* S = seq_state(e, sequence);
  std::lock_guard<std::mutex> seq_lock(S->mu);
  finish_pending_desc(S);   // a defer_desc call that was never followed by dani_frontend_match_resident
  std::vector<std::string> class_names, dynamic_names;
  if (p->use_yolo) {
    if (S != S0) { std::lock_guard<std::mutex> lk0(S0->mu); class_names = S0->class_names; dynamic_names = S0->dynamic_names; }
    else { class_names = S0->class_names; dynamic_names = S0->dynamic_names; }
    if (class_names.empty()) return fail(e, DANI_ERR_INVALID, "call dani_yolo_set_classes first (coco.names)");
  }
  HIP_TRY(e, hipSetDevice(e->device));
  if (!S->ctx) {
    Context* c = new Context();
    if (context_create_sync(c) != hipSuccess) { delete c; return fail(e, DANI_ERR_HIP, "stream / event creation failed for sequence %d", sequence); }
    c->busy = true;   // never handed out by acquire()
    { std::lock_guard<std::mutex> lk(e->mu); e->ctxs.push_back(c); }
    S->ctx = c;
  }
  Context* c = S->ctx;
  t_last_seq = S;
  trace_begin();
  const auto t0 = std::chrono::steady_clock::now();
  auto ms_since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - a).count(); };

  // ---- 0. enqueue the SuperPoint network first: it does not need the boxes, and running it on the main stream
  //         while YOLO's ~70 small kernels run on the second stream fills the CUs those leave idle
  std::vector<int32_t>& octave = S->octave;
  std::vector<float>& size = S->size;
  if (octave.size() < (size_t)B * cap) { octave.resize((size_t)B * cap); size.resize((size_t)B * cap); }
  SpCall sp_call;
  sp_call.ctx = c; sp_call.keep_desc_on_device = true; sp_call.phase = 1;
  sp_call.host_cull = p->use_yolo && p->sp.nlevels == 1;
  sp_call.side_free = !p->use_yolo;
  if (p->use_yolo) {   // tuning knobs of the slot reservation (defaults below are the measured optimum)
    static const int pad_kb = getenv("DANI_SP_LDS_PAD_KB") ? atoi(getenv("DANI_SP_LDS_PAD_KB")) : 0;
    static const unsigned pad_mask = getenv("DANI_SP_LDS_PAD_MASK") ? (unsigned)strtoul(getenv("DANI_SP_LDS_PAD_MASK"), nullptr, 0) : 0u;
    sp_call.reserve_lds_bytes = pad_kb * 1024; sp_call.reserve_lds_mask = pad_mask;
  }
  // YOLO's first stages (190x320 / 95x160 resolution: its largest kernels) suffer most from sharing the chip with conv1b's
  // 9600 workgroups, and the YOLO chain, not SuperPoint, is the critical path of this phase.  With head > 0 the chain's first
  // `head` stages are enqueued first and SuperPoint's MFMA layers wait for them (conv1a, HBM-bound, does not).
  static const int yolo_head = getenv("DANI_YOLO_HEAD") ? atoi(getenv("DANI_YOLO_HEAD")) : 0;
  const bool head_first = p->use_yolo && yolo_head > 0 && yolo_head <= 7;
  int rc = 0;
  auto enqueue_sp_network = [&]() -> int {
    return sp_extract_impl(e, sp_call, gray, on_device, B, W, H, gray_stride, &p->sp, nullptr, nullptr, p->lapping, kpts_xy, resp,
                           octave.data(), size.data(), desc, cap, n_out, mono_index);
  };
  if (!head_first && (rc = enqueue_sp_network())) return rc;
  trace_mark("sp_enqueued");

  // ---- 1. YOLO -> dynamic boxes (Tracking.cc:1535-1539)
  std::vector<int32_t> boxes, offs(B + 1, 0);
  if (p->use_yolo) {
    YoloCall yolo_call;
    yolo_call.ctx = c; yolo_call.alt_stream = true;
    const int dcap = 4096;
    std::vector<float>& dets = S->dets;
    std::vector<int32_t>& tmp = S->frame_boxes;            // per frame: up to dcap + 1 boxes
    const size_t tstride = (size_t)(dcap + 1) * 4;
    if (dets.size() < (size_t)B * dcap * 6) { dets.resize((size_t)B * dcap * 6); tmp.resize((size_t)B * tstride); }
    std::vector<int32_t> nd(B);
    std::vector<const char*> cn, dn;
    for (auto& s : class_names) cn.push_back(s.c_str());
    for (auto& s : dynamic_names) dn.push_back(s.c_str());
    std::vector<int> kk(B, 0);
    // box bookkeeping + RefineDynamicBoxes run in the same worker task as the frame's NMS
    std::function<void(int, const float*, int)> areas = [&](int b, const float* d6, int n) {
      kk[b] = yolo_dynamic_areas(d6, n, W, H, cn.data(), (int)cn.size(), dn.data(), (int)dn.size(), nullptr, tmp.data() + (size_t)b * tstride, dcap + 1);
    };
    yolo_call.post = &areas;
    // while the YOLO chain is still running: fetch SuperPoint's candidates and run the quadtrees without boxes (sp phase 3)
    std::function<void()> speculate = [&] {
      if (!sp_call.host_cull) return;
      SpCall sc = sp_call; sc.phase = 3;
      (void)sp_extract_impl(e, sc, gray, on_device, B, W, H, gray_stride, &p->sp, nullptr, nullptr, p->lapping, kpts_xy, resp,
                            octave.data(), size.data(), desc, cap, n_out, mono_index);
    };
    yolo_call.before_sync = &speculate;
    int mid_rc = 0;
    std::function<void()> mid = [&] {   // YOLO's head is enqueued: SuperPoint's network follows, its MFMA layers behind the head
      (void)hipEventRecord(c->ev_y0, c->stream2);
      sp_call.wait_event = c->ev_y0; sp_call.wait_layer = 1;
      mid_rc = enqueue_sp_network();
      sp_call.wait_event = nullptr;
    };
    if (head_first) { yolo_call.mid_chain = &mid; yolo_call.mid_after = yolo_head; }
    rc = yolo_detect_impl(e, yolo_call, bgr, on_device, B, W, H, bgr_stride, p->yolo_score_thresh, p->yolo_iou_thresh, dets.data(), dcap, nd.data());
    if (rc) return rc;
    if (mid_rc) return mid_rc;
    trace_mark("yolo_returned");
    for (int b = 0; b < B; b++) {
      const int k = kk[b];
      n_boxes[b] = k;
      if (k > box_cap) { return fail(e, DANI_ERR_CAPACITY, "frame %d has %d dynamic boxes (> %d)", b, k, box_cap); }
      memcpy(dyn_boxes + (size_t)b * box_cap * 4, tmp.data() + (size_t)b * tstride, (size_t)k * 16);
      boxes.insert(boxes.end(), tmp.begin() + (size_t)b * tstride, tmp.begin() + (size_t)b * tstride + (size_t)k * 4);
      offs[b + 1] = offs[b] + k;
    }
  } else if (n_boxes) {
    for (int b = 0; b < B; b++) n_boxes[b] = 0;
  }
  S->ms[0] = ms_since(t0);
  trace_mark("boxes_done");

  // ---- 2. SuperPoint with the boxes as mvDynamicArea; descriptors stay resident for LightGlue
  const auto t1 = std::chrono::steady_clock::now();
  sp_call.phase = 2; sp_call.defer_desc_sync = true;
  // the read-back's eight copies are enqueued from move_desc, i.e. behind LightGlue's launches when there is a LightGlue pass
  static const bool old_order = getenv("DANI_FE_OLD_ORDER") && atoi(getenv("DANI_FE_OLD_ORDER")) != 0;   // A/B: rounds 1-3's order
  sp_call.defer_desc_copies = desc != nullptr && !old_order;
  rc = sp_extract_impl(e, sp_call, gray, on_device, B, W, H, gray_stride, &p->sp, boxes.empty() ? nullptr : boxes.data(),
                       boxes.empty() ? nullptr : offs.data(), p->lapping, kpts_xy, resp, octave.data(), size.data(),
                       desc, cap, n_out, mono_index);
  if (rc) return rc;
  S->ms[1] = ms_since(t1);
  trace_mark("phase2_done");

  // ---- 3. LightGlue: frame b-1 (query) vs frame b (train); slot 0 of the feature slabs = previous call's last frame
  const auto t2 = std::chrono::steady_clock::now();
  // descriptor read-back: stream2 fills the pinned staging while LightGlue is being launched; the copy into the
  // caller's array then runs on this thread under the forward pass
  bool desc_moved = false;
  bool cache_enqueued = false;
  auto enqueue_cache = [&] {   // the last frame's features become the next call's predecessor (slot 0 of the slabs)
    if (cache_enqueued) return;
    cache_enqueued = true;
    if (!p->match_previous || (p->match_previous && p->pair_mode == 1)) return;   // extract-only / stereo calls cache nothing
    (void)hipMemcpyAsync(c->d_kimg_all, c->d_kimg + (size_t)(B - 1) * c->kcap * 2, (size_t)c->kcap * 2 * 4, hipMemcpyDeviceToDevice, c->stream);
    (void)hipMemcpyAsync(c->d_desc_all, c->d_descout + (size_t)(B - 1) * c->kcap * 256, (size_t)c->kcap * 256 * 4, hipMemcpyDeviceToDevice, c->stream);
  };
  bool copies_enqueued = false;
  auto enqueue_desc_copies = [&] {
    if (copies_enqueued || !desc) return;
    copies_enqueued = true;
    (void)sp_enqueue_desc_readback(e, c, n_out, B);   // a failure surfaces at the stream2 synchronise below / at the call's end
  };
  std::function<void()> move_desc = [&] {
    enqueue_cache();           // behind the LightGlue kernels in stream order, ahead of the call's final synchronise
    enqueue_desc_copies();
    if (desc_moved || !desc) return;
    desc_moved = true;
    if (hipStreamSynchronize(c->stream2) != hipSuccess) return;
    host_pool().run(B, [&](int b) {
      if (n_out[b]) memcpy(desc + (size_t)b * cap * 256, c->h_desc + (size_t)b * c->kcap * 256, (size_t)n_out[b] * 1024);
    });
  };
  const bool stereo = p->match_previous && p->pair_mode == 1;
  if (p->match_previous && !stereo) {
    std::vector<int32_t> n0(B), n1(B);
    for (int b = 0; b < B; b++) { n0[b] = b ? n_out[b - 1] : c->prev_n; n1[b] = n_out[b]; }
    const float sz[2] = {(float)W, (float)H};
    LgCall lg_call;
    lg_call.ctx = c; lg_call.overlap = &move_desc;
    // keypoints as the reference hands them to matchDescriptors: mvKeys (scaled by the level's factor, mono / stereo
    // partition order, src/SPextractor.cpp:164-216) next to the un-permuted descriptor rows (:159)
    rc = lg_match_impl(e, lg_call, B, c->kcap, 1, c->d_kimg_all, c->d_desc_all, n0.data(), c->d_kimg, c->d_descout, n1.data(), sz, sz,
                             p->lg_threshold, matches, scores, match_cap, k_out);
    if (rc) { move_desc(); (void)hipStreamSynchronize(c->stream); return rc; }
  } else if (stereo) {
    // left = frame 2k, right = frame 2k+1: the slabs of one pair lie two frame slots apart, so LightGlue's per-pair stride
    // (in_cap) is two slabs, and its per-pair output stride two match slots, starting at the right frame's slot
    const int P = B / 2;
    std::vector<int32_t> n0(P), n1(P), kk(P, 0);
    for (int k = 0; k < P; k++) { n0[k] = n_out[2 * k]; n1[k] = n_out[2 * k + 1]; }
    const float sz[2] = {(float)W, (float)H};
    LgCall lg_call;
    lg_call.ctx = c; lg_call.overlap = &move_desc;
    lg_call.out_stride = 2 * match_cap;   // a pair writes into the right frame's slot only: at most match_cap entries
    rc = lg_match_impl(e, lg_call, P, 2 * c->kcap, 1, c->d_kimg, c->d_descout, n0.data(), c->d_kimg + (size_t)c->kcap * 2,
                       c->d_descout + (size_t)c->kcap * 256, n1.data(), sz, sz, p->lg_threshold, matches + (size_t)match_cap * 2,
                       scores + match_cap, match_cap, kk.data());
    for (int k = 0; k < P; k++) { k_out[2 * k] = 0; k_out[2 * k + 1] = kk[k]; }   // also on DANI_ERR_CAPACITY: the sizes needed
    if (rc) { move_desc(); (void)hipStreamSynchronize(c->stream); return rc; }
  }
  const bool synced = cache_enqueued;   // LightGlue ran: its final synchronise covered the cache copies
  const bool defer = p->defer_desc && !p->match_previous && desc;   // split form: dani_frontend_match_resident finishes the read-back
  if (defer) { enqueue_cache(); enqueue_desc_copies(); desc_moved = true; }
  move_desc();
  if (!synced) HIP_TRY(e, hipStreamSynchronize(c->stream));
  if (!defer) HIP_TRY(e, hipStreamSynchronize(c->stream2));
  HIP_TRY(e, hipGetLastError());
  if (e->profiling) sp_read_timers(c, true);
  if (p->match_previous) c->prev_n = stereo ? 0 : n_out[B - 1];   // an extract-only call leaves the predecessor to dani_frontend_match_resident
  S->res_B = B; S->res_W = W; S->res_H = H; S->res_n.assign(n_out, n_out + B);
  if (defer) { S->pend_desc = desc; S->pend_cap = cap; }
  S->ms[2] = ms_since(t2);
  S->ms[3] = ms_since(t0);
  trace_mark("end");
  trace_end();
  return DANI_OK;
}

int dani_frontend_export_features(dani_engine* e, int32_t sequence, int32_t frame, float* dst, int32_t dst_on_device, int32_t cap,
                                  int32_t* n_out) {
  if (!e) return DANI_ERR_INVALID;
  if (!dst || cap < 1) return fail(e, DANI_ERR_INVALID, "bad dani_frontend_export_features argument");
  SeqState* S = seq_state(e, sequence);
  std::lock_guard<std::mutex> seq_lock(S->mu);
  Context* c = S->ctx;
  if (!c || frame < 0 || frame >= S->res_B) return fail(e, DANI_ERR_INVALID, "sequence %d holds no resident frame %d", sequence, frame);
  const int n = S->res_n[frame];
  if (n > cap) { if (n_out) *n_out = n; return fail(e, DANI_ERR_CAPACITY, "frame has %d keypoints (> %d)", n, cap); }
  HIP_TRY(e, hipSetDevice(e->device));
  const hipMemcpyKind kind = dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  const float cnt = (float)n;
  HIP_TRY(e, hipMemcpyAsync(dst, &cnt, 4, dst_on_device ? hipMemcpyHostToDevice : hipMemcpyHostToHost, c->stream));
  if (n) {
    HIP_TRY(e, hipMemcpyAsync(dst + 1, c->d_kimg + (size_t)frame * c->kcap * 2, (size_t)n * 8, kind, c->stream));
    HIP_TRY(e, hipMemcpyAsync(dst + 1 + (size_t)cap * 2, c->d_descout + (size_t)frame * c->kcap * 256, (size_t)n * 1024, kind, c->stream));
  }
  HIP_TRY(e, hipStreamSynchronize(c->stream));   // &cnt is a stack variable; the slab is ready for the collective on return
  if (n_out) *n_out = n;
  return DANI_OK;
}

int dani_frontend_match_resident(dani_engine* e, int32_t sequence, const float* pred, int32_t pred_on_device, int32_t pred_cap,
                                 float lg_threshold, int32_t* matches, float* scores, int32_t match_cap, int32_t* k_out) {
  if (!e) return DANI_ERR_INVALID;
  if (!matches || !scores || !k_out || match_cap < 1 || (pred && pred_cap < 1)) return fail(e, DANI_ERR_INVALID, "bad dani_frontend_match_resident argument");
  SeqState* S = seq_state(e, sequence);
  std::lock_guard<std::mutex> seq_lock(S->mu);
  return match_resident_locked(e, S, sequence, pred, pred_on_device, pred_cap, lg_threshold, matches, scores, match_cap, k_out);
}

}  // extern "C"

// ================================================================================================
// descriptor-distance scans of the projection searches
// ================================================================================================
namespace {

// rt_desc / rt_level: the train set already in HBM (a retained frame): t_desc / t_level are then not uploaded
int desc_scan(dani_engine* e, const float* q_desc, int32_t nq, const float* t_desc, int32_t nt, const int32_t* t_level,
              const int32_t* cand_offsets, const int32_t* cand_idx, float* best_dist, float* best_dist2, int32_t* best_idx,
              int32_t* best_level, int32_t* best_level2, float* cand_dist, const float* rt_desc = nullptr, const int32_t* rt_level = nullptr) {
  if (!e) return DANI_ERR_INVALID;
  if (nq < 0 || nt < 0 || (nq && (!q_desc || !cand_offsets || !best_dist || !best_dist2 || !best_idx || !best_level || !best_level2)))
    return fail(e, DANI_ERR_INVALID, "bad dani_desc_best2 argument");
  if (nq == 0) return DANI_OK;
  const int total = cand_offsets[nq];
  if (total < 0 || (total && (!cand_idx || (!t_desc && !rt_desc)))) return fail(e, DANI_ERR_INVALID, "bad candidate lists");
  for (int i = 0; i < total; i++)
    if (cand_idx[i] < 0 || cand_idx[i] >= nt) return fail(e, DANI_ERR_INVALID, "candidate index %d outside [0, %d)", cand_idx[i], nt);
  HIP_TRY(e, hipSetDevice(e->device));
  Context* c = acquire(e);
  if (!c) return fail(e, DANI_ERR_INVALID, "all contexts busy");
  struct Releaser { dani_engine* e; Context* c; ~Releaser() { release(e, c); } } rel{e, c};
  hipStream_t s = c->stream;
  const size_t bq = (size_t)nq * 1024, bt = rt_desc ? 1024 : (size_t)std::max(nt, 1) * 1024, bl = (size_t)std::max(nt, 1) * 4;
  const size_t bo = (size_t)(nq + 1) * 4, bc = (size_t)std::max(total, 1) * 4, bout = (size_t)nq * 4;
  const size_t need = bq + bt + bl + bo + 2 * bc + 5 * bout;   // ... + per-candidate distances
  if (need > c->scan_ws_bytes) {   // grow-only: a tracking thread calls this every frame with similar sizes
    if (c->scan_ws) (void)hipFree(c->scan_ws);
    c->scan_ws = nullptr; c->scan_ws_bytes = 0;
    HIP_TRY(e, hipMalloc((void**)&c->scan_ws, need + need / 4));
    c->scan_ws_bytes = need + need / 4;
  }
  char* d = c->scan_ws;
  float* dq = (float*)d; float* dt = (float*)(d + bq); int32_t* dl = (int32_t*)(d + bq + bt);
  int32_t* doff = (int32_t*)(d + bq + bt + bl); int32_t* dci = (int32_t*)(d + bq + bt + bl + bo);
  char* o = d + bq + bt + bl + bo + bc;
  float* dcd = (float*)(o + 5 * bout);
  HIP_TRY(e, hipMemcpyAsync(dq, q_desc, bq, hipMemcpyHostToDevice, s));
  if (nt && !rt_desc) HIP_TRY(e, hipMemcpyAsync(dt, t_desc, (size_t)nt * 1024, hipMemcpyHostToDevice, s));
  if (t_level && nt && !rt_desc) HIP_TRY(e, hipMemcpyAsync(dl, t_level, (size_t)nt * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(e, hipMemcpyAsync(doff, cand_offsets, bo, hipMemcpyHostToDevice, s));
  if (total) HIP_TRY(e, hipMemcpyAsync(dci, cand_idx, (size_t)total * 4, hipMemcpyHostToDevice, s));
  launch_desc_best2(dq, rt_desc ? rt_desc : dt, rt_desc ? rt_level : (t_level ? dl : nullptr), doff, dci, nq, (float*)o, (float*)(o + bout), (int32_t*)(o + 2 * bout),
                    (int32_t*)(o + 3 * bout), (int32_t*)(o + 4 * bout), cand_dist ? dcd : nullptr, s);
  HIP_TRY(e, hipGetLastError());
  HIP_TRY(e, hipMemcpyAsync(best_dist, o, bout, hipMemcpyDeviceToHost, s));
  HIP_TRY(e, hipMemcpyAsync(best_dist2, o + bout, bout, hipMemcpyDeviceToHost, s));
  HIP_TRY(e, hipMemcpyAsync(best_idx, o + 2 * bout, bout, hipMemcpyDeviceToHost, s));
  HIP_TRY(e, hipMemcpyAsync(best_level, o + 3 * bout, bout, hipMemcpyDeviceToHost, s));
  HIP_TRY(e, hipMemcpyAsync(best_level2, o + 4 * bout, bout, hipMemcpyDeviceToHost, s));
  if (cand_dist && total) HIP_TRY(e, hipMemcpyAsync(cand_dist, dcd, (size_t)total * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(e, hipStreamSynchronize(s));
  return DANI_OK;
}

}  // namespace

extern "C" {

int dani_desc_best2(dani_engine* e, const float* q_desc, int32_t nq, const float* t_desc, int32_t nt,
                    const int32_t* t_level, const int32_t* cand_offsets, const int32_t* cand_idx,
                    float* best_dist, float* best_dist2, int32_t* best_idx, int32_t* best_level,
                    int32_t* best_level2) {
  return desc_scan(e, q_desc, nq, t_desc, nt, t_level, cand_offsets, cand_idx, best_dist, best_dist2, best_idx, best_level,
                   best_level2, nullptr);
}

int dani_desc_mean(dani_engine* e, const float* rows, int32_t nrows, const int32_t* offsets, const int32_t* row_idx, int32_t nq,
                   float* mean_out) {
  if (!e) return DANI_ERR_INVALID;
  if (nq < 0 || nrows < 0 || (nq && (!offsets || !mean_out))) return fail(e, DANI_ERR_INVALID, "bad dani_desc_mean argument");
  if (nq == 0) return DANI_OK;
  const int total = offsets[nq];
  if (total < 0 || (total && (!row_idx || !rows))) return fail(e, DANI_ERR_INVALID, "bad observation lists");
  for (int i = 0; i < total; i++)
    if (row_idx[i] < 0 || row_idx[i] >= nrows) return fail(e, DANI_ERR_INVALID, "row index %d outside [0, %d)", row_idx[i], nrows);
  HIP_TRY(e, hipSetDevice(e->device));
  CtxLease lease(e, nullptr);
  Context* c = lease.c;
  if (!c) return fail(e, DANI_ERR_INVALID, "all contexts busy");
  hipStream_t s = c->stream;
  const size_t br = (size_t)std::max(nrows, 1) * 1024, bo = (size_t)(nq + 1) * 4, bi = (size_t)std::max(total, 1) * 4, bout = (size_t)nq * 1024;
  const size_t need = br + bo + bi + bout;
  if (need > c->scan_ws_bytes) {
    if (c->scan_ws) (void)hipFree(c->scan_ws);
    c->scan_ws = nullptr; c->scan_ws_bytes = 0;
    HIP_TRY(e, hipMalloc((void**)&c->scan_ws, need + need / 4));
    c->scan_ws_bytes = need + need / 4;
  }
  char* d = c->scan_ws;
  float* dr = (float*)d; int32_t* doff = (int32_t*)(d + br); int32_t* di = (int32_t*)(d + br + bo); float* dout = (float*)(d + br + bo + bi);
  if (nrows) HIP_TRY(e, hipMemcpyAsync(dr, rows, (size_t)nrows * 1024, hipMemcpyHostToDevice, s));
  HIP_TRY(e, hipMemcpyAsync(doff, offsets, bo, hipMemcpyHostToDevice, s));
  if (total) HIP_TRY(e, hipMemcpyAsync(di, row_idx, (size_t)total * 4, hipMemcpyHostToDevice, s));
  launch_desc_mean(dr, doff, di, nq, dout, s);
  HIP_TRY(e, hipGetLastError());
  HIP_TRY(e, hipMemcpyAsync(mean_out, dout, bout, hipMemcpyDeviceToHost, s));
  HIP_TRY(e, hipStreamSynchronize(s));
  return DANI_OK;
}

int dani_desc_scan(dani_engine* e, const float* q_desc, int32_t nq, const float* t_desc, int32_t nt,
                   const int32_t* t_level, const int32_t* cand_offsets, const int32_t* cand_idx,
                   float* best_dist, float* best_dist2, int32_t* best_idx, int32_t* best_level,
                   int32_t* best_level2, float* cand_dist) {
  if (!cand_dist && nq > 0 && cand_offsets && cand_offsets[nq] > 0) return fail(e, DANI_ERR_INVALID, "dani_desc_scan needs cand_dist");
  return desc_scan(e, q_desc, nq, t_desc, nt, t_level, cand_offsets, cand_idx, best_dist, best_dist2, best_idx, best_level,
                   best_level2, cand_dist);
}


/* ---- DBoW3 transform of float descriptors ---------------------------------------------------- */

struct dani_vocab {
  dani_engine* e = nullptr;
  int n_nodes = 0, L = 0;
  int32_t* d_child_off = nullptr; int32_t* d_child_ids = nullptr; float* d_desc = nullptr;
  std::vector<int32_t> child_off;      // host copies of what the host half of the transform needs
  std::vector<double> weight;
  std::vector<int32_t> word_id;
  float* d_feats = nullptr; int32_t* d_out = nullptr; int cap = 0;   // per-call staging, grow-only
  std::mutex mu;
};

int dani_vocab_create(dani_engine* e, int32_t n_nodes, int32_t L, const int32_t* child_offsets, const int32_t* child_ids,
                      const float* node_desc, const double* node_weight, const int32_t* node_word_id, dani_vocab** out) {
  if (!e) return DANI_ERR_INVALID;
  if (!out || n_nodes < 1 || L < 1 || !child_offsets || !child_ids || !node_desc || !node_weight || !node_word_id)
    return fail(e, DANI_ERR_INVALID, "bad dani_vocab_create argument");
  const int nchild = child_offsets[n_nodes];
  if (child_offsets[0] != 0 || nchild < 0 || nchild > n_nodes) return fail(e, DANI_ERR_INVALID, "bad child lists");
  for (int i = 0; i < n_nodes; i++) if (child_offsets[i + 1] < child_offsets[i]) return fail(e, DANI_ERR_INVALID, "child offsets not monotone");
  for (int i = 0; i < nchild; i++) if (child_ids[i] <= 0 || child_ids[i] >= n_nodes) return fail(e, DANI_ERR_INVALID, "child id %d outside (0, %d)", child_ids[i], n_nodes);
  // every descent must terminate: a child id larger than its parent's is what DBoW3's builder produces (nodes are
  // appended as they are created); require it, so that no input can make the kernel loop for ever
  for (int p = 0; p < n_nodes; p++)
    for (int c = child_offsets[p]; c < child_offsets[p + 1]; c++)
      if (child_ids[c] <= p) return fail(e, DANI_ERR_INVALID, "node %d has child %d: ids must grow down the tree", p, child_ids[c]);
  HIP_TRY(e, hipSetDevice(e->device));
  dani_vocab* v = new dani_vocab();
  v->e = e; v->n_nodes = n_nodes; v->L = L;
  v->child_off.assign(child_offsets, child_offsets + n_nodes + 1);
  v->weight.assign(node_weight, node_weight + n_nodes);
  v->word_id.assign(node_word_id, node_word_id + n_nodes);
  if (hipMalloc((void**)&v->d_child_off, (size_t)(n_nodes + 1) * 4) != hipSuccess || hipMalloc((void**)&v->d_child_ids, (size_t)std::max(nchild, 1) * 4) != hipSuccess ||
      hipMalloc((void**)&v->d_desc, (size_t)n_nodes * 1024) != hipSuccess) { dani_vocab_destroy(v); return fail(e, DANI_ERR_HIP, "hipMalloc failed (vocabulary)"); }
  HIP_TRY(e, hipMemcpy(v->d_child_off, child_offsets, (size_t)(n_nodes + 1) * 4, hipMemcpyHostToDevice));
  if (nchild) HIP_TRY(e, hipMemcpy(v->d_child_ids, child_ids, (size_t)nchild * 4, hipMemcpyHostToDevice));
  HIP_TRY(e, hipMemcpy(v->d_desc, node_desc, (size_t)n_nodes * 1024, hipMemcpyHostToDevice));
  *out = v;
  return DANI_OK;
}

void dani_vocab_destroy(dani_vocab* v) {
  if (!v) return;
  if (v->e) (void)hipSetDevice(v->e->device);
  for (void* p : {(void*)v->d_child_off, (void*)v->d_child_ids, (void*)v->d_desc, (void*)v->d_feats, (void*)v->d_out}) if (p) (void)hipFree(p);
  delete v;
}

static int bow_transform_impl(dani_engine* e, dani_vocab* v, const float* feats, const float* d_resident, int32_t n, int32_t levelsup, int32_t weighting, int32_t norm,
                       int32_t* word_ids, double* word_values, int32_t* n_words, int32_t* fv_node_ids, int32_t* fv_offsets,
                       int32_t* fv_features, int32_t* n_fv_nodes) {
  if (!e) return DANI_ERR_INVALID;
  if (!v || v->e != e || n < 0 || (n && !feats && !d_resident) || weighting < 0 || weighting > 3 || norm < 0 || norm > 2 || !word_ids || !word_values || !n_words ||
      !fv_node_ids || !fv_offsets || !fv_features || !n_fv_nodes) return fail(e, DANI_ERR_INVALID, "bad dani_bow_transform argument");
  *n_words = 0; *n_fv_nodes = 0; fv_offsets[0] = 0;
  if (n == 0) return DANI_OK;
  std::lock_guard<std::mutex> lk(v->mu);
  HIP_TRY(e, hipSetDevice(e->device));
  CtxLease lease(e, nullptr);
  Context* c = lease.c;
  if (!c) return fail(e, DANI_ERR_INVALID, "all contexts busy");
  hipStream_t s = c->stream;
  if (n > v->cap) {
    if (v->d_feats) (void)hipFree(v->d_feats);
    if (v->d_out) (void)hipFree(v->d_out);
    v->d_feats = nullptr; v->d_out = nullptr; v->cap = 0;
    HIP_TRY(e, hipMalloc((void**)&v->d_feats, (size_t)n * 1024));
    HIP_TRY(e, hipMalloc((void**)&v->d_out, (size_t)n * 8));
    v->cap = n;
  }
  if (!d_resident) HIP_TRY(e, hipMemcpyAsync(v->d_feats, feats, (size_t)n * 1024, hipMemcpyHostToDevice, s));
  launch_bow_descend(d_resident ? d_resident : v->d_feats, n, v->d_child_off, v->d_child_ids, v->d_desc, v->L - levelsup, v->d_out, v->d_out + n, s);
  HIP_TRY(e, hipGetLastError());
  std::vector<int32_t> leaf_nid((size_t)2 * n);
  HIP_TRY(e, hipMemcpyAsync(leaf_nid.data(), v->d_out, (size_t)n * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(e, hipStreamSynchronize(s));
  // BowVector / FeatureVector bookkeeping in feature order (Vocabulary.cpp:752-824, BowVector.cpp:34-84,
  // FeatureVector.cpp:31-45): std::map semantics, double weights
  std::map<int32_t, double> bow;
  std::map<int32_t, std::vector<int32_t>> fv;
  const bool tf = weighting <= 1;   // 0 TF_IDF, 1 TF: addWeight; 2 IDF, 3 BINARY: addIfNotExist
  for (int i = 0; i < n; i++) {
    const int leaf = leaf_nid[i], nid = leaf_nid[n + i];
    const double w = v->weight[leaf];
    if (!(w > 0)) continue;            // stopped word
    const int32_t wid = v->word_id[leaf];
    if (tf) bow[wid] += w; else bow.emplace(wid, w);
    fv[nid].push_back(i);
  }
  if (tf && !bow.empty() && norm == 0) { const double nd = (double)bow.size(); for (auto& kv : bow) kv.second /= nd; }
  if (norm != 0) {
    double nn = 0.0;
    if (norm == 1) for (auto& kv : bow) nn += std::fabs(kv.second);
    else { for (auto& kv : bow) nn += kv.second * kv.second; nn = std::sqrt(nn); }
    if (nn > 0.0) for (auto& kv : bow) kv.second /= nn;
  }
  int k = 0;
  for (auto& kv : bow) { word_ids[k] = kv.first; word_values[k] = kv.second; k++; }
  *n_words = k;
  int m = 0, off = 0;
  for (auto& kv : fv) {
    fv_node_ids[m] = kv.first;
    for (int32_t fi : kv.second) fv_features[off++] = fi;
    fv_offsets[++m] = off;
  }
  *n_fv_nodes = m;
  return DANI_OK;
}

int dani_bow_transform(dani_engine* e, dani_vocab* v, const float* feats, int32_t n, int32_t levelsup, int32_t weighting, int32_t norm,
                       int32_t* word_ids, double* word_values, int32_t* n_words, int32_t* fv_node_ids, int32_t* fv_offsets,
                       int32_t* fv_features, int32_t* n_fv_nodes) {
  return bow_transform_impl(e, v, feats, nullptr, n, levelsup, weighting, norm, word_ids, word_values, n_words, fv_node_ids, fv_offsets,
                            fv_features, n_fv_nodes);
}

/* ---- device-resident consumers: retained frames (SURVEY 8f, ranks 1-3 without the PCIe bounce) --------------------- */

struct dani_features {
  dani_engine* e = nullptr;
  int n = 0;
  float* d_kxy = nullptr;       // [n][2] coordinates used by the grid and the window tests (undistorted when supplied)
  float* d_desc = nullptr;      // [n][256]
  int32_t* d_octave = nullptr;  // [n]
  bool has_grid = false;
  dani_grid grid{};
  int32_t* d_cell_offsets = nullptr; int32_t* d_cell_idx = nullptr;
  std::mutex mu;                // the per-call scratch below
  char* ws = nullptr; size_t ws_bytes = 0;
};

void dani_features_release(dani_features* f) {
  if (!f) return;
  if (f->e) (void)hipSetDevice(f->e->device);
  for (void* p : {(void*)f->d_kxy, (void*)f->d_desc, (void*)f->d_octave, (void*)f->d_cell_offsets, (void*)f->d_cell_idx, (void*)f->ws})
    if (p) (void)hipFree(p);
  delete f;
}

int dani_features_count(const dani_features* f) { return f ? f->n : DANI_ERR_INVALID; }

int dani_frontend_retain(dani_engine* e, int32_t sequence, int32_t frame, const float* kpts_un, const int32_t* octave, const dani_grid* grid,
                         dani_features** out) {
  if (!e) return DANI_ERR_INVALID;
  if (!out || (grid && (grid->cols < 1 || grid->rows < 1 || (int64_t)grid->cols * grid->rows > (1 << 20))))
    return fail(e, DANI_ERR_INVALID, "bad dani_frontend_retain argument");
  *out = nullptr;
  SeqState* S = seq_state(e, sequence);
  std::lock_guard<std::mutex> seq_lock(S->mu);
  Context* c = S->ctx;
  if (!c || frame < 0 || frame >= S->res_B) return fail(e, DANI_ERR_INVALID, "sequence %d holds no resident frame %d", sequence, frame);
  HIP_TRY(e, hipSetDevice(e->device));
  const int n = S->res_n[frame], m = std::max(n, 1);
  dani_features* f = new dani_features();
  f->e = e; f->n = n;
  auto bail = [&](const char* what) { dani_features_release(f); return fail(e, DANI_ERR_HIP, "%s failed (retain)", what); };
  if (hipMalloc((void**)&f->d_kxy, (size_t)m * 8) != hipSuccess || hipMalloc((void**)&f->d_desc, (size_t)m * 1024) != hipSuccess ||
      hipMalloc((void**)&f->d_octave, (size_t)m * 4) != hipSuccess) return bail("hipMalloc");
  hipStream_t s = c->stream;
  if (n) {
    if (hipMemcpyAsync(f->d_desc, c->d_descout + (size_t)frame * c->kcap * 256, (size_t)n * 1024, hipMemcpyDeviceToDevice, s) != hipSuccess) return bail("descriptor copy");
    if ((kpts_un ? hipMemcpyAsync(f->d_kxy, kpts_un, (size_t)n * 8, hipMemcpyHostToDevice, s)
                 : hipMemcpyAsync(f->d_kxy, c->d_kimg + (size_t)frame * c->kcap * 2, (size_t)n * 8, hipMemcpyDeviceToDevice, s)) != hipSuccess) return bail("keypoint copy");
    if ((octave ? hipMemcpyAsync(f->d_octave, octave, (size_t)n * 4, hipMemcpyHostToDevice, s) : hipMemsetAsync(f->d_octave, 0, (size_t)n * 4, s)) != hipSuccess) return bail("octave copy");
  }
  if (grid) {
    const int ncells = grid->cols * grid->rows;
    int32_t* cell_of = nullptr;
    if (hipMalloc((void**)&f->d_cell_offsets, (size_t)(ncells + 1) * 4) != hipSuccess || hipMalloc((void**)&f->d_cell_idx, (size_t)m * 4) != hipSuccess ||
        hipMalloc((void**)&cell_of, (size_t)m * 4) != hipSuccess) { if (cell_of) (void)hipFree(cell_of); return bail("hipMalloc"); }
    launch_grid_assign(f->d_kxy, n, grid->min_x, grid->min_y, grid->inv_w, grid->inv_h, grid->cols, grid->rows, cell_of, f->d_cell_offsets, f->d_cell_idx, s);
    const hipError_t st = hipStreamSynchronize(s);
    (void)hipFree(cell_of);
    if (st != hipSuccess || hipGetLastError() != hipSuccess) return bail("grid kernel");
    f->grid = *grid; f->has_grid = true;
  } else if (hipStreamSynchronize(s) != hipSuccess) return bail("copy");
  *out = f;
  return DANI_OK;
}

int dani_features_undistort(dani_engine* e, dani_features* f, const float* K4, const float* dist, int32_t ndist, const dani_grid* grid,
                            float* kpts_un_out) {
  if (!e) return DANI_ERR_INVALID;
  if (!f || f->e != e || !K4 || ndist < 0 || ndist > 5 || (ndist && !dist) || !(K4[0] != 0.0f) || !(K4[1] != 0.0f) ||
      (grid && (grid->cols < 1 || grid->rows < 1 || (int64_t)grid->cols * grid->rows > (1 << 20))))
    return fail(e, DANI_ERR_INVALID, "bad dani_features_undistort argument");
  HIP_TRY(e, hipSetDevice(e->device));
  std::lock_guard<std::mutex> lk(f->mu);
  CtxLease lease(e, nullptr);
  hipStream_t s = lease.c->stream;
  const int n = f->n, m = std::max(n, 1);
  if (n && ndist >= 1 && dist[0] != 0.0f) {
    float* tmp = nullptr;
    HIP_TRY(e, hipMalloc((void**)&tmp, (size_t)m * 8));
    launch_undistort(f->d_kxy, n, K4, dist, ndist, tmp, s);
    hipError_t st = hipStreamSynchronize(s);
    if (st == hipSuccess) st = hipGetLastError();
    if (st != hipSuccess) { (void)hipFree(tmp); return fail(e, DANI_ERR_HIP, "undistort kernel failed: %s", hipGetErrorString(st)); }
    (void)hipFree(f->d_kxy);
    f->d_kxy = tmp;
  }
  if (grid) {
    const int ncells = grid->cols * grid->rows;
    for (void* p : {(void*)f->d_cell_offsets, (void*)f->d_cell_idx}) if (p) (void)hipFree(p);
    f->d_cell_offsets = nullptr; f->d_cell_idx = nullptr; f->has_grid = false;
    int32_t* cell_of = nullptr;
    HIP_TRY(e, hipMalloc((void**)&f->d_cell_offsets, (size_t)(ncells + 1) * 4));
    HIP_TRY(e, hipMalloc((void**)&f->d_cell_idx, (size_t)m * 4));
    HIP_TRY(e, hipMalloc((void**)&cell_of, (size_t)m * 4));
    launch_grid_assign(f->d_kxy, n, grid->min_x, grid->min_y, grid->inv_w, grid->inv_h, grid->cols, grid->rows, cell_of, f->d_cell_offsets, f->d_cell_idx, s);
    const hipError_t st = hipStreamSynchronize(s);
    (void)hipFree(cell_of);
    if (st != hipSuccess || hipGetLastError() != hipSuccess) return fail(e, DANI_ERR_HIP, "grid kernel failed");
    f->grid = *grid; f->has_grid = true;
  }
  if (kpts_un_out && n) HIP_TRY(e, hipMemcpy(kpts_un_out, f->d_kxy, (size_t)n * 8, hipMemcpyDeviceToHost));
  return DANI_OK;
}

int dani_features_fetch_grid(dani_features* f, int32_t* cell_offsets, int32_t* cell_idx) {
  if (!f || !f->has_grid || !cell_offsets) return DANI_ERR_INVALID;
  (void)hipSetDevice(f->e->device);
  const int ncells = f->grid.cols * f->grid.rows;
  if (hipMemcpy(cell_offsets, f->d_cell_offsets, (size_t)(ncells + 1) * 4, hipMemcpyDeviceToHost) != hipSuccess) return DANI_ERR_HIP;
  const int total = cell_offsets[ncells];
  if (total && cell_idx && hipMemcpy(cell_idx, f->d_cell_idx, (size_t)total * 4, hipMemcpyDeviceToHost) != hipSuccess) return DANI_ERR_HIP;
  return total;
}

int dani_desc_scan_resident(dani_engine* e, const float* q_desc, int32_t nq, const dani_features* train, const int32_t* cand_offsets,
                            const int32_t* cand_idx, float* best_dist, float* best_dist2, int32_t* best_idx, int32_t* best_level,
                            int32_t* best_level2, float* cand_dist) {
  if (!e) return DANI_ERR_INVALID;
  if (!train || train->e != e) return fail(e, DANI_ERR_INVALID, "bad retained frame");
  return desc_scan(e, q_desc, nq, nullptr, train->n, nullptr, cand_offsets, cand_idx, best_dist, best_dist2, best_idx, best_level, best_level2,
                   cand_dist, train->d_desc, train->d_octave);
}

int dani_search_windows_resident(dani_engine* e, const dani_features* train_c, const float* q_desc, int32_t nq, const float* qx, const float* qy,
                                 const float* qr, const int32_t* min_level, const int32_t* max_level, float* best_dist, float* best_dist2,
                                 int32_t* best_idx, int32_t* best_level, int32_t* best_level2, int32_t* cand_offsets, int32_t* cand_idx,
                                 float* cand_dist, int32_t cand_cap) {
  if (!e) return DANI_ERR_INVALID;
  dani_features* f = const_cast<dani_features*>(train_c);
  if (!f || f->e != e || !f->has_grid) return fail(e, DANI_ERR_INVALID, "the retained frame has no grid (pass one to dani_frontend_retain)");
  if (nq < 0 || (nq && (!q_desc || !qx || !qy || !qr || !best_dist || !best_dist2 || !best_idx || !best_level || !best_level2)) || cand_cap < 0 ||
      (cand_cap && (!cand_offsets || !cand_idx))) return fail(e, DANI_ERR_INVALID, "bad dani_search_windows_resident argument");
  if (nq == 0) { if (cand_offsets) cand_offsets[0] = 0; return DANI_OK; }
  HIP_TRY(e, hipSetDevice(e->device));
  std::lock_guard<std::mutex> lk(f->mu);
  CtxLease lease(e, nullptr);
  Context* c = lease.c;
  hipStream_t s = c->stream;
  // scratch: q_desc | qx qy qr | levels | counts | offsets | outputs(5) ; candidates (idx + dist) grow separately
  const size_t bq = (size_t)nq * 1024, b4 = (size_t)nq * 4, boff = (size_t)(nq + 1) * 4;
  const size_t fixed = bq + 3 * b4 + 2 * b4 + b4 + boff + 5 * b4 + 256;
  const size_t cand_room = (size_t)std::max(f->n, 1) * 64 * 8;     // first guess: 64 candidates per query ... grown below when short
  auto ensure = [&](size_t need) -> int {
    if (need <= f->ws_bytes) return 0;
    if (f->ws) (void)hipFree(f->ws);
    f->ws = nullptr; f->ws_bytes = 0;
    HIP_TRY(e, hipMalloc((void**)&f->ws, need));
    f->ws_bytes = need;
    return 0;
  };
  int rc = ensure(fixed + std::max(cand_room, (size_t)nq * 32 * 8));
  if (rc) return rc;
  auto carve = [&](char* base) {
    struct P { float *q, *qx, *qy, *qr; int32_t *minl, *maxl, *counts, *offs; char* out; char* cand; } p;
    p.q = (float*)base; char* w = base + bq;
    p.qx = (float*)w; w += b4; p.qy = (float*)w; w += b4; p.qr = (float*)w; w += b4;
    p.minl = (int32_t*)w; w += b4; p.maxl = (int32_t*)w; w += b4; p.counts = (int32_t*)w; w += b4;
    p.offs = (int32_t*)w; w += boff; w = (char*)(((uintptr_t)w + 15) & ~(uintptr_t)15);
    p.out = w; w += 5 * b4; w = (char*)(((uintptr_t)w + 15) & ~(uintptr_t)15);
    p.cand = w;
    return p;
  };
  auto P = carve(f->ws);
  const dani_grid& g = f->grid;
  auto upload = [&]() -> int {
    HIP_TRY(e, hipMemcpyAsync(P.q, q_desc, bq, hipMemcpyHostToDevice, s));
    HIP_TRY(e, hipMemcpyAsync(P.qx, qx, b4, hipMemcpyHostToDevice, s));
    HIP_TRY(e, hipMemcpyAsync(P.qy, qy, b4, hipMemcpyHostToDevice, s));
    HIP_TRY(e, hipMemcpyAsync(P.qr, qr, b4, hipMemcpyHostToDevice, s));
    if (min_level) HIP_TRY(e, hipMemcpyAsync(P.minl, min_level, b4, hipMemcpyHostToDevice, s));
    if (max_level) HIP_TRY(e, hipMemcpyAsync(P.maxl, max_level, b4, hipMemcpyHostToDevice, s));
    return 0;
  };
  if ((rc = upload())) return rc;
  launch_area_count(f->d_kxy, f->d_octave, g.min_x, g.min_y, g.inv_w, g.inv_h, g.cols, g.rows, f->d_cell_offsets, f->d_cell_idx, P.qx, P.qy, P.qr,
                    min_level ? P.minl : nullptr, max_level ? P.maxl : nullptr, nq, P.counts, P.offs, s);
  int32_t total = 0;
  HIP_TRY(e, hipMemcpyAsync(&total, P.offs + nq, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(e, hipStreamSynchronize(s));
  HIP_TRY(e, hipGetLastError());
  const size_t need = fixed + (size_t)std::max(total, 1) * 8 + 64;
  if (need > f->ws_bytes) {          // rare: more candidates than the first guess -> regrow and redo the (cheap) count pass
    if ((rc = ensure(need + need / 4))) return rc;
    P = carve(f->ws);
    if ((rc = upload())) return rc;
    launch_area_count(f->d_kxy, f->d_octave, g.min_x, g.min_y, g.inv_w, g.inv_h, g.cols, g.rows, f->d_cell_offsets, f->d_cell_idx, P.qx, P.qy, P.qr,
                      min_level ? P.minl : nullptr, max_level ? P.maxl : nullptr, nq, P.counts, P.offs, s);
  }
  int32_t* d_cidx = (int32_t*)P.cand;
  float* d_cdist = (float*)(P.cand + (size_t)std::max(total, 1) * 4);
  launch_area_fill(f->d_kxy, f->d_octave, g.min_x, g.min_y, g.inv_w, g.inv_h, g.cols, g.rows, f->d_cell_offsets, f->d_cell_idx, P.qx, P.qy, P.qr,
                   min_level ? P.minl : nullptr, max_level ? P.maxl : nullptr, nq, P.offs, d_cidx, s);
  launch_desc_best2(P.q, f->d_desc, f->d_octave, P.offs, d_cidx, nq, (float*)P.out, (float*)(P.out + b4), (int32_t*)(P.out + 2 * b4),
                    (int32_t*)(P.out + 3 * b4), (int32_t*)(P.out + 4 * b4), (cand_dist && cand_cap) ? d_cdist : nullptr, s);
  HIP_TRY(e, hipGetLastError());
  HIP_TRY(e, hipMemcpyAsync(best_dist, P.out, b4, hipMemcpyDeviceToHost, s));
  HIP_TRY(e, hipMemcpyAsync(best_dist2, P.out + b4, b4, hipMemcpyDeviceToHost, s));
  HIP_TRY(e, hipMemcpyAsync(best_idx, P.out + 2 * b4, b4, hipMemcpyDeviceToHost, s));
  HIP_TRY(e, hipMemcpyAsync(best_level, P.out + 3 * b4, b4, hipMemcpyDeviceToHost, s));
  HIP_TRY(e, hipMemcpyAsync(best_level2, P.out + 4 * b4, b4, hipMemcpyDeviceToHost, s));
  const bool lists = cand_cap > 0 && total <= cand_cap;
  if (cand_offsets) HIP_TRY(e, hipMemcpyAsync(cand_offsets, P.offs, boff, hipMemcpyDeviceToHost, s));
  if (lists && total) {
    HIP_TRY(e, hipMemcpyAsync(cand_idx, d_cidx, (size_t)total * 4, hipMemcpyDeviceToHost, s));
    if (cand_dist) HIP_TRY(e, hipMemcpyAsync(cand_dist, d_cdist, (size_t)total * 4, hipMemcpyDeviceToHost, s));
  }
  HIP_TRY(e, hipStreamSynchronize(s));
  if (cand_cap > 0 && total > cand_cap) return fail(e, DANI_ERR_CAPACITY, "%d candidates (> %d)", total, cand_cap);
  return DANI_OK;
}

int dani_bow_transform_resident(dani_engine* e, dani_vocab* v, const dani_features* f, int32_t levelsup, int32_t weighting, int32_t norm,
                                int32_t* word_ids, double* word_values, int32_t* n_words, int32_t* fv_node_ids, int32_t* fv_offsets,
                                int32_t* fv_features, int32_t* n_fv_nodes) {
  if (!e) return DANI_ERR_INVALID;
  if (!f || f->e != e) return fail(e, DANI_ERR_INVALID, "bad retained frame");
  return bow_transform_impl(e, v, nullptr, f->d_desc, f->n, levelsup, weighting, norm, word_ids, word_values, n_words, fv_node_ids, fv_offsets,
                            fv_features, n_fv_nodes);
}

}  // extern "C"
