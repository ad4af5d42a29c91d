// engine_internal.h -- shared state and helpers of the C-ABI implementation (engine.cpp, sp_engine.cpp, lg_engine.cpp,
// yolo_engine.cpp, frontend_engine.cpp).  Not part of the ABI: include/dani_frontend.h is.
#pragma once
#include "../../include/dani_frontend.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

using namespace dani;


namespace dani {
namespace eng {

extern thread_local std::string g_create_error;

// Every stream wait of the engine goes through here (engine.cpp).  dani_set_host_wait / DANI_HOST_WAIT, measured with per-thread
// CPU clocks on the 8-frame front-end call (bench.py: host_cpu_ms_per_step_top_threads; profiles/r04_bench_host_wait_*.json,
// r04_bench_final.json):
//   0 "block"   hipStreamSynchronize.  On this ROCm build that is NOT a sleep that frees the core: 10.8 ms of caller CPU + 9.9 ms
//               of a runtime thread per 10.8 ms call -- and it still yields the core now and then: on a loaded host 5 % of the
//               calls come back 3-4 ms late (one scheduler tick).
//   1 "spin"    (default) poll hipStreamQuery from user space: the same CPU (10.7 + 9.9 ms), 0.03-0.05 ms per call faster on a
//               quiet host and free of the late wake-ups on a loaded one (mean 10.76 against 10.94 ms on the same box).
//   2 "hybrid"  two stages: SLEEP (40 us naps polling an event) until a marker that the engine records a few hundred microseconds
//               of GPU work ahead of the point it is going to wait for (near_mark below: behind SuperPoint's detector head, in
//               front of LightGlue's last FFN), then SPIN for the rest; waits without a marker spin for at most 150 us and then
//               hand over to hipStreamSynchronize.  2.1 + 4.7 ms of CPU per call -- one core instead of 2.3 -- and the median of
//               the other two, but a thread that sleeps can be scheduled late: the same 3-4 ms outliers as mode 0 on a loaded host.
//               For deployments that need the core more than the last per cent.
extern std::atomic<int> g_host_wait;   // -1 = the environment's choice (DANI_HOST_WAIT=block|spin|hybrid; default spin)
int wait_mode_now();                   // the mode in force
hipError_t stream_wait(hipStream_t s, void (*before_spin)(void*) = nullptr, void* arg = nullptr);   // before_spin: called once when the naps are over (or at once without a marker / in the pure modes)
void near_mark(hipStream_t s);     // "the next wait on s is a few hundred microseconds of GPU work away": records s's marker event
void near_forget(hipStream_t s);   // the stream is about to be destroyed

constexpr int CAND_FAST = 4096;  // candidates per frame read back together with the counts (typical frames: 3-4 k)

struct Context {
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;   // YOLO (concurrent with the SuperPoint conv stack) and overlapped D2H copies
  hipEvent_t ev_desc = nullptr, ev_y0 = nullptr, ev_y1 = nullptr, ev_lg0 = nullptr, ev_lg1 = nullptr;
  hipEvent_t ev_fork0 = nullptr, ev_fork1 = nullptr;   // fork / join of SuperPoint's two heads (ordering only, timing disabled)
  int W = 0, H = 0, B = 0;  // shapes the buffers are built for
  int nlevels = 0;
  float scale_factor = 0.0f;
  struct Level {              // one pyramid level (ComputePyramid, src/SPextractor.cpp:631-656)
    int W = 0, H = 0;
    ActDesc act[12];          // output of layer i (layers 1,3,5: pooled)
    float* prob = nullptr;
    float* desc_nhwc = nullptr;
    float* desc_nchw = nullptr;
    std::vector<CellRect> cells;
    CellRect* d_cells = nullptr;
    int slot_stride = 0;
    int cand_cap = 0;                 // candidates per frame = sum of the cell areas (every pixel of every cell): cannot overflow
    float* slot_xyr = nullptr;
    int32_t* slot_count = nullptr;
    float* cand = nullptr;
    int32_t* cand_count = nullptr;
    float* h_cand = nullptr;          // pinned
    int32_t* h_cand_count = nullptr;  // pinned
    uint8_t* d_level_img = nullptr;   // level > 0: this level's image [B][H][W] (resized from level - 1)
    const uint8_t* last_img = nullptr; int last_stride = 0, last_B = 0; size_t last_frame = 0;   // the last call's level image (device)
    bool act0_valid = false;          // act[0] (conv1a) holds the last call's activation (false when conv1b computed it on the fly)
    bool heads_valid = false;         // act[9] / act[11] (semi, coarse descriptors) hold the last call's layers (false when the heads' epilogues consumed them)
  };
  std::vector<Level> lv;
  int32_t* d_boxes = nullptr;
  int32_t* d_box_offsets = nullptr;
  int box_cap = 0;
  uint8_t* d_img = nullptr;
  size_t d_img_bytes = 0;
  float* d_kxy_all = nullptr;
  float* d_desc_all = nullptr;
  int prev_n = 0;            // keypoints of the cached previous frame (slot 0)
  float* d_kxy = nullptr;
  float* d_kimg_all = nullptr;   // [B + 1][kcap][2] keypoint->pt in image coordinates, output order (slot 0: previous frame)
  float* d_kimg = nullptr;       // slot 1
  float* h_kimg = nullptr;       // pinned staging of slots 1..B
  int32_t* d_nk = nullptr;
  float* d_sampled = nullptr;
  float* d_descout = nullptr;
  int kcap = 0;
  // pinned host staging
  float* h_kxy = nullptr;  // [B][kcap][2] followed by per-level (offset, count) ints [nlevels][2][B]
  bool cands_enqueued = false;   // phase 1 already ran cell NMS + candidate read-back (SpCall::host_cull)
  bool spec_valid = false;       // phase 3 ran: spec_sels[b] = quadtree selection of frame b computed WITHOUT the boxes
  std::vector<std::vector<int>> spec_sels;
  float* h_desc = nullptr; // [B][kcap][256]: descriptor read-back of the fused call (copied out under LightGlue)
  int32_t* h_boxes = nullptr;
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t lev[13] = {};
  float ms[5] = {0, 0, 0, 0, 0};
  float layer_ms[12] = {};
  std::vector<void*> dev_allocs, host_allocs;
  bool busy = false;
  char* scan_ws = nullptr;              // grow-only workspace of dani_desc_best2
  size_t scan_ws_bytes = 0;
  struct LgContext* lg = nullptr;       // LightGlue / YOLO workspaces of this context (created on first use, owned)
  struct YoloContext* yolo = nullptr;
};

struct LgWeights;
struct YoloWeights;
struct CommState;   // comm_engine.cpp: the RCCL communicator of a sharded batch
// one image sequence (camera) of dani_frontend_batch: own context (stream, workspaces, cached predecessor frame)
struct SeqState {
  std::mutex mu;
  Context* ctx = nullptr;
  std::vector<std::string> class_names, dynamic_names;
  float ms[4] = {0, 0, 0, 0};
  // per-call host scratch, kept between calls (a fresh 0.8 MB vector per call costs a memset on the critical path)
  std::vector<float> dets, size;
  std::vector<int32_t> octave, frame_boxes;
  // resident features of the last dani_frontend_batch call (for dani_frontend_export_features / _match_resident)
  int res_B = 0, res_W = 0, res_H = 0;
  std::vector<int32_t> res_n;
  // descriptor read-back left running by a defer_desc call: stream2 is filling the pinned staging, `pend_desc` is the caller's array
  float* pend_desc = nullptr;
  int pend_cap = 0;
};


}  // namespace eng
}  // namespace dani

struct dani_engine {
  dani_config cfg;
  int device = 0;
  std::string err;
  std::map<std::string, std::vector<float>> tensors;
  bool finalized = false;
  ConvLayerPlan plan[12];
  ConvLayerPlan plan_plain_pb, plan_plain_db;   // convPb / convDb as plain layers when plan[9] / plan[11] are the fused heads (debug hooks)
  bool fused_heads = false;
  float* w1a = nullptr;
  float* b1a = nullptr;
  std::vector<void*> weight_allocs;
  std::vector<dani::eng::Context*> ctxs;
  std::mutex mu;
  std::mutex err_mu;                    // guards err (per-thread copies are handed out, engine.cpp)
  std::condition_variable ctx_free;     // signalled by release(): acquire() waits on it under mu
  std::atomic<bool> profiling{false};   // read by concurrent calls
  std::atomic<int> lg_gemm{0};   // DANI_LG_GEMM_*: 0 = f32 MFMA in the canonical order, 1 = bf16x3 split linears (lg_x3.hip), 2 = + attention
  dani::eng::LgWeights* lgw = nullptr;             // packed LightGlue / YOLO weights (owned; device buffers are in weight_allocs)
  dani::eng::YoloWeights* yw = nullptr;
  std::map<int, dani::eng::SeqState*> seqs;        // dani_frontend_batch sequences (own context + cached previous frame), under mu
  dani::eng::CommState* comm = nullptr;            // dani_comm_create: RCCL communicator + gather buffers (owned)
};


namespace dani {
namespace eng {

int fail(dani_engine* e, int code, const char* fmt, ...);

// Host timeline of one front-end call (DANI_TRACE_HOST=1, tools/debug/host_trace.py): labels with the microseconds since the call's
// start, printed to stderr as one line when the call ends.  Off: one predictable branch per mark.
struct HostTrace {
  bool on = false;
  std::chrono::steady_clock::time_point t0;
  int n = 0;
  const char* label[40];
  float us[40];
};
extern thread_local HostTrace t_trace;
inline void trace_begin() {
  static const bool env_on = getenv("DANI_TRACE_HOST") && atoi(getenv("DANI_TRACE_HOST")) != 0;
  t_trace.on = env_on;
  if (env_on) { t_trace.t0 = std::chrono::steady_clock::now(); t_trace.n = 0; }
}
inline void trace_mark(const char* label) {
  if (!t_trace.on || t_trace.n >= 40) return;
  t_trace.label[t_trace.n] = label;
  t_trace.us[t_trace.n++] = std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - t_trace.t0).count();
}
inline void trace_end() {
  if (!t_trace.on) return;
  fprintf(stderr, "dani-trace:");
  for (int i = 0; i < t_trace.n; i++) fprintf(stderr, " %s=%.1f", t_trace.label[i], t_trace.us[i]);
  fprintf(stderr, "\n");
  t_trace.on = false;
}

#define HIP_TRY(e, call)                                                                            \
  do {                                                                                              \
    hipError_t _st = (call);                                                                        \
    if (_st != hipSuccess)                                                                          \
      return fail((e), DANI_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_st), __FILE__, __LINE__); \
  } while (0)

// Small persistent worker pool for the per-frame host stages (quadtree, NMS, box refinement): one pool per calling
// thread, so concurrent sequences never share one.  Spawning std::threads per call cost ~150 us per stage.
class HostPool {
 public:
  explicit HostPool(int workers) {
    for (int i = 0; i < workers; i++) th_.emplace_back([this] { loop(); });
  }
  ~HostPool() {
    { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  // Wake the workers ahead of a run() that is known to come within a few hundred microseconds (the caller is about to
  // block on a stream synchronise whose completion starts the host stage): a sleeping worker takes 30-60 us to get
  // going, long enough for the caller to have run several of the 8 per-frame tasks itself.  Awake workers poll for
  // `spin_us` and go back to sleep if nothing arrives.
  void prewake(int spin_us = 400) {
    spin_until_.store(now_ns() + (int64_t)spin_us * 1000, std::memory_order_relaxed);
    { std::lock_guard<std::mutex> lk(m_); wake_epoch_++; }
    cv_.notify_all();
  }
  // runs fn(0..count-1); the caller takes part; returns when all are done
  void run(int count, const std::function<void(int)>& fn) {
    if (count <= 1 || th_.empty()) { for (int i = 0; i < count; i++) fn(i); return; }
    {
      std::lock_guard<std::mutex> lk(m_);
      fn_ = &fn; count_ = count; next_ = 0; remaining_ = count;
      gen_.fetch_add(1, std::memory_order_release);
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [this] { return remaining_ == 0; });
    fn_ = nullptr;
  }

 private:
  static int64_t now_ns() {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
  }
  void work() {
    for (;;) {
      int i;
      const std::function<void(int)>* f;
      {
        std::lock_guard<std::mutex> lk(m_);
        if (!fn_ || next_ >= count_) return;
        i = next_++; f = fn_;
      }
      (*f)(i);
      {
        std::lock_guard<std::mutex> lk(m_);
        if (--remaining_ == 0) done_.notify_all();
      }
    }
  }
  void loop() {
    uint64_t seen = 0, seen_wake = 0;
    for (;;) {
      // awake: poll for a job until the prewake deadline passes
      while (gen_.load(std::memory_order_acquire) == seen && now_ns() < spin_until_.load(std::memory_order_relaxed)) {
        for (int k = 0; k < 64; k++) __builtin_ia32_pause();
      }
      if (gen_.load(std::memory_order_acquire) == seen) {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return stop_ || gen_.load(std::memory_order_acquire) != seen || wake_epoch_ != seen_wake; });
        if (stop_) return;
        seen_wake = wake_epoch_;
        if (gen_.load(std::memory_order_acquire) == seen) continue;   // pre-woken: go and poll
      }
      seen = gen_.load(std::memory_order_acquire);
      work();
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  const std::function<void(int)>* fn_ = nullptr;
  int count_ = 0, next_ = 0, remaining_ = 0;
  std::atomic<uint64_t> gen_{0};
  std::atomic<int64_t> spin_until_{0};
  uint64_t wake_epoch_ = 0;
  bool stop_ = false;
};

HostPool& host_pool();   // one pool per calling thread

extern thread_local Context* t_last_ctx;
Context* last_ctx(dani_engine* e);   // the calling thread's last context, if it still belongs to engine e

// How the fused front-end (dani_frontend_batch) drives the three stages; the public ABI entry points pass the defaults.
struct SpCall {
  Context* ctx = nullptr;             // run on this context (the sequence's own) instead of acquiring a free one
  int phase = 0;                      // 0: whole extractor; 1: enqueue the network only; 2: everything after it;
                                      // 3 (between 1 and 2, host_cull calls): fetch the candidates and run the quadtrees
                                      // speculatively without boxes, while the YOLO chain is still running
  bool keep_desc_on_device = false;   // `desc` may be null: descriptors are consumed on the device (LightGlue)
  bool defer_desc_sync = false;       // descriptor read-back into pinned staging on stream2, no final synchronise
  bool defer_desc_copies = false;     // ... and the copies themselves are enqueued by the caller (sp_enqueue_desc_readback) once it has
                                      // enqueued what comes next: eight hipMemcpyAsync calls are 29 us of host time with the GPU idle
  hipEvent_t wait_event = nullptr;    // the conv layers from index wait_layer on wait for this event (recorded on another stream)
  int wait_layer = 1;
  int reserve_lds_bytes = 0;          // fused call with YOLO on the second stream: unused dynamic LDS added to the conv layers in
  unsigned reserve_lds_mask = 0;      // reserve_lds_mask (bit i = layer i), so that a CU keeps a free slot for YOLO's workgroups
  bool side_free = false;             // fused call without YOLO: the context's second stream is idle during the network (phase 0: always)
  bool host_cull = false;             // single-level calls: candidate NMS + read-back are enqueued in phase 1, without the
                                      // boxes; the dynamic-box cull (exact and idempotent at scale 1) runs on the host in phase 2
};
int sp_enqueue_desc_readback(dani_engine* e, Context* c, const int32_t* n_out, int B);   // the fused call's descriptor read-back (stream2, behind ev_desc)
struct YoloCall {
  Context* ctx = nullptr;
  bool alt_stream = false;            // run on the context's second (high-priority) stream, next to the SuperPoint network
  std::function<void(int, const float*, int)>* post = nullptr;   // per frame, inside the NMS worker: (frame, dets6, n)
  std::function<void()>* mid_chain = nullptr;     // called once, after the first `mid_after` stages of the chain have been enqueued
  int mid_after = 0;                              // (stage = model.0, model.1, C3 model.2, model.3, C3 model.4, ...)
  std::function<void()>* before_sync = nullptr;   // host work of the caller to run after the chain is enqueued, before waiting for it
};
struct LgCall {
  Context* ctx = nullptr;
  std::function<void()>* overlap = nullptr;   // host work of the caller to run while the forward pass executes
  int32_t out_stride = 0;   // matches / scores of pair p start at p * out_stride (0: p * cap); `cap` stays the per-pair write limit
};

hipError_t create_priority_stream(hipStream_t* s);
hipError_t context_create_sync(Context* c);   // both streams + every event, all checked (nothing left behind on failure)
void context_destroy_sync(Context* c);
Context* acquire(dani_engine* e);
void release(dani_engine* e, Context* c);

// a context for one call: the caller's own (fused front-end) or a free one from the pool, released on scope exit
struct CtxLease {
  dani_engine* e; Context* c; bool owned;
  CtxLease(dani_engine* e_, Context* given) : e(e_), c(given ? given : acquire(e_)), owned(!given) {}
  ~CtxLease() { if (owned && c) release(e, c); }
  CtxLease(const CtxLease&) = delete;
  CtxLease& operator=(const CtxLease&) = delete;
};

int dev_alloc(dani_engine* e, Context* c, void** p, size_t bytes, bool zero);
int host_alloc(dani_engine* e, Context* c, void** p, size_t bytes);
void free_buffers(Context* c);

// stage implementations (the ABI entry points pass default call options; dani_frontend_batch drives them explicitly)
int sp_extract_impl(dani_engine* e, const SpCall& call, const uint8_t* imgs, int32_t imgs_on_device, int32_t B, int32_t W, int32_t H,
                    int32_t stride, const dani_sp_params* p, const int32_t* boxes, const int32_t* box_offsets,
                    const int32_t lapping[2], float* kpts_xy, float* resp, int32_t* octave, float* size,
                    float* desc, int32_t cap, int32_t* n_out, int32_t* mono_index);
int lg_match_impl(dani_engine* e, const LgCall& call, int32_t P, int32_t in_cap, int32_t on_device, const float* kpts0,
                  const float* desc0, const int32_t* n0, const float* kpts1, const float* desc1, const int32_t* n1,
                  const float* size0, const float* size1, float threshold, int32_t* matches, float* scores,
                  int32_t cap, int32_t* k_out);
int yolo_detect_impl(dani_engine* e, const YoloCall& call, const uint8_t* bgr, int32_t on_device, int32_t B, int32_t W, int32_t H,
                     int32_t stride, float score_thresh, float iou_thresh, float* dets6, int32_t cap, int32_t* n_out);
void sp_read_timers(Context* c, bool prof);
// weights_spec.cpp: a tensor set (name -> float count) against the SuperPoint (required) / LightGlue / YOLOv5s (optional groups) specs;
// msg names the first missing or mis-shaped tensor; *groups: bit 0 SuperPoint, 1 LightGlue, 2 YOLO present
int check_tensor_counts(const std::map<std::string, size_t>& have, std::string& msg, int* groups, int* ignored);
int lg_finalize(dani_engine* e);     // pack the LightGlue / YOLO tensors of e->tensors (dani_finalize_weights, sp_engine.cpp)
int yolo_finalize(dani_engine* e);
void lg_release(dani_engine* e);     // workspaces and packed weights die with the engine
void yolo_release(dani_engine* e);
void seq_release(dani_engine* e);
void comm_release(dani_engine* e);   // comm_engine.cpp
SeqState* seq_state(dani_engine* e, int id = 0);   // frontend_engine.cpp
void finish_pending_desc(SeqState* S);
int match_resident_locked(dani_engine* e, SeqState* S, int32_t sequence, const float* pred, int32_t pred_on_device, int32_t pred_cap,
                          float lg_threshold, int32_t* matches, float* scores, int32_t match_cap, int32_t* k_out);
extern thread_local SeqState* t_last_seq;
}  // namespace eng
void launch_pack_features(const float* kimg, const float* desc, int n, int cap, float* dst, hipStream_t s);   // handoff_kernels.hip
namespace eng {
// device-side packing of one frame's features into the slab layout of dani_frontend_export_features (handoff_kernels.hip):
// dst[0] = n, kpts [cap][2], desc [cap][256]; the unused tail of both arrays is zero-filled

}  // namespace eng
}  // namespace dani

// (after every declaration above: the engine's translation units call hipStreamSynchronize by name)
#define hipStreamSynchronize(s_) dani::eng::stream_wait(s_)
