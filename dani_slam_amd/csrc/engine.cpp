// engine.cpp -- implementation of the C ABI in include/dani_frontend.h (SuperPoint part).
//
// Host side of the reference calls replaced here:
//   SPextractor::SPextractor / operator() / ComputeKeyPointsOctTree   /root/reference/src/SPextractor.cpp:80-386
//   SPDetector::detect / getKeyPoints / computeDescriptors             src/SuperPoint.cpp:186-274
// There is no CPU compute path in this file: every stage of the network runs in the HIP kernels of
// conv_mfma.hip / sp_post.hip; only the order-defining quadtree (host_logic.cpp) runs on the host, as it does
// in the reference.
#include "../../include/dani_frontend.h"

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

namespace {

thread_local std::string g_create_error;

struct SpLayerDef { const char* name; int cin, cout, k; };
const SpLayerDef kSp[12] = {
    {"conv1a", 1, 64, 3},    {"conv1b", 64, 64, 3},   {"conv2a", 64, 64, 3},    {"conv2b", 64, 64, 3},
    {"conv3a", 64, 128, 3},  {"conv3b", 128, 128, 3}, {"conv4a", 128, 128, 3},  {"conv4b", 128, 128, 3},
    {"convPa", 128, 256, 3}, {"convPb", 256, 65, 1},  {"convDa", 128, 256, 3},  {"convDb", 256, 256, 1}};
const ConvVariant kSpVariant[12] = {
    CONV3_ROW32_RELU /*unused for 1a*/, CONV3_ROW32_RELU_POOL, CONV3_ROW32_RELU, CONV3_ROW32_RELU_POOL,
    CONV3_ROW32_RELU, CONV3_ROW32_RELU_POOL, CONV3_BLK16_RELU, CONV3_BLK16_RELU,
    CONV3_BLK16_RELU, CONV1_BLK16_LINEAR, CONV3_BLK16_RELU, CONV1_BLK16_LINEAR};

constexpr int CAND_CAP = 32768;  // candidates per frame kept on the device (reference reserves 10*nfeatures)
constexpr int CAND_FAST = 4096;  // candidates per frame read back together with the counts (typical frames: 3-4 k)

struct Context {
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;   // YOLO (concurrent with the SuperPoint conv stack) and overlapped D2H copies
  hipEvent_t ev_desc = nullptr, ev_y0 = nullptr, ev_y1 = nullptr, ev_lg0 = nullptr, ev_lg1 = nullptr;
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
    float* slot_xyr = nullptr;
    int32_t* slot_count = nullptr;
    float* cand = nullptr;
    int32_t* cand_count = nullptr;
    float* h_cand = nullptr;          // pinned
    int32_t* h_cand_count = nullptr;  // pinned
    uint8_t* d_level_img = nullptr;   // level > 0: this level's image [B][H][W] (resized from level - 1)
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
  int32_t* d_nk = nullptr;
  float* d_sampled = nullptr;
  float* d_descout = nullptr;
  int kcap = 0;
  // pinned host staging
  float* h_kxy = nullptr;  // [B][kcap][2] followed by per-level (offset, count) ints [nlevels][2][B]
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
struct SeqState;

}  // namespace

struct dani_engine {
  dani_config cfg;
  int device = 0;
  std::string err;
  std::map<std::string, std::vector<float>> tensors;
  bool finalized = false;
  ConvLayerPlan plan[12];
  float* w1a = nullptr;
  float* b1a = nullptr;
  std::vector<void*> weight_allocs;
  std::vector<Context*> ctxs;
  std::mutex mu;
  bool profiling = false;
  LgWeights* lgw = nullptr;             // packed LightGlue / YOLO weights (owned; device buffers are in weight_allocs)
  YoloWeights* yw = nullptr;
  std::map<int, SeqState*> seqs;        // dani_frontend_batch sequences (own context + cached previous frame), under mu
};

namespace {

int fail(dani_engine* e, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (e) e->err = buf; else g_create_error = buf;
  return code;
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
  // runs fn(0..count-1); the caller takes part; returns when all are done
  void run(int count, const std::function<void(int)>& fn) {
    if (count <= 1 || th_.empty()) { for (int i = 0; i < count; i++) fn(i); return; }
    {
      std::lock_guard<std::mutex> lk(m_);
      fn_ = &fn; count_ = count; next_ = 0; remaining_ = count; gen_++;
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [this] { return remaining_ == 0; });
    fn_ = nullptr;
  }

 private:
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
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
        if (stop_) return;
        seen = gen_;
      }
      work();
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  const std::function<void(int)>* fn_ = nullptr;
  int count_ = 0, next_ = 0, remaining_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};

HostPool& host_pool() {
  thread_local HostPool pool(7);
  return pool;
}

thread_local Context* t_last_ctx = nullptr;
// the debug / timing entry points read "the last context of this thread": only if it still belongs to engine e
Context* last_ctx(dani_engine* e) {
  Context* c = t_last_ctx;
  if (!e || !c) return nullptr;
  std::lock_guard<std::mutex> lk(e->mu);
  for (Context* x : e->ctxs) if (x == c) return c;
  return nullptr;
}
// How the fused front-end (dani_frontend_batch) drives the three stages; the public ABI entry points pass the defaults.
struct SpCall {
  Context* ctx = nullptr;             // run on this context (the sequence's own) instead of acquiring a free one
  int phase = 0;                      // 0: whole extractor; 1: enqueue the network only; 2: everything after it
  bool keep_desc_on_device = false;   // `desc` may be null: descriptors are consumed on the device (LightGlue)
  bool defer_desc_sync = false;       // descriptor read-back into pinned staging on stream2, no final synchronise
};
struct YoloCall {
  Context* ctx = nullptr;
  bool alt_stream = false;            // run on the context's second (high-priority) stream, next to the SuperPoint network
  std::function<void(int, const float*, int)>* post = nullptr;   // per frame, inside the NMS worker: (frame, dets6, n)
};
struct LgCall {
  Context* ctx = nullptr;
  std::function<void()>* overlap = nullptr;   // host work of the caller to run while the forward pass executes
};

// stream2 carries YOLO's ~70 small dependent kernels while the main stream runs SuperPoint's few huge ones: give it
// the highest priority so that its workgroups get the CU slots that free up first (otherwise the chain starves
// behind conv1b's 9600 workgroups and the overlap buys nothing -- measured).
hipError_t create_priority_stream(hipStream_t* s) {
  int lo = 0, hi = 0;
  if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
  return hipStreamCreateWithPriority(s, hipStreamNonBlocking, hi);
}

Context* acquire(dani_engine* e) {
  std::lock_guard<std::mutex> lk(e->mu);
  for (Context* c : e->ctxs)
    if (!c->busy) { c->busy = true; return c; }
  return nullptr;
}
void release(dani_engine* e, Context* c) {
  std::lock_guard<std::mutex> lk(e->mu);
  c->busy = false;
}

// a context for one call: the caller's own (fused front-end) or a free one from the pool, released on scope exit
struct CtxLease {
  dani_engine* e; Context* c; bool owned;
  CtxLease(dani_engine* e_, Context* given) : e(e_), c(given ? given : acquire(e_)), owned(!given) {}
  ~CtxLease() { if (owned && c) release(e, c); }
  CtxLease(const CtxLease&) = delete;
  CtxLease& operator=(const CtxLease&) = delete;
};

int dev_alloc(dani_engine* e, Context* c, void** p, size_t bytes, bool zero) {
  HIP_TRY(e, hipMalloc(p, bytes));
  c->dev_allocs.push_back(*p);
  if (zero) HIP_TRY(e, hipMemsetAsync(*p, 0, bytes, c->stream));
  return 0;
}
int host_alloc(dani_engine* e, Context* c, void** p, size_t bytes) {
  HIP_TRY(e, hipHostMalloc(p, bytes, hipHostMallocDefault));
  c->host_allocs.push_back(*p);
  return 0;
}

void free_buffers(Context* c) {
  for (void* p : c->dev_allocs) (void)hipFree(p);
  for (void* p : c->host_allocs) (void)hipHostFree(p);
  c->dev_allocs.clear();
  c->host_allocs.clear();
  c->lv.clear();
  c->W = c->H = c->B = 0;
  c->nlevels = 0;
}

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
    const int nc = std::max<int>(1, (int)L.cells.size());
    if ((rc = dev_alloc(e, c, (void**)&L.d_cells, sizeof(CellRect) * nc, false))) return rc;
    if (!L.cells.empty())
      HIP_TRY(e, hipMemcpyAsync(L.d_cells, L.cells.data(), sizeof(CellRect) * L.cells.size(), hipMemcpyHostToDevice, c->stream));
    if ((rc = dev_alloc(e, c, (void**)&L.slot_xyr, (size_t)B * nc * L.slot_stride * 3 * 4, false))) return rc;
    if ((rc = dev_alloc(e, c, (void**)&L.slot_count, (size_t)B * nc * 4, true))) return rc;
    if ((rc = dev_alloc(e, c, (void**)&L.cand, (size_t)B * CAND_CAP * 3 * 4, false))) return rc;
    if ((rc = dev_alloc(e, c, (void**)&L.cand_count, (size_t)B * 4, true))) return rc;
    if ((rc = host_alloc(e, c, (void**)&L.h_cand, (size_t)B * CAND_CAP * 3 * 4))) return rc;
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

}  // namespace

// ================================================================================================
extern "C" {

const char* dani_version(void) { return "dani-frontend-hip 0.1 (gfx950)"; }

const char* dani_last_error(const dani_engine* e) { return e ? e->err.c_str() : g_create_error.c_str(); }

int dani_engine_create(const dani_config* cfg, dani_engine** out) {
  if (!cfg || !out) return fail(nullptr, DANI_ERR_INVALID, "null argument");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, DANI_ERR_NO_DEVICE, "no HIP device visible; this engine has no CPU fallback");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, DANI_ERR_NO_DEVICE, "device %d out of range (%d devices)", cfg->device, ndev);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return fail(nullptr, DANI_ERR_NO_DEVICE, "hipGetDeviceProperties failed");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, DANI_ERR_NO_DEVICE, "device %d is %s; the kernels are built for gfx950 (MI355X) only", cfg->device, prop.gcnArchName);
  dani_engine* e = new dani_engine();
  e->cfg = *cfg;
  if (e->cfg.max_batch < 1) e->cfg.max_batch = 1;
  if (e->cfg.n_contexts < 1) e->cfg.n_contexts = 2;
  if (e->cfg.max_width < 80) e->cfg.max_width = 640;
  if (e->cfg.max_height < 80) e->cfg.max_height = 480;
  e->device = cfg->device;
  if (hipSetDevice(e->device) != hipSuccess) { delete e; return fail(nullptr, DANI_ERR_HIP, "hipSetDevice failed"); }
  for (int i = 0; i < e->cfg.n_contexts; i++) {
    Context* c = new Context();
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        create_priority_stream(&c->stream2) != hipSuccess) { delete c; dani_engine_destroy(e); return fail(nullptr, DANI_ERR_HIP, "hipStreamCreate failed"); }
    (void)hipEventCreate(&c->ev_desc); (void)hipEventCreate(&c->ev_y0); (void)hipEventCreate(&c->ev_y1);
    (void)hipEventCreate(&c->ev_lg0); (void)hipEventCreate(&c->ev_lg1);
    for (auto& ev : c->ev) (void)hipEventCreate(&ev);
    for (auto& ev : c->lev) (void)hipEventCreate(&ev);
    e->ctxs.push_back(c);
  }
  *out = e;
  return DANI_OK;
}

namespace {
int lg_finalize(dani_engine* e); int yolo_finalize(dani_engine* e);
// the LightGlue / YOLO / sequence side tables are keyed by engine and context addresses: they must die with them
void lg_release(dani_engine* e); void yolo_release(dani_engine* e); void seq_release(dani_engine* e);
}

void dani_engine_destroy(dani_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  for (Context* c : e->ctxs) {
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
  }
  lg_release(e); yolo_release(e); seq_release(e);
  for (Context* c : e->ctxs) {
    free_buffers(c);
    for (auto& ev : c->ev) if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : c->lev) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : {c->ev_desc, c->ev_y0, c->ev_y1, c->ev_lg0, c->ev_lg1}) if (ev) (void)hipEventDestroy(ev);
    if (c->scan_ws) (void)hipFree(c->scan_ws);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    delete c;
  }
  for (void* p : e->weight_allocs) (void)hipFree(p);
  delete e;
}

int dani_set_tensor(dani_engine* e, const char* name, const float* data, int64_t count) {
  if (!e || !name || !data || count <= 0) return fail(e, DANI_ERR_INVALID, "bad dani_set_tensor argument");
  e->tensors[name].assign(data, data + count);
  e->finalized = false;
  return DANI_OK;
}

int dani_load_weights_file(dani_engine* e, const char* path) {
  // .dsw container: "DSW1", u32 n, then n x { u32 name_len, name, u64 count, count x f32 }
  if (!e || !path) return fail(e, DANI_ERR_INVALID, "null argument");
  FILE* f = fopen(path, "rb");
  if (!f) return fail(e, DANI_ERR_WEIGHTS, "cannot open weight file %s", path);
  char magic[4];
  uint32_t n = 0;
  bool ok = fread(magic, 1, 4, f) == 4 && memcmp(magic, "DSW1", 4) == 0 && fread(&n, 4, 1, f) == 1;
  for (uint32_t i = 0; ok && i < n; i++) {
    uint32_t nl = 0;
    uint64_t cnt = 0;
    ok = fread(&nl, 4, 1, f) == 1 && nl < 256;
    std::string name(nl, '\0');
    ok = ok && fread(&name[0], 1, nl, f) == nl && fread(&cnt, 8, 1, f) == 1 && cnt < (1ull << 28);
    if (!ok) break;
    std::vector<float> v(cnt);
    ok = fread(v.data(), 4, cnt, f) == cnt;
    if (ok) e->tensors[name].swap(v);
  }
  fclose(f);
  if (!ok) return fail(e, DANI_ERR_WEIGHTS, "malformed weight file %s", path);
  e->finalized = false;
  return DANI_OK;
}

}  // extern "C"

extern "C" {

int dani_finalize_weights(dani_engine* e) {
  if (!e) return DANI_ERR_INVALID;
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
    L.Cin = d.cin; L.Cout = d.cout; L.ksize = d.k; L.CinPad = d.cin;
    L.CC = conv_variant_cc(L.variant);
    const int nco = conv_variant_nco(L.variant);
    L.cout_tiles = (d.cout + nco - 1) / nco;
    L.w_packed_floats = (size_t)L.cout_tiles * nco * d.cin * d.k * d.k;
    std::vector<float> wp(L.w_packed_floats), bp((size_t)L.cout_tiles * nco);
    pack_conv_weights(wi->second.data(), bi->second.data(), d.cin, d.cout, d.k, L.CC, nco, wp.data(), bp.data());
    HIP_TRY(e, hipMalloc((void**)&L.w_packed, wp.size() * 4)); e->weight_allocs.push_back(L.w_packed);
    HIP_TRY(e, hipMalloc((void**)&L.bias_packed, bp.size() * 4)); e->weight_allocs.push_back(L.bias_packed);
    HIP_TRY(e, hipMemcpy(L.w_packed, wp.data(), wp.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(L.bias_packed, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
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

namespace {

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
  if (phase != 2) {
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
      if (lprof) HIP_TRY(e, hipEventRecord(c->lev[0], s));
      launch_conv1a(limg, B, L.H, L.W, lstride, lframe, e->w1a, e->b1a, L.act[0], s);
      if (lprof) HIP_TRY(e, hipEventRecord(c->lev[1], s));
      const int src_of[12] = {-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 7, 10};
      for (int i = 1; i < 12; i++) {
        launch_conv_mfma(e->plan[i], L.act[src_of[i]], L.act[i], B, s);
        if (lprof) HIP_TRY(e, hipEventRecord(c->lev[i + 1], s));
      }
      if (l == NL - 1) HIP_TRY(e, hipEventRecord(c->ev[1], s));
      launch_softmax_d2s(L.act[9], L.prob, B, s);
      launch_desc_norm(L.act[11], L.desc_nhwc, L.desc_nchw, B, s);
    }
    HIP_TRY(e, hipGetLastError());
    if (phase == 1) return DANI_OK;   // network enqueued; the caller continues with phase 2 once the boxes are known
  }
  int nboxes_total = 0;
  if (boxes && box_offsets) nboxes_total = box_offsets[B];
  if (nboxes_total > c->box_cap) return fail(e, DANI_ERR_INVALID, "too many dynamic boxes (%d)", nboxes_total);
  {
    int32_t* hb = c->h_boxes;
    if (nboxes_total) memcpy(hb, boxes, (size_t)nboxes_total * 16);
    int32_t* ho = hb + c->box_cap * 4;
    for (int b = 0; b <= B; b++) ho[b] = (boxes && box_offsets) ? box_offsets[b] : 0;
    if (nboxes_total) HIP_TRY(e, hipMemcpyAsync(c->d_boxes, hb, (size_t)nboxes_total * 16, hipMemcpyHostToDevice, s));
    HIP_TRY(e, hipMemcpyAsync(c->d_box_offsets, ho, (size_t)(B + 1) * 4, hipMemcpyHostToDevice, s));
  }
  for (int l = 0; l < NL; l++) {
    Context::Level& L = c->lv[l];
    const int PH = (L.H / 8) * 8, PW = (L.W / 8) * 8, ncells = (int)L.cells.size();
    if (ncells > 0) {
      launch_cell_nms(L.prob, PH, PW, B, L.d_cells, ncells, p->ini_th, p->min_th, c->d_boxes, c->d_box_offsets,
                      19 - 3, scales[l], L.slot_xyr, L.slot_count, L.slot_stride, s);
      launch_cell_compact(L.slot_xyr, L.slot_count, ncells, L.slot_stride, B, L.cand, L.cand_count, CAND_CAP, s);
      HIP_TRY(e, hipMemcpyAsync(L.h_cand_count, L.cand_count, (size_t)B * 4, hipMemcpyDeviceToHost, s));
      // optimistic read-back of the first CAND_FAST candidates of every frame in the same stream pass (one sync
      // instead of two); frames with more candidates get a second copy below
      HIP_TRY(e, hipMemcpy2DAsync(L.h_cand, (size_t)CAND_CAP * 12, L.cand, (size_t)CAND_CAP * 12, (size_t)CAND_FAST * 12, B, hipMemcpyDeviceToHost, s));
    }
  }
  HIP_TRY(e, hipEventRecord(c->ev[2], s));
  HIP_TRY(e, hipGetLastError());
  HIP_TRY(e, hipStreamSynchronize(s));
  bool more = false;
  for (int l = 0; l < NL; l++) {
    Context::Level& L = c->lv[l];
    for (int b = 0; b < B; b++) {
      int nc = L.cells.empty() ? 0 : L.h_cand_count[b];
      if (nc > CAND_CAP) return fail(e, DANI_ERR_CAPACITY, "frame %d level %d has %d candidates (> %d)", b, l, nc, CAND_CAP);
      L.h_cand_count[b] = nc;
      if (nc > CAND_FAST) {
        more = true;
        HIP_TRY(e, hipMemcpyAsync(L.h_cand + ((size_t)b * CAND_CAP + CAND_FAST) * 3, L.cand + ((size_t)b * CAND_CAP + CAND_FAST) * 3, (size_t)(nc - CAND_FAST) * 12, hipMemcpyDeviceToHost, s));
      }
    }
  }
  if (more) HIP_TRY(e, hipStreamSynchronize(s));

  // ---- host: quadtree per (frame, level), keypoint assembly (SPextractor.cpp:359-376, :164-216)
  const auto t0 = std::chrono::steady_clock::now();
  const int minBorderX = 19 - 3, minBorderY = minBorderX;
  int32_t* h_lv = reinterpret_cast<int32_t*>(c->h_kxy + (size_t)MB * c->kcap * 2);   // [level][0: offset, 1: count][MB]
  int status = DANI_OK;
  std::vector<std::vector<int>> sels((size_t)B * NL);
  host_pool().run(B * NL, [&](int t) {
    const int b = t / NL, l = t % NL;
    const Context::Level& L = c->lv[l];
    const HostKey* keys = reinterpret_cast<const HostKey*>(L.h_cand + (size_t)b * CAND_CAP * 3);
    distribute_octtree(keys, L.h_cand_count[b], minBorderX, L.W - 19 + 3, minBorderY, L.H - 19 + 3, per_level[l], sels[t]);
  });
  for (int b = 0; b < B; b++) {
    int n = 0;
    for (int l = 0; l < NL; l++) n += (int)sels[(size_t)b * NL + l].size();
    n_out[b] = n;
    const bool fits = n <= cap && n <= c->kcap;
    if (!fits) status = DANI_ERR_CAPACITY;
    float* kx = c->h_kxy + (size_t)b * c->kcap * 2;
    float* ok = kpts_xy + (size_t)b * cap * 2;
    float* orr = resp + (size_t)b * cap;
    // the mono / stereo partition runs over the levels in order (operator(), :175-213); descriptor rows keep the
    // pre-partition order
    int monoIndex = 0, stereoIndex = n - 1, off = 0;
    const int l0 = lapping ? lapping[0] : 0, l1 = lapping ? lapping[1] : 0;
    for (int l = 0; l < NL; l++) {
      const Context::Level& L = c->lv[l];
      const HostKey* keys = reinterpret_cast<const HostKey*>(L.h_cand + (size_t)b * CAND_CAP * 3);
      const std::vector<int>& sel = sels[(size_t)b * NL + l];
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
        if (octave) octave[(size_t)b * cap + dst] = l;
        if (size) size[(size_t)b * cap + dst] = (float)scaledPatchSize;
      }
      off += nl;
    }
    if (mono_index) mono_index[b] = monoIndex;
  }
  c->ms[2] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (status != DANI_OK) return fail(e, status, "output capacity %d too small", cap);

  // ---- descriptors: each level samples its own map and is normalised over its own keypoints (Q7)
  HIP_TRY(e, hipMemcpyAsync(c->d_kxy, c->h_kxy, ((size_t)MB * c->kcap * 2 + (size_t)NL * 2 * MB) * 4, hipMemcpyHostToDevice, s));
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
  if (desc) {   // descriptor read-back on the second stream so that it can run under the next stage's kernels
    HIP_TRY(e, hipEventRecord(c->ev_desc, s));
    HIP_TRY(e, hipStreamWaitEvent(c->stream2, c->ev_desc, 0));
    // fused call: into pinned staging (a copy into pageable memory would block this thread until it is done); the
    // caller moves it to `desc` while LightGlue runs
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

}  // namespace

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
  if (desc_map) HIP_TRY(e, hipMemcpy(desc_map, L.desc_nchw + (size_t)b * 256 * Hc * Wc, (size_t)256 * Hc * Wc * 4, hipMemcpyDeviceToHost));
  if (n_cand) *n_cand = L.h_cand_count[b];
  if (cand_xyr) {
    const int n = std::min(L.h_cand_count[b], cand_cap);
    memcpy(cand_xyr, L.h_cand + (size_t)b * CAND_CAP * 3, (size_t)n * 12);
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
  float* tmp = nullptr;
  HIP_TRY(e, hipMalloc((void**)&tmp, need * 4));
  launch_copy_layer_dense(a, b, tmp, c->stream);
  hipError_t st = hipMemcpyAsync(out, tmp, need * 4, hipMemcpyDeviceToHost, c->stream);
  if (st == hipSuccess) st = hipStreamSynchronize(c->stream);
  (void)hipFree(tmp);
  if (st != hipSuccess) return fail(e, DANI_ERR_HIP, "debug layer copy failed: %s", hipGetErrorString(st));
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
  distribute_octtree(reinterpret_cast<const HostKey*>(in_xyr), n, min_x, max_x, min_y, max_y, n_features, sel);
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

int dani_host_features_per_level(int32_t nfeatures, float scale_factor, int32_t nlevels, int32_t* per_level,
                                 float* scale_factors) {
  if (nlevels < 1 || !per_level) return DANI_ERR_INVALID;
  std::vector<int> f; std::vector<float> s;
  features_per_level(nfeatures, scale_factor, nlevels, f, s);
  for (int i = 0; i < nlevels; i++) { per_level[i] = f[i]; if (scale_factors) scale_factors[i] = s[i]; }
  return DANI_OK;
}

}  // extern "C"

// ================================================================================================
// LightGlue
// ================================================================================================
namespace {

constexpr int LG_LAYERS = 9;

struct LgLinear { float* w = nullptr; float* b = nullptr; int out = 0, in = 0; };
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
  double attn_flops = 0.0, lin_flops = 0.0;
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
int lg_pack_raw(dani_engine* e, const std::vector<float>& w, const std::vector<float>& b, int out, int in, LgLinear* L) {
  const int tiles = (out + 63) / 64;
  std::vector<float> wp((size_t)tiles * 64 * in), bp((size_t)tiles * 64);
  pack_conv_weights(w.data(), b.data(), in, out, 1, 32, 64, wp.data(), bp.data());
  L->out = out; L->in = in;
  int rc;
  if ((rc = lg_upload(e, wp, &L->w))) return rc;
  return lg_upload(e, bp, &L->b);
}

int lg_get_linear(dani_engine* e, const std::string& name, int out, int in, std::vector<float>& w, std::vector<float>& b) {
  auto wi = e->tensors.find(name + ".weight"), bi = e->tensors.find(name + ".bias");
  if (wi == e->tensors.end() || bi == e->tensors.end()) return fail(e, DANI_ERR_WEIGHTS, "missing tensor %s.weight/.bias", name.c_str());
  if (wi->second.size() != (size_t)out * in || bi->second.size() != (size_t)out)
    return fail(e, DANI_ERR_WEIGHTS, "tensor %s.weight has %zu floats, expected [%d,%d]", name.c_str(), wi->second.size(), out, in);
  w = wi->second; b = bi->second;
  return 0;
}

int lg_pack_linear(dani_engine* e, const std::string& name, int out, int in, LgLinear* L, const std::vector<int>* perm = nullptr) {
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
  return lg_pack_raw(e, w, b, out, in, L);
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
  return lg_pack_raw(e, w, b, 512, 512, L);
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
  return lg_pack_raw(e, w, b, 512, 256, L);
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
    if ((rc = lg_pack_linear(e, ps + ".Wqkv", 768, 256, &S.a, &perm))) return rc;
    if ((rc = lg_pack_ffn0_folded(e, ps + ".ffn.0", ps + ".out_proj", &S.ffn0))) return rc;
    if ((rc = lg_vec(e, ps + ".ffn.1.weight", 512, &S.ln_g))) return rc;
    if ((rc = lg_vec(e, ps + ".ffn.1.bias", 512, &S.ln_b))) return rc;
    if ((rc = lg_pack_linear(e, ps + ".ffn.3", 256, 512, &S.ffn3))) return rc;
    if ((rc = lg_pack_qkv_cross(e, pc, &C.a))) return rc;
    if ((rc = lg_pack_ffn0_folded(e, pc + ".ffn.0", pc + ".to_out", &C.ffn0))) return rc;
    if ((rc = lg_vec(e, pc + ".ffn.1.weight", 512, &C.ln_g))) return rc;
    if ((rc = lg_vec(e, pc + ".ffn.1.bias", 512, &C.ln_b))) return rc;
    if ((rc = lg_pack_linear(e, pc + ".ffn.3", 256, 512, &C.ffn3))) return rc;
  }
  const std::string pa = "log_assignment." + std::to_string(LG_LAYERS - 1);
  if ((rc = lg_pack_linear(e, pa + ".final_proj", 256, 256, &W->final_proj))) return rc;
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
struct LgShape { int nsets, max_n, max_mp, max_span, max_tail, T; };   // T = all main regions + the tail region

void lg_forward(const LgWeights* W, LgContext* L, const LgShape& sh, hipStream_t s, bool prof) {
  const int ld = L->ld, nsets = sh.nsets, max_n = sh.max_n, T = sh.T, max_np = round_up(max_n, 32);
  if (prof && !L->aev_ready) {
    for (auto& ev : L->aev) (void)hipEventCreate(&ev);
    for (auto& ev : L->lnev) (void)hipEventCreate(&ev);
    L->aev_ready = true;
  }
  int ai = 0, li = 0;
  auto attn = [&](const float* Q, const float* K, const float* V, int cross, float scale) {
    if (prof) (void)hipEventRecord(L->aev[ai++], s);
    launch_attention(Q, K, V, L->ctx, ld, L->d_sets, nsets, sh.max_mp, sh.max_tail, cross, scale, s);
    if (prof) (void)hipEventRecord(L->aev[ai++], s);
  };
  launch_desc_to_fm(L->d_sets, nsets, sh.max_span, L->x, ld, s);
  launch_posenc(L->d_sets, nsets, sh.max_span, W->wr, L->cosb, L->sinb, ld, s);
  auto lin = [&](const LgLinear& l, const float* x0, int K0, const float* x1, int K1, float* y, const float* res, int rot, float alpha) {
    if (prof) (void)hipEventRecord(L->lnev[li++], s);
    launch_linear_fm(x0, K0, x1, K1, l.w, l.b, y, l.out, T, ld, res, L->cosb, L->sinb, rot, alpha, s);
    if (prof) (void)hipEventRecord(L->lnev[li++], s);
  };
  for (int i = 0; i < LG_LAYERS; i++) {
    const LgBlockW& S = W->self_[i]; const LgBlockW& C = W->cross_[i];
    // self block: Wqkv (+rotary) -> attention -> [out_proj folded into] ffn.0 -> LN+GELU -> ffn.3 + residual
    lin(S.a, L->x, 256, nullptr, 0, L->qkv, nullptr, 512, 1.0f);
    attn(L->qkv, L->qkv + (size_t)256 * ld, L->qkv + (size_t)512 * ld, 0, 0.125f);
    lin(S.ffn0, L->x, 256, L->ctx, 256, L->hbuf, nullptr, 0, 1.0f);
    launch_ln_gelu(L->hbuf, S.ln_g, S.ln_b, T, ld, s);
    lin(S.ffn3, L->hbuf, 512, nullptr, 0, L->x, L->x, 0, 1.0f);
    // cross block: [to_qk*s | to_v] in one GEMM -> attention over the partner set -> [to_out folded into] ffn.0 ...
    lin(C.a, L->x, 256, nullptr, 0, L->qkv, nullptr, 0, 1.0f);
    attn(L->qkv, L->qkv, L->qkv + (size_t)256 * ld, 1, 1.0f);
    lin(C.ffn0, L->x, 256, L->ctx, 256, L->hbuf, nullptr, 0, 1.0f);
    launch_ln_gelu(L->hbuf, C.ln_g, C.ln_b, T, ld, s);
    lin(C.ffn3, L->hbuf, 512, nullptr, 0, L->x, L->x, 0, 1.0f);
  }
  lin(W->final_proj, L->x, 256, nullptr, 0, L->md, nullptr, 0, 0.25f);   // / d^0.25, d = 256
  launch_matchability(L->x, W->match_w, W->match_b, L->logsig, T, ld, s);
  const size_t pstride = (size_t)(round_up(L->kmax, 32) + 64) * L->sim_ld;
  launch_lg_densify(L->md, L->logsig, L->d_sets, nsets, max_n, L->md_dense, L->logsig_dense, ld, s);
  launch_sim(L->md_dense, ld, L->d_sets, nsets / 2, max_np, L->sim, L->sim_ld, pstride, s);
  launch_assignment_stats(L->sim, L->sim_ld, pstride, L->d_sets, nsets / 2, max_n, L->logsig_dense, L->lse_row, L->lse_col,
                          L->m0, L->max0, L->m1, ld, s);
}

thread_local LgContext* t_last_lg = nullptr;
LgContext* last_lg(dani_engine* e) {
  if (!e || !t_last_lg) return nullptr;
  std::lock_guard<std::mutex> lk(e->mu);
  for (Context* c : e->ctxs) if (c->lg && c->lg == t_last_lg) return t_last_lg;
  return nullptr;
}

}  // namespace

extern "C" {

}  // extern "C"

namespace {

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
  LgShape sh{2 * P, 0, 0, 0, 0, 0};
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
  sh.max_n = max_n; sh.T = T;
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
  // padding columns of the tail region: keep the residual stream finite there (they are never read as keys)
  if (T > T_main + ntail)
    HIP_TRY(e, hipMemset2DAsync(L->x + T_main + ntail, (size_t)L->ld * 4, 0, (size_t)(T - T_main - ntail) * 4, 256, s));
  memcpy(L->h_sets, L->last_sets.data(), sizeof(LgSet) * 2 * P);
  HIP_TRY(e, hipMemcpyAsync(L->d_sets, L->h_sets, sizeof(LgSet) * 2 * P, hipMemcpyHostToDevice, s));
  HIP_TRY(e, hipEventRecord(c->ev_lg0, s));
  const bool prof = e->profiling;
  if (max_n > 0) lg_forward(W, L, sh, s, prof);
  HIP_TRY(e, hipEventRecord(c->ev_lg1, s));
  HIP_TRY(e, hipGetLastError());
  HIP_TRY(e, hipMemcpy2DAsync(L->h_m0, (size_t)L->ld * 4, L->m0, (size_t)L->ld * 4, (size_t)doff * 4, 3, hipMemcpyDeviceToHost, s));
  if (call.overlap) (*call.overlap)();   // host work of the caller that can run under the forward pass
  HIP_TRY(e, hipStreamSynchronize(s));
  (void)hipEventElapsedTime(&c->ms[0], c->ev_lg0, c->ev_lg1);
  c->ms[4] = c->ms[0];
  t_last_ctx = c;
  if (prof && max_n > 0) {
    float tot = 0.0f;
    for (int i = 0; i < 36; i += 2) { float m = 0.0f; (void)hipEventElapsedTime(&m, L->aev[i], L->aev[i + 1]); tot += m; }
    L->attn_ms = tot;
    double fl = 0.0;   // algorithmic: per launch, per set: 4 heads x (QK^T + PV) = 2 * 2 * nq * nk * 256 flop
    for (const LgSet& st : L->last_sets) fl += 9.0 * 4.0 * 256.0 * st.n * ((double)st.n + (double)st.partner_n);
    L->attn_flops = fl;
    tot = 0.0f;
    for (int i = 0; i < 2 * LG_LINEARS; i += 2) { float m = 0.0f; (void)hipEventElapsedTime(&m, L->lnev[i], L->lnev[i + 1]); tot += m; }
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
    int32_t* mo = matches + (size_t)p * cap * 2;
    float* so = scores + (size_t)p * cap;
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

}  // namespace

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

int dani_lg_attention_profile(dani_engine* e, float* total_ms, double* total_flop, int32_t* launches) {
  LgContext* L = last_lg(e);
  if (!e || !L) return DANI_ERR_INVALID;
  if (total_ms) *total_ms = L->attn_ms;
  if (total_flop) *total_flop = L->attn_flops;
  if (launches) *launches = 18;
  return DANI_OK;
}

int dani_lg_linear_profile(dani_engine* e, float* total_ms, double* total_flop, int32_t* launches) {
  LgContext* L = last_lg(e);
  if (!e || !L) return DANI_ERR_INVALID;
  if (total_ms) *total_ms = L->lin_ms;
  if (total_flop) *total_flop = L->lin_flops;
  if (launches) *launches = LG_LINEARS;
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

// ================================================================================================
// YOLOv5s dynamic-object detector
// ================================================================================================
namespace {

constexpr int YOLO_W = 640, YOLO_H = 380;   // cv::resize(mRGB, img, cv::Size(640, 380)), src/YoloDetect.cpp:39
constexpr int YOLO_CAND_CAP = 15120;
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
  std::vector<void*> dev_allocs, host_allocs;
};

void yolo_release(dani_engine* e) {
  std::lock_guard<std::mutex> lk(e->mu);
  for (Context* c : e->ctxs) {
    YoloContext* Y = c->yolo;
    if (!Y) continue;
    for (void* p : Y->dev_allocs) (void)hipFree(p);
    for (void* p : Y->host_allocs) (void)hipHostFree(p);
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
  HIP_TRY(e, hipMalloc((void**)&Y->cand_count, (size_t)B * 4)); Y->dev_allocs.push_back(Y->cand_count);
  HIP_TRY(e, hipHostMalloc((void**)&Y->h_cand, (size_t)B * YOLO_CAND_CAP * 8 * 4, hipHostMallocDefault)); Y->host_allocs.push_back(Y->h_cand);
  HIP_TRY(e, hipHostMalloc((void**)&Y->h_count, (size_t)B * 4, hipHostMallocDefault)); Y->host_allocs.push_back(Y->h_count);
  Y->d_img_bytes = std::max(img_bytes, Y->d_img_bytes);
  HIP_TRY(e, hipMalloc((void**)&Y->d_img, Y->d_img_bytes)); Y->dev_allocs.push_back(Y->d_img);
  // the zero fills above ran on the null stream, which the engine's non-blocking streams do not wait for
  HIP_TRY(e, hipDeviceSynchronize());
  Y->B = B;
  return 0;
}

// a view of channels [c0, c0+C) of buffer `a`
struct YView { const ActDesc* a; int c0, C; };

void yolo_conv(const YoloWeights* W, const std::string& name, YView in, YView out, const YView* res, int B, hipStream_t s) {
  const ConvLayerPlan& L = W->convs.at(name).plan;
  ConvIO io;
  const size_t iplane = (size_t)in.a->Hp * in.a->Wp, oplane = (size_t)out.a->Hp * out.a->Wp;
  io.in = in.a->p + (size_t)in.c0 * iplane; io.out = out.a->p + (size_t)out.c0 * oplane;
  io.res = res ? res->a->p + (size_t)res->c0 * ((size_t)res->a->Hp * res->a->Wp) : nullptr;
  io.H = out.a->H; io.W = out.a->W;
  io.iHp = in.a->Hp; io.iWp = in.a->Wp; io.ipady = in.a->pady; io.oHp = out.a->Hp; io.oWp = out.a->Wp; io.opady = out.a->pady;
  io.in_fs = in.a->frame_stride(); io.out_fs = out.a->frame_stride(); io.res_fs = res ? res->a->frame_stride() : 0;
  launch_conv_io(L, io, B, s);
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

void yolo_forward(const YoloWeights* W, YoloContext* Y, int B, float score_thresh, hipStream_t s) {
  auto& b = Y->buf;
  auto V = [&](const char* n) { const ActDesc& a = b.at(n); return YView{&a, 0, a.C}; };
  auto S = [&](const char* n, int c0, int C) { return YView{&b.at(n), c0, C}; };
  (void)hipMemsetAsync(Y->cand_count, 0, (size_t)B * 4, s);   // candidate counters of the decode at the end
  yolo_conv(W, "model.0.conv", YView{&b.at("in"), 0, 4}, V("b0"), nullptr, B, s);
  yolo_conv(W, "model.1.conv", V("b0"), V("b1"), nullptr, B, s);
  yolo_c3(W, Y, kC3[0], V("b1"), V("b2"), B, s);
  yolo_conv(W, "model.3.conv", V("b2"), V("b3"), nullptr, B, s);
  yolo_c3(W, Y, kC3[1], V("b3"), S("cat16", 128, 128), B, s);            // x4 -> second half of cat16
  yolo_conv(W, "model.5.conv", S("cat16", 128, 128), V("b5"), nullptr, B, s);
  yolo_c3(W, Y, kC3[2], V("b5"), S("cat12", 256, 256), B, s);            // x6 -> second half of cat12
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
  yolo_c3(W, Y, kC3[5], V("cat16"), V("b17"), B, s);
  yolo_conv(W, "model.18.conv", V("b17"), S("cat19", 0, 128), nullptr, B, s);
  yolo_c3(W, Y, kC3[6], V("cat19"), V("b20"), B, s);
  yolo_conv(W, "model.21.conv", V("b20"), S("cat22", 0, 256), nullptr, B, s);
  yolo_c3(W, Y, kC3[7], V("cat22"), V("b23"), B, s);
  yolo_conv(W, "model.24.m.0", V("b17"), V("raw0"), nullptr, B, s);
  yolo_conv(W, "model.24.m.1", V("b20"), V("raw1"), nullptr, B, s);
  yolo_conv(W, "model.24.m.2", V("b23"), V("raw2"), nullptr, B, s);
  const ActDesc* raws[3] = {&b.at("raw0"), &b.at("raw1"), &b.at("raw2")};
  launch_yolo_decode(raws, score_thresh, Y->cand, Y->cand_count, YOLO_CAND_CAP, B, s);
}

thread_local YoloContext* t_last_yolo = nullptr;
YoloContext* last_yolo(dani_engine* e) {
  if (!e || !t_last_yolo) return nullptr;
  std::lock_guard<std::mutex> lk(e->mu);
  for (Context* c : e->ctxs) if (c->yolo && c->yolo == t_last_yolo) return t_last_yolo;
  return nullptr;
}

}  // namespace

extern "C" {

}  // extern "C"

namespace {

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
  launch_yolo_preproc(dimg, B, W, H, stride, fbytes, Y->buf.at("in"), s);
  yolo_forward(Wt, Y, B, score_thresh, s);
  HIP_TRY(e, hipEventRecord(c->ev_y1, s));
  HIP_TRY(e, hipGetLastError());
  // read-back: the counts and, optimistically, the first YOLO_CAND_FAST records of every frame in one round trip;
  // a frame with more candidates than that costs a second copy
  HIP_TRY(e, hipMemcpyAsync(Y->h_count, Y->cand_count, (size_t)B * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(e, hipMemcpy2DAsync(Y->h_cand, (size_t)YOLO_CAND_CAP * 32, Y->cand, (size_t)YOLO_CAND_CAP * 32, (size_t)YOLO_CAND_FAST * 32, B,
                              hipMemcpyDeviceToHost, s));
  HIP_TRY(e, hipStreamSynchronize(s));
  bool more = false;
  for (int b = 0; b < B; b++) {
    if (Y->h_count[b] > YOLO_CAND_CAP) Y->h_count[b] = YOLO_CAND_CAP;
    if (Y->h_count[b] > YOLO_CAND_FAST) {
      more = true;
      HIP_TRY(e, hipMemcpyAsync(Y->h_cand + ((size_t)b * YOLO_CAND_CAP + YOLO_CAND_FAST) * 8, Y->cand + ((size_t)b * YOLO_CAND_CAP + YOLO_CAND_FAST) * 8,
                                (size_t)(Y->h_count[b] - YOLO_CAND_FAST) * 32, hipMemcpyDeviceToHost, s));
    }
  }
  if (more) HIP_TRY(e, hipStreamSynchronize(s));
  if (!call.alt_stream) { (void)hipEventElapsedTime(&c->ms[0], c->ev_y0, c->ev_y1); c->ms[4] = c->ms[0]; }
  int status = DANI_OK;
  // greedy NMS is O(k n) per frame and independent across frames
  host_pool().run(B, [&](int b) {
    n_out[b] = yolo_nms_host(Y->h_cand + (size_t)b * YOLO_CAND_CAP * 8, Y->h_count[b], iou_thresh, dets6 + (size_t)b * cap * 6, cap);
    if (call.post && n_out[b] <= cap) (*call.post)(b, dets6 + (size_t)b * cap * 6, n_out[b]);
  });
  for (int b = 0; b < B; b++)
    if (n_out[b] > cap) status = DANI_ERR_CAPACITY;
  if (status != DANI_OK) return fail(e, status, "detection capacity %d too small", cap);
  return DANI_OK;
}

}  // namespace

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

// ================================================================================================
// Fused per-frame front-end (sequence mode)
// ================================================================================================
namespace {

struct SeqState {
  std::mutex mu;
  Context* ctx = nullptr;
  std::vector<std::string> class_names, dynamic_names;
  float ms[4] = {0, 0, 0, 0};
  // per-call host scratch, kept between calls (a fresh 0.8 MB vector per call costs a memset on the critical path)
  std::vector<float> dets, size;
  std::vector<int32_t> octave, frame_boxes;
};
// one SeqState (context + cached previous frame) per (engine, sequence id); id 0 also holds the class lists
SeqState* seq_state(dani_engine* e, int id = 0) {
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
  for (auto& kv : e->seqs) { if (t_last_seq == kv.second) t_last_seq = nullptr; delete kv.second; }
  e->seqs.clear();
}

}  // namespace

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
  if (S->ctx) S->ctx->prev_n = 0;
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
  SeqState* S0 = seq_state(e, 0);
  SeqState* S = seq_state(e, sequence);
  std::lock_guard<std::mutex> seq_lock(S->mu);
  std::vector<std::string> class_names, dynamic_names;
  if (p->use_yolo) {
    if (S != S0) { std::lock_guard<std::mutex> lk0(S0->mu); class_names = S0->class_names; dynamic_names = S0->dynamic_names; }
    else { class_names = S0->class_names; dynamic_names = S0->dynamic_names; }
    if (class_names.empty()) return fail(e, DANI_ERR_INVALID, "call dani_yolo_set_classes first (coco.names)");
  }
  HIP_TRY(e, hipSetDevice(e->device));
  if (!S->ctx) {
    Context* c = new Context();
    HIP_TRY(e, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(e, create_priority_stream(&c->stream2));
    (void)hipEventCreate(&c->ev_desc); (void)hipEventCreate(&c->ev_y0); (void)hipEventCreate(&c->ev_y1);
    (void)hipEventCreate(&c->ev_lg0); (void)hipEventCreate(&c->ev_lg1);
    for (auto& ev : c->ev) (void)hipEventCreate(&ev);
    for (auto& ev : c->lev) (void)hipEventCreate(&ev);
    c->busy = true;   // never handed out by acquire()
    { std::lock_guard<std::mutex> lk(e->mu); e->ctxs.push_back(c); }
    S->ctx = c;
  }
  Context* c = S->ctx;
  t_last_seq = S;
  const auto t0 = std::chrono::steady_clock::now();
  auto ms_since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - a).count(); };

  // ---- 0. enqueue the SuperPoint network first: it does not need the boxes, and running it on the main stream
  //         while YOLO's ~70 small kernels run on the second stream fills the CUs those leave idle
  std::vector<int32_t>& octave = S->octave;
  std::vector<float>& size = S->size;
  if (octave.size() < (size_t)B * cap) { octave.resize((size_t)B * cap); size.resize((size_t)B * cap); }
  SpCall sp_call;
  sp_call.ctx = c; sp_call.keep_desc_on_device = true; sp_call.phase = 1;
  int rc = sp_extract_impl(e, sp_call, gray, on_device, B, W, H, gray_stride, &p->sp, nullptr, nullptr, p->lapping, kpts_xy, resp,
                           octave.data(), size.data(), desc, cap, n_out, mono_index);
  if (rc) return rc;

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
    rc = yolo_detect_impl(e, yolo_call, bgr, on_device, B, W, H, bgr_stride, p->yolo_score_thresh, p->yolo_iou_thresh, dets.data(), dcap, nd.data());
    if (rc) return rc;
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

  // ---- 2. SuperPoint with the boxes as mvDynamicArea; descriptors stay resident for LightGlue
  const auto t1 = std::chrono::steady_clock::now();
  sp_call.phase = 2; sp_call.defer_desc_sync = true;
  rc = sp_extract_impl(e, sp_call, gray, on_device, B, W, H, gray_stride, &p->sp, boxes.empty() ? nullptr : boxes.data(),
                       boxes.empty() ? nullptr : offs.data(), p->lapping, kpts_xy, resp, octave.data(), size.data(),
                       desc, cap, n_out, mono_index);
  if (rc) return rc;
  S->ms[1] = ms_since(t1);

  // ---- 3. LightGlue: frame b-1 (query) vs frame b (train); slot 0 of the feature slabs = previous call's last frame
  const auto t2 = std::chrono::steady_clock::now();
  // descriptor read-back: stream2 fills the pinned staging while LightGlue is being launched; the copy into the
  // caller's array then runs on this thread under the forward pass
  bool desc_moved = false;
  bool cache_enqueued = false;
  auto enqueue_cache = [&] {   // the last frame's features become the next call's predecessor (slot 0 of the slabs)
    if (cache_enqueued) return;
    cache_enqueued = true;
    (void)hipMemcpyAsync(c->d_kxy_all, c->d_kxy + (size_t)(B - 1) * c->kcap * 2, (size_t)c->kcap * 2 * 4, hipMemcpyDeviceToDevice, c->stream);
    (void)hipMemcpyAsync(c->d_desc_all, c->d_descout + (size_t)(B - 1) * c->kcap * 256, (size_t)c->kcap * 256 * 4, hipMemcpyDeviceToDevice, c->stream);
  };
  std::function<void()> move_desc = [&] {
    enqueue_cache();           // behind the LightGlue kernels in stream order, ahead of the call's final synchronise
    if (desc_moved || !desc) return;
    desc_moved = true;
    if (hipStreamSynchronize(c->stream2) != hipSuccess) return;
    host_pool().run(B, [&](int b) {
      if (n_out[b]) memcpy(desc + (size_t)b * cap * 256, c->h_desc + (size_t)b * c->kcap * 256, (size_t)n_out[b] * 1024);
    });
  };
  if (p->match_previous) {
    std::vector<int32_t> n0(B), n1(B);
    for (int b = 0; b < B; b++) { n0[b] = b ? n_out[b - 1] : c->prev_n; n1[b] = n_out[b]; }
    const float sz[2] = {(float)W, (float)H};
    LgCall lg_call;
    lg_call.ctx = c; lg_call.overlap = &move_desc;
    rc = lg_match_impl(e, lg_call, B, c->kcap, 1, c->d_kxy_all, c->d_desc_all, n0.data(), c->d_kxy, c->d_descout, n1.data(), sz, sz,
                             p->lg_threshold, matches, scores, match_cap, k_out);
    if (rc) { move_desc(); (void)hipStreamSynchronize(c->stream); return rc; }
  }
  const bool synced = cache_enqueued;   // LightGlue ran: its final synchronise covered the cache copies
  move_desc();
  if (!synced) HIP_TRY(e, hipStreamSynchronize(c->stream));
  HIP_TRY(e, hipStreamSynchronize(c->stream2));
  HIP_TRY(e, hipGetLastError());
  if (e->profiling) sp_read_timers(c, true);
  c->prev_n = n_out[B - 1];
  S->ms[2] = ms_since(t2);
  S->ms[3] = ms_since(t0);
  return DANI_OK;
}

}  // extern "C"

// ================================================================================================
// descriptor-distance scans of the projection searches
// ================================================================================================
extern "C" int dani_desc_best2(dani_engine* e, const float* q_desc, int32_t nq, const float* t_desc, int32_t nt,
                               const int32_t* t_level, const int32_t* cand_offsets, const int32_t* cand_idx,
                               float* best_dist, float* best_dist2, int32_t* best_idx, int32_t* best_level,
                               int32_t* best_level2) {
  if (!e) return DANI_ERR_INVALID;
  if (nq < 0 || nt < 0 || (nq && (!q_desc || !cand_offsets || !best_dist || !best_dist2 || !best_idx || !best_level || !best_level2)))
    return fail(e, DANI_ERR_INVALID, "bad dani_desc_best2 argument");
  if (nq == 0) return DANI_OK;
  const int total = cand_offsets[nq];
  if (total < 0 || (total && (!cand_idx || !t_desc))) return fail(e, DANI_ERR_INVALID, "bad candidate lists");
  for (int i = 0; i < total; i++)
    if (cand_idx[i] < 0 || cand_idx[i] >= nt) return fail(e, DANI_ERR_INVALID, "candidate index %d outside [0, %d)", cand_idx[i], nt);
  HIP_TRY(e, hipSetDevice(e->device));
  Context* c = acquire(e);
  if (!c) return fail(e, DANI_ERR_INVALID, "all contexts busy");
  struct Releaser { dani_engine* e; Context* c; ~Releaser() { release(e, c); } } rel{e, c};
  hipStream_t s = c->stream;
  const size_t bq = (size_t)nq * 1024, bt = (size_t)std::max(nt, 1) * 1024, bl = (size_t)std::max(nt, 1) * 4;
  const size_t bo = (size_t)(nq + 1) * 4, bc = (size_t)std::max(total, 1) * 4, bout = (size_t)nq * 4;
  const size_t need = bq + bt + bl + bo + bc + 5 * bout;
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
  HIP_TRY(e, hipMemcpyAsync(dq, q_desc, bq, hipMemcpyHostToDevice, s));
  if (nt) HIP_TRY(e, hipMemcpyAsync(dt, t_desc, (size_t)nt * 1024, hipMemcpyHostToDevice, s));
  if (t_level && nt) HIP_TRY(e, hipMemcpyAsync(dl, t_level, (size_t)nt * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(e, hipMemcpyAsync(doff, cand_offsets, bo, hipMemcpyHostToDevice, s));
  if (total) HIP_TRY(e, hipMemcpyAsync(dci, cand_idx, (size_t)total * 4, hipMemcpyHostToDevice, s));
  launch_desc_best2(dq, dt, t_level ? dl : nullptr, doff, dci, nq, (float*)o, (float*)(o + bout), (int32_t*)(o + 2 * bout),
                    (int32_t*)(o + 3 * bout), (int32_t*)(o + 4 * bout), s);
  HIP_TRY(e, hipGetLastError());
  HIP_TRY(e, hipMemcpyAsync(best_dist, o, bout, hipMemcpyDeviceToHost, s));
  HIP_TRY(e, hipMemcpyAsync(best_dist2, o + bout, bout, hipMemcpyDeviceToHost, s));
  HIP_TRY(e, hipMemcpyAsync(best_idx, o + 2 * bout, bout, hipMemcpyDeviceToHost, s));
  HIP_TRY(e, hipMemcpyAsync(best_level, o + 3 * bout, bout, hipMemcpyDeviceToHost, s));
  HIP_TRY(e, hipMemcpyAsync(best_level2, o + 4 * bout, bout, hipMemcpyDeviceToHost, s));
  HIP_TRY(e, hipStreamSynchronize(s));
  return DANI_OK;
}
