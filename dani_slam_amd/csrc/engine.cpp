// engine.cpp -- engine lifetime, weights container, contexts and the small shared helpers of the C ABI in
// include/dani_frontend.h.  The stages live in sp_engine.cpp (SuperPoint), lg_engine.cpp (LightGlue), yolo_engine.cpp
// (YOLOv5s) and frontend_engine.cpp (fused per-frame call, descriptor scans).  There is no CPU compute path anywhere:
// every network stage runs in the HIP kernels; only order-defining host logic (host_logic.cpp) runs on the host, as it
// does in the reference.
#include "engine_internal.h"
#include <ctime>
#include <unordered_map>

using namespace dani;
using namespace dani::eng;

namespace dani {
namespace eng {

std::atomic<int> g_host_wait{-1};
thread_local HostTrace t_trace;
thread_local std::string g_create_error;
// Error text is kept per calling thread (several host threads share one engine: stereo extractors, mapping threads):
// the message of this thread's last failed call, tagged with the engine it belongs to; e->err (under err_mu) keeps the
// engine's most recent one for threads that have none of their own.
thread_local std::string t_err;
thread_local const dani_engine* t_err_engine = nullptr;
thread_local std::string t_err_copy;

int fail(dani_engine* e, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (e) {
    t_err = buf; t_err_engine = e;
    std::lock_guard<std::mutex> lk(e->err_mu);
    e->err = buf;
  } else g_create_error = buf;
  return code;
}

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
// stream2 carries YOLO's ~70 small dependent kernels while the main stream runs SuperPoint's few huge ones: give it
// the highest priority so that its workgroups get the CU slots that free up first (otherwise the chain starves
// behind conv1b's 9600 workgroups and the overlap buys nothing -- measured).
hipError_t create_priority_stream(hipStream_t* s) {
  int lo = 0, hi = 0;
  if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
  return hipStreamCreateWithPriority(s, hipStreamNonBlocking, hi);
}

// Streams and events of a context, every creation checked: a context either has all of them or is torn down again.
hipError_t context_create_sync(Context* c) {
  hipError_t rc = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (rc == hipSuccess) rc = create_priority_stream(&c->stream2);
  for (hipEvent_t* ev : {&c->ev_desc, &c->ev_y0, &c->ev_y1, &c->ev_lg0, &c->ev_lg1})
    if (rc == hipSuccess) rc = hipEventCreate(ev);
  for (hipEvent_t* ev : {&c->ev_fork0, &c->ev_fork1})   // ordering only (SuperPoint head fork), never timed
    if (rc == hipSuccess) rc = hipEventCreateWithFlags(ev, hipEventDisableTiming);
  for (auto& ev : c->ev) if (rc == hipSuccess) rc = hipEventCreate(&ev);
  for (auto& ev : c->lev) if (rc == hipSuccess) rc = hipEventCreate(&ev);
  if (rc != hipSuccess) context_destroy_sync(c);
  return rc;
}
void context_destroy_sync(Context* c) {
  for (auto& ev : c->ev) if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
  for (auto& ev : c->lev) if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
  for (hipEvent_t* ev : {&c->ev_desc, &c->ev_y0, &c->ev_y1, &c->ev_lg0, &c->ev_lg1, &c->ev_fork0, &c->ev_fork1})
    if (*ev) { (void)hipEventDestroy(*ev); *ev = nullptr; }
  if (c->stream) { near_forget(c->stream); (void)hipStreamDestroy(c->stream); c->stream = nullptr; }
  if (c->stream2) { near_forget(c->stream2); (void)hipStreamDestroy(c->stream2); c->stream2 = nullptr; }
}

// A free context of the pool; when all n_contexts are in use the caller WAITS for one (calls are short and always
// release): more concurrent callers than contexts is a throughput matter, never an error.
Context* acquire(dani_engine* e) {
  std::unique_lock<std::mutex> lk(e->mu);
  for (;;) {
    for (Context* c : e->ctxs)
      if (!c->busy) { c->busy = true; return c; }
    e->ctx_free.wait(lk);
  }
}
void release(dani_engine* e, Context* c) {
  { std::lock_guard<std::mutex> lk(e->mu); c->busy = false; }
  e->ctx_free.notify_one();
}

int dev_alloc(dani_engine* e, Context* c, void** p, size_t bytes, bool zero) {
  HIP_TRY(e, hipMalloc(p, bytes));
  c->dev_allocs.push_back(*p);
  if (zero) HIP_TRY(e, hipMemsetAsync(*p, 0, bytes, c->stream));
  return 0;
}
// ---- host waits (engine_internal.h) --------------------------------------------------------------------------------------
namespace {
struct NearMark { hipEvent_t ev = nullptr; bool armed = false; };
std::mutex g_near_mu;
std::unordered_map<hipStream_t, NearMark> g_near;
inline int wait_mode() {
  static const int env_mode = [] {
    const char* m = getenv("DANI_HOST_WAIT");
    if (m) return !strcmp(m, "block") ? 0 : (!strcmp(m, "hybrid") ? 2 : 1);
    return 1;   // spin: what the runtime's own wait costs in CPU on this ROCm build, without its wake-up outliers (engine_internal.h)
  }();
  const int mode = g_host_wait.load(std::memory_order_relaxed);
  return mode < 0 ? env_mode : mode;
}
inline double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

int wait_mode_now() { return wait_mode(); }

void near_mark(hipStream_t s) {
  if (wait_mode() != 2) return;
  std::lock_guard<std::mutex> lk(g_near_mu);
  NearMark& m = g_near[s];
  if (!m.ev && hipEventCreateWithFlags(&m.ev, hipEventDisableTiming) != hipSuccess) { m.ev = nullptr; return; }
  m.armed = hipEventRecord(m.ev, s) == hipSuccess;
}
void near_forget(hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_near_mu);
  auto it = g_near.find(s);
  if (it == g_near.end()) return;
  if (it->second.ev) (void)hipEventDestroy(it->second.ev);
  g_near.erase(it);
}
hipError_t stream_wait(hipStream_t s, void (*before_spin)(void*), void* arg) {
  const int mode = wait_mode();
  if (mode == 0) { if (before_spin) before_spin(arg); return (hipStreamSynchronize)(s); }
  auto spin = [&](double limit_us, bool* done) -> hipError_t {   // poll until the stream is idle or limit_us have passed
    const double t0 = now_us();
    for (;;) {
      const hipError_t st = hipStreamQuery(s);
      if (st != hipErrorNotReady) { *done = true; return st; }
      for (int k = 0; k < 32; k++) __builtin_ia32_pause();
      if (limit_us > 0 && now_us() - t0 > limit_us) { *done = false; return hipSuccess; }
    }
  };
  bool done = false;
  if (mode == 1) { if (before_spin) before_spin(arg); return spin(0, &done); }
  hipEvent_t ev = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_near_mu);
    auto it = g_near.find(s);
    if (it != g_near.end() && it->second.armed) { ev = it->second.ev; it->second.armed = false; }
  }
  if (ev) {   // stage 1: nap until the marker has passed (a stream error ends the naps; stage 2 then reports it)
    while (hipEventQuery(ev) == hipErrorNotReady) {
      timespec ts{0, 40000};
      nanosleep(&ts, nullptr);
    }
    if (before_spin) before_spin(arg);
    const hipError_t st = spin(3000.0, &done);   // stage 2: the marker sits 0.2-0.6 ms ahead of the end; 3 ms = something else is queued behind it
    if (done) return st;
    return (hipStreamSynchronize)(s);
  }
  if (before_spin) before_spin(arg);
  const hipError_t st = spin(150.0, &done);
  if (done) return st;
  return (hipStreamSynchronize)(s);
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


}  // namespace eng
}  // namespace dani

// ================================================================================================
extern "C" {

int dani_set_host_wait(int32_t mode) {
  if (mode < 0 || mode > 2) return DANI_ERR_INVALID;
  dani::eng::g_host_wait.store(mode, std::memory_order_relaxed);
  return DANI_OK;
}

int dani_get_host_wait(void) { return dani::eng::wait_mode_now(); }

const char* dani_version(void) { return "dani-frontend-hip 0.1 (gfx950)"; }

const char* dani_last_error(const dani_engine* e) {
  if (!e) return g_create_error.c_str();
  if (t_err_engine == e) return t_err.c_str();          // this thread's own last failure on this engine
  std::lock_guard<std::mutex> lk(const_cast<dani_engine*>(e)->err_mu);
  t_err_copy = e->err;                                   // another thread's: a private copy, never a pointer into shared state
  return t_err_copy.c_str();
}

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
    if (context_create_sync(c) != hipSuccess) { delete c; dani_engine_destroy(e); return fail(nullptr, DANI_ERR_HIP, "stream / event creation failed"); }
    e->ctxs.push_back(c);
  }
  *out = e;
  return DANI_OK;
}

void dani_engine_destroy(dani_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  for (Context* c : e->ctxs) {
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
  }
  comm_release(e);   // after the stream synchronises above: no collective is in flight
  lg_release(e); yolo_release(e); seq_release(e);
  for (Context* c : e->ctxs) {
    free_buffers(c);
    if (c->scan_ws) (void)hipFree(c->scan_ws);
    context_destroy_sync(c);
    delete c;
  }
  for (void* p : e->weight_allocs) (void)hipFree(p);
  delete e;
}

int dani_set_tensor(dani_engine* e, const char* name, const float* data, int64_t count) {
  if (!e || !name || !data || count <= 0) return fail(e, DANI_ERR_INVALID, "bad dani_set_tensor argument");
  try {
    e->tensors[name].assign(data, data + count);
  } catch (const std::exception&) {
    return fail(e, DANI_ERR_INVALID, "out of host memory for tensor %s (%lld floats)", name, (long long)count);
  }
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
  try {   // no exception crosses the ABI: a malformed or huge file is an error code, not a std::bad_alloc
    for (uint32_t i = 0; ok && i < n; i++) {
      uint32_t nl = 0;
      uint64_t cnt = 0;
      if (fread(&nl, 4, 1, f) != 1 || nl == 0 || nl >= 256) { ok = false; break; }   // checked BEFORE anything is sized by it
      std::string name(nl, '\0');
      ok = fread(&name[0], 1, nl, f) == nl && fread(&cnt, 8, 1, f) == 1 && cnt > 0 && cnt < (1ull << 28);
      if (!ok) break;
      std::vector<float> v(cnt);
      ok = fread(v.data(), 4, cnt, f) == cnt;
      if (ok) e->tensors[name].swap(v);
    }
  } catch (const std::exception&) {
    ok = false;
  }
  fclose(f);
  if (!ok) return fail(e, DANI_ERR_WEIGHTS, "malformed weight file %s", path);
  e->finalized = false;
  return DANI_OK;
}

}  // extern "C"

