// comm_engine.cpp -- the one collective of the front-end, native: an RCCL all-gather of shard-boundary features over xGMI,
// behind the C ABI (dani_comm_*, dani_frontend_exchange_boundary, dani_frontend_match_exchanged).
//
// No reference counterpart: the reference is one process on one device.  BASELINE config #4 is ONE batch
// of consecutive frames sharded over the GPUs of a node; LightGlue matches every frame against its predecessor
// (LGmatcher::SearchByBoW / Frame::ComputeStereoMatches style pair matching, /root/reference/src/Frame.cc:813-860), so the pair
// (last frame of rank r-1, first frame of rank r) needs features that live on another GPU:
//     pack kernel (count | keypoints | descriptors, 1.06 MB)  ->  ncclAllGather on the sequence's own stream  ->
//     LightGlue of the resident frames with slab r-1 as predecessor, ordered behind the collective by that stream.
// Nothing of it crosses the host, and there is no host synchronise between the three steps.
//
// RCCL is loaded with dlopen when the first communicator is created: a single-GPU tracker never needs it, and the library
// must bind to the HIP runtime THIS library runs on (a process that imported PyTorch first runs on the runtime PyTorch
// bundles, together with the RCCL it bundles; a plain C++ host gets ROCm's own pair through this library's RUNPATH --
// the dynamic loader resolves "librccl.so.1" consistently in both cases).
#include "engine_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

using namespace dani;
using namespace dani::eng;

namespace dani {
namespace eng {

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  std::string err;
};

static RcclApi* rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {getenv("DANI_RCCL_LIB"), "librccl.so.1", "librccl.so"};
    for (const char* n : names) {
      if (!n || !*n) continue;
      api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (api.lib) break;
      api.err = dlerror();
    }
    if (!api.lib) return;
    auto sym = [&](const char* s) { void* p = dlsym(api.lib, s); if (!p) api.err = std::string("librccl has no symbol ") + s; return p; };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    api.GetVersion = (decltype(api.GetVersion))sym("ncclGetVersion");
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.GetErrorString) { dlclose(api.lib); api.lib = nullptr; }
  });
  return &api;
}

struct CommState {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  int version = 0;
  long collectives = 0;                 // ncclAllGather calls enqueued so far
  struct Buf { float* send = nullptr; float* gathered = nullptr; int cap = 0; bool valid = false; };
  std::map<int, Buf> bufs;              // per sequence: send slab + [world][slab] receive buffer, engine-owned
  float* carry = nullptr;               // rank 0: the last rank's slab of the previous exchange (predecessor of the next batch's first frame)
  int carry_cap = 0;
  bool carry_valid = false;
  hipEvent_t carry_ev = nullptr;        // recorded behind the copy into `carry`: sequences alternate (two streams), the reader waits for it
  std::mutex mu;
};

static size_t slab_floats(int cap) { return (size_t)1 + (size_t)cap * 258; }

void comm_release(dani_engine* e) {
  CommState* C = e->comm;
  if (!C) return;
  e->comm = nullptr;
  for (auto& kv : C->bufs) { if (kv.second.send) (void)hipFree(kv.second.send); if (kv.second.gathered) (void)hipFree(kv.second.gathered); }
  if (C->carry) (void)hipFree(C->carry);
  if (C->carry_ev) (void)hipEventDestroy(C->carry_ev);
  if (C->comm) (void)rccl_api()->CommDestroy(C->comm);
  delete C;
}

}  // namespace eng
}  // namespace dani

#define RCCL_TRY(e, api, call)                                                                                           \
  do {                                                                                                                   \
    ncclResult_t _st = (call);                                                                                           \
    if (_st != ncclSuccess) return fail((e), DANI_ERR_HIP, "%s failed: %s (%s:%d)", #call, (api)->GetErrorString(_st), __FILE__, __LINE__); \
  } while (0)

extern "C" {

int dani_comm_available(void) {
  RcclApi* api = rccl_api();
  if (!api->lib) return fail(nullptr, DANI_ERR_UNSUPPORTED, "librccl could not be loaded: %s", api->err.c_str());
  return DANI_OK;
}

int dani_comm_unique_id(void* id, int32_t id_bytes) {
  if (!id || id_bytes < DANI_COMM_ID_BYTES) return DANI_ERR_INVALID;
  RcclApi* api = rccl_api();
  if (!api->lib) return fail(nullptr, DANI_ERR_UNSUPPORTED, "librccl could not be loaded: %s", api->err.c_str());
  static_assert(sizeof(ncclUniqueId) == DANI_COMM_ID_BYTES, "RCCL's unique id is 128 bytes");
  ncclUniqueId u;
  const ncclResult_t st = api->GetUniqueId(&u);
  if (st != ncclSuccess) return fail(nullptr, DANI_ERR_HIP, "ncclGetUniqueId failed: %s", api->GetErrorString(st));
  memcpy(id, &u, sizeof u);
  return DANI_OK;
}

int dani_comm_create(dani_engine* e, const void* id, int32_t id_bytes, int32_t rank, int32_t world) {
  if (!e) return DANI_ERR_INVALID;
  if (!id || id_bytes < DANI_COMM_ID_BYTES || world < 1 || rank < 0 || rank >= world) return fail(e, DANI_ERR_INVALID, "bad dani_comm_create argument (rank %d of %d)", rank, world);
  if (e->comm) return fail(e, DANI_ERR_INVALID, "the engine already has a communicator (dani_comm_destroy first)");
  RcclApi* api = rccl_api();
  if (!api->lib) return fail(e, DANI_ERR_UNSUPPORTED, "librccl could not be loaded: %s", api->err.c_str());
  HIP_TRY(e, hipSetDevice(e->device));
  ncclUniqueId u;
  memcpy(&u, id, sizeof u);
  CommState* C = new CommState();
  C->rank = rank; C->world = world;
  const ncclResult_t st = api->CommInitRank(&C->comm, world, u, rank);   // collective over the `world` ranks: every rank calls it
  if (st != ncclSuccess) { delete C; return fail(e, DANI_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, api->GetErrorString(st)); }
  if (api->GetVersion) (void)api->GetVersion(&C->version);
  if (hipEventCreateWithFlags(&C->carry_ev, hipEventDisableTiming) != hipSuccess) {
    (void)api->CommDestroy(C->comm); delete C;
    return fail(e, DANI_ERR_HIP, "hipEventCreate failed (communicator)");
  }
  e->comm = C;
  return DANI_OK;
}

int dani_comm_destroy(dani_engine* e) {
  if (!e) return DANI_ERR_INVALID;
  if (!e->comm) return DANI_OK;
  (void)hipSetDevice(e->device);
  {   // collectives still in flight on the sequences' streams must have finished before their communicator goes
    std::lock_guard<std::mutex> lk(e->mu);
    for (auto& kv : e->seqs) if (kv.second->ctx && kv.second->ctx->stream) (void)hipStreamSynchronize(kv.second->ctx->stream);
  }
  comm_release(e);
  return DANI_OK;
}

int dani_comm_info(dani_engine* e, int32_t* rank, int32_t* world, int32_t* rccl_version, int64_t* collectives) {
  if (!e) return DANI_ERR_INVALID;
  CommState* C = e->comm;
  if (!C) return fail(e, DANI_ERR_INVALID, "no communicator (dani_comm_create)");
  if (rank) *rank = C->rank;
  if (world) *world = C->world;
  if (rccl_version) *rccl_version = C->version;
  if (collectives) *collectives = C->collectives;
  return DANI_OK;
}

int dani_frontend_exchange_boundary(dani_engine* e, int32_t sequence, int32_t frame, int32_t cap, float* gathered_dev,
                                    const float** gathered_out, int32_t* n_out) {
  if (!e) return DANI_ERR_INVALID;
  CommState* C = e->comm;
  if (!C) return fail(e, DANI_ERR_INVALID, "no communicator (dani_comm_create)");
  if (cap < 1) return fail(e, DANI_ERR_INVALID, "bad dani_frontend_exchange_boundary argument");
  RcclApi* api = rccl_api();
  SeqState* S = seq_state(e, sequence);
  std::lock_guard<std::mutex> seq_lock(S->mu);
  Context* c = S->ctx;
  if (!c || frame < 0 || frame >= S->res_B) return fail(e, DANI_ERR_INVALID, "sequence %d holds no resident frame %d", sequence, frame);
  const int n = S->res_n[frame];
  if (n_out) *n_out = n;
  if (n > cap) return fail(e, DANI_ERR_CAPACITY, "frame has %d keypoints (> %d)", n, cap);
  HIP_TRY(e, hipSetDevice(e->device));
  std::lock_guard<std::mutex> lk(C->mu);
  CommState::Buf& b = C->bufs[sequence];
  const size_t nfl = slab_floats(cap);
  if (b.cap != cap) {
    HIP_TRY(e, hipStreamSynchronize(c->stream));
    if (b.send) (void)hipFree(b.send);
    if (b.gathered) (void)hipFree(b.gathered);
    b = CommState::Buf();
    HIP_TRY(e, hipMalloc((void**)&b.send, nfl * 4));
    HIP_TRY(e, hipMalloc((void**)&b.gathered, nfl * 4 * C->world));
    b.cap = cap;
  }
  float* recv = gathered_dev ? gathered_dev : b.gathered;
  launch_pack_features(c->d_kimg + (size_t)frame * c->kcap * 2, c->d_descout + (size_t)frame * c->kcap * 256, n, cap, b.send, c->stream);
  HIP_TRY(e, hipGetLastError());
  // every rank's slab to every rank: 1 MB per rank, one direct exchange over the xGMI links (RCCL picks the algorithm); enqueued on
  // the sequence's stream, so the LightGlue pass that reads the result is ordered behind it without a host wait
  RCCL_TRY(e, api, api->AllGather(b.send, recv, nfl, ncclFloat, C->comm, c->stream));
  C->collectives++;
  b.valid = !gathered_dev;
  if (gathered_out) *gathered_out = recv;
  return DANI_OK;
}

int dani_frontend_match_exchanged(dani_engine* e, int32_t sequence, float lg_threshold, int32_t* matches, float* scores,
                                  int32_t match_cap, int32_t* k_out) {
  if (!e) return DANI_ERR_INVALID;
  CommState* C = e->comm;
  if (!C) return fail(e, DANI_ERR_INVALID, "no communicator (dani_comm_create)");
  if (!matches || !scores || !k_out || match_cap < 1) return fail(e, DANI_ERR_INVALID, "bad dani_frontend_match_exchanged argument");
  SeqState* S = seq_state(e, sequence);
  std::lock_guard<std::mutex> seq_lock(S->mu);
  Context* c = S->ctx;
  if (!c || S->res_B < 1) return fail(e, DANI_ERR_INVALID, "sequence %d holds no resident frames (call dani_frontend_batch first)", sequence);
  HIP_TRY(e, hipSetDevice(e->device));
  const float* pred = nullptr;
  int cap = 0;
  const float* last_slab = nullptr;
  {
    std::lock_guard<std::mutex> lk(C->mu);
    auto it = C->bufs.find(sequence);
    if (it == C->bufs.end() || !it->second.valid) return fail(e, DANI_ERR_INVALID, "sequence %d has no exchanged slabs (call dani_frontend_exchange_boundary with gathered_dev = NULL first)", sequence);
    CommState::Buf& b = it->second;
    cap = b.cap;
    const size_t nfl = slab_floats(cap);
    if (C->rank > 0) pred = b.gathered + (size_t)(C->rank - 1) * nfl;                 // the neighbouring rank's last frame
    else if (C->carry_valid && C->carry_cap == cap) {                                 // rank 0: the last rank's final frame of the previous batch
      pred = C->carry;
      HIP_TRY(e, hipStreamWaitEvent(c->stream, C->carry_ev, 0));                      // ... written on the stream of the sequence that matched before
    }
    last_slab = b.gathered + (size_t)(C->world - 1) * nfl;
    b.valid = false;
  }
  if (!pred) c->prev_n = 0;            // the very first batch: frame 0 of rank 0 has no predecessor
  int rc = match_resident_locked(e, S, sequence, pred, 1, cap, lg_threshold, matches, scores, match_cap, k_out);
  if (rc) return rc;
  if (C->rank == 0) {                  // keep the last rank's slab: the next exchange of this sequence overwrites the gather buffer
    std::lock_guard<std::mutex> lk(C->mu);
    const size_t nfl = slab_floats(cap);
    if (C->carry_cap != cap) {
      if (C->carry) { HIP_TRY(e, hipStreamSynchronize(c->stream)); (void)hipFree(C->carry); C->carry = nullptr; }
      HIP_TRY(e, hipMalloc((void**)&C->carry, nfl * 4));
      C->carry_cap = cap;
    }
    HIP_TRY(e, hipMemcpyAsync(C->carry, last_slab, nfl * 4, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(e, hipEventRecord(C->carry_ev, c->stream));
    C->carry_valid = true;
  }
  return DANI_OK;
}

int dani_comm_reset_carry(dani_engine* e) {
  if (!e) return DANI_ERR_INVALID;
  CommState* C = e->comm;
  if (!C) return fail(e, DANI_ERR_INVALID, "no communicator (dani_comm_create)");
  std::lock_guard<std::mutex> lk(C->mu);
  C->carry_valid = false;
  return DANI_OK;
}

}  // extern "C"
