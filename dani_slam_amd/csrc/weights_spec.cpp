// weights_spec.cpp -- the tensor names and shapes the engine expects, in one place, and a host-only check of a
// tensor set / .dsw file against them (no GPU needed: used by dani_finalize_weights before anything is packed, and by
// dani_check_weights_file, which tools/convert_weights.py and the CPU tests call on freshly converted archives).
//
// What the reference loads, and fails on (the shims map DANI_ERR_WEIGHTS to the same std::exit):
//   SuperPoint  torch::load(model, "/sly_slam/superpoint_new.pt")   /root/reference/src/SPextractor.cpp:86-91
//               parameter names = the register_module names of      src/SuperPoint.cpp:37-52
//   LightGlue   torch::jit::load("/sly_slam/lightglue_scripted_v2.pt")   src/LightGlue.cpp:9   (public checkpoint names)
//   YOLOv5s     torch::jit::load("yolov5s.torchscript.pt")               src/YoloDetect.cpp:9  (public fused names)
#include "engine_internal.h"

namespace dani {
namespace eng {

struct TensorSpec { std::string name; int d[4]; int nd; };

static size_t spec_count(const TensorSpec& t) {
  size_t n = 1;
  for (int i = 0; i < t.nd; i++) n *= (size_t)t.d[i];
  return n;
}
static std::string spec_shape(const TensorSpec& t) {
  std::string s = "[";
  for (int i = 0; i < t.nd; i++) s += (i ? "," : "") + std::to_string(t.d[i]);
  return s + "]";
}
static void add(std::vector<TensorSpec>& v, const std::string& name, int a, int b = 0, int c = 0, int d = 0) {
  TensorSpec t{name, {a, b, c, d}, 1 + (b > 0) + (c > 0) + (d > 0)};
  v.push_back(t);
}
static void add_linear(std::vector<TensorSpec>& v, const std::string& name, int out, int in) {
  add(v, name + ".weight", out, in);
  add(v, name + ".bias", out);
}
static void add_conv(std::vector<TensorSpec>& v, const std::string& name, int cout, int cin, int k) {
  add(v, name + ".weight", cout, cin, k, k);
  add(v, name + ".bias", cout);
}

// SuperPoint: the 12 Conv2d of struct SuperPoint (src/SuperPoint.cpp:16-33, registered :37-52), OIHW
void superpoint_spec(std::vector<TensorSpec>& v) {
  static const struct { const char* n; int cin, cout, k; } L[12] = {
      {"conv1a", 1, 64, 3},    {"conv1b", 64, 64, 3},   {"conv2a", 64, 64, 3},    {"conv2b", 64, 64, 3},
      {"conv3a", 64, 128, 3},  {"conv3b", 128, 128, 3}, {"conv4a", 128, 128, 3},  {"conv4b", 128, 128, 3},
      {"convPa", 128, 256, 3}, {"convPb", 256, 65, 1},  {"convDa", 128, 256, 3},  {"convDb", 256, 256, 1}};
  for (const auto& l : L) add_conv(v, l.n, l.cout, l.cin, l.k);
}

// LightGlue (SuperPoint variant of the public model: d = 256, 4 heads, 9 layers; only the last layer's assignment head
// is used -- the engine runs all 9 layers without early exit, lg_engine.cpp lg_finalize)
void lightglue_spec(std::vector<TensorSpec>& v) {
  add(v, "posenc.Wr.weight", 32, 2);
  for (int i = 0; i < 9; i++) {
    const std::string ps = "transformers." + std::to_string(i) + ".self_attn", pc = "transformers." + std::to_string(i) + ".cross_attn";
    add_linear(v, ps + ".Wqkv", 768, 256);
    add_linear(v, ps + ".out_proj", 256, 256);
    add_linear(v, ps + ".ffn.0", 512, 512);
    add(v, ps + ".ffn.1.weight", 512); add(v, ps + ".ffn.1.bias", 512);
    add_linear(v, ps + ".ffn.3", 256, 512);
    add_linear(v, pc + ".to_qk", 256, 256);
    add_linear(v, pc + ".to_v", 256, 256);
    add_linear(v, pc + ".to_out", 256, 256);
    add_linear(v, pc + ".ffn.0", 512, 512);
    add(v, pc + ".ffn.1.weight", 512); add(v, pc + ".ffn.1.bias", 512);
    add_linear(v, pc + ".ffn.3", 256, 512);
  }
  add_linear(v, "log_assignment.8.final_proj", 256, 256);
  add(v, "log_assignment.8.matchability.weight", 1, 256);
  add(v, "log_assignment.8.matchability.bias", 1);
}

// YOLOv5s v6 (width 0.5, depth 0.33), BatchNorm folded into the convolutions (tools/convert_weights.py does the folding)
void yolo_spec(std::vector<TensorSpec>& v) {
  auto conv = [&](const std::string& n, int cin, int cout, int k) { add_conv(v, n + ".conv", cout, cin, k); };
  auto c3 = [&](const std::string& n, int c1, int c2, int reps) {
    const int c_ = c2 / 2;
    conv(n + ".cv1", c1, c_, 1); conv(n + ".cv2", c1, c_, 1); conv(n + ".cv3", 2 * c_, c2, 1);
    for (int j = 0; j < reps; j++) { conv(n + ".m." + std::to_string(j) + ".cv1", c_, c_, 1); conv(n + ".m." + std::to_string(j) + ".cv2", c_, c_, 3); }
  };
  conv("model.0", 3, 32, 6); conv("model.1", 32, 64, 3); c3("model.2", 64, 64, 1);
  conv("model.3", 64, 128, 3); c3("model.4", 128, 128, 2);
  conv("model.5", 128, 256, 3); c3("model.6", 256, 256, 3);
  conv("model.7", 256, 512, 3); c3("model.8", 512, 512, 1);
  conv("model.9.cv1", 512, 256, 1); conv("model.9.cv2", 1024, 512, 1);
  conv("model.10", 512, 256, 1); c3("model.13", 512, 256, 1);
  conv("model.14", 256, 128, 1); c3("model.17", 256, 128, 1);
  conv("model.18", 128, 128, 3); c3("model.20", 256, 256, 1);
  conv("model.21", 256, 256, 3); c3("model.23", 512, 512, 1);
  const int dch[3] = {128, 256, 512};
  for (int i = 0; i < 3; i++) add_conv(v, "model.24.m." + std::to_string(i), 255, dch[i], 1);
}

// A tensor set (name -> number of floats) against the three specs.  SuperPoint is required; LightGlue and YOLO are
// groups that are either absent (their key tensor is missing: an SP-only / no-YOLO engine, as lg_finalize / yolo_finalize
// treat it) or complete.  msg names the first missing or mis-shaped tensor.  *ignored counts tensors no spec knows.
int check_tensor_counts(const std::map<std::string, size_t>& have, std::string& msg, int* groups, int* ignored) {
  std::vector<TensorSpec> sp, lgs, ys;
  superpoint_spec(sp); lightglue_spec(lgs); yolo_spec(ys);
  std::map<std::string, bool> known;
  int present = 0;
  auto check = [&](const std::vector<TensorSpec>& spec, const char* what) -> bool {
    for (const TensorSpec& t : spec) {
      known[t.name] = true;
      auto it = have.find(t.name);
      if (it == have.end()) { msg = std::string("missing tensor ") + t.name + " " + spec_shape(t) + " (" + what + ")"; return false; }
      if (it->second != spec_count(t)) {
        msg = "tensor " + t.name + " has " + std::to_string(it->second) + " floats, expected " + std::to_string(spec_count(t)) + " " + spec_shape(t) + " (" + what + ")";
        return false;
      }
    }
    return true;
  };
  if (!check(sp, "SuperPoint, src/SuperPoint.cpp:37-52")) return DANI_ERR_WEIGHTS;
  present |= 1;
  if (have.count("posenc.Wr.weight")) {
    if (!check(lgs, "LightGlue, public checkpoint names")) return DANI_ERR_WEIGHTS;
    present |= 2;
  } else for (const TensorSpec& t : lgs) known[t.name] = true;
  if (have.count("model.0.conv.weight")) {
    if (!check(ys, "YOLOv5s, fused public names")) return DANI_ERR_WEIGHTS;
    present |= 4;
  } else for (const TensorSpec& t : ys) known[t.name] = true;
  int ign = 0;
  std::string first;
  for (const auto& kv : have)
    if (!known.count(kv.first)) { if (!ign) first = kv.first; ign++; }
  if (groups) *groups = present;
  if (ignored) *ignored = ign;
  msg = ign ? std::to_string(ign) + " tensor(s) the engine does not use are ignored (first: " + first + ")" : std::string();
  return DANI_OK;
}

}  // namespace eng
}  // namespace dani

extern "C" {

// declared in include/dani_frontend.h
int dani_check_weights_file(const char* path, int32_t* groups, char* msg, int32_t msg_cap) {
  using namespace dani::eng;
  auto say = [&](const std::string& s) { if (msg && msg_cap > 0) { snprintf(msg, (size_t)msg_cap, "%s", s.c_str()); } };
  if (groups) *groups = 0;
  if (!path) { say("null path"); return DANI_ERR_INVALID; }
  FILE* f = fopen(path, "rb");
  if (!f) { say(std::string("cannot open weight file ") + path); return DANI_ERR_WEIGHTS; }
  std::map<std::string, size_t> have;
  char magic[4];
  uint32_t n = 0;
  bool ok = fread(magic, 1, 4, f) == 4 && memcmp(magic, "DSW1", 4) == 0 && fread(&n, 4, 1, f) == 1;
  for (uint32_t i = 0; ok && i < n; i++) {
    uint32_t nl = 0;
    uint64_t cnt = 0;
    if (fread(&nl, 4, 1, f) != 1 || nl == 0 || nl >= 256) { ok = false; break; }
    std::string name(nl, '\0');
    ok = fread(&name[0], 1, nl, f) == nl && fread(&cnt, 8, 1, f) == 1 && cnt > 0 && cnt < (1ull << 28) && fseek(f, (long)(cnt * 4), SEEK_CUR) == 0;
    if (ok) have[name] = (size_t)cnt;
  }
  if (ok) {   // the last payload must really be there (fseek past the end succeeds)
    const long pos = ftell(f);
    ok = fseek(f, 0, SEEK_END) == 0 && ftell(f) >= pos;
  }
  fclose(f);
  if (!ok) { say(std::string("malformed weight file ") + path); return DANI_ERR_WEIGHTS; }
  std::string m;
  int ignored = 0, g = 0;
  const int rc = check_tensor_counts(have, m, &g, &ignored);
  if (groups) *groups = g;
  say(m);
  return rc;
}

}  // extern "C"
