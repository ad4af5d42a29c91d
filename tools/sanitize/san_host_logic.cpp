// Host-logic edge cases under AddressSanitizer + UBSan (CPU build only; the GPU pool refuses sanitizer runs).
// Build and run:  tools/sanitize/run.sh
#include "common.h"
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
using namespace dani;
int main() {
  std::mt19937 rng0(7);
  std::vector<HostKey> keys(3277);
  for (auto& k : keys) { k.x = (float)(rng0() % 608); k.y = (float)(rng0() % 448); k.r = (rng0() % 100000) / 100000.0f; }
  std::vector<int> out;
  for (int N : {1, 5, 100, 1024, 3000, 5000}) for (int n : {0, 1, 2, 50, 3277}) { distribute_octtree(keys.data(), n, 16, 624, 16, 464, N, out); }
  printf("quadtree ok %zu\n", out.size());
  // portrait regions (round(width / height) = 0: the reference indexes an empty vector, src/SPextractor.cpp:410-433) and
  // keys outside the region must be refused, not walked
  bool refused = !distribute_octtree(keys.data(), 50, 16, 84, 16, 384, 100, out) && out.empty();      // 100x400 image
  refused = refused && !distribute_octtree(keys.data(), 50, 16, 144, 16, 464, 100, out);              // 160x480 image
  HostKey bad[3] = {{10.f, 10.f, 0.5f}, {608.f, 10.f, 0.5f}, {-1.f, 3.f, 0.2f}};
  refused = refused && !distribute_octtree(bad, 2, 16, 624, 16, 464, 100, out) && !distribute_octtree(bad + 2, 1, 16, 624, 16, 464, 100, out);
  refused = refused && !distribute_octtree(bad, 1, 16, 16, 16, 464, 100, out);                        // empty region
  if (!refused) { printf("portrait / out-of-range keys were not refused\n"); return 1; }
  printf("portrait ok\n");
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> ux(-10, 650), uy(-10, 490);
  std::vector<float> k(2 * 1025); for (int i = 0; i < 1025; i++) { k[2*i] = ux(rng); k[2*i+1] = uy(rng); }
  std::vector<int32_t> off(64 * 48 + 1), idx(1025), oct(1025, 0);
  assign_features_to_grid(k.data(), 1025, 0, 0, 0.1f, 0.1f, 64, 48, off.data(), idx.data());
  std::vector<int32_t> cand;
  for (int q = 0; q < 2000; q++) features_in_area(k.data(), oct.data(), 0, 0, 0.1f, 0.1f, 64, 48, off.data(), idx.data(), ux(rng), uy(rng), 20.0f, -1, 0, cand);
  printf("grid ok %zu\n", cand.size());
  std::vector<int32_t> m(2 * 300); std::vector<float> s(300), ul(500), ur(480), u(500), d(500);
  for (auto& v : ul) v = ux(rng); for (auto& v : ur) v = ux(rng);
  for (int i = 0; i < 300; i++) { m[2*i] = rng() % 500; m[2*i+1] = rng() % 480; s[i] = (rng() % 1000) / 1000.0f; }
  stereo_from_matches(ul.data(), 500, ur.data(), 480, m.data(), s.data(), 300, 0.1f, 40.0f, u.data(), d.data());
  {   // UndistortKeyPoints: TUM1-like coefficients, in place, and the degenerate inputs
    const float K4[4] = {517.3f, 516.5f, 318.6f, 255.3f}, dist[5] = {0.2624f, -0.9531f, -0.0054f, 0.0026f, 1.1633f}, zero[4] = {0, 0.3f, 0, 0};
    std::vector<float> kk(k.begin(), k.begin() + 2 * 1025), un(2 * 1025);
    undistort_keypoints(kk.data(), 1025, K4, dist, 5, un.data());
    undistort_keypoints(kk.data(), 1025, K4, dist, 4, kk.data());
    undistort_keypoints(kk.data(), 1025, K4, zero, 4, un.data());
    undistort_keypoints(nullptr, 0, K4, dist, 5, nullptr);
    printf("undistort ok %f\n", un[0]);
  }
  // YOLO host
  std::vector<float> c8(8 * 400), dets(6 * 400);
  for (int i = 0; i < 400; i++) { int id = i; memcpy(&c8[8*i], &id, 4); c8[8*i+1] = ux(rng); c8[8*i+2] = uy(rng); c8[8*i+3] = 50; c8[8*i+4] = 60; c8[8*i+5] = 0.9f; c8[8*i+6] = 0.8f; int cl = i % 80; memcpy(&c8[8*i+7], &cl, 4); }
  int nd = yolo_nms_host(c8.data(), 400, 0.5f, dets.data(), 400);
  printf("nms ok %d\n", nd);
  return 0;
}
