// Drives include/dani_shims.hpp (SPextractor, LightGlue, YoloDetection) from C++ through the C ABI, against the mock
// OpenCV core types of tests/cpp/mock_opencv (OpenCV is absent from this image).  Used by tests/test_shims_gpu.py.
//   shim_smoke <gray0.bin> <gray1.bin> <bgr0.bin> <out.bin>      (640x480 frames; DANI_WEIGHTS names the .dsw)
#include "dani_shims.hpp"

#include <cstdio>

static std::vector<unsigned char> slurp(const char* p, size_t n) {
  std::vector<unsigned char> v(n);
  FILE* f = fopen(p, "rb");
  if (!f || fread(v.data(), 1, n, f) != n) { fprintf(stderr, "cannot read %s\n", p); std::exit(2); }
  fclose(f);
  return v;
}

int main(int argc, char** argv) {
  if (argc != 5) return 2;
  const int W = 640, H = 480;
  auto g0 = slurp(argv[1], (size_t)W * H), g1 = slurp(argv[2], (size_t)W * H), c0 = slurp(argv[3], (size_t)W * H * 3);
  FILE* out = fopen(argv[4], "wb");
  auto put_i = [&](int32_t v) { fwrite(&v, 4, 1, out); };
  auto put_f = [&](const float* p, size_t n) { fwrite(p, 4, n, out); };

  // YoloDetection::Detect -> mvDynamicArea (src/Tracking.cc:1535-1539)
  YoloDetection det;
  cv::Mat rgb(H, W, CV_8UC3, c0.data());
  det.GetImage(rgb);
  const bool ok = det.Detect();
  put_i(ok ? 1 : 0); put_i((int)det.mvDynamicArea.size());
  for (auto& r : det.mvDynamicArea) { put_i(r.x); put_i(r.y); put_i(r.width); put_i(r.height); }

  // SPextractor::operator() on two frames; the first with the dynamic boxes
  ORB_SLAM3::ORBextractor ex(300, 1.2f, 1, 0.015f, 0.007f);
  std::vector<int> lap = {0, 0};
  std::vector<cv::KeyPoint> k0, k1;
  cv::Mat d0, d1, m0(H, W, CV_8UC1, g0.data()), m1(H, W, CV_8UC1, g1.data());
  ex.mvDynamicArea = det.mvDynamicArea;
  const int mono0 = ex(m0, cv::Mat(), k0, d0, lap);
  ex.mvDynamicArea.clear();
  const int mono1 = ex(m1, cv::Mat(), k1, d1, lap);
  for (auto* pr : {&k0, &k1}) {
    put_i((int)pr->size());
    for (auto& k : *pr) { put_f(&k.pt.x, 1); put_f(&k.pt.y, 1); put_f(&k.response, 1); put_f(&k.size, 1); put_i(k.octave); }
  }
  put_i(mono0); put_i(mono1);
  put_f(d0.ptr<float>(), (size_t)d0.rows * 256); put_f(d1.ptr<float>(), (size_t)d1.rows * 256);

  // LightGlue::matchDescriptors with explicit sizes
  ORB_SLAM3::LightGlue lg;
  auto matches = lg.matchDescriptors(k0, d0.ptr<float>(), d0.rows, k1, d1.ptr<float>(), d1.rows, (float)W, (float)H, (float)W, (float)H);
  put_i((int)matches.size());
  for (auto& m : matches) { put_i(m.queryIdx); put_i(m.trainIdx); put_f(&m.distance, 1); }
  // empty image -> -1 (src/SPextractor.cpp:133-134)
  std::vector<cv::KeyPoint> ke; cv::Mat de;
  put_i(ex(cv::Mat(), cv::Mat(), ke, de, lap));
  // an 8-level pyramid (Examples/Stereo-Inertial/RealSense_D435i.yaml: nFeatures 1250, nLevels 8): DistributeOctTree may
  // overshoot by up to 3 keypoints per level, which the shim's output capacity has to cover
  ORB_SLAM3::ORBextractor ex8(1250, 1.2f, 8, 0.015f, 0.007f);
  std::vector<cv::KeyPoint> k8; cv::Mat d8;
  const int mono8 = ex8(m0, cv::Mat(), k8, d8, lap);
  put_i(ex8.mLastStatus); put_i(mono8); put_i((int)k8.size()); put_i(d8.rows);
  for (auto& k : k8) { put_f(&k.pt.x, 1); put_f(&k.pt.y, 1); put_i(k.octave); }
  // a frame the engine refuses (64 x 48: below its 80 x 80 minimum): handed back empty like an empty image, logged and counted
  cv::Mat tiny(48, 64, CV_8UC1, g0.data());
  std::vector<cv::KeyPoint> kt; cv::Mat dt;
  put_i((int)ex.DroppedFrames());
  put_i(ex(tiny, cv::Mat(), kt, dt, lap)); put_i(ex.mLastStatus); put_i((int)ex.DroppedFrames()); put_i((int)kt.size());
  fclose(out);
  printf("shim_smoke ok: %zu + %zu keypoints, %zu matches, %zu dynamic boxes\n", k0.size(), k1.size(), matches.size(), det.mvDynamicArea.size());
  return 0;
}
