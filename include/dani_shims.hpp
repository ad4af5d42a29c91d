// dani_shims.hpp -- header-only C++17 shims that keep the reference's class names and method signatures on top
// of the C ABI in dani_frontend.h.  They replace include/SPextractor.h (+ SuperPoint.h) of the reference; see
// INTEGRATION.md.  This header needs OpenCV (cv::Mat, cv::KeyPoint) exactly like the headers it replaces.  OpenCV is not
// present in this repo's image: here the header is compiled and run end to end against a mock of the few core types it
// touches (tests/cpp/mock_opencv, tests/test_shims_gpu.py -- results identical to the ctypes path); against the real
// library it is compiled inside the reference's build tree.
#pragma once
#include <opencv2/core/core.hpp>

#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <string>
#include <mutex>
#include <vector>

#include "dani_frontend.h"

namespace dani_shim {

// Process-wide engine: weights are loaded once (the reference re-loads its TorchScript/archives per object
// construction, quirk Q12: src/LGmatcher.cpp:39, src/Frame.cc:827).  DANI_WEIGHTS names the .dsw produced by
// tools/convert_weights.py from /sly_slam/superpoint_new.pt (src/SPextractor.cpp:86).
inline dani_engine* engine() {
  static dani_engine* e = nullptr;
  static std::once_flag once;
  std::call_once(once, [] {
    dani_config cfg{0, 8, 1280, 1024, 4};
    const char* w = std::getenv("DANI_WEIGHTS");
    if (dani_engine_create(&cfg, &e) != DANI_OK || dani_load_weights_file(e, w ? w : "/sly_slam/dani_frontend.dsw") != DANI_OK ||
        dani_finalize_weights(e) != DANI_OK) {
      // same behaviour as the reference on a model-load failure: src/SPextractor.cpp:88-91
      std::cerr << "Error loading the model: " << dani_last_error(e) << std::endl;
      std::exit(EXIT_FAILURE);
    }
  });
  return e;
}

}  // namespace dani_shim

namespace ORB_SLAM3 {

// Replaces class SPextractor, include/SPextractor.h:24-96 (same public surface; no torch member).
class SPextractor {
 public:
  SPextractor(int nfeatures_, float scaleFactor_, int nlevels_, float iniThFAST_, float minThFAST_)
      : nfeatures(nfeatures_), scaleFactor(scaleFactor_), nlevels(nlevels_), iniThFAST(iniThFAST_), minThFAST(minThFAST_) {
    dani_shim::engine();
    std::cout << "[SPDetector::detect] Model loaded successfully." << std::endl;
    mvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels);
    mnFeaturesPerLevel.resize(nlevels);
    dani_host_features_per_level(nfeatures, scaleFactor_, nlevels, mnFeaturesPerLevel.data(), mvScaleFactor.data());
    mvInvScaleFactor.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
    for (int i = 0; i < nlevels; i++) {
      mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
      mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
      mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
    }
    mvImagePyramid.resize(nlevels);
  }
  ~SPextractor() {}

  // src/SPextractor.cpp:130-217
  int operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint>& _keypoints,
                 cv::OutputArray _descriptors, std::vector<int>& vLappingArea) {
    if (_image.empty()) return -1;
    cv::Mat image = _image.getMat();
    CV_Assert(image.type() == CV_8UC1);
    // DistributeOctTree may return up to 3 keypoints more than asked PER LEVEL (src/SPextractor.cpp:405-629); should a
    // frame still need more, the engine reports the count (DANI_ERR_CAPACITY) and the call is repeated with it
    int cap = nfeatures + 8 + 4 * nlevels;
    std::vector<float> kxy, resp, size, desc;
    std::vector<int32_t> octave, boxes(4 * mvDynamicArea.size());
    for (size_t i = 0; i < mvDynamicArea.size(); i++) {
      boxes[4 * i] = mvDynamicArea[i].x; boxes[4 * i + 1] = mvDynamicArea[i].y;
      boxes[4 * i + 2] = mvDynamicArea[i].width; boxes[4 * i + 3] = mvDynamicArea[i].height;
    }
    const dani_sp_params p{nfeatures, (float)scaleFactor, nlevels, iniThFAST, minThFAST};
    const int32_t lap[2] = {vLappingArea[0], vLappingArea[1]};
    int32_t n = 0, mono = 0;
    int rc = DANI_OK;
    for (int attempt = 0; attempt < 2; attempt++) {
      kxy.resize(2 * (size_t)cap); resp.resize(cap); size.resize(cap); desc.resize((size_t)cap * 256); octave.resize(cap);
      rc = dani_sp_extract(dani_shim::engine(), image.data, image.cols, image.rows, (int)image.step, &p,
                           boxes.data(), (int)mvDynamicArea.size(), lap, kxy.data(), resp.data(),
                           octave.data(), size.data(), desc.data(), cap, &n, &mono);
      if (rc != DANI_ERR_CAPACITY || n <= cap) break;
      cap = n;
    }
    mLastStatus = rc;
    if (rc != DANI_OK) {
      // A run-time engine error on one frame must not end the SLAM process (the reference exits only when the MODEL
      // cannot be loaded, src/SPextractor.cpp:88-91): report it and hand back what operator() returns for a frame it
      // cannot process (-1, no keypoints, :133-134); Frame's constructors return early on an empty keypoint set.
      // Every such frame is logged and counted (mnDroppedFrames / DroppedFrames()); since round 4 a dense frame is not
      // among them: the candidate buffers hold every pixel of every cell (csrc/sp_engine.cpp).
      ++mnDroppedFrames;
      std::cerr << "[SPextractor] frame dropped (" << mnDroppedFrames << " so far, status " << rc << "): "
                << dani_last_error(dani_shim::engine()) << std::endl;
      _keypoints.clear(); _descriptors.release();
      return -1;
    }
    if (n == 0) _descriptors.release();
    else {
      _descriptors.create(n, 256, CV_32F);
      cv::Mat(n, 256, CV_32F, desc.data()).copyTo(_descriptors.getMat());
    }
    _keypoints = std::vector<cv::KeyPoint>(n);
    for (int i = 0; i < n; i++)
      _keypoints[i] = cv::KeyPoint(kxy[2 * i], kxy[2 * i + 1], size[i], -1, resp[i], octave[i]);
    return mono;
  }

  int inline GetLevels() { return nlevels; }
  float inline GetScaleFactor() { return scaleFactor; }
  std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
  std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
  std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
  std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

  int mLastStatus = DANI_OK;                 // dani_status of the last operator() call (not in the reference)
  long mnDroppedFrames = 0;                  // frames this extractor handed back empty because the engine reported an error
  long DroppedFrames() const { return mnDroppedFrames; }
  std::vector<cv::Mat> mvImagePyramid;       // kept for source compatibility (not filled: the pyramid lives in HBM)
  std::vector<cv::Rect2i> mvDynamicArea;     // written by Tracking::GrabImageRGBD, src/Tracking.cc:1539

 protected:
  int nfeatures; double scaleFactor; int nlevels; float iniThFAST, minThFAST;
  std::vector<int> mnFeaturesPerLevel;
  std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
};

#ifndef ORB_SLAM3_ORBEXTRACTOR_DEFINED
#define ORB_SLAM3_ORBEXTRACTOR_DEFINED
typedef SPextractor ORBextractor;   // include/SPextractor.h:100
#endif

// Replaces class LightGlue, include/LightGlue.h:11-36.  Construction is O(1) (the reference jit::loads the blob in
// every constructor: 13 LGmatcher constructions per frame path + one per stereo frame, quirk Q12).  The three call
// sites pass pointer + count instead of torch::Tensor (src/Frame.cc:823-860, src/LGmatcher.cpp:299-314, :971-983):
//     lg.matchDescriptors(mvKeys, mDescriptors.ptr<float>(), mDescriptors.rows,
//                         mvKeysRight, mDescriptorsRight.ptr<float>(), mDescriptorsRight.rows,
//                         (float)imgWidth, (float)imgHeight, (float)imgWidth, (float)imgHeight);
class LightGlue {
 public:
  LightGlue() { dani_shim::engine(); }
  ~LightGlue() {}
  int mLastStatus = DANI_OK;   // dani_status of the last matchDescriptors() (not in the reference)
  // size w/h <= 0 selects the reference's size.numel()==0 branch (1 + max - min of the keypoints)
  std::vector<cv::DMatch> matchDescriptors(const std::vector<cv::KeyPoint>& keypoints1, const float* descriptors1, int n1,
                                           const std::vector<cv::KeyPoint>& keypoints2, const float* descriptors2, int n2,
                                           float w1, float h1, float w2, float h2, float threshold = 0.05f) {
    std::vector<cv::DMatch> out;
    if (n1 <= 0 || n2 <= 0) return out;
    std::vector<float> k1(2 * (size_t)n1), k2(2 * (size_t)n2);
    for (int i = 0; i < n1; i++) { k1[2 * i] = keypoints1[i].pt.x; k1[2 * i + 1] = keypoints1[i].pt.y; }   // keypointsToTensor
    for (int i = 0; i < n2; i++) { k2[2 * i] = keypoints2[i].pt.x; k2[2 * i + 1] = keypoints2[i].pt.y; }
    const float s1[2] = {w1, h1}, s2[2] = {w2, h2};
    const int cap = n1;
    std::vector<int32_t> m(2 * (size_t)cap);
    std::vector<float> sc(cap);
    int32_t k = 0;
    const int rc = dani_lg_match(dani_shim::engine(), k1.data(), descriptors1, n1, k2.data(), descriptors2, n2,
                                 w1 > 0 ? s1 : nullptr, w2 > 0 ? s2 : nullptr, threshold, m.data(), sc.data(), cap, &k);
    mLastStatus = rc;
    if (rc != DANI_OK) { std::cerr << "[LightGlue] " << dani_last_error(dani_shim::engine()) << std::endl; return out; }
    out.reserve(k);
    for (int i = 0; i < k; i++) out.emplace_back(m[2 * i], m[2 * i + 1], sc[i]);   // DMatch(queryIdx, trainIdx, distance = score)
    return out;
  }
};

}  // namespace ORB_SLAM3

// Replaces class YoloDetection, include/YoloDetect.h:16-41 (global namespace, as in the reference).
class YoloDetection {
 public:
  YoloDetection() {
    dani_shim::engine();
    std::ifstream f("coco.names");                 // src/YoloDetect.cpp:12-16
    std::string name;
    while (std::getline(f, name)) mClassnames.push_back(name);
    mvDynamicNames = {"person", "car", "motorbike", "bus", "train", "truck", "boat", "bird", "cat",
                      "dog", "horse", "sheep", "crow", "bear"};   // :17-18
  }
  ~YoloDetection() {}
  void GetImage(cv::Mat& RGB) { mRGB = RGB; }
  void ClearImage() { mRGB = 0; }
  void ClearArea() { mvPersonArea.clear(); }
  bool Detect() {                                  // src/YoloDetect.cpp:27-82
    if (mRGB.empty()) { std::cout << "Read RGB failed!" << std::endl; return false; }
    const int cap = 4096;
    std::vector<float> dets(6 * (size_t)cap);
    int32_t n = 0;
    mLastStatus = dani_yolo_detect(dani_shim::engine(), mRGB.data, mRGB.cols, mRGB.rows, (int)mRGB.step, 0.4f, 0.5f, dets.data(), cap, &n);
    if (mLastStatus != DANI_OK) {   // engine error: no boxes this frame, and the caller can see it (false + mLastStatus)
      std::cerr << "[YoloDetection] " << dani_last_error(dani_shim::engine()) << std::endl;
      return false;
    }
    if (n > 0) {
      std::vector<const char*> cn, dn;
      for (auto& s : mClassnames) cn.push_back(s.c_str());
      for (auto& s : mvDynamicNames) dn.push_back(s.c_str());
      std::vector<int32_t> rects(4 * (size_t)n), dyn(4 * (size_t)(n + 1));
      const int k = dani_host_dynamic_areas(dets.data(), n, mRGB.cols, mRGB.rows, cn.data(), (int)cn.size(), dn.data(),
                                            (int)dn.size(), rects.data(), dyn.data(), n + 1);
      for (int i = 0; i < n; i++)
        mmDetectMap[mClassnames[(int)dets[6 * i + 5]]].push_back(cv::Rect2i(rects[4 * i], rects[4 * i + 1], rects[4 * i + 2], rects[4 * i + 3]));
      mvDynamicArea.clear();
      for (int i = 0; i < k; i++) mvDynamicArea.push_back(cv::Rect2i(dyn[4 * i], dyn[4 * i + 1], dyn[4 * i + 2], dyn[4 * i + 3]));
    }
    return true;
  }
  int mLastStatus = DANI_OK;   // dani_status of the last Detect() (not in the reference)
  std::vector<cv::Rect2i> mvPersonArea = {};
  cv::Mat mRGB, mDepth;
  std::vector<std::string> mClassnames, mvDynamicNames;
  std::vector<cv::Rect2i> mvDynamicArea;
  std::map<std::string, std::vector<cv::Rect2i>> mmDetectMap;
};
