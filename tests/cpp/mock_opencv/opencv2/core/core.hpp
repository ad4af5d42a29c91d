// MOCK of the handful of OpenCV core types that include/dani_shims.hpp touches -- TEST SCAFFOLDING ONLY.
// OpenCV is absent from this image, so the shims cannot be compiled against the real headers here; this file exists so
// that tests/test_shims_gpu.py can at least compile the shim header as C++17 and drive it end to end through the C ABI
// from a C++ translation unit.  It mirrors the documented OpenCV semantics of the members used and nothing else; it is
// not used to build anything of the reference.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8UC1 0
#define CV_8UC3 16
#define CV_32F 5
#define CV_Assert(expr) do { if (!(expr)) std::abort(); } while (0)

namespace cv {

struct Point2f { float x = 0, y = 0; };

struct KeyPoint {
  Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1;
  KeyPoint() {}
  KeyPoint(float x, float y, float size_, float angle_ = -1, float response_ = 0, int octave_ = 0, int class_id_ = -1)
      : size(size_), angle(angle_), response(response_), octave(octave_), class_id(class_id_) { pt.x = x; pt.y = y; }
};

struct DMatch {
  int queryIdx = -1, trainIdx = -1, imgIdx = -1; float distance = 0;
  DMatch() {}
  DMatch(int q, int t, float d) : queryIdx(q), trainIdx(t), distance(d) {}
};

struct Rect2i {
  int x = 0, y = 0, width = 0, height = 0;
  Rect2i() {}
  Rect2i(int x_, int y_, int w_, int h_) : x(x_), y(y_), width(w_), height(h_) {}
};

class Mat {
 public:
  int rows = 0, cols = 0; size_t step = 0; unsigned char* data = nullptr;
  Mat() {}
  Mat(int r, int c, int type, void* ext) : rows(r), cols(c), step((size_t)c * esz(type)), data((unsigned char*)ext), type_(type) {}
  Mat(int r, int c, int type) { create(r, c, type); }
  Mat& operator=(int) { rows = cols = 0; step = 0; data = nullptr; own_.reset(); return *this; }   // `mRGB = 0`
  void create(int r, int c, int type) {
    rows = r; cols = c; type_ = type; step = (size_t)c * esz(type);
    own_ = std::shared_ptr<unsigned char>(new unsigned char[(size_t)r * step], std::default_delete<unsigned char[]>());
    data = own_.get();
  }
  void release() { *this = 0; }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  int type() const { return type_; }
  void copyTo(Mat dst) const { for (int r = 0; r < rows; r++) memcpy(dst.data + (size_t)r * dst.step, data + (size_t)r * step, (size_t)cols * esz(type_)); }
  template <class T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + (size_t)r * step); }
 private:
  static size_t esz(int type) { return type == CV_32F ? 4 : (type == CV_8UC3 ? 3 : 1); }
  int type_ = 0;
  std::shared_ptr<unsigned char> own_;
};

// InputArray / OutputArray as thin Mat references (the shims only call empty / getMat / create / release)
class _InputArray {
 public:
  _InputArray() {}
  _InputArray(const Mat& m) : m_(const_cast<Mat*>(&m)) {}
  bool empty() const { return !m_ || m_->empty(); }
  Mat getMat() const { return m_ ? *m_ : Mat(); }
 protected:
  Mat* m_ = nullptr;
};
class _OutputArray : public _InputArray {
 public:
  _OutputArray(Mat& m) { m_ = &m; }
  void create(int r, int c, int type) const { m_->create(r, c, type); }
  void release() const { m_->release(); }
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;

}  // namespace cv
