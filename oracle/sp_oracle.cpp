// ============================================================================
// oracle/sp_oracle.cpp  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the reference's SuperPoint front-end path.  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
// library; the product (dani_slam_amd/) never links, imports or calls it.
//
// What it restates (all citations are into /root/reference):
//   SuperPoint::forward                 src/SuperPoint.cpp:56-99
//   SPDetector::detect                  src/SuperPoint.cpp:190-208
//   SPDetector::getKeyPoints            src/SuperPoint.cpp:211-243
//   NMS2 (incl. quirk Q4, see below)    src/SuperPoint.cpp:277-363
//   SPDetector::computeDescriptors      src/SuperPoint.cpp:246-274
//   SPextractor ctor tables             src/SPextractor.cpp:80-128
//   SPextractor::operator()             src/SPextractor.cpp:130-217
//   ComputeKeyPointsOctTree             src/SPextractor.cpp:219-386
//   ExtractorNode::DivideNode           src/SPextractor.cpp:21-77
//   DistributeOctTree / compareNodes    src/SPextractor.cpp:388-629
//
// PARITY PIN STATUS: the reference ships no golden vectors, no tests and no
// weights for this path, and its sources cannot be built
// here (OpenCV absent).  The network part of this file is pinned against the
// ATen CPU operators the reference calls (conv2d / max_pool2d / softmax / norm /
// grid_sampler, via tests/golden/make_golden_sp.py run with torch 2.10 CPU);
// the host logic (NMS2, cell loop, cull, quadtree) is a literal restatement of
// the cited source lines and is "parity unpinned" by any reference fixture.
//
// CANONICAL F32 EVALUATION ORDER (shared spec with the HIP kernels, stated in
// DESIGN.md section 3).  LibTorch-CPU's own accumulation order is opaque
// (MKL-DNN blocking), so the spec fixes one valid f32 order:
//   conv:    acc = bias[co]; for cp in 0..ceil(Cin/2): for ky: for kx:
//            for cc in 0..1: c = 2cp+cc (<Cin): acc = fmaf(in, w, acc)
//            with zero padding taken literally (fmaf(0, w, acc)).
//   softmax: m = max_c x; e_c = dani_expf(x_c - m); s = ((e_0+e_1)+e_2)+...;
//            p_c = e_c / s.
//   norms:   s = fmaf(d_i, d_i, s) sequentially, sqrtf, IEEE divide.
// The gfx950 instruction v_mfma_f32_32x32x2_f32 evaluates exactly this chain
// (measured: tools/probes/mfma_chain_probe.hip, 0/1024 mismatching bits), so
// the HIP path is bit-identical to this file, and this file agrees with ATen to
// f32 rounding (tests/test_oracle_golden.py).
// ============================================================================
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <utility>
#include <vector>

#define ORC_API extern "C" __attribute__((visibility("default")))

// ---------------------------------------------------------------------------
// dani_expf: exp(x) for x <= 0 using only exact-rounded f32 ops (fmaf, rintf,
// ldexpf), so that CPU and GPU produce identical bits.  x is clamped to >= -87
// so the result stays a normal float.  Accuracy ~1 ulp (degree-7 Taylor on
// |r| <= ln2/2).
// ---------------------------------------------------------------------------
static inline float dani_expf(float x) {
  x = fmaxf(x, -87.0f);
  const float n = rintf(x * 1.44269504088896341f);
  float r = __builtin_fmaf(n, -0.693145751953125f, x);        // ln2 high part (exact in 12 bits)
  r = __builtin_fmaf(n, -1.42860682030941723212e-6f, r);      // ln2 low part
  float p = 1.0f / 5040.0f;
  p = __builtin_fmaf(p, r, 1.0f / 720.0f);
  p = __builtin_fmaf(p, r, 1.0f / 120.0f);
  p = __builtin_fmaf(p, r, 1.0f / 24.0f);
  p = __builtin_fmaf(p, r, 1.0f / 6.0f);
  p = __builtin_fmaf(p, r, 0.5f);
  p = __builtin_fmaf(p, r, 1.0f);
  p = __builtin_fmaf(p, r, 1.0f);
  return ldexpf(p, (int)n);
}
ORC_API float orc_expf(float x) { return dani_expf(x); }

// ---------------------------------------------------------------------------
// conv2d, stride 1, "same" zero padding, ksize in {1,3}, canonical order.
// in [Cin][H][W], w [Cout][Cin][k][k] (OIHW as torch), out [Cout][H][W].
// SuperPoint.cpp:16-33 (layer definitions), :58-77 (use).
// ---------------------------------------------------------------------------
ORC_API int orc_conv2d(const float* in, int Cin, int H, int W, const float* w, const float* b,
                       int Cout, int ksize, int relu, float* out) {
  if (ksize != 1 && ksize != 3) return -1;
  const int pad = ksize / 2;
  const int Hp = H + 2 * pad, Wp = W + 2 * pad;
  // literal zero padding
  std::vector<float> xp((size_t)Cin * Hp * Wp, 0.0f);
  for (int c = 0; c < Cin; c++)
    for (int y = 0; y < H; y++)
      memcpy(&xp[((size_t)c * Hp + y + pad) * Wp + pad], &in[((size_t)c * H + y) * W], sizeof(float) * W);
  const int RB = 4;  // rows per block (keeps the accumulators in L1)
  const int ncp = (Cin + 1) / 2;
#pragma omp parallel for schedule(dynamic, 1)
  for (int co = 0; co < Cout; co++) {
    std::vector<float> acc((size_t)RB * W);
    for (int y0 = 0; y0 < H; y0 += RB) {
      const int rb = std::min(RB, H - y0);
      for (int i = 0; i < rb * W; i++) acc[i] = b ? b[co] : 0.0f;
      for (int cp = 0; cp < ncp; cp++)
        for (int ky = 0; ky < ksize; ky++)
          for (int kx = 0; kx < ksize; kx++)
            for (int cc = 0; cc < 2; cc++) {
              const int c = 2 * cp + cc;
              if (c >= Cin) continue;
              const float wv = w[(((size_t)co * Cin + c) * ksize + ky) * ksize + kx];
              for (int r = 0; r < rb; r++) {
                const float* src = &xp[((size_t)c * Hp + y0 + r + ky) * Wp + kx];
                float* a = &acc[(size_t)r * W];
#pragma omp simd
                for (int x = 0; x < W; x++) a[x] = __builtin_fmaf(src[x], wv, a[x]);
              }
            }
      for (int r = 0; r < rb; r++) {
        float* dst = &out[((size_t)co * H + y0 + r) * W];
        const float* a = &acc[(size_t)r * W];
        if (relu)
          for (int x = 0; x < W; x++) dst[x] = a[x] > 0.0f ? a[x] : 0.0f;  // torch::relu
        else
          for (int x = 0; x < W; x++) dst[x] = a[x];
      }
    }
  }
  return 0;
}

// max_pool2d(x, 2, 2), floor mode.  SuperPoint.cpp:60,64,68.
ORC_API void orc_maxpool2(const float* in, int C, int H, int W, float* out) {
  const int Ho = H / 2, Wo = W / 2;
  for (int c = 0; c < C; c++)
    for (int y = 0; y < Ho; y++)
      for (int x = 0; x < Wo; x++) {
        const float* p = &in[((size_t)c * H + 2 * y) * W + 2 * x];
        out[((size_t)c * Ho + y) * Wo + x] = fmaxf(fmaxf(p[0], p[1]), fmaxf(p[W], p[W + 1]));
      }
}

// softmax over 65 channels, drop the dustbin, depth-to-space 8x8.  SuperPoint.cpp:82-93.
// semi [65][Hc][Wc] -> prob [8Hc][8Wc]; prob[r][c] = softmax(semi)[(r%8)*8 + c%8][r/8][c/8].
ORC_API void orc_softmax_d2s(const float* semi, int Hc, int Wc, float* prob) {
  const int HW = Hc * Wc, W = Wc * 8;
  for (int cy = 0; cy < Hc; cy++)
    for (int cx = 0; cx < Wc; cx++) {
      const float* s = semi + (size_t)cy * Wc + cx;
      float m = s[0];
      for (int c = 1; c < 65; c++) m = fmaxf(m, s[(size_t)c * HW]);
      float e[65];
      float sum = 0.0f;
      for (int c = 0; c < 65; c++) {
        e[c] = dani_expf(s[(size_t)c * HW] - m);
        sum = sum + e[c];
      }
      for (int c = 0; c < 64; c++) prob[(size_t)(cy * 8 + c / 8) * W + cx * 8 + c % 8] = e[c] / sum;
    }
}

// desc /= ||desc||_2 over channels, no epsilon (Q8).  SuperPoint.cpp:79-80.
ORC_API void orc_desc_l2norm(float* desc, int C, int HW) {
  for (int p = 0; p < HW; p++) {
    float s = 0.0f;
    for (int c = 0; c < C; c++) s = __builtin_fmaf(desc[(size_t)c * HW + p], desc[(size_t)c * HW + p], s);
    const float dn = sqrtf(s);
    for (int c = 0; c < C; c++) desc[(size_t)c * HW + p] = desc[(size_t)c * HW + p] / dn;
  }
}

// Names and order of the 24 weight tensors, as register_module() names them
// (SuperPoint.cpp:37-52): <layer>.weight, <layer>.bias.
static const char* const kLayerNames[12] = {"conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b",
                                            "conv4a", "conv4b", "convPa", "convPb", "convDa", "convDb"};
static const int kCin[12] = {1, 64, 64, 64, 64, 128, 128, 128, 128, 256, 128, 256};
static const int kCout[12] = {64, 64, 64, 64, 128, 128, 128, 128, 256, 65, 256, 256};
static const int kK[12] = {3, 3, 3, 3, 3, 3, 3, 3, 3, 1, 3, 1};

ORC_API const char* orc_layer_name(int i) { return (i >= 0 && i < 12) ? kLayerNames[i] : nullptr; }
ORC_API int orc_layer_dims(int i, int* cin, int* cout, int* k) {
  if (i < 0 || i >= 12) return -1;
  *cin = kCin[i]; *cout = kCout[i]; *k = kK[i];
  return 0;
}

// SPDetector::detect + SuperPoint::forward.  img u8 [H][stride]; weights = 24 pointers
// (w0,b0,w1,b1,...).  prob [8*(H/8)][8*(W/8)], desc [256][H/8][W/8] (already channel-normalised).
// `taps` (optional, may be null): 12 pointers receiving each layer's post-activation output
// (pre-pool), used by the per-layer parity tests.
ORC_API int orc_superpoint_forward(const uint8_t* img, int H, int W, int stride, const float* const* weights,
                                   float* prob, float* desc, float* const* taps) {
  if (H < 8 || W < 8) return -1;
  std::vector<float> x((size_t)H * W);
  for (int y = 0; y < H; y++)
    for (int c = 0; c < W; c++) x[(size_t)y * W + c] = (float)img[(size_t)y * stride + c] / 255.0f;  // :192-193
  std::vector<float> a, bbuf;
  int h = H, w = W;
  std::vector<float> cur = x;
  int curC = 1;
  auto conv = [&](int li, const std::vector<float>& in, int relu, std::vector<float>& out) {
    out.assign((size_t)kCout[li] * h * w, 0.0f);
    orc_conv2d(in.data(), kCin[li], h, w, weights[2 * li], weights[2 * li + 1], kCout[li], kK[li], relu, out.data());
    if (taps && taps[li]) memcpy(taps[li], out.data(), out.size() * sizeof(float));
  };
  auto pool = [&](std::vector<float>& t, int C) {
    std::vector<float> o((size_t)C * (h / 2) * (w / 2));
    orc_maxpool2(t.data(), C, h, w, o.data());
    h /= 2; w /= 2;
    t.swap(o);
  };
  std::vector<float> t0, t1;
  conv(0, cur, 1, t0); conv(1, t0, 1, t1); pool(t1, 64);
  conv(2, t1, 1, t0);  conv(3, t0, 1, t1); pool(t1, 64);
  conv(4, t1, 1, t0);  conv(5, t0, 1, t1); pool(t1, 128);
  conv(6, t1, 1, t0);  conv(7, t0, 1, t1);                      // t1 = x after conv4b
  (void)curC;
  std::vector<float> cPa, semi, cDa, d;
  conv(8, t1, 1, cPa);  conv(9, cPa, 0, semi);
  conv(10, t1, 1, cDa); conv(11, cDa, 0, d);
  orc_desc_l2norm(d.data(), 256, h * w);
  memcpy(desc, d.data(), d.size() * sizeof(float));
  orc_softmax_d2s(semi.data(), h, w, prob);
  return 0;
}

// ---------------------------------------------------------------------------
// Host-side extraction logic
// ---------------------------------------------------------------------------
struct OKey {  // the cv::KeyPoint fields the path uses
  float x, y, size, angle, response;
  int octave;
};

// NMS2 with the reference's literal indexing (quirk Q4): `confidence` is the UNPADDED w*h
// buffer but is indexed with PADDED coordinates through the unchecked cv::Mat::at, i.e. the
// flat element (vv+k)*w + (uu+j).  Reads past the end of the buffer are undefined behaviour in
// the reference; this restatement DEFINES them as 0.0f (documented deviation).
// SuperPoint.cpp:277-363.
static void nms2(const std::vector<OKey>& det, const std::vector<float>& conf, std::vector<OKey>& pts, int border,
                 int dist_thresh, int img_width, int img_height) {
  const int w = img_width, h = img_height, d = dist_thresh;
  const int gw = w + 2 * d, gh = h + 2 * d;
  std::vector<int> px(det.size()), py(det.size());
  for (size_t i = 0; i < det.size(); i++) { px[i] = (int)det[i].x; py[i] = (int)det[i].y; }   // :283-289
  std::vector<int8_t> grid((size_t)gw * gh, 0);        // already "copyMakeBorder"-ed (:311)
  std::vector<uint16_t> inds((size_t)w * h, 0);
  std::vector<float> confidence((size_t)w * h, 0.0f);
  for (size_t i = 0; i < det.size(); i++) {            // :300-309
    grid[(size_t)(py[i] + d) * gw + px[i] + d] = 1;
    inds[(size_t)py[i] * w + px[i]] = (uint16_t)i;
    confidence[(size_t)py[i] * w + px[i]] = conf[i];
  }
  auto conf_at = [&](int row, int col) -> float {      // unchecked Mat::at<float>(row, col) on a w-wide buffer
    const long idx = (long)row * w + col;
    return (idx >= 0 && idx < (long)w * h) ? confidence[idx] : 0.0f;
  };
  for (size_t i = 0; i < det.size(); i++) {            // :313-330
    const int uu = px[i] + d, vv = py[i] + d;
    if (grid[(size_t)vv * gw + uu] != 1) continue;
    for (int k = -d; k < d + 1; k++)
      for (int j = -d; j < d + 1; j++) {
        if (j == 0 && k == 0) continue;
        if (conf_at(vv + k, uu + j) < conf_at(vv, uu)) grid[(size_t)(vv + k) * gw + uu + j] = 0;
      }
    grid[(size_t)vv * gw + uu] = 2;
  }
  for (int v = 0; v < h + d; v++)                      // :335-352
    for (int u = 0; u < w + d; u++) {
      if (u - d >= w - border || u - d < border || v - d >= h - border || v - d < border) continue;
      if (grid[(size_t)v * gw + u] == 2) {
        const int sel = inds[(size_t)(v - d) * w + (u - d)];
        OKey k; k.x = (float)px[sel]; k.y = (float)py[sel]; k.size = 8.0f; k.angle = -1.0f;
        k.response = conf[sel]; k.octave = 0;
        pts.push_back(k);
      }
    }
}

// SPDetector::getKeyPoints (SuperPoint.cpp:211-243): threshold the cell, nonzero in row-major
// order, NMS2 with border 0, dist_thresh 4.
static void get_keypoints(const float* prob, int PH, int PW, float th, int iniX, int maxX, int iniY, int maxY,
                          std::vector<OKey>& out) {
  // torch slice semantics: clamp to the tensor
  iniX = std::max(0, std::min(iniX, PW)); maxX = std::max(iniX, std::min(maxX, PW));
  iniY = std::max(0, std::min(iniY, PH)); maxY = std::max(iniY, std::min(maxY, PH));
  std::vector<OKey> raw;
  std::vector<float> conf;
  for (int y = iniY; y < maxY; y++)
    for (int x = iniX; x < maxX; x++) {
      const float p = prob[(size_t)y * PW + x];
      if (p > th) {
        OKey k; k.x = (float)(x - iniX); k.y = (float)(y - iniY); k.size = 8.0f; k.angle = -1.0f;
        k.response = p; k.octave = 0;
        raw.push_back(k);
        conf.push_back(p);
      }
    }
  nms2(raw, conf, out, 0, 4, maxX - iniX, maxY - iniY);
}

ORC_API int orc_cell_keypoints(const float* prob, int PH, int PW, float th, int iniX, int maxX, int iniY, int maxY,
                               float* out_xyr, int cap) {
  std::vector<OKey> v;
  get_keypoints(prob, PH, PW, th, iniX, maxX, iniY, maxY, v);
  const int n = (int)v.size();
  for (int i = 0; i < n && i < cap; i++) { out_xyr[3 * i] = v[i].x; out_xyr[3 * i + 1] = v[i].y; out_xyr[3 * i + 2] = v[i].response; }
  return n;
}

// cvRound / saturate_cast<int>(float): round half to even (OpenCV uses lrint / cvtss2si).
static inline int cv_round(float v) { return (int)lrintf(v); }
static inline int cv_round_d(double v) { return (int)lrint(v); }

// cv::Rect2i::contains(cv::Point2f): the float point is converted to Point2i with
// saturate_cast<int> first (Q6), then x <= px < x+w && y <= py < y+h.
static inline bool rect_contains(const int32_t* r, float fx, float fy) {
  const int x = cv_round(fx), y = cv_round(fy);
  return r[0] <= x && x < r[0] + r[2] && r[1] <= y && y < r[1] + r[3];
}

static const int EDGE_THRESHOLD = 19;  // SPextractor.cpp:17
static const int PATCH_SIZE = 31;      // SPextractor.cpp:15

// The 204-cell loop of ComputeKeyPointsOctTree incl. the in-loop dynamic-box cull replayed
// literally (Q5, Q6).  SPextractor.cpp:232-354.  `cols`/`rows` = pyramid-level image size.
static void collect_candidates(const float* prob, int PH, int PW, int cols, int rows, float iniTh, float minTh,
                               const int32_t* boxes, int nboxes, float scale, std::vector<OKey>& vToDistributeKeys) {
  const float W = 35;
  const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
  const int maxBorderX = cols - EDGE_THRESHOLD + 3, maxBorderY = rows - EDGE_THRESHOLD + 3;
  const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
  const int nCols = (int)(width / W), nRows = (int)(height / W);
  const int wCell = (int)ceilf(width / nCols), hCell = (int)ceilf(height / nRows);
  for (int i = 0; i < nRows; i++) {
    const float iniY = (float)(minBorderY + i * hCell);
    float maxY = iniY + hCell + 6;
    if (iniY >= maxBorderY - 3) continue;
    if (maxY > maxBorderY) maxY = (float)maxBorderY;
    for (int j = 0; j < nCols; j++) {
      const float iniX = (float)(minBorderX + j * wCell);
      float maxX = iniX + wCell + 6;
      if (iniX >= maxBorderX - 6) continue;
      if (maxX > maxBorderX) maxX = (float)maxBorderX;
      std::vector<OKey> cell;
      get_keypoints(prob, PH, PW, iniTh, (int)iniX, (int)maxX, (int)iniY, (int)maxY, cell);
      if (cell.empty()) get_keypoints(prob, PH, PW, minTh, (int)iniX, (int)maxX, (int)iniY, (int)maxY, cell);
      for (auto& k : cell) {
        k.x += j * wCell; k.y += i * hCell;
        vToDistributeKeys.push_back(k);
      }
      // cull over ALL accumulated points, inside the loop (:314-352)
      for (auto& k : vToDistributeKeys) {
        k.x += minBorderX; k.y += minBorderY;
        k.x *= scale; k.y *= scale;
      }
      {
        std::vector<OKey> kept;
        kept.reserve(vToDistributeKeys.size());
        for (const auto& k : vToDistributeKeys) {
          bool find = false;
          for (int b = 0; b < nboxes; b++)
            if (rect_contains(boxes + 4 * b, k.x, k.y)) { find = true; break; }
          if (!find) kept.push_back(k);
        }
        vToDistributeKeys.swap(kept);
      }
      const float scale_inverse = 1 / scale;
      for (auto& k : vToDistributeKeys) {
        k.x *= scale_inverse; k.y *= scale_inverse;
        k.x -= minBorderX; k.y -= minBorderY;
      }
    }
  }
}

ORC_API int orc_collect_candidates(const float* prob, int PH, int PW, int cols, int rows, float iniTh, float minTh,
                                   const int32_t* boxes, int nboxes, float scale, float* out_xyr, int cap) {
  std::vector<OKey> v;
  collect_candidates(prob, PH, PW, cols, rows, iniTh, minTh, boxes, nboxes, scale, v);
  const int n = (int)v.size();
  for (int i = 0; i < n && i < cap; i++) { out_xyr[3 * i] = v[i].x; out_xyr[3 * i + 1] = v[i].y; out_xyr[3 * i + 2] = v[i].response; }
  return n;
}

// ---- quadtree: literal std::list restatement of SPextractor.cpp:21-77, 388-629 ----
struct ONode {
  std::vector<OKey> vKeys;
  int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
  std::list<ONode>::iterator lit;
  bool bNoMore = false;
};

static void divide_node(const ONode& p, ONode& n1, ONode& n2, ONode& n3, ONode& n4) {
  const int halfX = (int)ceilf(static_cast<float>(p.URx - p.ULx) / 2);
  const int halfY = (int)ceilf(static_cast<float>(p.BRy - p.ULy) / 2);
  n1.ULx = p.ULx; n1.ULy = p.ULy; n1.URx = p.ULx + halfX; n1.URy = p.ULy;
  n1.BLx = p.ULx; n1.BLy = p.ULy + halfY; n1.BRx = p.ULx + halfX; n1.BRy = p.ULy + halfY;
  n2.ULx = n1.URx; n2.ULy = n1.URy; n2.URx = p.URx; n2.URy = p.URy;
  n2.BLx = n1.BRx; n2.BLy = n1.BRy; n2.BRx = p.URx; n2.BRy = p.ULy + halfY;
  n3.ULx = n1.BLx; n3.ULy = n1.BLy; n3.URx = n1.BRx; n3.URy = n1.BRy;
  n3.BLx = p.BLx; n3.BLy = p.BLy; n3.BRx = n1.BRx; n3.BRy = p.BLy;
  n4.ULx = n3.URx; n4.ULy = n3.URy; n4.URx = n2.BRx; n4.URy = n2.BRy;
  n4.BLx = n3.BRx; n4.BLy = n3.BRy; n4.BRx = p.BRx; n4.BRy = p.BRy;
  for (const OKey& kp : p.vKeys) {
    if (kp.x < n1.URx) {
      if (kp.y < n1.BRy) n1.vKeys.push_back(kp); else n3.vKeys.push_back(kp);
    } else if (kp.y < n1.BRy) n2.vKeys.push_back(kp);
    else n4.vKeys.push_back(kp);
  }
  if (n1.vKeys.size() == 1) n1.bNoMore = true;
  if (n2.vKeys.size() == 1) n2.bNoMore = true;
  if (n3.vKeys.size() == 1) n3.bNoMore = true;
  if (n4.vKeys.size() == 1) n4.bNoMore = true;
}

static bool compare_nodes(std::pair<int, ONode*>& e1, std::pair<int, ONode*>& e2) {  // :388-403
  if (e1.first < e2.first) return true;
  if (e1.first > e2.first) return false;
  return e1.second->ULx < e2.second->ULx;
}

static std::vector<OKey> distribute_octtree(const std::vector<OKey>& keys, int minX, int maxX, int minY, int maxY, int N) {
  const int nIni = (int)roundf(static_cast<float>(maxX - minX) / (maxY - minY));
  const float hX = static_cast<float>(maxX - minX) / nIni;
  std::list<ONode> lNodes;
  std::vector<ONode*> vpIniNodes(nIni);
  for (int i = 0; i < nIni; i++) {
    ONode ni;
    ni.ULx = (int)(hX * static_cast<float>(i)); ni.ULy = 0;
    ni.URx = (int)(hX * static_cast<float>(i + 1)); ni.URy = 0;
    ni.BLx = ni.ULx; ni.BLy = maxY - minY;
    ni.BRx = ni.URx; ni.BRy = maxY - minY;
    lNodes.push_back(ni);
    vpIniNodes[i] = &lNodes.back();
  }
  for (const OKey& kp : keys) vpIniNodes[(int)(kp.x / hX)]->vKeys.push_back(kp);
  auto lit = lNodes.begin();
  while (lit != lNodes.end()) {
    if (lit->vKeys.size() == 1) { lit->bNoMore = true; lit++; }
    else if (lit->vKeys.empty()) lit = lNodes.erase(lit);
    else lit++;
  }
  bool bFinish = false;
  std::vector<std::pair<int, ONode*>> vSizeAndPointerToNode;
  auto add_child = [&](ONode& n, int* nToExpand) {
    if (n.vKeys.size() > 0) {
      lNodes.push_front(n);
      if (n.vKeys.size() > 1) {
        if (nToExpand) (*nToExpand)++;
        vSizeAndPointerToNode.push_back(std::make_pair((int)n.vKeys.size(), &lNodes.front()));
        lNodes.front().lit = lNodes.begin();
      }
    }
  };
  while (!bFinish) {
    int prevSize = (int)lNodes.size();
    lit = lNodes.begin();
    int nToExpand = 0;
    vSizeAndPointerToNode.clear();
    while (lit != lNodes.end()) {
      if (lit->bNoMore) { lit++; continue; }
      ONode n1, n2, n3, n4;
      divide_node(*lit, n1, n2, n3, n4);
      add_child(n1, &nToExpand); add_child(n2, &nToExpand); add_child(n3, &nToExpand); add_child(n4, &nToExpand);
      lit = lNodes.erase(lit);
    }
    if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) {
      bFinish = true;
    } else if (((int)lNodes.size() + nToExpand * 3) > N) {
      while (!bFinish) {
        prevSize = (int)lNodes.size();
        std::vector<std::pair<int, ONode*>> vPrev = vSizeAndPointerToNode;
        vSizeAndPointerToNode.clear();
        std::sort(vPrev.begin(), vPrev.end(), compare_nodes);
        for (int j = (int)vPrev.size() - 1; j >= 0; j--) {
          ONode n1, n2, n3, n4;
          divide_node(*vPrev[j].second, n1, n2, n3, n4);
          add_child(n1, nullptr); add_child(n2, nullptr); add_child(n3, nullptr); add_child(n4, nullptr);
          lNodes.erase(vPrev[j].second->lit);
          if ((int)lNodes.size() >= N) break;
        }
        if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
      }
    }
  }
  std::vector<OKey> res;
  for (auto& nd : lNodes) {
    const OKey* best = &nd.vKeys[0];
    float maxResponse = best->response;
    for (size_t k = 1; k < nd.vKeys.size(); k++)
      if (nd.vKeys[k].response > maxResponse) { best = &nd.vKeys[k]; maxResponse = nd.vKeys[k].response; }
    res.push_back(*best);
  }
  return res;
}

// in_xyr: n x (x, y, response); out_xyr: selected keypoints in list order; returns count.
ORC_API int orc_distribute_octtree(const float* in_xyr, int n, int minX, int maxX, int minY, int maxY, int N,
                                   float* out_xyr, int cap) {
  std::vector<OKey> keys(n);
  for (int i = 0; i < n; i++) { keys[i] = OKey{in_xyr[3 * i], in_xyr[3 * i + 1], 8.0f, -1.0f, in_xyr[3 * i + 2], 0}; }
  if (n == 0) return 0;
  std::vector<OKey> r = distribute_octtree(keys, minX, maxX, minY, maxY, N);
  const int m = (int)r.size();
  for (int i = 0; i < m && i < cap; i++) { out_xyr[3 * i] = r[i].x; out_xyr[3 * i + 1] = r[i].y; out_xyr[3 * i + 2] = r[i].response; }
  return m;
}

// SPDetector::computeDescriptors (SuperPoint.cpp:246-274): grid_sampler(bilinear, zeros,
// align_corners=false) at the keypoints, then EACH CHANNEL ROW divided by its L2 norm over the
// n keypoints (Q7), transposed to [n][256].
// desc_map [C][Hc][Wc]; kxy n x (x,y) in prob-map pixel coordinates; PW/PH = prob dims.
ORC_API void orc_compute_descriptors(const float* desc_map, int C, int Hc, int Wc, int PH, int PW, const float* kxy,
                                     int n, float* out) {
  std::vector<float> samp((size_t)C * n);
  for (int i = 0; i < n; i++) {
    const float gx = 2.0f * kxy[2 * i] / (float)PW - 1.0f;       // :257
    const float gy = 2.0f * kxy[2 * i + 1] / (float)PH - 1.0f;   // :258
    const float ix = ((gx + 1.0f) * (float)Wc - 1.0f) / 2.0f;    // grid_sampler_unnormalize, align_corners=false
    const float iy = ((gy + 1.0f) * (float)Hc - 1.0f) / 2.0f;
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float wx = ix - x0f, wy = iy - y0f;                    // ATen CPU kernel: w = x - x_w, e = 1 - w
    const float ex = 1.0f - wx, sy = 1.0f - wy;
    const float nw = sy * ex, ne = sy * wx, sw = wy * ex, se = wy * wx;
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    auto at = [&](int c, int y, int x) -> float {
      return (x >= 0 && x < Wc && y >= 0 && y < Hc) ? desc_map[((size_t)c * Hc + y) * Wc + x] : 0.0f;
    };
    for (int c = 0; c < C; c++) {
      float v = at(c, y0, x0) * nw;
      v = __builtin_fmaf(at(c, y0, x1), ne, v);
      v = __builtin_fmaf(at(c, y1, x0), sw, v);
      v = __builtin_fmaf(at(c, y1, x1), se, v);
      samp[(size_t)c * n + i] = v;
    }
  }
  for (int c = 0; c < C; c++) {                                  // :261-264, norm over dim 1 of [256][n]
    float s = 0.0f;
    for (int i = 0; i < n; i++) s = __builtin_fmaf(samp[(size_t)c * n + i], samp[(size_t)c * n + i], s);
    const float dn = sqrtf(s);
    for (int i = 0; i < n; i++) out[(size_t)i * C + c] = samp[(size_t)c * n + i] / dn;
  }
}

// SPextractor constructor tables (SPextractor.cpp:97-126).
static void features_per_level(int nfeatures, float scaleFactorF, int nlevels, std::vector<int>& out,
                               std::vector<float>& scales) {
  const double scaleFactor = (double)scaleFactorF;  // member is double, initialised from float
  scales.assign(nlevels, 1.0f);
  for (int i = 1; i < nlevels; i++) scales[i] = (float)(scales[i - 1] * scaleFactor);
  out.assign(nlevels, 0);
  float factor = (float)(1.0f / scaleFactor);
  float nDesired = (float)(nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels)));
  int sum = 0;
  for (int level = 0; level < nlevels - 1; level++) {
    out[level] = cv_round(nDesired);
    sum += out[level];
    nDesired *= factor;
  }
  out[nlevels - 1] = std::max(nfeatures - sum, 0);
}

ORC_API int orc_features_per_level(int nfeatures, float scaleFactor, int nlevels, int* out, float* scales) {
  std::vector<int> f; std::vector<float> s;
  features_per_level(nfeatures, scaleFactor, nlevels, f, s);
  for (int i = 0; i < nlevels; i++) { out[i] = f[i]; if (scales) scales[i] = s[i]; }
  return 0;
}

static void resize_bilinear_u8(const uint8_t* src, int sw, int sh, int sstride, int cn, uint8_t* dst, int dw, int dh);

// Whole SPextractor::operator() (src/SPextractor.cpp:130-217) incl. ComputePyramid (:631-656, level k =
// cv::resize(level k-1, INTER_LINEAR) to cvRound(dim * invScale_k); the REFLECT_101 border only served ORB patches
// and is never read by the SuperPoint path) and the per-level loop of ComputeKeyPointsOctTree (:219-386).
// Outputs: kpts n x (x, y) in level-0 coordinates, resp n, octave n, size n, desc n x 256 (levels concatenated, each
// level normalised over its own keypoints, Q7); returns n (<= cap) or <0.  mono_index as operator() returns it
// (:164-216), incl. the "descriptors are not permuted" quirk.  prob_out / descmap_out: level 0 only.
ORC_API int orc_sp_extract(const uint8_t* img, int W, int H, int stride, const float* const* weights, int nfeatures,
                           float scaleFactor, int nlevels, float iniTh, float minTh, const int32_t* boxes, int nboxes,
                           int lap0, int lap1, float* kpts_xy, float* resp, int32_t* octave, float* size,
                           float* desc, int cap, int* mono_index, float* prob_out, float* descmap_out) {
  if (!img || W <= 0 || H <= 0) return -1;  // operator() returns -1 on an empty image (:135-136)
  if (nlevels < 1) return -2;
  std::vector<int> fpl; std::vector<float> scales;
  features_per_level(nfeatures, scaleFactor, nlevels, fpl, scales);
  const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
  std::vector<OKey> all;
  std::vector<uint8_t> cur((size_t)W * H), nxt;
  for (int y = 0; y < H; y++) memcpy(&cur[(size_t)y * W], img + (size_t)y * stride, W);
  int cw = W, ch = H, n_total = 0;
  for (int level = 0; level < nlevels; level++) {
    if (level > 0) {
      const float inv = 1.0f / scales[level];                       // mvInvScaleFactor[level]
      const int nw = cv_round((float)W * inv), nh = cv_round((float)H * inv);
      nxt.assign((size_t)nw * nh, 0);
      resize_bilinear_u8(cur.data(), cw, ch, cw, 1, nxt.data(), nw, nh);
      cur.swap(nxt); cw = nw; ch = nh;
    }
    if (cw < 8 || ch < 8) return -5;
    const int PH = (ch / 8) * 8, PW = (cw / 8) * 8, Hc = ch / 8, Wc = cw / 8;
    std::vector<float> prob((size_t)PH * PW), dmap((size_t)256 * Hc * Wc);
    if (orc_superpoint_forward(cur.data(), ch, cw, cw, weights, prob.data(), dmap.data(), nullptr)) return -3;
    if (level == 0 && prob_out) memcpy(prob_out, prob.data(), prob.size() * sizeof(float));
    if (level == 0 && descmap_out) memcpy(descmap_out, dmap.data(), dmap.size() * sizeof(float));
    std::vector<OKey> cand;
    collect_candidates(prob.data(), PH, PW, cw, ch, iniTh, minTh, boxes, nboxes, scales[level], cand);
    const int maxBorderX = cw - EDGE_THRESHOLD + 3, maxBorderY = ch - EDGE_THRESHOLD + 3;
    std::vector<OKey> keys;
    if (!cand.empty()) keys = distribute_octtree(cand, minBorderX, maxBorderX, minBorderY, maxBorderY, fpl[level]);
    const int scaledPatchSize = (int)(PATCH_SIZE * scales[level]);
    for (auto& k : keys) { k.x += minBorderX; k.y += minBorderY; k.octave = level; k.size = (float)scaledPatchSize; }
    const int n = (int)keys.size();
    if (n_total + n > cap) return -4;
    if (n > 0) {
      std::vector<float> kxy((size_t)2 * n);
      for (int i = 0; i < n; i++) { kxy[2 * i] = keys[i].x; kxy[2 * i + 1] = keys[i].y; }
      orc_compute_descriptors(dmap.data(), 256, Hc, Wc, PH, PW, kxy.data(), n, desc + (size_t)n_total * 256);
    }
    if (level != 0)
      for (auto& k : keys) { k.x *= scales[level]; k.y *= scales[level]; }       // keypoint->pt *= scale (:190-192)
    all.insert(all.end(), keys.begin(), keys.end());
    n_total += n;
  }
  // mono / stereo-lapping partition of the KEYPOINTS only (:164-216)
  const int n = n_total;
  int monoIndex = 0, stereoIndex = n - 1;
  std::vector<OKey> outk(n);
  for (int i = 0; i < n; i++) {
    if (all[i].x >= lap0 && all[i].x <= lap1) outk[stereoIndex--] = all[i];
    else outk[monoIndex++] = all[i];
  }
  for (int i = 0; i < n; i++) {
    kpts_xy[2 * i] = outk[i].x; kpts_xy[2 * i + 1] = outk[i].y; resp[i] = outk[i].response;
    octave[i] = outk[i].octave; size[i] = outk[i].size;
  }
  if (mono_index) *mono_index = monoIndex;
  return n;
}

// ============================================================================
// General convolution for the YOLOv5s restatement (oracle/yolo_oracle.py): any k, stride, padding, canonical
// chain order (cp, ky, kx, cc) from acc = bias, activation 0 none / 1 ReLU / 2 SiLU.
// SiLU canonical form: x / (1 + dani_expf(-x)).
// Long contractions are a two-level sum: the input channels are cut into `kseg` equal contiguous
// ranges -- 3x3 stride 1: 4 ranges from 256 channels, 2 from 128; 3x3 stride 2: 2 from 128; 1x1: 2 from 512; otherwise
// 1 -- each range is one chain in
// the order above (the bias opens range 0, the others start from 0), and the range sums are added left to right.
// ============================================================================
static int orc_kseg(int cin, int k, int stride) {
  if (k == 3 && stride == 1) return cin >= 256 ? 4 : (cin >= 128 ? 2 : 1);
  if (k == 3) return cin >= 128 ? 2 : 1;
  if (k == 1) return cin >= 512 ? 2 : 1;
  return 1;
}
ORC_API int orc_conv2d_ex(const float* in, int Cin, int H, int W, const float* w, const float* b, int Cout, int k,
                          int stride, int pad, int act, float* out) {
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  const int Hp = H + 2 * pad, Wp = W + 2 * pad + stride * 8;
  std::vector<float> xp((size_t)Cin * Hp * Wp, 0.0f);
  for (int c = 0; c < Cin; c++)
    for (int y = 0; y < H; y++)
      memcpy(&xp[((size_t)c * Hp + y + pad) * Wp + pad], &in[((size_t)c * H + y) * W], sizeof(float) * W);
  const int ncp = (Cin + 1) / 2;
  const int kseg = (ncp % orc_kseg(Cin, k, stride) == 0) ? orc_kseg(Cin, k, stride) : 1, seg_cp = ncp / kseg;
#pragma omp parallel for schedule(dynamic, 1)
  for (int co = 0; co < Cout; co++) {
    std::vector<float> acc(Wo), tot(Wo);
    for (int y = 0; y < Ho; y++) {
      for (int sg = 0; sg < kseg; sg++) {
      for (int x = 0; x < Wo; x++) acc[x] = (sg == 0 && b) ? b[co] : 0.0f;
      for (int cp = sg * seg_cp; cp < (sg + 1) * seg_cp; cp++)
        for (int ky = 0; ky < k; ky++)
          for (int kx = 0; kx < k; kx++)
            for (int cc = 0; cc < 2; cc++) {
              const int c = 2 * cp + cc;
              if (c >= Cin) continue;
              const float wv = w[(((size_t)co * Cin + c) * k + ky) * k + kx];
              const float* src = &xp[((size_t)c * Hp + y * stride + ky) * Wp + kx];
              if (stride == 1) {
#pragma omp simd
                for (int x = 0; x < Wo; x++) acc[x] = __builtin_fmaf(src[x], wv, acc[x]);
              } else {
                for (int x = 0; x < Wo; x++) acc[x] = __builtin_fmaf(src[x * stride], wv, acc[x]);
              }
            }
      for (int x = 0; x < Wo; x++) tot[x] = sg ? tot[x] + acc[x] : acc[x];
      }
      float* dst = &out[((size_t)co * Ho + y) * Wo];
      for (int x = 0; x < Wo; x++) {
        float v = tot[x];
        if (act == 1) v = v > 0.0f ? v : 0.0f;
        else if (act == 2) v = v / (1.0f + dani_expf(-v));
        dst[x] = v;
      }
    }
  }
  return 0;
}

// nn.MaxPool2d(kernel 5, stride 1, padding 2) (implicit -inf padding), SPPF.
ORC_API void orc_maxpool5(const float* in, int C, int H, int W, float* out) {
  for (int c = 0; c < C; c++)
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) {
        float m = -INFINITY;
        for (int dy = -2; dy <= 2; dy++)
          for (int dx = -2; dx <= 2; dx++) {
            const int yy = y + dy, xx = x + dx;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) m = fmaxf(m, in[((size_t)c * H + yy) * W + xx]);
          }
        out[((size_t)c * H + y) * W + x] = m;
      }
}

// cv::resize(src, dst, Size(dw, dh)) with INTER_LINEAR on 8UC3, restated from OpenCV's fixed-point path
// (INTER_RESIZE_COEF_BITS = 11: horizontal pass into int32 with coefficients scaled by 2048, vertical pass
//  dst = ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2).  Used by YoloDetection::Detect,
// /root/reference/src/YoloDetect.cpp:39.  (OpenCV itself is absent from this image; YOLO parity is box-level.)
static void resize_bilinear_u8(const uint8_t* src, int sw, int sh, int sstride, int cn, uint8_t* dst, int dw, int dh) {
  // cv::resize: inv_scale = (double)dsize / ssize; scale = 1. / inv_scale   (imgproc/resize.cpp)
  const double sx = 1.0 / ((double)dw / sw), sy = 1.0 / ((double)dh / sh);
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<short> xa(2 * dw), ya(2 * dh);
  auto coefs = [](int d, double scale, int ssize, int& ofs, short* ab) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= s;
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= ssize - 1) { s = ssize - 1; f = 0.f; }
    ofs = s;
    ab[0] = (short)lrintf((1.f - f) * 2048.f);
    ab[1] = (short)lrintf(f * 2048.f);
  };
  for (int x = 0; x < dw; x++) coefs(x, sx, sw, xofs[x], &xa[2 * x]);
  for (int y = 0; y < dh; y++) coefs(y, sy, sh, yofs[y], &ya[2 * y]);
  std::vector<int> r0((size_t)dw * cn), r1((size_t)dw * cn);
  auto hrow = [&](int srow, std::vector<int>& o) {
    srow = std::min(std::max(srow, 0), sh - 1);
    const uint8_t* S = src + (size_t)srow * sstride;
    for (int x = 0; x < dw; x++) {
      const int s0 = xofs[x], s1 = std::min(s0 + 1, sw - 1);
      for (int c = 0; c < cn; c++) o[cn * x + c] = S[cn * s0 + c] * xa[2 * x] + S[cn * s1 + c] * xa[2 * x + 1];
    }
  };
  for (int y = 0; y < dh; y++) {
    hrow(yofs[y], r0); hrow(yofs[y] + 1, r1);
    const int b0 = ya[2 * y], b1 = ya[2 * y + 1];
    for (int i = 0; i < dw * cn; i++) {
      const int v = (((b0 * (r0[i] >> 4)) >> 16) + ((b1 * (r1[i] >> 4)) >> 16) + 2) >> 2;
      dst[(size_t)y * dw * cn + i] = (uint8_t)std::min(std::max(v, 0), 255);
    }
  }
}
ORC_API void orc_resize_bilinear_u8c3(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh) {
  resize_bilinear_u8(src, sw, sh, sstride, 3, dst, dw, dh);
}
ORC_API void orc_resize_bilinear_u8c1(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh) {
  resize_bilinear_u8(src, sw, sh, sstride, 1, dst, dw, dh);
}

// canonical sigmoid 1 / (1 + dani_expf(-x)) (Detect layer), elementwise
ORC_API void orc_sigmoid(const float* in, float* out, long n) {
  for (long i = 0; i < n; i++) out[i] = 1.0f / (1.0f + dani_expf(-in[i]));
}
