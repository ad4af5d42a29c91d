// host_logic.cpp -- the order-sensitive host stages of the extractor, implemented for speed with flat
// arrays (the reference uses std::list<ExtractorNode> holding std::vector<cv::KeyPoint> copies):
//   cell grid of ComputeKeyPointsOctTree      /root/reference/src/SPextractor.cpp:232-266
//   DistributeOctTree / DivideNode            src/SPextractor.cpp:21-77, 388-629
//   mnFeaturesPerLevel / scale tables         src/SPextractor.cpp:97-126
// The quadtree output ORDER is part of the contract (it is the keypoint order seen by Frame), so the
// list discipline of the reference (push_front of children n1..n4, erase of the parent, largest-first
// expansion after std::sort by (size, UL.x)) is reproduced exactly with an index-linked list.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <utility>
#include <vector>

#include "common.h"

namespace dani {

int cell_grid(int cols, int rows, std::vector<CellRect>& cells) {
  cells.clear();
  const float W = 35;
  const int EDGE = 19;
  const int minBorderX = EDGE - 3, minBorderY = minBorderX;
  const int maxBorderX = cols - EDGE + 3, maxBorderY = rows - EDGE + 3;
  const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
  const int nCols = (int)(width / W), nRows = (int)(height / W);
  if (nCols <= 0 || nRows <= 0) return 0;
  const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
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
      CellRect c;
      c.iniX = (int)iniX; c.maxX = (int)maxX; c.iniY = (int)iniY; c.maxY = (int)maxY;
      c.offX = j * wCell; c.offY = i * hCell;
      cells.push_back(c);
    }
  }
  return (int)cells.size();
}

void features_per_level(int nfeatures, float scaleFactorF, int nlevels, std::vector<int>& per_level,
                        std::vector<float>& scales) {
  const double scaleFactor = (double)scaleFactorF;
  scales.assign(nlevels, 1.0f);
  for (int i = 1; i < nlevels; i++) scales[i] = (float)(scales[i - 1] * scaleFactor);
  per_level.assign(nlevels, 0);
  float factor = (float)(1.0f / scaleFactor);
  float nDesired = (float)(nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels)));
  int sum = 0;
  for (int level = 0; level < nlevels - 1; level++) {
    per_level[level] = (int)std::lrintf(nDesired);
    sum += per_level[level];
    nDesired *= factor;
  }
  per_level[nlevels - 1] = std::max(nfeatures - sum, 0);
}

namespace {

struct QNode {
  int ulx, uly, brx, bry;  // UL and BR corners (UR.x == BR.x, BL.y == BR.y for every node)
  int kbeg, kcnt;          // key index range in the pool
  int prev, next;          // list links
  bool no_more;
};

struct QTree {
  const HostKey* keys;
  std::vector<QNode> nodes;
  // key (index, x, y) records; children are stable partitions of their parent.  The coordinates travel with the index
  // so that a split streams through contiguous memory, and the quadrant is computed without branches: on real
  // candidate sets the two data-dependent branches of the obvious version mispredict half the time and were ~60 % of
  // the quadtree's 200 us per frame.
  struct PKey { int k; float x, y; };
  std::vector<PKey> pool;
  std::vector<unsigned char> quad;   // scratch: quadrant of every key of the node being split
  int head = -1, tail = -1, size = 0;

  int new_node() { nodes.emplace_back(); return (int)nodes.size() - 1; }
  void push_front(int n) {
    nodes[n].prev = -1; nodes[n].next = head;
    if (head >= 0) nodes[head].prev = n; else tail = n;
    head = n; size++;
  }
  void push_back(int n) {
    nodes[n].next = -1; nodes[n].prev = tail;
    if (tail >= 0) nodes[tail].next = n; else head = n;
    tail = n; size++;
  }
  int erase(int n) {  // returns the successor
    const int p = nodes[n].prev, q = nodes[n].next;
    if (p >= 0) nodes[p].next = q; else head = q;
    if (q >= 0) nodes[q].prev = p; else tail = p;
    size--;
    return q;
  }
  // DivideNode: children ids in ch[0..3] (n1..n4), created but not linked
  void divide(int parent, int ch[4]) {
    const QNode P = nodes[parent];
    const int halfX = (int)std::ceil(static_cast<float>(P.brx - P.ulx) / 2);
    const int halfY = (int)std::ceil(static_cast<float>(P.bry - P.uly) / 2);
    const int mx = P.ulx + halfX, my = P.uly + halfY;
    for (int i = 0; i < 4; i++) ch[i] = new_node();
    QNode& n1 = nodes[ch[0]]; QNode& n2 = nodes[ch[1]]; QNode& n3 = nodes[ch[2]]; QNode& n4 = nodes[ch[3]];
    n1.ulx = P.ulx; n1.uly = P.uly; n1.brx = mx;    n1.bry = my;
    n2.ulx = mx;    n2.uly = P.uly; n2.brx = P.brx; n2.bry = my;
    n3.ulx = P.ulx; n3.uly = my;    n3.brx = mx;    n3.bry = P.bry;
    n4.ulx = mx;    n4.uly = my;    n4.brx = P.brx; n4.bry = P.bry;
    int cnt[4] = {0, 0, 0, 0};
    const int beg = P.kbeg, end = P.kbeg + P.kcnt;
    const float fmx = (float)mx, fmy = (float)my;
    if ((int)quad.size() < P.kcnt) quad.resize(P.kcnt);
    for (int i = beg; i < end; i++) {     // n1: x < mx, y < my; n2: x >= mx, y < my; n3: x < mx, y >= my; n4: the rest
      const unsigned char q = (unsigned char)((pool[i].x >= fmx ? 1 : 0) | (pool[i].y >= fmy ? 2 : 0));
      quad[i - beg] = q;
      cnt[q]++;
    }
    int start[4];
    const int base = (int)pool.size();
    start[0] = base; start[1] = start[0] + cnt[0]; start[2] = start[1] + cnt[1]; start[3] = start[2] + cnt[2];
    pool.resize(base + P.kcnt);
    int w[4] = {start[0], start[1], start[2], start[3]};
    for (int i = beg; i < end; i++) pool[w[quad[i - beg]]++] = pool[i];
    for (int i = 0; i < 4; i++) {
      QNode& c = nodes[ch[i]];
      c.kbeg = start[i]; c.kcnt = cnt[i]; c.no_more = (cnt[i] == 1); c.prev = c.next = -1;
    }
  }
};

}  // namespace

// Returns false (nothing written) where the reference itself has undefined behaviour: nIni = round(width / height) = 0
// for portrait regions (width / height < 0.5: hX = inf and vpIniNodes is empty, src/SPextractor.cpp:410-428), or a key
// whose x lies outside [0, maxX - minX) (vpIniNodes[kp.pt.x / hX] out of range, :433).
bool distribute_octtree(const HostKey* keys, int n, int minX, int maxX, int minY, int maxY, int N,
                        std::vector<int>& out_idx) {
  out_idx.clear();
  if (n <= 0) return true;
  if (maxX <= minX || maxY <= minY) return false;
  // per-thread scratch kept between calls: ~0.8 MB of node and key-record storage; allocating it per call (mmap +
  // first-touch page faults) cost as much as the algorithm
  thread_local QTree t;
  t.keys = keys;
  t.nodes.clear(); t.pool.clear();
  t.head = t.tail = -1; t.size = 0;
  t.nodes.reserve(8 * (size_t)std::max(N, 64) + 64);
  t.pool.reserve((size_t)n * 12);
  const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
  if (nIni < 1) return false;
  const float hX = static_cast<float>(maxX - minX) / nIni;
  // initial nodes + assignment (stable per node)
  thread_local std::vector<int> which;
  which.resize(n);
  std::vector<int> cnt(std::max(nIni, 1), 0);
  for (int i = 0; i < n; i++) {
    const float q = keys[i].x / hX;
    if (!(q >= 0.0f) || !(q < (float)nIni)) return false;
    which[i] = (int)q; cnt[which[i]]++;
  }
  std::vector<int> start(nIni + 1, 0);
  for (int i = 0; i < nIni; i++) start[i + 1] = start[i] + cnt[i];
  t.pool.resize(n);
  { std::vector<int> w(start.begin(), start.end() - 1); for (int i = 0; i < n; i++) t.pool[w[which[i]]++] = QTree::PKey{i, keys[i].x, keys[i].y}; }
  for (int i = 0; i < nIni; i++) {
    const int id = t.new_node();
    QNode& nd = t.nodes[id];
    nd.ulx = (int)(hX * static_cast<float>(i)); nd.uly = 0;
    nd.brx = (int)(hX * static_cast<float>(i + 1)); nd.bry = maxY - minY;
    nd.kbeg = start[i]; nd.kcnt = cnt[i]; nd.no_more = false;
    t.push_back(id);
  }
  for (int it = t.head; it >= 0;) {
    QNode& nd = t.nodes[it];
    if (nd.kcnt == 1) { nd.no_more = true; it = nd.next; }
    else if (nd.kcnt == 0) it = t.erase(it);
    else it = nd.next;
  }

  std::vector<std::pair<int, int>> expandable;  // (size, node id) in creation order
  auto link_children = [&](const int ch[4], int* nToExpand) {
    for (int i = 0; i < 4; i++) {
      const int c = ch[i];
      if (t.nodes[c].kcnt > 0) {
        t.push_front(c);
        if (t.nodes[c].kcnt > 1) {
          if (nToExpand) (*nToExpand)++;
          expandable.push_back(std::make_pair(t.nodes[c].kcnt, c));
        }
      }
    }
  };
  auto cmp = [&](const std::pair<int, int>& a, const std::pair<int, int>& b) {
    if (a.first < b.first) return true;
    if (a.first > b.first) return false;
    return t.nodes[a.second].ulx < t.nodes[b.second].ulx;
  };

  bool finish = false;
  while (!finish) {
    int prevSize = t.size;
    int nToExpand = 0;
    expandable.clear();
    for (int it = t.head; it >= 0;) {
      if (t.nodes[it].no_more) { it = t.nodes[it].next; continue; }
      int ch[4];
      t.divide(it, ch);
      link_children(ch, &nToExpand);
      it = t.erase(it);
    }
    if (t.size >= N || t.size == prevSize) {
      finish = true;
    } else if (t.size + nToExpand * 3 > N) {
      while (!finish) {
        prevSize = t.size;
        std::vector<std::pair<int, int>> prev = expandable;
        expandable.clear();
        std::sort(prev.begin(), prev.end(), cmp);
        for (int j = (int)prev.size() - 1; j >= 0; j--) {
          int ch[4];
          t.divide(prev[j].second, ch);
          link_children(ch, nullptr);
          t.erase(prev[j].second);
          if (t.size >= N) break;
        }
        if (t.size >= N || t.size == prevSize) finish = true;
      }
    }
  }
  out_idx.reserve(t.size);
  for (int it = t.head; it >= 0; it = t.nodes[it].next) {
    const QNode& nd = t.nodes[it];
    int best = t.pool[nd.kbeg].k;
    float maxResponse = keys[best].r;
    for (int k = 1; k < nd.kcnt; k++) {
      const int idx = t.pool[nd.kbeg + k].k;
      if (keys[idx].r > maxResponse) { best = idx; maxResponse = keys[idx].r; }
    }
    out_idx.push_back(best);
  }
  return true;
}

}  // namespace dani

// =================================================================================================
// YOLO host stages: order-defining greedy NMS and integer box bookkeeping (reference keeps them on the host too).
// =================================================================================================
#include <cstring>
#include <map>
#include <string>

namespace dani {

// cand8 records: {bitcast pred index, cx, cy, w, h, obj, max class prob, bitcast class}.  Restates
// non_max_suppression (/root/reference/src/YoloDetect.cpp:84-146) after its score filter: xywh -> xyxy in place,
// score = obj * max class, descending sort, repeat {keep first; keep only boxes with IoU <= thresh}.
// A NaN IoU (degenerate boxes) fails `<=` and is dropped, as `ious <= iou_thresh` does in the reference.
int yolo_nms_host(const float* cand8, int n, float iou_thresh, float* dets6, int cap) {
  struct Det { int idx; float x1, y1, x2, y2, score, cls, area; };
  std::vector<Det> d(n);
  for (int i = 0; i < n; i++) {
    const float* c = cand8 + (size_t)i * 8;
    int idx, cls;
    memcpy(&idx, &c[0], 4); memcpy(&cls, &c[7], 4);
    Det& o = d[i];
    o.idx = idx;
    o.x1 = c[1] - c[3] / 2.0f; o.y1 = c[2] - c[4] / 2.0f;
    o.x2 = o.x1 + c[3]; o.y2 = o.y1 + c[4];
    o.score = c[5] * c[6]; o.cls = (float)cls;
    o.area = (o.y2 - o.y1) * (o.x2 - o.x1);
  }
  std::sort(d.begin(), d.end(), [](const Det& a, const Det& b) { return a.idx < b.idx; });   // prediction order
  std::vector<int> order(n);
  for (int i = 0; i < n; i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return d[a].score > d[b].score; });
  int k = 0;
  std::vector<int> rest;
  while (!order.empty()) {
    const Det& f = d[order[0]];
    if (k < cap) { float* o = dets6 + (size_t)k * 6; o[0] = f.x1; o[1] = f.y1; o[2] = f.x2; o[3] = f.y2; o[4] = f.score; o[5] = f.cls; }
    k++;
    rest.clear();
    for (size_t i = 1; i < order.size(); i++) {
      const Det& g = d[order[i]];
      const float left = std::max(f.x1, g.x1), top = std::max(f.y1, g.y1);
      const float right = std::min(f.x2, g.x2), bottom = std::min(f.y2, g.y2);
      const float w = std::max(0.0f, right - left), h = std::max(0.0f, bottom - top);
      const float overlap = w * h;
      const float iou = overlap / (f.area + g.area - overlap);
      if (iou <= iou_thresh) rest.push_back(order[i]);
    }
    order.swap(rest);
  }
  return k;
}

namespace {
struct IRect { int x, y, w, h; int area() const { return w * h; } };
IRect rect_and(const IRect& a, const IRect& b) {   // cv::Rect operator&
  const int x = std::max(a.x, b.x), y = std::max(a.y, b.y);
  const int w = std::min(a.x + a.w, b.x + b.w) - x, h = std::min(a.y + a.h, b.y + b.h) - y;
  if (w <= 0 || h <= 0) return IRect{0, 0, 0, 0};
  return IRect{x, y, w, h};
}
}  // namespace

int yolo_dynamic_areas(const float* dets6, int n, int cols, int rows, const char* const* class_names, int n_classes,
                       const char* const* dynamic_names, int n_dynamic, int32_t* det_rects, int32_t* dyn_boxes, int cap) {
  if (n <= 0) return 0;   // `if (dets.size() > 0)`: nothing is pushed, not even the placeholder
  auto is_dynamic = [&](const std::string& nm) {
    for (int i = 0; i < n_dynamic; i++) if (nm == dynamic_names[i]) return true;
    return false;
  };
  std::map<std::string, std::vector<IRect>> detect_map;   // std::map: static classes are applied in name order
  std::vector<IRect> dynamic;
  for (int i = 0; i < n; i++) {
    const float* d = dets6 + (size_t)i * 6;
    const float left = d[0] * (float)cols / 640, top = d[1] * (float)rows / 384;     // quirk Q10: rows / 384
    const float right = d[2] * (float)cols / 640, bottom = d[3] * (float)rows / 384;
    const int cid = (int)d[5];
    const IRect r{(int)left, (int)top, (int)(right - left), (int)(bottom - top)};     // float -> int truncation
    if (det_rects) { det_rects[4 * i] = r.x; det_rects[4 * i + 1] = r.y; det_rects[4 * i + 2] = r.w; det_rects[4 * i + 3] = r.h; }
    const std::string nm = (cid >= 0 && cid < n_classes) ? class_names[cid] : "";
    detect_map[nm].push_back(r);
    if (is_dynamic(nm)) dynamic.push_back(r);
  }
  if (dynamic.empty()) dynamic.push_back(IRect{1, 1, 1, 1});
  int k = 0;
  for (const IRect& D : dynamic) {
    std::vector<IRect> sub{D}, nw;
    for (const auto& kv : detect_map) {
      if (is_dynamic(kv.first)) continue;
      for (const IRect& S : kv.second) {
        nw.clear();
        for (const IRect& region : sub) {
          const IRect in = rect_and(region, S);
          if (in.area() > 0) {
            if (in.y > region.y) nw.push_back(IRect{region.x, region.y, region.w, in.y - region.y});
            if (in.y + in.h < region.y + region.h) nw.push_back(IRect{region.x, in.y + in.h, region.w, region.y + region.h - (in.y + in.h)});
            if (in.x > region.x) nw.push_back(IRect{region.x, region.y, in.x - region.x, region.h});
            if (in.x + in.w < region.x + region.w) nw.push_back(IRect{in.x + in.w, region.y, region.x + region.w - (in.x + in.w), region.h});
          } else nw.push_back(region);
        }
        sub = nw;
      }
    }
    if (!sub.empty()) {
      IRect best = sub[0];
      for (size_t i = 1; i < sub.size(); i++) if (best.area() < sub[i].area()) best = sub[i];
      if (k < cap) { dyn_boxes[4 * k] = best.x; dyn_boxes[4 * k + 1] = best.y; dyn_boxes[4 * k + 2] = best.w; dyn_boxes[4 * k + 3] = best.h; }
      k++;
    }
  }
  return k;
}

}  // namespace dani

// ================================================================================================
// feature hand-off formats (host, order-defining; literal restatements)
// ================================================================================================
namespace dani {

// Frame::AssignFeaturesToGrid + Frame::PosInGrid (/root/reference/src/Frame.cc:387-418,727-737), case Nleft == -1:
// posX = round((x - mnMinX) * mfGridElementWidthInv) with C round() (half away from zero); keypoints outside the grid
// are dropped; mGrid[col][row] lists hold keypoint indices in ascending order.  CSR: cell = col * rows + row.
void assign_features_to_grid(const float* kxy, int n, float min_x, float min_y, float inv_w, float inv_h, int cols, int rows,
                             int32_t* cell_offsets, int32_t* cell_idx) {
  std::vector<int> cell(n, -1), count((size_t)cols * rows + 1, 0);
  for (int i = 0; i < n; i++) {
    const int px = (int)std::round((kxy[2 * i] - min_x) * inv_w), py = (int)std::round((kxy[2 * i + 1] - min_y) * inv_h);
    if (px < 0 || px >= cols || py < 0 || py >= rows) continue;
    cell[i] = px * rows + py;
    count[cell[i] + 1]++;
  }
  for (int c = 0; c < cols * rows; c++) count[c + 1] += count[c];
  for (int c = 0; c <= cols * rows; c++) cell_offsets[c] = count[c];
  std::vector<int> fill(count.begin(), count.end() - 1);
  for (int i = 0; i < n; i++)
    if (cell[i] >= 0) cell_idx[fill[cell[i]]++] = i;
}

// Frame::GetFeaturesInArea (src/Frame.cc:659-725) for one query window; appends to `out` in the reference's order
// (cells ix outer / iy inner, cell lists in index order); returns false when the window misses the grid.
void features_in_area(const float* kxy, const int32_t* octave, float min_x, float min_y, float inv_w, float inv_h, int cols, int rows,
                      const int32_t* cell_offsets, const int32_t* cell_idx, float x, float y, float r, int min_level, int max_level,
                      std::vector<int32_t>& out) {
  const int nMinCellX = std::max(0, (int)std::floor((x - min_x - r) * inv_w));
  if (nMinCellX >= cols) return;
  const int nMaxCellX = std::min(cols - 1, (int)std::ceil((x - min_x + r) * inv_w));
  if (nMaxCellX < 0) return;
  const int nMinCellY = std::max(0, (int)std::floor((y - min_y - r) * inv_h));
  if (nMinCellY >= rows) return;
  const int nMaxCellY = std::min(rows - 1, (int)std::ceil((y - min_y + r) * inv_h));
  if (nMaxCellY < 0) return;
  const bool check_levels = (min_level > 0) || (max_level >= 0);
  for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
    for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
      const int c = ix * rows + iy;
      for (int j = cell_offsets[c]; j < cell_offsets[c + 1]; j++) {
        const int k = cell_idx[j];
        if (check_levels) {
          const int o = octave ? octave[k] : 0;
          if (o < min_level) continue;
          if (max_level >= 0 && o > max_level) continue;
        }
        const float distx = kxy[2 * k] - x, disty = kxy[2 * k + 1] - y;
        if (std::fabs(distx) < r && std::fabs(disty) < r) out.push_back(k);
      }
    }
}

// Frame::UndistortKeyPoints (src/Frame.cc:749-782): identity when the first distortion coefficient is exactly 0,
// else cv::undistortPoints(pts, pts, K, distCoef, noArray(), K).  OpenCV is not part of the reference tree (find_package
// OpenCV 4.2); this restates the published algorithm of its calib3d module for that call shape (R = identity, P = K,
// default criteria = 5 fixed-point iterations): everything in double, inputs and outputs float.
//   x = (u - cx) / fx, y = (v - cy) / fy;  five times:  r2 = x^2 + y^2,
//   icdist = (1 + ((k6 r2 + k5) r2 + k4) r2) / (1 + ((k3 r2 + k2) r2 + k1) r2)   (k4..k6 = 0 here),
//   dx = 2 p1 x y + p2 (r2 + 2 x^2),  dy = p1 (r2 + 2 y^2) + 2 p2 x y,  (x, y) = ((x0 - dx) icdist, (y0 - dy) icdist)
//   (a negative icdist abandons the point at its normalised input);  out = (fx x + cx, fy y + cy).
void undistort_keypoints(const float* xy, int n, const float K[4], const float* dist, int ndist, float* out) {
  if (ndist < 1 || dist[0] == 0.0f) {
    if (out != xy) memcpy(out, xy, (size_t)n * 8);
    return;
  }
  double k[5] = {0, 0, 0, 0, 0};   // k1 k2 p1 p2 k3
  for (int i = 0; i < ndist && i < 5; i++) k[i] = (double)dist[i];
  const double fx = K[0], fy = K[1], cx = K[2], cy = K[3], ifx = 1.0 / fx, ify = 1.0 / fy;
  for (int i = 0; i < n; i++) {
    double x = ((double)xy[2 * i] - cx) * ifx, y = ((double)xy[2 * i + 1] - cy) * ify;
    const double x0 = x, y0 = y;
    for (int it = 0; it < 5; it++) {
      const double r2 = x * x + y * y;
      const double icdist = 1.0 / (1.0 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
      if (icdist < 0) { x = x0; y = y0; break; }
      const double dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
      const double dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
      x = (x0 - dx) * icdist;
      y = (y0 - dy) * icdist;
    }
    const double xx = fx * x + cx, yy = fy * y + cy, ww = 1.0 / 1.0;   // P = K, third row (0 0 1)
    out[2 * i] = (float)(xx * ww);
    out[2 * i + 1] = (float)(yy * ww);
  }
}

// Frame::ComputeStereoMatches after the LightGlue call (src/Frame.cc:862-914): disparity gate [0, mbf / mb), depth =
// mbf / max-ish(disparity) (non-positive -> 0.01), then the matches whose (1 - score) reaches 1.5 x the median of the
// sorted (distance, left index) pairs are cleared again.
void stereo_from_matches(const float* ul_x, int nl, const float* ur_x, int nr, const int32_t* matches, const float* scores, int k,
                         float mb, float mbf, float* u_right, float* depth) {
  for (int i = 0; i < nl; i++) { u_right[i] = -1.0f; depth[i] = -1.0f; }
  const float minZ = mb, minD = 0, maxD = mbf / minZ;
  std::vector<std::pair<float, int>> disp;
  disp.reserve(k);
  for (int m = 0; m < k; m++) {
    const int iL = matches[2 * m], iR = matches[2 * m + 1];
    if (iL >= nl || iR >= nr || iL < 0 || iR < 0) continue;
    const float uL = ul_x[iL], uR = ur_x[iR];
    float disparity = uL - uR;
    if (disparity >= minD && disparity < maxD) {
      if (disparity <= 0) disparity = 0.01;
      depth[iL] = mbf / disparity;
      u_right[iL] = uR;
      disp.push_back(std::make_pair(1.0f - scores[m], iL));
    }
  }
  if (disp.empty()) return;
  std::sort(disp.begin(), disp.end());
  const float median = disp[disp.size() / 2].first, th = 1.5f * median;
  for (int i = (int)disp.size() - 1; i >= 0; i--) {
    if (disp[i].first < th) break;
    u_right[disp[i].second] = -1;
    depth[disp[i].second] = -1;
  }
}

}  // namespace dani
