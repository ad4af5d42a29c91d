// handoff_kernels.hip -- SURVEY.md section 8(f) rank 2 on the device: Frame's keypoint grid and the windowed candidate
// search that feeds the projection-search distance scans, over features that never left HBM.
//   grid_assign_kernel    Frame::AssignFeaturesToGrid + PosInGrid      /root/reference/src/Frame.cc:387-418, 727-737
//   area_candidates_kernel Frame::GetFeaturesInArea                    src/Frame.cc:659-725
// Both are order-defining (cell lists in ascending keypoint index; candidates in cell order ix outer / iy inner, then list
// order) and integer / compare work: HBM / latency bound, no matrix cores.  The host restatements in host_logic.cpp define
// the expected results bit for bit (tests/test_resident_gpu.py).
#include "common.h"

namespace dani {

// One workgroup builds the whole grid of one frame (n keypoints, cols x rows cells; 64 x 48 = 3072 in the reference):
//   1. cell id of every keypoint: (int) roundf((x - min_x) * inv_w) -- C round(), half away from zero, as the reference
//      -- or -1 outside the grid;
//   2. every thread owns cells t, t + 1024, ...: it counts its cells' keypoints by scanning the id array, the block scans
//      the counts, and the same walk fills the lists -- ascending keypoint index by construction, no atomics, no sort.
// n * ncells / 1024 compares per thread (3 k for 1 k keypoints): ~10 us, once per retained frame.
constexpr int GRID_THREADS = 1024;
__global__ __launch_bounds__(GRID_THREADS) void grid_assign_kernel(const float* kxy, int n, float min_x, float min_y, float inv_w,
                                                                   float inv_h, int cols, int rows, int32_t* cell_of /*[n] scratch*/,
                                                                   int32_t* cell_offsets /*[ncells+1]*/, int32_t* cell_idx /*[n]*/) {
  __shared__ int32_t part[GRID_THREADS];
  const int tid = threadIdx.x, ncells = cols * rows;
  for (int i = tid; i < n; i += GRID_THREADS) {
    const int px = (int)roundf((kxy[2 * i] - min_x) * inv_w), py = (int)roundf((kxy[2 * i + 1] - min_y) * inv_h);
    cell_of[i] = (px < 0 || px >= cols || py < 0 || py >= rows) ? -1 : px * rows + py;
  }
  __syncthreads();
  // contiguous cell range per thread, so that the prefix over threads is the prefix over cells
  const int per = (ncells + GRID_THREADS - 1) / GRID_THREADS, c0 = min(tid * per, ncells), c1 = min(c0 + per, ncells);
  int mine = 0;
  for (int i = 0; i < n; i++) { const int c = cell_of[i]; mine += (c >= c0 && c < c1) ? 1 : 0; }
  part[tid] = mine;
  __syncthreads();
  for (int off = 1; off < GRID_THREADS; off <<= 1) {      // Hillis-Steele inclusive scan
    const int v = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int base = part[tid] - mine;
  for (int c = c0; c < c1; c++) {
    cell_offsets[c] = base;
    for (int i = 0; i < n; i++)
      if (cell_of[i] == c) cell_idx[base++] = i;
  }
  if (tid == GRID_THREADS - 1) cell_offsets[ncells] = part[tid];
}

void launch_grid_assign(const float* kxy, int n, float min_x, float min_y, float inv_w, float inv_h, int cols, int rows,
                        int32_t* cell_of, int32_t* cell_offsets, int32_t* cell_idx, hipStream_t s) {
  hipLaunchKernelGGL(grid_assign_kernel, dim3(1), dim3(GRID_THREADS), 0, s, kxy, n, min_x, min_y, inv_w, inv_h, cols, rows, cell_of,
                     cell_offsets, cell_idx);
}

// Frame::UndistortKeyPoints (src/Frame.cc:749-782) = cv::undistortPoints(pts, pts, K, distCoef, noArray(), K): the five
// fixed-point iterations of OpenCV's calib3d in double precision, operation for operation as undistort_keypoints in
// host_logic.cpp (IEEE double add / multiply / divide, no contraction: -ffp-contract=off), so the two agree bit for bit.
struct UndistortArgs { double fx, fy, cx, cy, k[5]; };
__global__ void undistort_kernel(const float* xy, int n, UndistortArgs a, float* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double ifx = 1.0 / a.fx, ify = 1.0 / a.fy;
  double x = ((double)xy[2 * i] - a.cx) * ifx, y = ((double)xy[2 * i + 1] - a.cy) * ify;
  const double x0 = x, y0 = y;
  for (int it = 0; it < 5; it++) {
    const double r2 = x * x + y * y;
    const double icdist = 1.0 / (1.0 + ((a.k[4] * r2 + a.k[1]) * r2 + a.k[0]) * r2);
    if (icdist < 0) { x = x0; y = y0; break; }
    const double dx = 2 * a.k[2] * x * y + a.k[3] * (r2 + 2 * x * x);
    const double dy = a.k[2] * (r2 + 2 * y * y) + 2 * a.k[3] * x * y;
    x = (x0 - dx) * icdist;
    y = (y0 - dy) * icdist;
  }
  const double xx = a.fx * x + a.cx, yy = a.fy * y + a.cy;
  out[2 * i] = (float)xx;
  out[2 * i + 1] = (float)yy;
}
void launch_undistort(const float* xy, int n, const float K4[4], const float* dist, int ndist, float* out, hipStream_t s) {
  if (n <= 0) return;
  UndistortArgs a;
  a.fx = K4[0]; a.fy = K4[1]; a.cx = K4[2]; a.cy = K4[3];
  for (int i = 0; i < 5; i++) a.k[i] = i < ndist ? (double)dist[i] : 0.0;
  hipLaunchKernelGGL(undistort_kernel, dim3((n + 255) / 256), dim3(256), 0, s, xy, n, a, out);
}

// GetFeaturesInArea for nq windows: one wavefront per window.  FILL = false counts (counts[q]); FILL = true writes the
// candidate indices at cand_offsets[q] in the reference's order.  The arithmetic mirrors features_in_area (host_logic.cpp).
template <bool FILL>
__global__ __launch_bounds__(64) void area_candidates_kernel(const float* kxy, const int32_t* octave, float min_x, float min_y, float inv_w,
                                                             float inv_h, int cols, int rows, const int32_t* cell_offsets,
                                                             const int32_t* cell_idx, const float* qx, const float* qy, const float* qr,
                                                             const int32_t* min_level, const int32_t* max_level, int nq,
                                                             int32_t* counts, const int32_t* cand_offsets, int32_t* cand_idx) {
  const int q = blockIdx.x, lane = threadIdx.x;
  if (q >= nq) return;
  const float x = qx[q], y = qy[q], r = qr[q];
  const int minl = min_level ? min_level[q] : -1, maxl = max_level ? max_level[q] : -1;
  int total = 0;
  const int nMinCellX = max(0, (int)floorf((x - min_x - r) * inv_w));
  const int nMaxCellX = min(cols - 1, (int)ceilf((x - min_x + r) * inv_w));
  const int nMinCellY = max(0, (int)floorf((y - min_y - r) * inv_h));
  const int nMaxCellY = min(rows - 1, (int)ceilf((y - min_y + r) * inv_h));
  const bool hit = nMinCellX < cols && nMaxCellX >= 0 && nMinCellY < rows && nMaxCellY >= 0;
  const bool check_levels = (minl > 0) || (maxl >= 0);
  const int wbase = FILL ? cand_offsets[q] : 0;
  if (hit)
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
      // the cells (ix, nMinCellY .. nMaxCellY) are adjacent in the CSR (cell = ix * rows + iy): one contiguous index range
      const int j0 = cell_offsets[ix * rows + nMinCellY], j1 = cell_offsets[ix * rows + nMaxCellY + 1];
      for (int j = j0; j < j1; j += 64) {
        const int jj = j + lane;
        bool keep = false;
        int k = 0;
        if (jj < j1) {
          k = cell_idx[jj];
          keep = true;
          if (check_levels) {
            const int o = octave ? octave[k] : 0;
            if (o < minl) keep = false;
            if (maxl >= 0 && o > maxl) keep = false;
          }
          const float distx = kxy[2 * k] - x, disty = kxy[2 * k + 1] - y;
          keep = keep && fabsf(distx) < r && fabsf(disty) < r;
        }
        const unsigned long long mask = __ballot(keep);
        if (FILL && keep) cand_idx[wbase + total + __popcll(mask & ((1ull << lane) - 1ull))] = k;
        total += __popcll(mask);
      }
    }
  if (!FILL && lane == 0) counts[q] = total;
}

// exclusive scan of counts [nq] -> offsets [nq + 1], one workgroup (nq is a few thousand map points)
__global__ __launch_bounds__(1024) void exclusive_scan_kernel(const int32_t* counts, int n, int32_t* offsets) {
  __shared__ int32_t part[1024];
  const int tid = threadIdx.x;
  const int per = (n + 1023) / 1024, i0 = min(tid * per, n), i1 = min(i0 + per, n);
  int mine = 0;
  for (int i = i0; i < i1; i++) mine += counts[i];
  part[tid] = mine;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int base = part[tid] - mine;
  for (int i = i0; i < i1; i++) { offsets[i] = base; base += counts[i]; }
  if (tid == 1023) offsets[n] = part[tid];
}

void launch_area_count(const float* kxy, const int32_t* octave, float min_x, float min_y, float inv_w, float inv_h, int cols, int rows,
                       const int32_t* cell_offsets, const int32_t* cell_idx, const float* qx, const float* qy, const float* qr,
                       const int32_t* min_level, const int32_t* max_level, int nq, int32_t* counts, int32_t* cand_offsets, hipStream_t s) {
  if (nq <= 0) return;
  hipLaunchKernelGGL(area_candidates_kernel<false>, dim3(nq), dim3(64), 0, s, kxy, octave, min_x, min_y, inv_w, inv_h, cols, rows, cell_offsets,
                     cell_idx, qx, qy, qr, min_level, max_level, nq, counts, nullptr, nullptr);
  hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, s, counts, nq, cand_offsets);
}
void launch_area_fill(const float* kxy, const int32_t* octave, float min_x, float min_y, float inv_w, float inv_h, int cols, int rows,
                      const int32_t* cell_offsets, const int32_t* cell_idx, const float* qx, const float* qy, const float* qr,
                      const int32_t* min_level, const int32_t* max_level, int nq, const int32_t* cand_offsets, int32_t* cand_idx, hipStream_t s) {
  if (nq <= 0) return;
  hipLaunchKernelGGL(area_candidates_kernel<true>, dim3(nq), dim3(64), 0, s, kxy, octave, min_x, min_y, inv_w, inv_h, cols, rows, cell_offsets,
                     cell_idx, qx, qy, qr, min_level, max_level, nq, nullptr, cand_offsets, cand_idx);
}

// ---------------------------------------------------------------------------------------------
// One frame's features packed into the slab the shard-boundary all-gather moves (comm_engine.cpp; layout of
// dani_frontend_export_features): dst[0] = n, kpts [cap][2] at dst + 1, desc [cap][256] at dst + 1 + 2 cap; rows >= n are
// zero-filled so that the gathered bytes are fully defined.  1.06 MB at cap = 1036: HBM-bound copy, ~2 us.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_features_kernel(const float* kimg, const float* desc, int n, int cap, float* dst) {
  const size_t total = (size_t)cap * 258;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i == 0) dst[0] = (float)n;
  if (i >= total) return;
  const size_t nk = (size_t)cap * 2;
  float v;
  if (i < nk) v = i < (size_t)n * 2 ? kimg[i] : 0.0f;
  else { const size_t j = i - nk; v = j < (size_t)n * 256 ? desc[j] : 0.0f; }
  dst[1 + i] = v;
}
void launch_pack_features(const float* kimg, const float* desc, int n, int cap, float* dst, hipStream_t s) {
  const size_t total = (size_t)cap * 258;
  hipLaunchKernelGGL(pack_features_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, kimg, desc, n, cap, dst);
}

}  // namespace dani
