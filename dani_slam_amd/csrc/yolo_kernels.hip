// yolo_kernels.hip -- the non-GEMM stages of the YOLOv5s dynamic-object detector (HBM / latency bound).
// Replaces, on the GPU side, the pre-processing and tensor plumbing of YoloDetection::Detect
// (/root/reference/src/YoloDetect.cpp:39-49) and the score filter of non_max_suppression (:91-93); the
// convolutions run in conv_mfma.hip, the order-defining greedy NMS and the box refinement stay on the host
// (host_logic.cpp) exactly as the quadtree does for SuperPoint.
#include "canon_math.h"
#include "common.h"

namespace dani {

// ---------------------------------------------------------------------------------------------
// cv::resize(BGR 8UC3 -> 640x380, INTER_LINEAR) restated from OpenCV's fixed-point path (coefficients scaled by
// 2^11, vertical pass ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2), then BGR->RGB, /255, HWC->CHW into the
// padded planar layout (4 planes: R, G, B, zero) with pady = 2 for the 6x6/2 stem.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void resize_coef(int d, double scale, int ssize, int& ofs, int& a0, int& a1) {
  float f = (float)((d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { s = 0; f = 0.0f; }
  if (s >= ssize - 1) { s = ssize - 1; f = 0.0f; }
  ofs = s;
  a0 = (int)(short)__float2int_rn((1.0f - f) * 2048.0f);
  a1 = (int)(short)__float2int_rn(f * 2048.0f);
}

__global__ __launch_bounds__(256) void yolo_preproc_kernel(const uint8_t* bgr, int sw, int sh, int sstride,
                                                           size_t sframe, float* out, int dw, int dh, int oHp, int oWp,
                                                           int opady) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
  if (x >= dw || y >= dh) return;
  int xo, xa0, xa1, yo, yb0, yb1;
  resize_coef(x, 1.0 / ((double)dw / sw), sw, xo, xa0, xa1);   // cv::resize: scale = 1 / ((double)dsize / ssize)
  resize_coef(y, 1.0 / ((double)dh / sh), sh, yo, yb0, yb1);
  const int x1 = min(xo + 1, sw - 1), y1 = min(yo + 1, sh - 1);
  const uint8_t* r0 = bgr + (size_t)b * sframe + (size_t)yo * sstride;
  const uint8_t* r1 = bgr + (size_t)b * sframe + (size_t)y1 * sstride;
  const size_t oplane = (size_t)oHp * oWp;
  float* o = out + (size_t)b * 4 * oplane + (size_t)(y + opady) * oWp + x + PADX;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const int h0 = r0[3 * xo + c] * xa0 + r0[3 * x1 + c] * xa1;
    const int h1 = r1[3 * xo + c] * xa0 + r1[3 * x1 + c] * xa1;
    int v = (((yb0 * (h0 >> 4)) >> 16) + ((yb1 * (h1 >> 4)) >> 16) + 2) >> 2;
    v = min(max(v, 0), 255);
    o[(size_t)(2 - c) * oplane] = (float)v / 255.0f;   // BGR -> RGB plane order
  }
}

void launch_yolo_preproc(const uint8_t* bgr, int B, int sw, int sh, int sstride, size_t sframe, const ActDesc& out,
                         hipStream_t s) {
  dim3 grid((out.W + 63) / 64, (out.H + 3) / 4, B);
  hipLaunchKernelGGL(yolo_preproc_kernel, grid, dim3(256), 0, s, bgr, sw, sh, sstride, sframe, out.p, out.W, out.H, out.Hp,
                     out.Wp, out.pady);
}

// nn.MaxPool2d(5, 1, 2) on C planes (SPPF); in/out are channel slices of the same concat buffer
__global__ __launch_bounds__(64) void maxpool5_kernel(const float* in, float* out, int H, int W, int Hp, int Wp,
                                                      size_t fs) {
  const int x = threadIdx.x, y = blockIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (x >= W) return;
  const float* p = in + (size_t)b * fs + (size_t)c * Hp * Wp;
  float m = -INFINITY;
  for (int dy = -2; dy <= 2; dy++) {
    const int yy = y + dy;
    if (yy < 0 || yy >= H) continue;
    for (int dx = -2; dx <= 2; dx++) {
      const int xx = x + dx;
      if (xx >= 0 && xx < W) m = fmaxf(m, p[(size_t)(yy + PADY) * Wp + xx + PADX]);
    }
  }
  out[(size_t)b * fs + (size_t)c * Hp * Wp + (size_t)(y + PADY) * Wp + x + PADX] = m;
}

void launch_maxpool5(const float* in, float* out, int C, int H, int W, int Hp, int Wp, size_t fs, int B, hipStream_t s) {
  hipLaunchKernelGGL(maxpool5_kernel, dim3(H, C, B), dim3(64), 0, s, in, out, H, W, Hp, Wp, fs);
}

// nn.Upsample(scale 2, nearest): src [C][H][W] slice -> dst [C][2H][2W] slice (only Hd x Wd written)
__global__ __launch_bounds__(64) void upsample2_kernel(const float* src, int sHp, int sWp, size_t sfs, float* dst, int Hd,
                                                       int Wd, int dHp, int dWp, size_t dfs) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y % Hd, c = blockIdx.y / Hd, b = blockIdx.z;
  if (x >= Wd) return;
  dst[(size_t)b * dfs + (size_t)c * dHp * dWp + (size_t)(y + PADY) * dWp + x + PADX] =
      src[(size_t)b * sfs + (size_t)c * sHp * sWp + (size_t)((y >> 1) + PADY) * sWp + (x >> 1) + PADX];
}

void launch_upsample2(const float* src, int sHp, int sWp, size_t sfs, float* dst, int C, int Hd, int Wd, int dHp, int dWp,
                      size_t dfs, int B, hipStream_t s) {
  hipLaunchKernelGGL(upsample2_kernel, dim3((Wd + 63) / 64, Hd * C, B), dim3(64), 0, s, src, sHp, sWp, sfs, dst, Hd, Wd, dHp,
                     dWp, dfs);
}

// ---------------------------------------------------------------------------------------------
// Detect decode + score filter: one thread per (anchor, cell) of one level.  sigmoid = 1/(1+dani_expf(-x)),
// xy = (s*2 - 0.5 + grid) * stride, wh = (s*2)^2 * anchor, score = obj * max(cls) > thresh -> candidate record
// {pred index, cx, cy, w, h, obj, max cls, argmax cls} appended with an atomic counter (the host sorts the few
// records by pred index, so the result does not depend on the append order).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void yolo_decode_kernel(const float* raw, int ny, int nx, int Hp, int Wp, size_t fs,
                                                         int level_base, float stride, float aw0, float ah0, float aw1,
                                                         float ah1, float aw2, float ah2, float score_thresh, float* cand,
                                                         int* cand_count, int cand_cap) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y % ny, a = blockIdx.y / ny, b = blockIdx.z;
  if (x >= nx) return;
  const size_t plane = (size_t)Hp * Wp;
  const float* p = raw + (size_t)b * fs + (size_t)(a * 85) * plane + (size_t)(y + PADY) * Wp + x + PADX;
  const float obj = 1.0f / (1.0f + dani_expf(-p[4 * plane]));
  float best = -1.0f;
  int bi = 0;
  for (int c = 0; c < 80; c++) {
    const float v = 1.0f / (1.0f + dani_expf(-p[(size_t)(5 + c) * plane]));
    if (v > best) { best = v; bi = c; }
  }
  const float score = obj * best;
  if (!(score > score_thresh)) return;
  const float sx = 1.0f / (1.0f + dani_expf(-p[0])), sy = 1.0f / (1.0f + dani_expf(-p[plane]));
  const float sw = 1.0f / (1.0f + dani_expf(-p[2 * plane])), sh = 1.0f / (1.0f + dani_expf(-p[3 * plane]));
  const float aw = a == 0 ? aw0 : (a == 1 ? aw1 : aw2), ah = a == 0 ? ah0 : (a == 1 ? ah1 : ah2);
  const float cx = (sx * 2.0f - 0.5f + (float)x) * stride, cy = (sy * 2.0f - 0.5f + (float)y) * stride;
  const float w2 = sw * 2.0f, h2 = sh * 2.0f;
  const float bw = (w2 * w2) * aw, bh = (h2 * h2) * ah;
  const int slot = atomicAdd(&cand_count[b], 1);
  if (slot < cand_cap) {
    float* o = cand + ((size_t)b * cand_cap + slot) * 8;
    o[0] = __int_as_float(level_base + (a * ny + y) * nx + x);
    o[1] = cx; o[2] = cy; o[3] = bw; o[4] = bh; o[5] = obj; o[6] = best; o[7] = __int_as_float(bi);
  }
}

void launch_yolo_decode(const float* raw, int ny, int nx, int Hp, int Wp, size_t fs, int level, int level_base,
                        float score_thresh, float* cand, int* cand_count, int cand_cap, int B, hipStream_t s) {
  static const float anchors[3][6] = {{10, 13, 16, 30, 33, 23}, {30, 61, 62, 45, 59, 119}, {116, 90, 156, 198, 373, 326}};
  static const float strides[3] = {8.0f, 16.0f, 32.0f};
  const float* an = anchors[level];
  hipLaunchKernelGGL(yolo_decode_kernel, dim3((nx + 63) / 64, ny * 3, B), dim3(64), 0, s, raw, ny, nx, Hp, Wp, fs, level_base,
                     strides[level], an[0], an[1], an[2], an[3], an[4], an[5], score_thresh, cand, cand_count, cand_cap);
}

}  // namespace dani
