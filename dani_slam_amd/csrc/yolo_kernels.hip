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

// SPPF: the three chained nn.MaxPool2d(5, 1, 2) of one channel plane in one workgroup (plane in LDS, -inf padding as
// in PyTorch): out1 = mp5(in), out2 = mp5(out1), out3 = mp5(out2), written to the three following 256-channel slices
// of the concat buffer.  One launch instead of three 24576-workgroup ones.
__global__ __launch_bounds__(256) void sppf_pool_kernel(float* cat, int C, int H, int W, int Hp, int Wp, size_t fs) {
  __shared__ float pl[2][1024];
  const int c = blockIdx.x, b = blockIdx.y, n = H * W;
  const size_t plane = (size_t)Hp * Wp;
  float* base = cat + (size_t)b * fs + (size_t)c * plane;
  for (int i = threadIdx.x; i < n; i += 256) pl[0][i] = base[(size_t)(i / W + PADY) * Wp + i % W + PADX];
  __syncthreads();
  for (int k = 1; k <= 3; k++) {
    const float* src = pl[(k - 1) & 1];
    float* dst = pl[k & 1];
    float* out = base + (size_t)(k * C) * plane;
    for (int i = threadIdx.x; i < n; i += 256) {
      const int y = i / W, x = i % W;
      float m = -INFINITY;
      for (int yy = max(y - 2, 0); yy <= min(y + 2, H - 1); yy++)
        for (int xx = max(x - 2, 0); xx <= min(x + 2, W - 1); xx++) m = fmaxf(m, src[yy * W + xx]);
      dst[i] = m;
      out[(size_t)(y + PADY) * Wp + x + PADX] = m;
    }
    __syncthreads();
  }
}

void launch_sppf_pool(float* cat, int C, int H, int W, int Hp, int Wp, size_t fs, int B, hipStream_t s) {
  hipLaunchKernelGGL(sppf_pool_kernel, dim3(C, B), dim3(256), 0, s, cat, C, H, W, Hp, Wp, fs);
}

// nn.Upsample(scale 2, nearest): src [C][H][W] slice -> dst [C][2H][2W] slice (only Hd x Wd written); one workgroup
// per (channel, frame) plane
__global__ __launch_bounds__(256) void upsample2_kernel(const float* src, int sHp, int sWp, size_t sfs, float* dst, int Hd,
                                                        int Wd, int dHp, int dWp, size_t dfs) {
  const int c = blockIdx.x, b = blockIdx.y;
  const float* sp = src + (size_t)b * sfs + (size_t)c * sHp * sWp;
  float* dp = dst + (size_t)b * dfs + (size_t)c * dHp * dWp;
  for (int i = threadIdx.x; i < Hd * Wd; i += 256) {
    const int y = i / Wd, x = i % Wd;
    dp[(size_t)(y + PADY) * dWp + x + PADX] = sp[(size_t)((y >> 1) + PADY) * sWp + (x >> 1) + PADX];
  }
}

void launch_upsample2(const float* src, int sHp, int sWp, size_t sfs, float* dst, int C, int Hd, int Wd, int dHp, int dWp,
                      size_t dfs, int B, hipStream_t s) {
  hipLaunchKernelGGL(upsample2_kernel, dim3(C, B), dim3(256), 0, s, src, sHp, sWp, sfs, dst, Hd, Wd, dHp, dWp, dfs);
}

// ---------------------------------------------------------------------------------------------
// Detect decode + score filter: one thread per (anchor, cell), all three levels in one launch.  sigmoid = 1/(1+dani_expf(-x)),
// xy = (s*2 - 0.5 + grid) * stride, wh = (s*2)^2 * anchor, score = obj * max(cls) > thresh -> candidate record
// {pred index, cx, cy, w, h, obj, max cls, argmax cls} appended with an atomic counter (the host sorts the few
// records by pred index, so the result does not depend on the append order).
// ---------------------------------------------------------------------------------------------
struct DecodeLevels {
  const float* raw[3];
  int ny[3], nx[3], Hp[3], Wp[3], base[3], row0[3];   // row0: first blockIdx.y of the level (rows = anchors x ny)
  size_t fs[3];
};

__global__ __launch_bounds__(64) void yolo_decode_kernel(DecodeLevels lv, float score_thresh, float* cand, float* cand_fast, int fast_cap,
                                                         int* cand_count, int cand_cap) {
  const int level = blockIdx.y >= lv.row0[2] ? 2 : (blockIdx.y >= lv.row0[1] ? 1 : 0);
  const int ny = lv.ny[level], nx = lv.nx[level], row = blockIdx.y - lv.row0[level];
  const int x = blockIdx.x * 64 + threadIdx.x, y = row % ny, a = row / ny, b = blockIdx.z;
  if (x >= nx) return;
  const float anchors[3][6] = {{10, 13, 16, 30, 33, 23}, {30, 61, 62, 45, 59, 119}, {116, 90, 156, 198, 373, 326}};
  const float stride = level == 0 ? 8.0f : (level == 1 ? 16.0f : 32.0f);
  const size_t plane = (size_t)lv.Hp[level] * lv.Wp[level];
  const float* p = lv.raw[level] + (size_t)b * lv.fs[level] + (size_t)(a * 85) * plane + (size_t)(y + PADY) * lv.Wp[level] + x + PADX;
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
  const float aw = anchors[level][2 * a], ah = anchors[level][2 * a + 1];
  const float cx = (sx * 2.0f - 0.5f + (float)x) * stride, cy = (sy * 2.0f - 0.5f + (float)y) * stride;
  const float w2 = sw * 2.0f, h2 = sh * 2.0f;
  const float bw = (w2 * w2) * aw, bh = (h2 * h2) * ah;
  const int slot = atomicAdd(&cand_count[b], 1);
  if (slot < cand_cap) {
    float* o = slot < fast_cap ? cand_fast + ((size_t)b * fast_cap + slot) * 8 : cand + ((size_t)b * cand_cap + slot) * 8;   // the first fast_cap records of a frame live in the contiguous read-back block
    o[0] = __int_as_float(lv.base[level] + (a * ny + y) * nx + x);
    o[1] = cx; o[2] = cy; o[3] = bw; o[4] = bh; o[5] = obj; o[6] = best; o[7] = __int_as_float(bi);
  }
}

// all three Detect levels in one launch
void launch_yolo_decode(const ActDesc* raws[3], float score_thresh, float* cand, float* cand_fast, int fast_cap, int* cand_count, int cand_cap, int B, hipStream_t s) {
  DecodeLevels lv;
  int base = 0, row = 0, maxnx = 0;
  for (int i = 0; i < 3; i++) {
    const ActDesc& r = *raws[i];
    lv.raw[i] = r.p; lv.ny[i] = r.H; lv.nx[i] = r.W; lv.Hp[i] = r.Hp; lv.Wp[i] = r.Wp; lv.fs[i] = r.frame_stride();
    lv.base[i] = base; lv.row0[i] = row;
    base += 3 * r.H * r.W; row += 3 * r.H;
    maxnx = r.W > maxnx ? r.W : maxnx;
  }
  hipLaunchKernelGGL(yolo_decode_kernel, dim3((maxnx + 63) / 64, row, B), dim3(64), 0, s, lv, score_thresh, cand, cand_fast, fast_cap, cand_count, cand_cap);
}

}  // namespace dani
