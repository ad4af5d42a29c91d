// sp_post.hip -- the non-GEMM stages of the SuperPoint extractor (HBM / latency bound, integer + compare
// work; not reshaped into GEMMs):
//   softmax65 + dustbin drop + depth-to-space     /root/reference/src/SuperPoint.cpp:82-93
//   descriptor-map channel L2 normalisation       src/SuperPoint.cpp:79-80
//   per-cell threshold / nonzero / NMS2 (Q4)      src/SuperPoint.cpp:211-243, 277-363
//   cell loop + dynamic-box cull ordering         src/SPextractor.cpp:232-354
//   grid_sampler + keypoint-axis normalisation    src/SuperPoint.cpp:246-274
#include "canon_math.h"
#include "common.h"

namespace dani {

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// ---------------------------------------------------------------------------------------------
// softmax over the 65 detector channels, drop channel 64, depth-to-space 8x8.
// One thread per 8x8 cell; canonical order: max, exp, sequential sum c = 0..64, IEEE divide.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void softmax_d2s_kernel(const float* semi, int Hc, int Wc, int Hp, int Wp, int Cpad,
                                                         float* prob) {
  const int cx = blockIdx.x * 64 + threadIdx.x, cy = blockIdx.y, b = blockIdx.z;
  if (cx >= Wc) return;
  const size_t plane = (size_t)Hp * Wp;
  const float* s = semi + (size_t)b * Cpad * plane + (size_t)(cy + PADY) * Wp + cx + PADX;
  float e[65];
  float m = s[0];
#pragma unroll
  for (int c = 1; c < 65; c++) m = fmaxf(m, s[(size_t)c * plane]);
  float sum = 0.0f;
#pragma unroll
  for (int c = 0; c < 65; c++) {
    e[c] = dani_expf(s[(size_t)c * plane] - m);
    sum = sum + e[c];
  }
  const int PW = Wc * 8;
  float* o = prob + (size_t)b * (Hc * 8) * PW + (size_t)(cy * 8) * PW + cx * 8;
#pragma unroll
  for (int r = 0; r < 8; r++) {
    float4 v0, v1;
    v0.x = e[r * 8 + 0] / sum; v0.y = e[r * 8 + 1] / sum; v0.z = e[r * 8 + 2] / sum; v0.w = e[r * 8 + 3] / sum;
    v1.x = e[r * 8 + 4] / sum; v1.y = e[r * 8 + 5] / sum; v1.z = e[r * 8 + 6] / sum; v1.w = e[r * 8 + 7] / sum;
    reinterpret_cast<float4*>(o + (size_t)r * PW)[0] = v0;
    reinterpret_cast<float4*>(o + (size_t)r * PW)[1] = v1;
  }
}

void launch_softmax_d2s(const ActDesc& semi, float* prob, int B, hipStream_t s) {
  dim3 grid((semi.W + 63) / 64, semi.H, B);
  hipLaunchKernelGGL(softmax_d2s_kernel, grid, dim3(64), 0, s, semi.p, semi.H, semi.W, semi.Hp, semi.Wp, semi.C, prob);
}

// ---------------------------------------------------------------------------------------------
// desc /= ||desc||_2 over the 256 channels (sequential fma chain c = 0..255, sqrt, divide), written
// as NHWC [pixel][256] so that the keypoint sampler reads whole 1 KiB rows.
// Block = 64 pixels x 4 channel quarters.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void desc_norm_kernel(const float* desc, int Hc, int Wc, int Hp, int Wp,
                                                        float* out_nhwc, float* out_nchw) {
  __shared__ float s_sum[64];
  __shared__ float tile[64][65];
  const int lp = threadIdx.x & 63, q = threadIdx.x >> 6, b = blockIdx.y;
  const int npix = Hc * Wc;
  const int p = blockIdx.x * 64 + lp;
  const bool ok = p < npix;
  const int y = ok ? p / Wc : 0, x = ok ? p % Wc : 0;
  const size_t plane = (size_t)Hp * Wp;
  const float* src = desc + (size_t)b * 256 * plane + (size_t)(y + PADY) * Wp + x + PADX;
  float v[64];
#pragma unroll
  for (int i = 0; i < 64; i++) v[i] = ok ? src[(size_t)(q * 64 + i) * plane] : 0.0f;
  for (int qq = 0; qq < 4; qq++) {
    if (q == qq) {
      float s = qq ? s_sum[lp] : 0.0f;
#pragma unroll
      for (int i = 0; i < 64; i++) s = __builtin_fmaf(v[i], v[i], s);
      s_sum[lp] = s;
    }
    __syncthreads();
  }
  const float dn = sqrtf(s_sum[lp]);
#pragma unroll
  for (int i = 0; i < 64; i++) v[i] = v[i] / dn;
  if (out_nchw && ok) {
#pragma unroll
    for (int i = 0; i < 64; i++) out_nchw[((size_t)b * 256 + q * 64 + i) * npix + p] = v[i];
  }
  // transpose through LDS, one channel quarter at a time
  for (int qq = 0; qq < 4; qq++) {
    __syncthreads();
    if (q == qq) {
#pragma unroll
      for (int i = 0; i < 64; i++) tile[i][lp] = v[i];
    }
    __syncthreads();
    // 256 threads write 64 pixels x 64 channels: thread -> (pixel row = t/64 + 4*k, channel = t%64)
    for (int k = 0; k < 16; k++) {
      const int pp = (threadIdx.x >> 6) + 4 * k, c = threadIdx.x & 63;
      const int gp = blockIdx.x * 64 + pp;
      if (gp < npix) out_nhwc[((size_t)b * npix + gp) * 256 + qq * 64 + c] = tile[c][pp];
    }
  }
}

void launch_desc_norm(const ActDesc& desc, float* desc_nhwc, float* desc_nchw_opt, int B, hipStream_t s) {
  dim3 grid((desc.H * desc.W + 63) / 64, B);
  hipLaunchKernelGGL(desc_norm_kernel, grid, dim3(256), 0, s, desc.p, desc.H, desc.W, desc.Hp, desc.Wp, desc_nhwc,
                     desc_nchw_opt);
}

// ---------------------------------------------------------------------------------------------
// One wavefront per (cell, frame): SPDetector::getKeyPoints + NMS2 with the reference's literal
// semantics, then the dynamic-box cull, emitted in row-major order into the cell's slot.
//
// NMS2 quirk Q4 (src/SuperPoint.cpp:311-330): the 9x9 suppression compares
//     conf_flat[(y+4+k)*w + (x+4+j)]  <  conf_flat[(y+4)*w + (x+4)]
// on the UNPADDED w*h confidence buffer (padded coordinates through an unchecked cv::Mat::at);
// reads past the end of the buffer are defined as 0.0f here, exactly as in the oracle.
// The candidate visit order (row-major, == torch::nonzero order) is sequential by definition; the wave
// walks it with a ballot mask per row and applies each active candidate's 80 neighbour updates across
// lanes.  A candidate only suppresses anything if its shifted centre is itself a candidate (> 0).
// ---------------------------------------------------------------------------------------------
constexpr int NMS_D = 4;
constexpr int CELL_MAX_W = 64, CELL_MAX_H = 64;  // one lane per cell column

__global__ __launch_bounds__(64) void cell_nms_kernel(const float* prob, int PH, int PW, const CellRect* cells,
                                                      int ncells, float iniTh, float minTh, const int32_t* boxes,
                                                      const int32_t* box_offsets, int minBorder, float scale,
                                                      float* slot_xyr, int32_t* slot_count, int slot_stride) {
  __shared__ float conf[CELL_MAX_W * CELL_MAX_H];
  __shared__ int sbox[256 * 4];   // this frame's dynamic boxes (first 256; the rest are read from global memory)
  __shared__ uint8_t grid[(CELL_MAX_H + 2 * NMS_D) * (CELL_MAX_W + 2 * NMS_D)];
  const int cell = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const CellRect c = cells[cell];
  const int w = c.maxX - c.iniX, h = c.maxY - c.iniY, gw = w + 2 * NMS_D, gh = h + 2 * NMS_D;
  const int n = w * h;
  const float* pb = prob + (size_t)b * PH * PW;
  float* slot = slot_xyr + ((size_t)b * ncells + cell) * slot_stride * 3;
  const int nb0 = box_offsets ? box_offsets[b] : 0, nb1 = box_offsets ? box_offsets[b + 1] : 0;
  const int nbl = min(nb1 - nb0, 256);
  for (int i = lane; i < nbl * 4; i += 64) sbox[i] = boxes[4 * nb0 + i];

  int emitted = 0;
  for (int attempt = 0; attempt < 2; attempt++) {
    const float th = attempt ? minTh : iniTh;
    __syncthreads();
    for (int i = lane; i < gw * gh; i += 64) grid[i] = 0;
    __syncthreads();
    int cnt = 0;
    for (int i = lane; i < n; i += 64) {
      const int y = i / w, x = i - y * w;
      const float p = pb[(size_t)(c.iniY + y) * PW + c.iniX + x];
      const bool cand = p > th;
      conf[i] = cand ? p : 0.0f;
      if (cand) { grid[(y + NMS_D) * gw + x + NMS_D] = 1; cnt++; }
    }
    __syncthreads();
    const int total = wave_sum(cnt);
    if (total > 0) {
      // sequential NMS2 visit
      for (int y = 0; y < h; y++) {
        // a pixel's "initial candidate" flag == conf > 0 (th >= 0, prob > 0); live state is re-read below
        unsigned long long mask = __ballot(lane < w && conf[y * w + (lane < w ? lane : 0)] > 0.0f);
        while (mask) {
          const int x = __builtin_ctzll(mask);
          mask &= mask - 1;
          const int gidx = (y + NMS_D) * gw + x + NMS_D;
          if (grid[gidx] != 1) continue;  // uniform
          const int cidx = (y + NMS_D) * w + x + NMS_D;
          const float c0 = cidx < n ? conf[cidx] : 0.0f;
          if (c0 > 0.0f) {
            for (int t = lane; t < 81; t += 64) {
              const int k = t / 9 - NMS_D, jj = t % 9 - NMS_D;
              if (k == 0 && jj == 0) continue;
              const int i2 = (y + NMS_D + k) * w + x + NMS_D + jj;
              const float cv = (i2 >= 0 && i2 < n) ? conf[i2] : 0.0f;
              if (cv < c0) grid[(y + NMS_D + k) * gw + x + NMS_D + jj] = 0;
            }
          }
          __syncthreads();
          if (lane == 0) grid[gidx] = 2;
          __syncthreads();
        }
      }
      // emission in row-major order + dynamic-box cull.  The reference (SPextractor.cpp:314-352) maps ALL accumulated
      // points p -> (p + border) * scale, erases those inside a box, and maps back p * (1/scale) - border after
      // EVERY non-skipped cell, i.e. a point emitted by cell ci goes through (ncells - ci) round trips.  At scale 1
      // the round trip is exact (one test suffices); otherwise the f32 drift and the per-trip test are replayed here
      // literally (quirk Q6) -- the drifted coordinates are what the quadtree and the keypoints see.
      emitted = 0;
      const float fb = (float)minBorder, sinv = 1.0f / scale;
      const int trips = (scale == 1.0f) ? 1 : (ncells - cell);
      for (int y = 0; y < h; y++) {
        bool keep = lane < w && grid[(y + NMS_D) * gw + lane + NMS_D] == 2;
        float fx = (float)(lane + c.offX), fy = (float)(y + c.offY);
        if (__ballot(keep)) {
          for (int t = 0; t < trips; t++) {
            float sx = fx + fb, sy = fy + fb;
            sx *= scale; sy *= scale;
            const int px = __float2int_rn(sx), py = __float2int_rn(sy);   // Rect::contains(Point2f): saturate_cast<int>
            bool inside = false;
            for (int bi = 0; bi < nbl; bi++) {               // uniform trip count, LDS broadcast reads
              const int bx = sbox[4 * bi], by = sbox[4 * bi + 1], bw = sbox[4 * bi + 2], bh = sbox[4 * bi + 3];
              inside |= (bx <= px && px < bx + bw && by <= py && py < by + bh);
            }
            for (int bi = nb0 + nbl; bi < nb1; bi++) {
              const int bx = boxes[4 * bi], by = boxes[4 * bi + 1], bw = boxes[4 * bi + 2], bh = boxes[4 * bi + 3];
              inside |= (bx <= px && px < bx + bw && by <= py && py < by + bh);
            }
            keep = keep && !inside;
            if (scale != 1.0f) { sx *= sinv; sy *= sinv; fx = sx - fb; fy = sy - fb; }
          }
        }
        const unsigned long long km = __ballot(keep);
        if (keep) {
          const int pos = emitted + __popcll(km & ((1ull << lane) - 1));
          slot[3 * pos + 0] = fx;
          slot[3 * pos + 1] = fy;
          slot[3 * pos + 2] = conf[y * w + lane];
        }
        emitted += __popcll(km);
      }
      // the reference retries with minTh only when the cell produced NO keypoint (vKeysCell.empty())
      // BEFORE the cull (SPextractor.cpp:285-304); NMS2 never empties a non-empty candidate set.
      break;
    }
  }
  if (lane == 0) slot_count[b * ncells + cell] = emitted;
}

void launch_cell_nms(const float* prob, int PH, int PW, int B, const CellRect* cells, int ncells, float iniTh,
                     float minTh, const int32_t* boxes, const int32_t* box_offsets, int minBorder, float scale,
                     float* slot_xyr, int32_t* slot_count, int slot_stride, hipStream_t s) {
  hipLaunchKernelGGL(cell_nms_kernel, dim3(ncells, B), dim3(64), 0, s, prob, PH, PW, cells, ncells, iniTh, minTh,
                     boxes, box_offsets, minBorder, scale, slot_xyr, slot_count, slot_stride);
}

// ordered compaction: cell c of frame b copies its slot to cand[b][prefix(c) ...]
__global__ __launch_bounds__(64) void cell_compact_kernel(const float* slot_xyr, const int32_t* slot_count, int ncells,
                                                          int slot_stride, float* cand_xyr, int32_t* cand_count,
                                                          int cand_cap) {
  const int cell = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const int32_t* cnt = slot_count + b * ncells;
  int pre = 0;
  for (int i = lane; i < cell; i += 64) pre += cnt[i];
  pre = wave_sum(pre);
  const int mine = cnt[cell];
  const float* slot = slot_xyr + ((size_t)b * ncells + cell) * slot_stride * 3;
  float* dst = cand_xyr + (size_t)b * cand_cap * 3;
  for (int i = lane; i < mine * 3; i += 64) {
    const int k = pre * 3 + i;
    if (k < cand_cap * 3) dst[k] = slot[i];
  }
  if (cell == ncells - 1 && lane == 0) cand_count[b] = pre + mine;
}

void launch_cell_compact(const float* slot_xyr, const int32_t* slot_count, int ncells, int slot_stride, int B,
                         float* cand_xyr, int32_t* cand_count, int cand_cap, hipStream_t s) {
  hipLaunchKernelGGL(cell_compact_kernel, dim3(ncells, B), dim3(64), 0, s, slot_xyr, slot_count, ncells, slot_stride,
                     cand_xyr, cand_count, cand_cap);
}

// ---------------------------------------------------------------------------------------------
// SPDetector::computeDescriptors: bilinear grid_sampler (zeros padding, align_corners = false) of the
// normalised descriptor map at the keypoints, then every CHANNEL is divided by its L2 norm over the
// frame's keypoints (quirk Q7), output [n][256].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void desc_sample_kernel(const float* desc_nhwc, int Hc, int Wc, int PH, int PW,
                                                          const float* kxy, const int32_t* koff, const int32_t* nk,
                                                          int cap, float* sampled) {
  const int b = blockIdx.y, c = threadIdx.x;
  if ((int)blockIdx.x >= nk[b]) return;
  const int i = koff[b] + blockIdx.x;
  const float kx = kxy[((size_t)b * cap + i) * 2], ky = kxy[((size_t)b * cap + i) * 2 + 1];
  const float gx = 2.0f * kx / (float)PW - 1.0f;
  const float gy = 2.0f * ky / (float)PH - 1.0f;
  const float ix = ((gx + 1.0f) * (float)Wc - 1.0f) / 2.0f;
  const float iy = ((gy + 1.0f) * (float)Hc - 1.0f) / 2.0f;
  const float x0f = floorf(ix), y0f = floorf(iy);
  const float wx = ix - x0f, wy = iy - y0f;
  const float ex = 1.0f - wx, sy = 1.0f - wy;
  const float nw = sy * ex, ne = sy * wx, sw = wy * ex, se = wy * wx;
  const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
  const float* base = desc_nhwc + (size_t)b * Hc * Wc * 256 + c;
  auto at = [&](int y, int x) -> float {
    return (x >= 0 && x < Wc && y >= 0 && y < Hc) ? base[((size_t)y * Wc + x) * 256] : 0.0f;
  };
  float v = at(y0, x0) * nw;
  v = __builtin_fmaf(at(y0, x1), ne, v);
  v = __builtin_fmaf(at(y1, x0), sw, v);
  v = __builtin_fmaf(at(y1, x1), se, v);
  sampled[((size_t)b * cap + i) * 256 + c] = v;
}

// one block per frame, thread = channel: sequential fma chain over the keypoints (canonical order).  The chain is
// serial by definition, but its loads are not: 32 rows are fetched ahead of the 32 dependent fmas that consume them.
__global__ __launch_bounds__(256) void desc_colnorm_kernel(const float* sampled, const int32_t* koff, const int32_t* nk,
                                                           int cap, float* dn) {
  const int b = blockIdx.x, c = threadIdx.x;
  const int n = nk[b];
  const float* s = sampled + ((size_t)b * cap + koff[b]) * 256 + c;
  float acc = 0.0f;
  int i = 0;
  for (; i + 32 <= n; i += 32) {
    float v[32];
#pragma unroll
    for (int k = 0; k < 32; k++) v[k] = s[(size_t)(i + k) * 256];
#pragma unroll
    for (int k = 0; k < 32; k++) acc = __builtin_fmaf(v[k], v[k], acc);
  }
  for (; i < n; i++) {
    const float v = s[(size_t)i * 256];
    acc = __builtin_fmaf(v, v, acc);
  }
  dn[b * 256 + c] = sqrtf(acc);
}

__global__ __launch_bounds__(256) void desc_scale_kernel(const float* sampled, const float* dn, const int32_t* koff,
                                                         const int32_t* nk, int cap, float* out) {
  const int b = blockIdx.y, c = threadIdx.x;
  if ((int)blockIdx.x >= nk[b]) return;
  const size_t o = ((size_t)b * cap + koff[b] + blockIdx.x) * 256 + c;
  out[o] = sampled[o] / dn[b * 256 + c];
}

void launch_descriptors(const float* desc_nhwc, int Hc, int Wc, int PH, int PW, const float* kxy, const int32_t* koff,
                        const int32_t* nk, int B, int cap, int max_n, float* sampled, float* out, hipStream_t s) {
  // dn scratch lives at the tail of `sampled` (B*cap*256 floats followed by B*256)
  float* dn = sampled + (size_t)B * cap * 256;
  hipLaunchKernelGGL(desc_sample_kernel, dim3(max_n, B), dim3(256), 0, s, desc_nhwc, Hc, Wc, PH, PW, kxy, koff, nk, cap,
                     sampled);
  hipLaunchKernelGGL(desc_colnorm_kernel, dim3(B), dim3(256), 0, s, sampled, koff, nk, cap, dn);
  hipLaunchKernelGGL(desc_scale_kernel, dim3(max_n, B), dim3(256), 0, s, sampled, dn, koff, nk, cap, out);
}

// ---------------------------------------------------------------------------------------------
// cv::resize(8UC1 -> dw x dh, INTER_LINEAR): OpenCV's fixed-point path (coefficients x 2048, vertical pass
// ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2), scale = 1 / ((double)dsize / ssize) as imgproc/resize.cpp.
// Builds pyramid level k from level k-1 (SPextractor::ComputePyramid, src/SPextractor.cpp:631-656).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void resize_coef1(int d, double scale, int ssize, int& ofs, int& a0, int& a1) {
  float f = (float)((d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { s = 0; f = 0.0f; }
  if (s >= ssize - 1) { s = ssize - 1; f = 0.0f; }
  ofs = s;
  a0 = (int)(short)__float2int_rn((1.0f - f) * 2048.0f);
  a1 = (int)(short)__float2int_rn(f * 2048.0f);
}

__global__ __launch_bounds__(256) void resize_u8c1_kernel(const uint8_t* src, int sw, int sh, int sstride, size_t sframe,
                                                          uint8_t* dst, int dw, int dh) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
  if (x >= dw || y >= dh) return;
  int xo, xa0, xa1, yo, yb0, yb1;
  resize_coef1(x, 1.0 / ((double)dw / sw), sw, xo, xa0, xa1);
  resize_coef1(y, 1.0 / ((double)dh / sh), sh, yo, yb0, yb1);
  const int x1 = min(xo + 1, sw - 1), y1 = min(yo + 1, sh - 1);
  const uint8_t* r0 = src + (size_t)b * sframe + (size_t)yo * sstride;
  const uint8_t* r1 = src + (size_t)b * sframe + (size_t)y1 * sstride;
  const int h0 = r0[xo] * xa0 + r0[x1] * xa1, h1 = r1[xo] * xa0 + r1[x1] * xa1;
  int v = (((yb0 * (h0 >> 4)) >> 16) + ((yb1 * (h1 >> 4)) >> 16) + 2) >> 2;
  dst[(size_t)b * dw * dh + (size_t)y * dw + x] = (uint8_t)min(max(v, 0), 255);
}

void launch_resize_u8c1(const uint8_t* src, int B, int sw, int sh, int sstride, size_t sframe, uint8_t* dst, int dw, int dh,
                        hipStream_t s) {
  hipLaunchKernelGGL(resize_u8c1_kernel, dim3((dw + 63) / 64, (dh + 3) / 4, B), dim3(256), 0, s, src, sw, sh, sstride, sframe,
                     dst, dw, dh);
}

// tests only: padded planar frame b -> dense [C][H][W]
__global__ void copy_dense_kernel(const float* a, int C, int H, int W, int Hp, int Wp, int pady, float* out) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y, c = blockIdx.z;
  if (x < W) out[((size_t)c * H + y) * W + x] = a[((size_t)c * Hp + y + pady) * Wp + x + PADX];
}
void launch_copy_layer_dense(const ActDesc& a, int b, float* out_dense, hipStream_t s) {
  hipLaunchKernelGGL(copy_dense_kernel, dim3((a.W + 63) / 64, a.H, a.C), dim3(64), 0, s, a.p + b * a.frame_stride(),
                     a.C, a.H, a.W, a.Hp, a.Wp, a.pady, out_dense);
}

}  // namespace dani

// =================================================================================================
// the descriptor-distance scans of the projection searches.
// LGmatcher::DescriptorDistance = (float) cv::norm(a, b, NORM_L2) over 256 floats (src/LGmatcher.cpp:2009-2013;
// OpenCV: float difference, squared and accumulated in double, sqrt) and the best / second-best bookkeeping of
// LGmatcher::SearchByProjection (src/LGmatcher.cpp:84-120): candidates are visited in list order, strict `<`.
// One wavefront per query (map-point descriptor), 4 dimensions per lane; bound: HBM (1 KiB per candidate row).
// =================================================================================================
namespace dani {

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, off); hi = __shfl_xor(hi, off);
    v += __hiloint2double(hi, lo);
  }
  return v;
}

__global__ __launch_bounds__(64) void desc_best2_kernel(const float* q, const float* t, const int32_t* t_level,
                                                        const int32_t* cand_off, const int32_t* cand_idx, int nq,
                                                        float* best_dist, float* best_dist2, int32_t* best_idx,
                                                        int32_t* best_level, int32_t* best_level2, float* cand_dist) {
  const int qi = blockIdx.x, lane = threadIdx.x;
  if (qi >= nq) return;
  const float4 a = reinterpret_cast<const float4*>(q + (size_t)qi * 256)[lane];
  float bd = 256.0f, bd2 = 256.0f;
  int bl = -1, bl2 = -1, bi = -1;
  const int c0 = cand_off[qi], c1 = cand_off[qi + 1];
  // one candidate: float difference, double accumulation (4 per lane, then the wave tree), sqrt, strict-< bookkeeping
  auto sq = [&](const float4& b) {
    const float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
    return (double)d0 * (double)d0 + (double)d1 * (double)d1 + (double)d2 * (double)d2 + (double)d3 * (double)d3;
  };
  auto visit = [&](double ssum, int idx, int c_) {
    const float dist = (float)sqrt(ssum);
    if (cand_dist && lane == 0) cand_dist[c_] = dist;     // every candidate's distance, for callers that replay the scan
    const int lvl = t_level ? t_level[idx] : 0;
    if (dist < bd) { bd2 = bd; bd = dist; bl2 = bl; bl = lvl; bi = idx; }
    else if (dist < bd2) { bl2 = lvl; bd2 = dist; }
  };
  int c = c0;
  // four candidate rows in flight per wave (the scan is latency bound: rows come from L2); visited in list order
  for (; c + 4 <= c1; c += 4) {
    int idx[4];
    float4 b[4];
#pragma unroll
    for (int u = 0; u < 4; u++) idx[u] = cand_idx[c + u];
#pragma unroll
    for (int u = 0; u < 4; u++) b[u] = reinterpret_cast<const float4*>(t + (size_t)idx[u] * 256)[lane];
    double sv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) sv[u] = wave_sum_f64(sq(b[u]));
#pragma unroll
    for (int u = 0; u < 4; u++) visit(sv[u], idx[u], c + u);
  }
  for (; c < c1; c++) {
    const int idx = cand_idx[c];
    visit(wave_sum_f64(sq(reinterpret_cast<const float4*>(t + (size_t)idx * 256)[lane])), idx, c);
  }
  if (lane == 0) {
    best_dist[qi] = bd; best_dist2[qi] = bd2; best_idx[qi] = bi; best_level[qi] = bl; best_level2[qi] = bl2;
  }
}

void launch_desc_best2(const float* q, const float* t, const int32_t* t_level, const int32_t* cand_off,
                       const int32_t* cand_idx, int nq, float* best_dist, float* best_dist2, int32_t* best_idx,
                       int32_t* best_level, int32_t* best_level2, float* cand_dist, hipStream_t s) {
  if (nq > 0)
    hipLaunchKernelGGL(desc_best2_kernel, dim3(nq), dim3(64), 0, s, q, t, t_level, cand_off, cand_idx, nq, best_dist, best_dist2,
                       best_idx, best_level, best_level2, cand_dist);
}

// ---------------------------------------------------------------------------------------------
// MapPoint::ComputeDistinctiveDescriptors (/root/reference/src/MapPoint.cc:330-379) for many map
// points at once: mean of the observing keyframes' descriptor rows, accumulated in f32 in list order (cv::Mat += per
// row, then /= (float)count) -> bit-identical to the host loop.  One wavefront per map point, 4 channels per lane, four
// rows in flight (HBM / L2 gather, 1 KiB per row).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void desc_mean_kernel(const float* rows, const int32_t* off, const int32_t* idx, int nq, float* out) {
  const int qi = blockIdx.x, lane = threadIdx.x;
  if (qi >= nq) return;
  const int c0 = off[qi], c1 = off[qi + 1];
  float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  int c = c0;
  for (; c + 4 <= c1; c += 4) {
    float4 b[4];
#pragma unroll
    for (int u = 0; u < 4; u++) b[u] = reinterpret_cast<const float4*>(rows + (size_t)idx[c + u] * 256)[lane];
#pragma unroll
    for (int u = 0; u < 4; u++) { acc.x += b[u].x; acc.y += b[u].y; acc.z += b[u].z; acc.w += b[u].w; }
  }
  for (; c < c1; c++) {
    const float4 b = reinterpret_cast<const float4*>(rows + (size_t)idx[c] * 256)[lane];
    acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
  }
  if (c1 > c0) { const float n = (float)(c1 - c0); acc.x /= n; acc.y /= n; acc.z /= n; acc.w /= n; }
  reinterpret_cast<float4*>(out + (size_t)qi * 256)[lane] = acc;
}

void launch_desc_mean(const float* rows, const int32_t* off, const int32_t* idx, int nq, float* out, hipStream_t s) {
  if (nq > 0) hipLaunchKernelGGL(desc_mean_kernel, dim3(nq), dim3(64), 0, s, rows, off, idx, nq, out);
}

// ---------------------------------------------------------------------------------------------
// the tree descent of DBoW3's Vocabulary::transform for float descriptors
// (/root/reference/Thirdparty/DBow3/src/Vocabulary.cpp:836-873, called through Frame::ComputeBoW, src/Frame.cc:740-747).
// One wavefront per feature: at every level the squared L2 distance to each child -- DescManip::distance,
// DescManip.cpp:120-129: float difference, float product, double accumulation -- is formed 4 dimensions per lane and
// reduced over the wave; children are visited in stored order with strict `<`.  (The double sum of 256 float products is
// exact unless their magnitudes span more than ~2^29, so the tree-shaped reduction equals the reference's sequential
// one; a last-bit difference could only flip a decision between two children that tie to 1 ulp of a double.)
// Outputs per feature: the leaf node reached and the ancestor at level `nid_level` (0 = root when nid_level <= 0).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void bow_descend_kernel(const float* feats, int n, const int32_t* child_off, const int32_t* child_ids,
                                                         const float* node_desc, int nid_level, int32_t* leaf_out, int32_t* nid_out) {
  const int f = blockIdx.x, lane = threadIdx.x;
  if (f >= n) return;
  const float4 a = reinterpret_cast<const float4*>(feats + (size_t)f * 256)[lane];
  int node = 0, level = 0, nid = 0;
  for (;;) {
    const int c0 = child_off[node], c1 = child_off[node + 1];
    if (c1 <= c0) break;                       // leaf
    ++level;
    double best = 1.7976931348623157e308;
    int best_id = child_ids[c0];
    for (int c = c0; c < c1; c++) {
      const int id = child_ids[c];
      const float4 b = reinterpret_cast<const float4*>(node_desc + (size_t)id * 256)[lane];
      const float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
      const float p0 = d0 * d0, p1 = d1 * d1, p2 = d2 * d2, p3 = d3 * d3;
      const double sd = wave_sum_f64((((double)p0 + (double)p1) + (double)p2) + (double)p3);
      if (sd < best) { best = sd; best_id = id; }
    }
    node = best_id;
    if (level == nid_level) nid = node;
  }
  if (lane == 0) { leaf_out[f] = node; nid_out[f] = nid; }
}

void launch_bow_descend(const float* feats, int n, const int32_t* child_off, const int32_t* child_ids, const float* node_desc,
                        int nid_level, int32_t* leaf_out, int32_t* nid_out, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(bow_descend_kernel, dim3(n), dim3(64), 0, s, feats, n, child_off, child_ids, node_desc, nid_level, leaf_out, nid_out);
}

}  // namespace dani
