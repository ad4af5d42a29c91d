// conv_mfma.hip -- implicit-GEMM convolutions on the gfx950 FP32 matrix cores.
//
// Replaces the 11 MFMA-shaped Conv2d layers of SuperPoint::forward
// (/root/reference/src/SuperPoint.cpp:16-33 definitions, :58-77 use; LibTorch-CPU conv in the reference).
//
// GEMM view:  D[cout][pixel] = sum_k  W[cout][k] * X[k][pixel],   k = (cin-pair cp, ky, kx, cc)
//   A operand (32 rows)  = weights,  lane l holds W[cout = l&31][k = l>>5]
//   B operand (32 cols)  = pixels,   lane l holds X[k = l>>5][pixel = l&31]
// One v_mfma_f32_32x32x2_f32 therefore consumes input channels (2cp, 2cp+1) of one filter tap.  The
// instruction is bit-for-bit an ordered fmaf chain (tools/probes/mfma_chain_probe.hip), so walking k in
// the order (cp, ky, kx, cc) with the accumulator initialised to the bias reproduces the repo's canonical
// f32 evaluation order exactly (oracle/sp_oracle.cpp orc_conv2d).  Nothing is re-associated behind
// the specification's back: where a YOLOv5s layer's order is specified as a sum of K segments (Cfg::SEG, conv_kseg), the
// segments are separate chains added in the specified order.  The output is bit-identical to the oracle.
//
// Work decomposition: a workgroup = 4 waves = 64 couts x (TH x TW) pixels of one frame; each wave owns
// 2 cout blocks x NPB pixel blocks of 32 pixels (4 independent 32x32 accumulators = 64 VGPRs).  Input
// channels are streamed through LDS in chunks of CC channels (haloed tile + the matching weight slab);
// the next chunk is prefetched global->VGPR while the current one feeds the MFMAs, and 2-4 workgroups
// per CU cover each other's barriers.  Roofline: MFMA-bound (arithmetic intensity of a tile is ~80 FLOP/B
// against L2, >1000 FLOP/B against HBM).
#include "canon_math.h"
#include "common.h"
#include <cstdio>
#include <cstdlib>

namespace dani {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
  const float* in;
  const float* w;      // packed
  const float* bias;   // packed
  float* out;
  const float* res;    // optional residual (same plane geometry as out), added after the activation
  int Cin, Cout, H, W; // Cin = (padded) input channels; H, W = OUTPUT dims before pooling
  int iHp, iWp, ipady, oHp, oWp, opady;
  size_t in_fs, out_fs, res_fs;   // per-frame strides in floats (tensors may be channel slices of concat buffers)
  int tiles_x, tiles_y, cout_tiles, nchunks;
  int ntiles, tiles_base, tiles_rem;   // pixel tiles of the launch (B * tiles_y * tiles_x) = 8 * tiles_base + tiles_rem
  // Cfg::F1A (conv1b with conv1a computed on the fly): the u8 image instead of `in`, and conv1a's raw parameters
  const uint8_t* img; int img_stride; size_t img_fs;
  const float* w1a; const float* b1a;   // [64][9], [64]
};

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2 };

#ifdef DANI_CONV_STAMPS
// Debug build only (make STAMPS=1, tools/debug/conv_stamps.py): workgroup 0 of every conv launch records core-clock and
// 100 MHz wall-clock stamps of its chunk loop into g_conv_stamps[launch][..]: [0] = chunks, [1] wall at entry, [2] clock at
// entry, then per chunk (clock when the chunk is ready in LDS, clock when its MFMAs are issued), [92] = grid, [93] = 1000 ACT + 100 KS +
// 10 S + accumulators per wave, last two = clock, wall at exit.
__device__ unsigned long long g_conv_stamps[256 * 96];
__device__ int g_conv_launches;
#define STAMP(i_, v_) do { if (stamp_row >= 0 && (i_) < 96) g_conv_stamps[stamp_row * 96 + (i_)] = (v_); } while (0)
#else
#define STAMP(i_, v_) do { } while (0)
#endif

// SEG_: K segments.  The canonical order of a YOLOv5s layer with a long contraction (conv_kseg(), common.h) is a TWO-level sum:
// the input channels are cut into SEG equal contiguous ranges, each range is accumulated in the usual (channel pair, tap,
// channel) order -- the bias opens range 0 -- and the range sums are added left to right, ((s0 + s1) + s2) + s3.  A
// workgroup then gives each range to its own wave(s): 4 / SEG waves along the pixels, SEG along K, every wave a chain
// 1 / SEG as long, SEG times as many workgroups.  (The deep YOLO levels launch 200-600 workgroups that each walk K = 1152 ...
// 2304 alone)  SEG = 1: the plain order, everything else in this file.
// FW_: flat pixel tiles.  The two deepest YOLOv5s levels are 24 x 40 and 12 x 20 pixels per frame: rectangular tiles of 32-pixel
// blocks cover them at 83 % (8 x 16 tiles on 24 x 40) and 62 % (4 x 32 on 12 x 20) -- the rest of every MFMA is padding.  With
// FW_ = the level's width a pixel block is 32 CONSECUTIVE pixels of the frame in row-major order (960 = 30 blocks exactly,
// 240 = 7.5), a workgroup stages the full-width input rows its pixels touch, and every lane carries its own (row, column)
// into the operand address -- the k order, hence every result bit, is unchanged.
// F1A_: conv1b computes its own input.  SuperPoint's conv1a (1 -> 64 channels, 9 taps, ReLU; /root/reference/src/SuperPoint.cpp:58
// with detect()'s u8 -> f32 / 255, :192-193) is 0.35 GFLOP that used to write 629 MB per 8-frame step for conv1b to read back
// (profiles/r03_pmc_summary.json: 661 MB of HBM traffic per conv1a_kernel launch).  With F1A_ a conv1b workgroup keeps the
// (TH + 4) x (TW + 4) u8 window of its tile in LDS and every thread forms the haloed conv1a values of the next 8-channel chunk on
// the vector ALU -- 9 fmaf per value in conv1a_kernel's own order, so bit for bit what that kernel stores -- under the current
// chunk's MFMAs.  Positions outside the image are conv1b's zero padding (NOT conv1a of a padded image).
// EPI_: SuperPoint's two heads finish inside their last 1 x 1 layer (a workgroup holds ALL output channels of its 128 cells):
//   EPI_SOFTMAX_D2S  convPb (65 channels, NCB_ = 3): softmax over the channels, dustbin dropped, depth-to-space 8 x 8 into the dense
//                    probability map (/root/reference/src/SuperPoint.cpp:82-93) -- softmax_d2s_kernel's arithmetic, value for value
//   EPI_L2NORM_NHWC  convDb (256 channels, NCB_ = 8): desc / ||desc||_2 per cell, written as [cell][256] for the keypoint sampler
//                    (:79-80) -- desc_norm_kernel's arithmetic
// The raw layer outputs (semi, coarse descriptors) never reach HBM; sp_engine.cpp re-runs the plain layer for the debug hooks.
enum { EPI_NONE = 0, EPI_SOFTMAX_D2S = 1, EPI_L2NORM_NHWC = 2 };
template <int KS_, int S_, int PBH_, int PBW_, int NPB_, int CC_, int ACT_, bool POOL_, int NCB_ = 2, int SEG_ = 1, int FW_ = 0, bool F1A_ = false, int EPI_ = 0>
struct Cfg {
  static constexpr bool F1A = F1A_;
  static constexpr int EPI = EPI_;
  static_assert(EPI_ == 0 || (KS_ == 1 && S_ == 1 && NPB_ == 1 && SEG_ == 1 && FW_ == 0 && !POOL_ && ACT_ == 0 && !F1A_), "head epilogues sit on a plain 1x1 layer");
  static_assert(EPI_ != 1 || NCB_ == 3, "softmax over 65 channels: three cout blocks per wave");
  static_assert(EPI_ != 2 || NCB_ == 8, "L2 norm over 256 channels: eight cout blocks per wave");
  static constexpr int KS = KS_, S = S_, PBH = PBH_, PBW = PBW_, NPB = NPB_, CC = CC_, ACT = ACT_;
  static constexpr bool POOL = POOL_;
  static constexpr int T = KS * KS;
  static constexpr bool HAS_RES = (KS == 3 && S == 1 && ACT == ACT_SILU);   // only YOLOv5's bottleneck cv2 takes a shortcut operand
  static constexpr int PAD = (KS == 1) ? 0 : (KS == 3) ? 1 : 2;   // 1x1 p0, 3x3 p1, 6x6 p2 (YOLOv5 stem)
  static constexpr int NCB = NCB_, NCO = 32 * NCB_;   // cout blocks per wave / couts per workgroup
  static constexpr int SEG = SEG_, WPX = 4 / SEG_;     // K segments / waves along the pixels
  static constexpr int FW = FW_, NPIX = 32 * WPX * NPB_;   // flat tiles: pixels per workgroup; they span at most MR image rows
  static constexpr int MR = FW_ ? (NPIX - 2) / FW_ + 2 : 0;
  static constexpr int TH = FW_ ? MR : WPX * NPB * PBH, TW = FW_ ? FW_ : PBW; // WPX waves stacked vertically (output pixels)
  static constexpr int LROWS = S * (TH - 1) + KS;
  static constexpr int XOFF = (KS == 1) ? 0 : PADX - PAD;          // LDS column of input col (S*px + kx - PAD)
  static constexpr int RS = (XOFF + S * (TW - 1) + KS + 3) / 4 * 4; // LDS row stride (floats), 16-B multiple
  static constexpr int GX0 = (KS == 1) ? PADX : 0;                 // first staged padded column minus S*x0
  static constexpr int IN_F4 = CC * LROWS * (RS / 4);   // per K segment
  static constexpr int W_F4 = CC * T * NCO / 4;         // per K segment
  static constexpr int NI = (SEG * IN_F4 + 255) / 256, NW = (SEG * W_F4 + 255) / 256;   // staging float4 per thread, all segments
  static constexpr int SEG_IN = CC * LROWS * RS, SEG_W = CC * T * NCO;                  // LDS floats per segment
  static constexpr int IMG_RS = (TW + 4 + 3) / 4 * 4, IMG_ROWS = LROWS + 2;             // F1A: u8 window of the tile, as f32 / 255
  static constexpr int HALO_W = S * (TW - 1) + KS, HALO_N = LROWS * HALO_W;             // F1A: haloed positions of one input channel
  static constexpr int NPOS = (HALO_N + 255) / 256;                                     // F1A: positions per thread
  static constexpr int LDS_FLOATS = SEG * (SEG_IN + SEG_W) + (F1A_ ? IMG_ROWS * IMG_RS : 0);   // [seg][input tile], [seg][weight slab] (, image window)
  static_assert(!F1A_ || (KS_ == 3 && S_ == 1 && SEG_ == 1 && FW_ == 0 && CC_ == 8), "conv1a feeds a plain 3x3 stride-1 layer");
  static_assert(SEG == 1 || SEG == 2 || SEG == 4, "K segments");
  static_assert(SEG == 1 || !POOL_, "pooled layers are not split");
  static_assert(FW_ == 0 || !POOL_, "pooled layers keep rectangular tiles");
  static_assert((SEG - 1) * WPX * 64 * (16 * NCB * NPB + 1) <= LDS_FLOATS, "the segment sums meet in the tile buffers");
  static_assert(PBH * PBW == 32, "a pixel block is one MFMA column block");
  static_assert(!POOL || (PBH == 1 && NPB == 2 && S == 1), "fused pooling needs 2 image rows per wave");
  static_assert(LDS_FLOATS * 4 <= 65536, "LDS budget per workgroup");
};

template <class C>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(ConvArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
  float* sIn = lds;                       // [SEG][CC][LROWS][RS]
  float* sW = lds + C::SEG * C::SEG_IN;   // [SEG][CC / 2][T][2][NCO]
  // The SiLU instantiations are YOLOv5s layers: a chain of ~60 small dependent launches that runs concurrently with
  // SuperPoint's few huge ones and is the critical path of that phase.  Raise their waves' issue priority so that a
  // YOLO workgroup sharing a SIMD with SuperPoint waves is not slowed to their pace.
  if constexpr (C::ACT == ACT_SILU) __builtin_amdgcn_s_setprio(3);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;

  // XCD-aware tile order.  Workgroup ids are dealt round-robin to the 8 XCDs (id % 8), each with its own L2.  In plain
  // order (cout tile fastest) the 2-16 workgroups that read the SAME input tile land on different XCDs and every L2 fetches
  // it again from the fabric: 122 MB per launch for YOLO's 1x1 layers against 16-31 MB algorithmic, 886 MB against 360 MB
  // for the pooled SuperPoint layers, whose neighbouring tiles' halo rows were spread the same way
  // (profiles/r02_pmc_summary.json).  Here XCD x owns a CONTIGUOUS range of pixel tiles and walks it in order, all cout
  // tiles of a pixel tile back to back: slot = id / 8 counts (pixel tile, cout tile) pairs inside the XCD.  Speed only.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int ct = slot % a.cout_tiles;
  const int tl = slot / a.cout_tiles;
  // balanced contiguous ranges: XCD x owns tiles_base (+1 for x < tiles_rem) tiles starting at x * tiles_base + min(x, tiles_rem)
  if (tl >= a.tiles_base + (xcd < a.tiles_rem ? 1 : 0)) return;   // padding of the grid (whole workgroup, before any barrier)
  int tile = xcd * a.tiles_base + min(xcd, a.tiles_rem) + tl;     // < ntiles by construction
  const int tx = tile % a.tiles_x; tile /= a.tiles_x;
  const int ty = tile % a.tiles_y;
  const int b = tile / a.tiles_y;
  // flat tiles (tiles_y = 1): pixels [p0, p0 + NPIX) of the frame, staged from the full-width rows y0 ..
  const int p0 = C::FW ? tx * C::NPIX : 0;
  const int y0 = C::FW ? p0 / (C::FW ? C::FW : 1) : ty * C::TH, x0 = C::FW ? 0 : tx * C::TW;

#ifdef DANI_CONV_STAMPS
  int stamp_row = -1;
  if (blockIdx.x == 0 && tid == 0) { stamp_row = atomicAdd(&g_conv_launches, 1); if (stamp_row >= 256) stamp_row = -1; }
  STAMP(0, (unsigned long long)a.nchunks); STAMP(1, __builtin_amdgcn_s_memrealtime()); STAMP(2, __builtin_readcyclecounter());
  STAMP(92, (unsigned long long)gridDim.x); STAMP(93, (unsigned long long)(C::ACT * 1000 + C::KS * 100 + C::S * 10 + C::NCB * C::NPB));
#endif
  const size_t iplane = (size_t)a.iHp * a.iWp;
  const float* inb = a.in + (size_t)b * a.in_fs;
  // first staged element (padded coordinates): row S*y0 + ipady - PAD, column S*x0 + GX0
  const int gy0 = C::S * y0 + a.ipady - C::PAD;
  const int gx0 = C::S * x0 + C::GX0;
  const float* wt = a.w + (size_t)ct * a.nchunks * (C::W_F4 * 4);
  const int seg_chunks = a.nchunks / C::SEG;                 // chunks per K segment (launch_cfg checks divisibility)
  const int wpx = wave % C::WPX, seg = wave / C::WPX;        // this wave's pixel sub-tile and K segment

  // per-thread staging offsets (independent of the chunk): float4 e of the LDS image belongs to segment e / IN_F4
  int goff[C::NI];
#pragma unroll
  for (int t = 0; t < C::NI; t++) {
    const int e = tid + 256 * t;
    const int sg = e / C::IN_F4, el = e % C::IN_F4;
    const int q = el % (C::RS / 4);
    const int r = (el / (C::RS / 4)) % C::LROWS;
    const int c = el / ((C::RS / 4) * C::LROWS) + sg * seg_chunks * C::CC;
    goff[t] = (e < C::SEG * C::IN_F4) ? (int)(c * iplane) + (gy0 + r) * a.iWp + gx0 + 4 * q : 0;  // tail lanes re-read element 0
  }
  int woff[C::NW];
#pragma unroll
  for (int t = 0; t < C::NW; t++) {
    const int e = tid + 256 * t;
    const int sg = e / C::W_F4, el = e % C::W_F4;
    woff[t] = (e < C::SEG * C::W_F4) ? sg * seg_chunks * C::W_F4 + el : 0;
  }

  f32x16 acc[C::NCB][C::NPB];
#pragma unroll
  for (int cb = 0; cb < C::NCB; cb++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const float bv = seg ? 0.0f : a.bias[ct * C::NCO + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];   // the bias opens segment 0
#pragma unroll
      for (int pb = 0; pb < C::NPB; pb++) acc[cb][pb][r] = bv;
    }

  f32x4 pin[C::F1A ? 1 : C::NI], pw[C::NW];
  // F1A: this thread's haloed positions (row r, column cx of the LROWS x HALO_W grid), their 3 x 3 image windows and the
  // conv1a values of the chunk being prepared
  float win[C::F1A ? C::NPOS : 1][9], nv[C::F1A ? C::NPOS : 1][C::CC];
  int lpos[C::F1A ? C::NPOS : 1];       // LDS offset of the position inside a channel's tile, or -1
  bool inimg[C::F1A ? C::NPOS : 1];
  if constexpr (C::F1A) {
    float* sImg = lds + C::SEG * (C::SEG_IN + C::SEG_W);
    const uint8_t* im = a.img + (size_t)b * a.img_fs;
    for (int e = tid; e < C::IMG_ROWS * (C::TW + 4); e += 256) {
      const int r = e / (C::TW + 4), c = e % (C::TW + 4);
      const int iy = y0 - 2 + r, ix = x0 - 2 + c;
      sImg[r * C::IMG_RS + c] = (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) ? (float)im[(size_t)iy * a.img_stride + ix] / 255.0f : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < C::NPOS; u++) {
      const int pos = tid + 256 * u;
      const bool has = pos < C::HALO_N;
      const int r = has ? pos / C::HALO_W : 0, cx = has ? pos % C::HALO_W : 0;
      lpos[u] = has ? r * C::RS + C::XOFF + cx : -1;
      const int iy = y0 - 1 + r, ix = x0 - 1 + cx;                     // conv1b's input coordinate of this position
      inimg[u] = has && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
#pragma unroll
      for (int t = 0; t < 9; t++) win[u][t] = sImg[(r + t / 3) * C::IMG_RS + cx + t % 3];
    }
  }
  // conv1a of chunk ch_ at this thread's positions: acc = bias, taps in (ky, kx) order, ReLU (conv1a_kernel's arithmetic)
  auto conv1a_chunk = [&](int ch_) {
    if constexpr (C::F1A) {
#pragma unroll
      for (int cc = 0; cc < C::CC; cc++) {
        const float* wv = a.w1a + (size_t)(ch_ * C::CC + cc) * 9;      // uniform: scalar loads
        const float bv = a.b1a[ch_ * C::CC + cc];
#pragma unroll
        for (int u = 0; u < C::NPOS; u++) {
          float v = bv;
#pragma unroll
          for (int t = 0; t < 9; t++) v = __builtin_fmaf(win[u][t], wv[t], v);
          nv[u][cc] = (inimg[u] && v > 0.0f) ? v : 0.0f;
        }
      }
    }
  };
  // global -> VGPR prefetch of chunk `ch` (loads are unconditional: tail lanes re-read element 0)
#define DANI_PREFETCH(ch_)                                                                   \
  {                                                                                          \
    if constexpr (!C::F1A) {                                                                 \
      const float* src_ = inb + (size_t)(ch_) * C::CC * iplane;                              \
      _Pragma("unroll") for (int t = 0; t < C::NI; t++)                                      \
        pin[t] = *reinterpret_cast<const f32x4*>(src_ + goff[t]);                            \
    }                                                                                        \
    const f32x4* wsrc_ = reinterpret_cast<const f32x4*>(wt + (size_t)(ch_) * (C::W_F4 * 4)); \
    _Pragma("unroll") for (int t = 0; t < C::NW; t++)                                        \
      pw[t] = wsrc_[woff[t]];                                                                \
    conv1a_chunk(ch_);                                                                       \
  }

  const float* bA = sW + seg * C::SEG_W + h * C::NCO + j;
  const float* bB = sIn + seg * C::SEG_IN + (h * C::LROWS + C::S * (wpx * C::NPB * C::PBH + j / C::PBW)) * C::RS + C::S * (j % C::PBW) + C::XOFF;
  // flat tiles: this lane's pixel of block pb has its own (row, column); a pixel past the frame's end reads the last one's operands
  const float* bBf[C::NPB];
  const int npix = a.H * a.W;
#pragma unroll
  for (int pb = 0; pb < C::NPB; pb++) {
    constexpr int fw = C::FW ? C::FW : 1;
    const int p = min(p0 + (wpx * C::NPB + pb) * 32 + j, npix - 1);
    bBf[pb] = sIn + seg * C::SEG_IN + (h * C::LROWS + C::S * (p / fw - y0)) * C::RS + C::S * (p % fw) + C::XOFF;
  }

  DANI_PREFETCH(0);
  for (int ch = 0; ch < seg_chunks; ch++) {
    __syncthreads();  // everyone finished reading the previous chunk
    if constexpr (C::F1A) {
#pragma unroll
      for (int u = 0; u < C::NPOS; u++)
        if (lpos[u] >= 0) {
#pragma unroll
          for (int cc = 0; cc < C::CC; cc++) sIn[cc * C::LROWS * C::RS + lpos[u]] = nv[u][cc];
        }
    } else {
#pragma unroll
      for (int t = 0; t < C::NI; t++)
        if (tid + 256 * t < C::SEG * C::IN_F4) reinterpret_cast<f32x4*>(sIn)[tid + 256 * t] = pin[t];
    }
#pragma unroll
    for (int t = 0; t < C::NW; t++)
      if (tid + 256 * t < C::SEG * C::W_F4) reinterpret_cast<f32x4*>(sW)[tid + 256 * t] = pw[t];
    __syncthreads();
    STAMP(3 + 2 * ch, __builtin_readcyclecounter());
    if (ch + 1 < seg_chunks) DANI_PREFETCH(ch + 1);

#pragma unroll
    for (int cp = 0; cp < C::CC / 2; cp++)
#pragma unroll
      for (int ky = 0; ky < C::KS; ky++)
#pragma unroll
        for (int kx = 0; kx < C::KS; kx++) {
          float av[C::NCB], bv[C::NPB];
#pragma unroll
          for (int cb = 0; cb < C::NCB; cb++) av[cb] = bA[((cp * C::T + ky * C::KS + kx) * 2) * C::NCO + cb * 32];
#pragma unroll
          for (int pb = 0; pb < C::NPB; pb++)
            bv[pb] = C::FW ? bBf[pb][(2 * cp * C::LROWS + ky) * C::RS + kx] : bB[(2 * cp * C::LROWS + C::S * pb * C::PBH + ky) * C::RS + kx];
#pragma unroll
          for (int cb = 0; cb < C::NCB; cb++)
#pragma unroll
            for (int pb = 0; pb < C::NPB; pb++)
              acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb], bv[pb], acc[cb][pb], 0, 0, 0);
        }
    STAMP(4 + 2 * ch, __builtin_readcyclecounter());
  }

#undef DANI_PREFETCH
  if constexpr (C::SEG > 1) {
    // the segment sums meet in LDS (the tile buffers are free after the barrier) and are added left to right by segment 0's waves
    __syncthreads();
    constexpr int XS = 16 * C::NCB * C::NPB + 1;
    if (seg) {
      float* x = lds + (((seg - 1) * C::WPX + wpx) * 64 + lane) * XS;
#pragma unroll
      for (int cb = 0; cb < C::NCB; cb++)
#pragma unroll
        for (int pb = 0; pb < C::NPB; pb++)
#pragma unroll
          for (int r = 0; r < 16; r++) x[(cb * C::NPB + pb) * 16 + r] = acc[cb][pb][r];
    }
    __syncthreads();
    if (seg) return;
#pragma unroll
    for (int sg = 1; sg < C::SEG; sg++) {
      const float* x = lds + (((sg - 1) * C::WPX + wpx) * 64 + lane) * XS;
#pragma unroll
      for (int cb = 0; cb < C::NCB; cb++)
#pragma unroll
        for (int pb = 0; pb < C::NPB; pb++)
#pragma unroll
          for (int r = 0; r < 16; r++) acc[cb][pb][r] += x[(cb * C::NPB + pb) * 16 + r];
    }
  }
  // ---- epilogue ----
  // Addresses: an accumulator register's output channel is wave-uniform up to the lane half (cu + 4h), so the channel
  // base goes to the scalar unit and the lane adds ONE 32-bit byte offset shared by all 16 x NCB registers.  Written per
  // element as out[(size_t)co * oplane + (size_t)y * oWp + x] every store cost six 64-bit vector address instructions
  // (203 of the 582 epilogue VALU instructions of the 3x3 kernel).  Per-frame tensors are < 2^32 bytes (launch_cfg).
  const unsigned oplane = (unsigned)a.oHp * (unsigned)a.oWp;
  float* outb = a.out + (size_t)b * a.out_fs;
  auto at_w = [](float* base, unsigned off) -> float& { return *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off); };
  auto at_r = [](const float* base, unsigned off) -> const float& { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + off); };
  if constexpr (C::EPI == EPI_SOFTMAX_D2S) {
    // Lane (j, h) holds channels 32 cb + 8 g + 4 h + i (register 4 g + i) of ONE cell; its partner lane (j, 1 - h) holds the other
    // half.  softmax_d2s_kernel's order: max, e_c = dani_expf(s_c - max), sum over c = 0 .. 64 in order, e_c / sum.
    const int cy = y0 + wpx * C::PBH + j / C::PBW, cx = x0 + j % C::PBW;
    const bool ok = cy < a.H && cx < a.W;
    const float c64 = acc[2][0][0];                 // channel 64 (the dustbin) lives on lane half 0
    float m = acc[0][0][0];
#pragma unroll
    for (int cb = 0; cb < 2; cb++)
#pragma unroll
      for (int r = 0; r < 16; r++) m = fmaxf(m, acc[cb][0][r]);
    if (h == 0) m = fmaxf(m, c64);
    m = fmaxf(m, __shfl_xor(m, 32));
    float e[2][16], pe[2][16];
#pragma unroll
    for (int cb = 0; cb < 2; cb++)
#pragma unroll
      for (int r = 0; r < 16; r++) e[cb][r] = dani_expf(acc[cb][0][r] - m);
    float e64 = dani_expf(c64 - m);
#pragma unroll
    for (int cb = 0; cb < 2; cb++)
#pragma unroll
      for (int r = 0; r < 16; r++) pe[cb][r] = __shfl_xor(e[cb][r], 32);
    const float pe64 = __shfl_xor(e64, 32);
    float sum = 0.0f;
#pragma unroll
    for (int cb = 0; cb < 2; cb++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
#pragma unroll
        for (int i = 0; i < 4; i++) sum = sum + (h ? pe[cb][4 * g + i] : e[cb][4 * g + i]);   // channels 32 cb + 8 g + i     (lane half 0)
#pragma unroll
        for (int i = 0; i < 4; i++) sum = sum + (h ? e[cb][4 * g + i] : pe[cb][4 * g + i]);   // channels 32 cb + 8 g + 4 + i (lane half 1)
      }
    sum = sum + (h ? pe64 : e64);
    if (ok) {
      const int PW = a.W * 8;
      float* o = outb + (size_t)(cy * 8) * PW + cx * 8 + 4 * h;      // row dy = 4 cb + g of the cell's 8 x 8 block, columns 4 h .. 4 h + 3
#pragma unroll
      for (int cb = 0; cb < 2; cb++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
          f32x4 v;
#pragma unroll
          for (int i = 0; i < 4; i++) v[i] = e[cb][4 * g + i] / sum;
          *reinterpret_cast<f32x4*>(o + (size_t)(4 * cb + g) * PW) = v;
        }
    }
  } else if constexpr (C::EPI == EPI_L2NORM_NHWC) {
    // desc_norm_kernel's order: s = fma(d_c, d_c, s) for c = 0 .. 255, dn = sqrt(s), d_c / dn.  Both lanes of a cell walk the whole
    // chain (the partner's sixteen values of a cout block arrive by one cross-half exchange each).
    const int cy = y0 + wpx * C::PBH + j / C::PBW, cx = x0 + j % C::PBW;
    const bool ok = cy < a.H && cx < a.W;
    float ss = 0.0f;
#pragma unroll
    for (int cb = 0; cb < 8; cb++) {
      float pv[16];
#pragma unroll
      for (int r = 0; r < 16; r++) pv[r] = __shfl_xor(acc[cb][0][r], 32);
#pragma unroll
      for (int g = 0; g < 4; g++) {
#pragma unroll
        for (int i = 0; i < 4; i++) { const float d = h ? pv[4 * g + i] : acc[cb][0][4 * g + i]; ss = __builtin_fmaf(d, d, ss); }
#pragma unroll
        for (int i = 0; i < 4; i++) { const float d = h ? acc[cb][0][4 * g + i] : pv[4 * g + i]; ss = __builtin_fmaf(d, d, ss); }
      }
    }
    const float dn = sqrtf(ss);
    if (ok) {
      float* o = outb + ((size_t)cy * a.W + cx) * 256 + 4 * h;
#pragma unroll
      for (int cb = 0; cb < 8; cb++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
          f32x4 v;
#pragma unroll
          for (int i = 0; i < 4; i++) v[i] = acc[cb][0][4 * g + i] / dn;
          *reinterpret_cast<f32x4*>(o + 32 * cb + 8 * g) = v;
        }
    }
  } else if constexpr (C::POOL) {
    const int yo = (y0 + wave * 2) >> 1, xo = (x0 + j) >> 1;
    const bool ok = ((j & 1) == 0) && yo < (a.H >> 1) && xo < (a.W >> 1);
    const unsigned lane_off = 4u * ((unsigned)(yo + a.opady) * (unsigned)a.oWp + (unsigned)(xo + PADX) + 4u * (unsigned)h * oplane);
#pragma unroll
    for (int cb = 0; cb < C::NCB; cb++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        float m = fmaxf(acc[cb][0][r], acc[cb][1][r]);
        m = fmaxf(m, __shfl_xor(m, 1));
        if (C::ACT == ACT_RELU) m = m > 0.0f ? m : 0.0f;
        const int cu = ct * C::NCO + cb * 32 + (r & 3) + 8 * (r >> 2);   // uniform; this lane's channel is cu + 4h
        if (ok && cu + 4 * h < a.Cout) at_w(outb + (size_t)cu * oplane, lane_off) = m;
      }
  } else {
    const float* resb = a.res ? a.res + (size_t)b * a.res_fs : nullptr;
    // this lane's pixel of block pb: rectangular tiles (y0 + block row, x0 + column) or, flat, pixel p0 + ... of the frame
    auto pixel = [&](int pb, int& y, int& x) -> bool {
      if constexpr (C::FW != 0) {
        const int p = p0 + (wpx * C::NPB + pb) * 32 + j;
        y = p / C::FW; x = p % C::FW;
        return p < npix;
      } else {
        y = y0 + wpx * C::NPB * C::PBH + pb * C::PBH + j / C::PBW; x = x0 + j % C::PBW;
        return y < a.H && x < a.W;
      }
    };
    // Residual operands first, all of them behind ONE uniform branch: loaded next to their use inside the loop below, each
    // load sits in its own branch with a vmcnt(0) behind it -- 16 dependent round trips per accumulator at the very end of
    // the workgroup (tools/debug/conv_stamps.py: 11-16 k cycles of epilogue in the shortcut layers against 5 k in the others).
    float rv[C::HAS_RES ? C::NPB : 1][C::HAS_RES ? C::NCB : 1][16];
    if (C::HAS_RES && resb) {
#pragma unroll
      for (int pb = 0; pb < C::NPB; pb++) {
        int y, x;
        const bool ok = pixel(pb, y, x);
        const unsigned lane_off = 4u * ((unsigned)(y + a.opady) * (unsigned)a.oWp + (unsigned)(x + PADX) + 4u * (unsigned)h * oplane);
#pragma unroll
        for (int cb = 0; cb < C::NCB; cb++)
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int cu = ct * C::NCO + cb * 32 + (r & 3) + 8 * (r >> 2);
            rv[pb][cb][r] = (ok && cu + 4 * h < a.Cout) ? at_r(resb + (size_t)cu * oplane, lane_off) : 0.0f;
          }
      }
    }
#pragma unroll
    for (int pb = 0; pb < C::NPB; pb++) {
      int y, x;
      const bool ok = pixel(pb, y, x);
      const unsigned lane_off = 4u * ((unsigned)(y + a.opady) * (unsigned)a.oWp + (unsigned)(x + PADX) + 4u * (unsigned)h * oplane);
      if constexpr (C::ACT == ACT_SILU) {
        // The canonical SiLU -- x / (1 + dani_expf(-x)): ~30 dependent instructions per element, IEEE division included -- as 16
        // INDEPENDENT chains per accumulator in one branch-free block, the stores behind it: with a (predicated) store after every
        // element each chain sat in its own basic block and ran at its full dependent latency, 5000 cycles per workgroup
        // (tools/debug/conv_stamps.py), on ~50 launches of YOLO's chain.  Same operations per element: bit-identical.
#pragma unroll
        for (int cb = 0; cb < C::NCB; cb++) {
          float e[16], v[16];
#pragma unroll
          for (int r = 0; r < 16; r++) e[r] = dani_expf(-acc[cb][pb][r]);
#pragma unroll
          for (int r = 0; r < 16; r++) v[r] = acc[cb][pb][r] / (1.0f + e[r]);
          if constexpr (C::HAS_RES) {
            if (resb) {
#pragma unroll
              for (int r = 0; r < 16; r++) v[r] = rv[pb][cb][r] + v[r];
            }
          }
          // (the values are pinned HERE: left alone, hipcc sinks each element's ldexp and its whole division into the
          // predicated block of that element's store, where they run as one dependent chain again)
#pragma unroll
          for (int r = 0; r < 16; r++) asm volatile("" : "+v"(v[r]));
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int cu = ct * C::NCO + cb * 32 + (r & 3) + 8 * (r >> 2);   // uniform; this lane's channel is cu + 4h
            if (ok && cu + 4 * h < a.Cout) at_w(outb + (size_t)cu * oplane, lane_off) = v[r];
          }
        }
      } else {
#pragma unroll
        for (int cb = 0; cb < C::NCB; cb++)
#pragma unroll
          for (int r = 0; r < 16; r++) {
            float v = acc[cb][pb][r];
            if (C::ACT == ACT_RELU) v = v > 0.0f ? v : 0.0f;
            const int cu = ct * C::NCO + cb * 32 + (r & 3) + 8 * (r >> 2);   // uniform; this lane's channel is cu + 4h
            if (ok && cu + 4 * h < a.Cout) at_w(outb + (size_t)cu * oplane, lane_off) = v;
          }
      }
    }
  }
  STAMP(94, __builtin_readcyclecounter()); STAMP(95, __builtin_amdgcn_s_memrealtime());
}

//                         KS S PBH PBW NPB CC  ACT       POOL
using CfgRow32Relu      = Cfg<3, 1, 1, 32, 2, 8, ACT_RELU, false>;
using CfgRow32ReluPool  = Cfg<3, 1, 1, 32, 2, 8, ACT_RELU, true>;
using CfgBlk16Relu      = Cfg<3, 1, 2, 16, 2, 8, ACT_RELU, false>;
using CfgConv1Blk16     = Cfg<1, 1, 2, 16, 2, 32, ACT_NONE, false>;
// YOLOv5s variants (SiLU)
using CfgStem6Row32     = Cfg<6, 2, 1, 32, 2, 4, ACT_SILU, false>;
using CfgS2Row32Silu    = Cfg<3, 2, 1, 32, 2, 8, ACT_SILU, false>;
using CfgS2Blk16Silu    = Cfg<3, 2, 2, 16, 2, 8, ACT_SILU, false>;
using CfgRow32Silu      = Cfg<3, 1, 1, 32, 2, 8, ACT_SILU, false>;
using CfgBlk16Silu      = Cfg<3, 1, 2, 16, 2, 8, ACT_SILU, false>;
using CfgConv1Row32Silu = Cfg<1, 1, 1, 32, 2, 32, ACT_SILU, false>;
using CfgConv1Blk16Silu = Cfg<1, 1, 2, 16, 2, 32, ACT_SILU, false>;
// "small" variants for layers whose grid would not fill the chip: 8x16-pixel x 32-cout workgroups (one 32x32
// accumulator per wave) -> 4x more, 4x shorter workgroups.  Same k order, so results are bit-identical.
using CfgBlk16sRelu      = Cfg<3, 1, 2, 16, 1, 8, ACT_RELU, false, 1>;
using CfgConv1Blk16s     = Cfg<1, 1, 2, 16, 1, 32, ACT_NONE, false, 1>;
using CfgBlk16sSilu      = Cfg<3, 1, 2, 16, 1, 8, ACT_SILU, false, 1>;
using CfgS2Blk16sSilu    = Cfg<3, 2, 2, 16, 1, 8, ACT_SILU, false, 1>;
using CfgConv1Blk16sSilu = Cfg<1, 1, 2, 16, 1, 32, ACT_SILU, false, 1>;
// "narrow" 8x32-tile variants for layers with <= 32 output channels (YOLO stem and the first C3): 32 couts per WG
using CfgStem6Row32n     = Cfg<6, 2, 1, 32, 2, 4, ACT_SILU, false, 1>;
using CfgRow32nSilu      = Cfg<3, 1, 1, 32, 2, 8, ACT_SILU, false, 1>;
using CfgConv1Row32nSilu = Cfg<1, 1, 1, 32, 2, 32, ACT_SILU, false, 1>;
// 4 x 32-pixel tiles (wave = one image row of 32 pixels, one accumulator) x 32 couts: the 12 x 20 level of YOLOv5s
using CfgR4Silu          = Cfg<3, 1, 1, 32, 1, 8, ACT_SILU, false, 1>;
using CfgS2R4Silu        = Cfg<3, 2, 1, 32, 1, 8, ACT_SILU, false, 1>;
using CfgConv1R4Silu     = Cfg<1, 1, 1, 32, 1, 32, ACT_SILU, false, 1>;
using CfgConv1R4         = Cfg<1, 1, 1, 32, 1, 32, ACT_NONE, false, 1>;
// K-segmented twins (Cfg::SEG): the long-K layers of YOLO's two deepest levels
using CfgBlk16sSiluK2    = Cfg<3, 1, 2, 16, 1, 8, ACT_SILU, false, 1, 2>;
using CfgS2Blk16sSiluK2  = Cfg<3, 2, 2, 16, 1, 8, ACT_SILU, false, 1, 2>;
using CfgR4SiluK4        = Cfg<3, 1, 1, 32, 1, 8, ACT_SILU, false, 1, 4>;
using CfgS2R4SiluK2      = Cfg<3, 2, 1, 32, 1, 8, ACT_SILU, false, 1, 2>;
using CfgConv1Blk16sSiluK2 = Cfg<1, 1, 2, 16, 1, 32, ACT_SILU, false, 1, 2>;
using CfgConv1R4SiluK2   = Cfg<1, 1, 1, 32, 1, 32, ACT_SILU, false, 1, 2>;
using CfgConv1R4K2       = Cfg<1, 1, 1, 32, 1, 32, ACT_NONE, false, 1, 2>;
// flat pixel tiles (Cfg::FW) for the 24 x 40 and 12 x 20 levels of YOLOv5s: one 32-pixel block and 32 couts per wave
//                                KS S        NPB CC  ACT       POOL  NCB SEG FW
using CfgF40Silu3K2      = Cfg<3, 1, 1, 32, 1, 8, ACT_SILU, false, 1, 2, 40>;
using CfgF40Silu3S2K2    = Cfg<3, 2, 1, 32, 1, 8, ACT_SILU, false, 1, 2, 40>;
using CfgF40Silu1        = Cfg<1, 1, 1, 32, 1, 32, ACT_SILU, false, 1, 1, 40>;
using CfgF40Lin1         = Cfg<1, 1, 1, 32, 1, 32, ACT_NONE, false, 1, 1, 40>;
using CfgF20Silu3K4      = Cfg<3, 1, 1, 32, 1, 8, ACT_SILU, false, 1, 4, 20>;
using CfgF20Silu3S2K2    = Cfg<3, 2, 1, 32, 1, 8, ACT_SILU, false, 1, 2, 20>;
using CfgF20Silu1K2      = Cfg<1, 1, 1, 32, 1, 32, ACT_SILU, false, 1, 2, 20>;
// SuperPoint at small batch: narrow (32-cout) twins of the 8x32 tiles and 4x32-pixel tiles
using CfgRow32nRelu      = Cfg<3, 1, 1, 32, 2, 8, ACT_RELU, false, 1>;
using CfgRow32nReluPool  = Cfg<3, 1, 1, 32, 2, 8, ACT_RELU, true, 1>;
using CfgR4Relu          = Cfg<3, 1, 1, 32, 1, 8, ACT_RELU, false, 1>;
// conv1b with conv1a computed inside (Cfg::F1A): the pooled 8x32 tile at 64 and at 32 couts per workgroup
using CfgRow32ReluPoolF1A  = Cfg<3, 1, 1, 32, 2, 8, ACT_RELU, true, 2, 1, 0, true>;
using CfgRow32nReluPoolF1A = Cfg<3, 1, 1, 32, 2, 8, ACT_RELU, true, 1, 1, 0, true>;

// SuperPoint's heads with their post-processing in the epilogue (Cfg::EPI): 8 x 16 cells x ALL couts per workgroup
using CfgConv1SoftmaxD2S   = Cfg<1, 1, 2, 16, 1, 32, ACT_NONE, false, 3, 1, 0, false, EPI_SOFTMAX_D2S>;
using CfgConv1L2NormNhwc   = Cfg<1, 1, 2, 16, 1, 32, ACT_NONE, false, 8, 1, 0, false, EPI_L2NORM_NHWC>;

template <class C>
static bool launch_cfg(const ConvLayerPlan& L, const ConvIO& io, int B, hipStream_t s, int reserve_lds) {
  ConvArgs a;
  a.in = io.in; a.w = L.w_packed; a.bias = L.bias_packed; a.out = io.out; a.res = io.res;
  a.Cin = L.CinPad; a.Cout = L.Cout; a.H = io.H; a.W = io.W;
  a.iHp = io.iHp; a.iWp = io.iWp; a.ipady = io.ipady; a.oHp = io.oHp; a.oWp = io.oWp; a.opady = io.opady;
  a.in_fs = io.in_fs; a.out_fs = io.out_fs; a.res_fs = io.res_fs;
  a.img = io.img; a.img_stride = io.img_stride; a.img_fs = io.img_fs; a.w1a = io.w1a; a.b1a = io.b1a;
  if (C::F1A && (!io.img || !io.w1a || !io.b1a || L.CinPad != 64)) return false;   // the fused layer reads the image, not `in`
  if (C::FW && io.W != C::FW) return false;   // a flat instantiation is built for one level width
  if (C::EPI && (L.cout_tiles != 1 || L.Cout != (C::EPI == EPI_SOFTMAX_D2S ? 65 : 256) || io.res)) return false;   // a head epilogue needs the whole channel axis in one workgroup
  a.tiles_x = C::FW ? (io.H * io.W + C::NPIX - 1) / C::NPIX : (io.W + C::TW - 1) / C::TW;
  a.tiles_y = C::FW ? 1 : (io.H + C::TH - 1) / C::TH;
  a.cout_tiles = L.cout_tiles;
  a.nchunks = L.CinPad / C::CC;
  // 32-bit byte offsets inside a frame (epilogue): refused, not launched; the engine reports DANI_ERR_UNSUPPORTED
  if ((size_t)(L.cout_tiles * C::NCO + 8) * io.oHp * io.oWp >= ((size_t)1 << 30)) return false;
  a.ntiles = B * a.tiles_x * a.tiles_y;
  a.tiles_base = a.ntiles / 8; a.tiles_rem = a.ntiles % 8;
  if (a.ntiles <= 0 || a.cout_tiles <= 0) return false;
  if (io.res && !C::HAS_RES) return false;   // this instantiation has no residual path
  if (a.nchunks % C::SEG) return false;       // K segments are whole chunks
  const int grid = 8 * (a.tiles_base + (a.tiles_rem ? 1 : 0)) * a.cout_tiles;   // 8 XCDs x (largest share of pixel tiles) x cout tiles
  // reserve_lds: dynamic LDS the kernel never touches.  It only lowers how many of THESE workgroups fit on a CU, so that
  // a slot (registers, waves and the rest of the LDS) stays free for the workgroups of a concurrent stream (the fused
  // front-end: YOLO's chain of small dependent kernels next to SuperPoint's huge ones, sp_engine.cpp).
  hipLaunchKernelGGL(conv_mfma_kernel<C>, dim3(grid), dim3(256), (size_t)reserve_lds, s, a);
  return true;
}

bool launch_conv_io(const ConvLayerPlan& L, const ConvIO& io, int B, hipStream_t s, int reserve_lds) {
  switch (L.variant) {
    case CONV3_ROW32_RELU: return launch_cfg<CfgRow32Relu>(L, io, B, s, reserve_lds);
    case CONV3_ROW32_RELU_POOL: return launch_cfg<CfgRow32ReluPool>(L, io, B, s, reserve_lds);
    case CONV3_BLK16_RELU: return launch_cfg<CfgBlk16Relu>(L, io, B, s, reserve_lds);
    case CONV1_BLK16_LINEAR: return launch_cfg<CfgConv1Blk16>(L, io, B, s, reserve_lds);
    case CONV6S2_ROW32_SILU: return launch_cfg<CfgStem6Row32>(L, io, B, s, reserve_lds);
    case CONV3S2_ROW32_SILU: return launch_cfg<CfgS2Row32Silu>(L, io, B, s, reserve_lds);
    case CONV3S2_BLK16_SILU: return launch_cfg<CfgS2Blk16Silu>(L, io, B, s, reserve_lds);
    case CONV3_ROW32_SILU: return launch_cfg<CfgRow32Silu>(L, io, B, s, reserve_lds);
    case CONV3_BLK16_SILU: return launch_cfg<CfgBlk16Silu>(L, io, B, s, reserve_lds);
    case CONV1_ROW32_SILU: return launch_cfg<CfgConv1Row32Silu>(L, io, B, s, reserve_lds);
    case CONV1_BLK16_SILU: return launch_cfg<CfgConv1Blk16Silu>(L, io, B, s, reserve_lds);
    case CONV3_BLK16S_RELU: return launch_cfg<CfgBlk16sRelu>(L, io, B, s, reserve_lds);
    case CONV1_BLK16S_LINEAR: return launch_cfg<CfgConv1Blk16s>(L, io, B, s, reserve_lds);
    case CONV3_BLK16S_SILU: return launch_cfg<CfgBlk16sSilu>(L, io, B, s, reserve_lds);
    case CONV3S2_BLK16S_SILU: return launch_cfg<CfgS2Blk16sSilu>(L, io, B, s, reserve_lds);
    case CONV1_BLK16S_SILU: return launch_cfg<CfgConv1Blk16sSilu>(L, io, B, s, reserve_lds);
    case CONV6S2_ROW32N_SILU: return launch_cfg<CfgStem6Row32n>(L, io, B, s, reserve_lds);
    case CONV3_ROW32N_SILU: return launch_cfg<CfgRow32nSilu>(L, io, B, s, reserve_lds);
    case CONV1_ROW32N_SILU: return launch_cfg<CfgConv1Row32nSilu>(L, io, B, s, reserve_lds);
    case CONV3_R4_SILU: return launch_cfg<CfgR4Silu>(L, io, B, s, reserve_lds);
    case CONV3S2_R4_SILU: return launch_cfg<CfgS2R4Silu>(L, io, B, s, reserve_lds);
    case CONV1_R4_SILU: return launch_cfg<CfgConv1R4Silu>(L, io, B, s, reserve_lds);
    case CONV1_R4_LINEAR: return launch_cfg<CfgConv1R4>(L, io, B, s, reserve_lds);
    case CONV3_BLK16S_SILU_K2: return launch_cfg<CfgBlk16sSiluK2>(L, io, B, s, reserve_lds);
    case CONV3S2_BLK16S_SILU_K2: return launch_cfg<CfgS2Blk16sSiluK2>(L, io, B, s, reserve_lds);
    case CONV3_R4_SILU_K4: return launch_cfg<CfgR4SiluK4>(L, io, B, s, reserve_lds);
    case CONV3S2_R4_SILU_K2: return launch_cfg<CfgS2R4SiluK2>(L, io, B, s, reserve_lds);
    case CONV1_BLK16S_SILU_K2: return launch_cfg<CfgConv1Blk16sSiluK2>(L, io, B, s, reserve_lds);
    case CONV1_R4_SILU_K2: return launch_cfg<CfgConv1R4SiluK2>(L, io, B, s, reserve_lds);
    case CONV1_R4_LINEAR_K2: return launch_cfg<CfgConv1R4K2>(L, io, B, s, reserve_lds);
    case CONV3_ROW32N_RELU: return launch_cfg<CfgRow32nRelu>(L, io, B, s, reserve_lds);
    case CONV3_ROW32N_RELU_POOL: return launch_cfg<CfgRow32nReluPool>(L, io, B, s, reserve_lds);
    case CONV3_R4_RELU: return launch_cfg<CfgR4Relu>(L, io, B, s, reserve_lds);
    case CONV3_ROW32_RELU_POOL_F1A: return launch_cfg<CfgRow32ReluPoolF1A>(L, io, B, s, reserve_lds);
    case CONV3_ROW32N_RELU_POOL_F1A: return launch_cfg<CfgRow32nReluPoolF1A>(L, io, B, s, reserve_lds);
    case CONV3_F40_SILU_K2: return launch_cfg<CfgF40Silu3K2>(L, io, B, s, reserve_lds);
    case CONV3S2_F40_SILU_K2: return launch_cfg<CfgF40Silu3S2K2>(L, io, B, s, reserve_lds);
    case CONV1_F40_SILU: return launch_cfg<CfgF40Silu1>(L, io, B, s, reserve_lds);
    case CONV1_F40_LINEAR: return launch_cfg<CfgF40Lin1>(L, io, B, s, reserve_lds);
    case CONV3_F20_SILU_K4: return launch_cfg<CfgF20Silu3K4>(L, io, B, s, reserve_lds);
    case CONV3S2_F20_SILU_K2: return launch_cfg<CfgF20Silu3S2K2>(L, io, B, s, reserve_lds);
    case CONV1_F20_SILU_K2: return launch_cfg<CfgF20Silu1K2>(L, io, B, s, reserve_lds);
    case CONV1_BLK16S_SOFTMAX_D2S: return launch_cfg<CfgConv1SoftmaxD2S>(L, io, B, s, reserve_lds);
    case CONV1_BLK16S_L2NORM_NHWC: return launch_cfg<CfgConv1L2NormNhwc>(L, io, B, s, reserve_lds);
  }
  return false;
}

// SuperPoint convenience form: whole tensors in the standard padded planar layout
bool launch_conv_mfma(const ConvLayerPlan& L, const ActDesc& in, const ActDesc& out, int B, hipStream_t s, int reserve_lds) {
  ConvIO io;
  io.in = in.p; io.out = out.p; io.res = nullptr;
  io.H = in.H; io.W = in.W;   // stride-1 layers: conv output dims == input dims (pooling happens in the epilogue)
  io.iHp = in.Hp; io.iWp = in.Wp; io.ipady = in.pady; io.oHp = out.Hp; io.oWp = out.Wp; io.opady = out.pady;
  io.in_fs = in.frame_stride(); io.out_fs = out.frame_stride(); io.res_fs = 0;
  return launch_conv_io(L, io, B, s, reserve_lds);
}

bool launch_conv_mfma_f1a(const ConvLayerPlan& L, const uint8_t* img, int img_stride, size_t img_fs, const float* w1a, const float* b1a,
                          const ActDesc& in, const ActDesc& out, int B, hipStream_t s, int reserve_lds) {
  ConvIO io;
  io.in = in.p; io.out = out.p; io.res = nullptr;
  io.H = in.H; io.W = in.W;
  io.iHp = in.Hp; io.iWp = in.Wp; io.ipady = in.pady; io.oHp = out.Hp; io.oWp = out.Wp; io.opady = out.pady;
  io.in_fs = in.frame_stride(); io.out_fs = out.frame_stride(); io.res_fs = 0;
  io.img = img; io.img_stride = img_stride; io.img_fs = img_fs; io.w1a = w1a; io.b1a = b1a;
  return launch_conv_io(L, io, B, s, reserve_lds);
}

// A SuperPoint head whose epilogue writes the finished map (L.variant is an EPI variant): dst = the dense probability map
// [B][8 Hc][8 Wc] or the normalised descriptors [B][Hc * Wc][256]; dst_fs = floats per frame
bool launch_conv_head(const ConvLayerPlan& L, const ActDesc& in, float* dst, size_t dst_fs, int B, hipStream_t s, int reserve_lds) {
  ConvIO io;
  io.in = in.p; io.out = dst; io.res = nullptr;
  io.H = in.H; io.W = in.W;
  io.iHp = in.Hp; io.iWp = in.Wp; io.ipady = in.pady; io.oHp = 0; io.oWp = 0; io.opady = 0;
  io.in_fs = in.frame_stride(); io.out_fs = dst_fs; io.res_fs = 0;
  if (L.variant != CONV1_BLK16S_SOFTMAX_D2S && L.variant != CONV1_BLK16S_L2NORM_NHWC) return false;
  return launch_conv_io(L, io, B, s, reserve_lds);
}

#ifdef DANI_CONV_STAMPS
}  // namespace dani
extern "C" __attribute__((visibility("default"))) int dani_debug_conv_stamps(unsigned long long* out, int reset) {
  if (reset) { int z = 0; return (int)hipMemcpyToSymbol(HIP_SYMBOL(dani::g_conv_launches), &z, 4); }
  int n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(dani::g_conv_launches), 4) != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(dani::g_conv_stamps), sizeof(unsigned long long) * 256 * 96) != hipSuccess) return -1;
  return n;
}
namespace dani {
#endif

const char* conv_kernel_name(ConvVariant v) {
  switch (v) {
    case CONV3_ROW32_RELU: return "conv_mfma_kernel<Cfg<3,1,1,32,2,8,RELU,false>>";
    case CONV3_ROW32_RELU_POOL: return "conv_mfma_kernel<Cfg<3,1,1,32,2,8,RELU,true>>";
    case CONV3_BLK16_RELU: return "conv_mfma_kernel<Cfg<3,1,2,16,2,8,RELU,false>>";
    case CONV1_BLK16_LINEAR: return "conv_mfma_kernel<Cfg<1,1,2,16,2,32,NONE,false>>";
    case CONV3_BLK16S_RELU: return "conv_mfma_kernel<Cfg<3,1,2,16,1,8,RELU,false,1>>";
    case CONV1_BLK16S_LINEAR: return "conv_mfma_kernel<Cfg<1,1,2,16,1,32,NONE,false,1>>";
    case CONV3_ROW32N_RELU: return "conv_mfma_kernel<Cfg<3,1,1,32,2,8,RELU,false,1>>";
    case CONV3_ROW32N_RELU_POOL: return "conv_mfma_kernel<Cfg<3,1,1,32,2,8,RELU,true,1>>";
    case CONV3_R4_RELU: return "conv_mfma_kernel<Cfg<3,1,1,32,1,8,RELU,false,1>>";
    case CONV3_ROW32_RELU_POOL_F1A: return "conv_mfma_kernel<Cfg<3,1,1,32,2,8,RELU,true,2,1,0,F1A>>+conv1a";
    case CONV3_ROW32N_RELU_POOL_F1A: return "conv_mfma_kernel<Cfg<3,1,1,32,2,8,RELU,true,1,1,0,F1A>>+conv1a";
    case CONV1_BLK16S_SOFTMAX_D2S: return "conv_mfma_kernel<Cfg<1,1,2,16,1,32,NONE,false,3,1,0,false,EPI_SOFTMAX_D2S>>+softmax_d2s";
    case CONV1_BLK16S_L2NORM_NHWC: return "conv_mfma_kernel<Cfg<1,1,2,16,1,32,NONE,false,8,1,0,false,EPI_L2NORM_NHWC>>+l2norm_nhwc";
    default: return "conv_mfma_kernel<yolo variant>";
  }
}

// OIHW -> [cout_tile][chunk][cp][tap][cc][NCO]; couts beyond Cout are zero rows.
void pack_conv_weights(const float* w, const float* bias, int Cin, int Cout, int ks, int CC, int NCO, float* wp,
                       float* bp) {
  const int T = ks * ks, cout_tiles = (Cout + NCO - 1) / NCO, nch = (Cin + CC - 1) / CC;   // Cin padded up to CC
  size_t o = 0;
  for (int ct = 0; ct < cout_tiles; ct++)
    for (int ch = 0; ch < nch; ch++)
      for (int cp = 0; cp < CC / 2; cp++)
        for (int t = 0; t < T; t++)
          for (int cc = 0; cc < 2; cc++)
            for (int co = 0; co < NCO; co++) {
              const int cout = ct * NCO + co, cin = ch * CC + 2 * cp + cc;
              wp[o++] = (cout < Cout && cin < Cin) ? w[((size_t)cout * Cin + cin) * T + t] : 0.0f;
            }
  for (int i = 0; i < cout_tiles * NCO; i++) bp[i] = (i < Cout) ? bias[i] : 0.0f;
}

// ---------------------------------------------------------------------------------------------
// conv1a: Cin = 1, K = 9 -- not MFMA-shaped (0.35 GFLOP, HBM-bound on its 78.6 MB output at 640x480).
// Fuses SPDetector::detect's u8 -> f32 / 255 (/root/reference/src/SuperPoint.cpp:192-193) with
// relu(conv1a(x)) (:58).  One thread per pixel, 64 couts in a loop, weights through the scalar cache.
// Canonical order for Cin = 1: acc = bias; taps in (ky, kx) order.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv1a_kernel(const uint8_t* img, int H, int W, int stride, size_t fstride,
                                                     const float* __restrict__ w, const float* __restrict__ bias,
                                                     float* out, int oHp, int oWp) {
  // one thread = 4 horizontally adjacent pixels: a 3 x 6 input window in registers, one 16-byte store per channel
  const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.z;
  if (x >= W || y >= H) return;
  const uint8_t* im = img + (size_t)b * fstride;
  float p[3][6];
#pragma unroll
  for (int ky = 0; ky < 3; ky++)
#pragma unroll
    for (int c = 0; c < 6; c++) {
      const int yy = y + ky - 1, xx = x + c - 1;
      p[ky][c] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? (float)im[(size_t)yy * stride + xx] / 255.0f : 0.0f;
    }
  const size_t oplane = (size_t)oHp * oWp;
  float* o = out + (size_t)b * 64 * oplane + (size_t)(y + PADY) * oWp + x + PADX;
  const bool full = x + 3 < W && ((oWp & 3) == 0);   // PADX = 4 and x % 4 == 0: the store is 16-byte aligned
#pragma unroll 2
  for (int co = 0; co < 64; co++) {
    const float bv = bias[co];
    float acc[4] = {bv, bv, bv, bv};
#pragma unroll
    for (int ky = 0; ky < 3; ky++)
#pragma unroll
      for (int kx = 0; kx < 3; kx++) {
        const float wv = w[co * 9 + ky * 3 + kx];
#pragma unroll
        for (int q = 0; q < 4; q++) acc[q] = __builtin_fmaf(p[ky][kx + q], wv, acc[q]);
      }
    f32x4 v;
#pragma unroll
    for (int q = 0; q < 4; q++) v[q] = acc[q] > 0.0f ? acc[q] : 0.0f;
    if (full) *reinterpret_cast<f32x4*>(o + (size_t)co * oplane) = v;
    else
      for (int q = 0; q < 4 && x + q < W; q++) o[(size_t)co * oplane + q] = v[q];
  }
}

void launch_conv1a(const uint8_t* img, int B, int H, int W, int stride, size_t fstride, const float* w,
                   const float* bias, const ActDesc& out, hipStream_t s) {
  dim3 grid((W + 255) / 256, (H + 3) / 4, B);
  hipLaunchKernelGGL(conv1a_kernel, grid, dim3(256), 0, s, img, H, W, stride, fstride, w, bias, out.p, out.Hp, out.Wp);
}

}  // namespace dani
