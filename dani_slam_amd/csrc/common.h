// Internal declarations shared by the HIP kernels and the engine (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

namespace dani {

// ---------------------------------------------------------------------------------------------
// Activation layout in HBM: "padded planar" f32  [B][C][Hp][Wp].
// Pixel (y, x) of channel c lives at row y + PADY, column x + PADX.  The border (row 0, columns
// 0..3, and everything right of W / below H) is zeroed once at allocation and never written, so the
// 3x3 halo loads of the conv kernels need no bounds checks and read the literal zeros of "same"
// padding; PADX = 4 keeps the interior and the halo loads 16-byte aligned.
// ---------------------------------------------------------------------------------------------
constexpr int PADX = 4;
constexpr int PADY = 1;

struct ActDesc {
  float* p = nullptr;
  int C = 0, H = 0, W = 0;  // logical dims
  int Hp = 0, Wp = 0;       // padded plane dims
  int pady = PADY;          // top padding rows (2 for the input of the 6x6 stride-2 YOLO stem)
  size_t frame_stride() const { return (size_t)C * Hp * Wp; }
  size_t bytes(int B) const { return frame_stride() * B * sizeof(float); }
};

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
inline void act_shape(ActDesc& a, int C, int H, int W) {
  a.C = C; a.H = H; a.W = W;
  a.Hp = PADY + round_up(H, 16) + 1;      // tiles are at most 16 rows tall
  a.Wp = PADX + round_up(W, 32) + 4;      // tiles are at most 32 columns wide; +4 right halo
}

// Which conv kernel variant a layer uses (see conv_mfma.hip).
enum ConvVariant {
  CONV3_ROW32_RELU = 0,       // 3x3, wave = 2 rows x 32 cols, WG tile 8x32, 64 couts, ReLU
  CONV3_ROW32_RELU_POOL = 1,  // same + fused 2x2 max-pool
  CONV3_BLK16_RELU = 2,       // 3x3, wave = 4 rows x 16 cols, WG tile 16x16, 64 couts, ReLU (1/8-res layers)
  CONV1_BLK16_LINEAR = 3,     // 1x1, WG tile 16x16, 64 couts, no activation
  // YOLOv5s (SiLU): 6x6 stride-2 stem, 3x3 stride-2, 3x3, 1x1; "ROW32" = 8x32 tiles, "BLK16" = 16x16 tiles
  CONV6S2_ROW32_SILU = 4, CONV3S2_ROW32_SILU = 5, CONV3S2_BLK16_SILU = 6, CONV3_ROW32_SILU = 7,
  CONV3_BLK16_SILU = 8, CONV1_ROW32_SILU = 9, CONV1_BLK16_SILU = 10,
  // small-grid variants: 8x16-pixel x 32-cout workgroups
  CONV3_BLK16S_RELU = 11, CONV1_BLK16S_LINEAR = 12, CONV3_BLK16S_SILU = 13, CONV3S2_BLK16S_SILU = 14, CONV1_BLK16S_SILU = 15,
  // narrow variants (32 couts per workgroup) of the 8x32-tile kernels, for layers with <= 32 output channels
  CONV6S2_ROW32N_SILU = 16, CONV3_ROW32N_SILU = 17, CONV1_ROW32N_SILU = 18,
  // 4 x 32-pixel x 32-cout workgroups (one image row per wave) for the 12 x 20 level of YOLOv5s: three tiles cover a
  // frame's 240 pixels at 62 % (the 8 x 16 tiles need four, 47 %)
  CONV3_R4_SILU = 19, CONV3S2_R4_SILU = 20, CONV1_R4_SILU = 21, CONV1_R4_LINEAR = 22,
  // K-segmented twins (conv_kseg): 2 segments on the 8x16 small tiles (-> 4x16-pixel workgroups), 4 segments on the 4x32
  // tiles (-> one 32-pixel row per workgroup), 2 segments for the long 1x1 layers
  CONV3_BLK16S_SILU_K2 = 23, CONV3S2_BLK16S_SILU_K2 = 24, CONV3_R4_SILU_K4 = 25, CONV3S2_R4_SILU_K2 = 26,
  CONV1_BLK16S_SILU_K2 = 27, CONV1_R4_SILU_K2 = 28, CONV1_R4_LINEAR_K2 = 29,
  // SuperPoint's 1/2- and 1/4-resolution layers at small batch: 32-cout twins of the 8x32 tiles (plain and pooled) and
  // 4x32-pixel tiles -- 2x / 4x the workgroups where one or two frames would leave most CUs with one workgroup or none
  CONV3_ROW32N_RELU = 30, CONV3_ROW32N_RELU_POOL = 31, CONV3_R4_RELU = 32,
  // flat pixel tiles (conv_mfma.hip, Cfg::FW): 32 consecutive pixels of a frame per wave, for YOLOv5s' 24 x 40 (F40) and
  // 12 x 20 (F20) levels; K segments as conv_kseg prescribes
  CONV3_F40_SILU_K2 = 33, CONV3S2_F40_SILU_K2 = 34, CONV1_F40_SILU = 35, CONV1_F40_LINEAR = 36,
  CONV3_F20_SILU_K4 = 37, CONV3S2_F20_SILU_K2 = 38, CONV1_F20_SILU_K2 = 39,
  // SuperPoint's conv1b with conv1a computed inside the kernel from the u8 image (conv_mfma.hip, Cfg::F1A)
  CONV3_ROW32_RELU_POOL_F1A = 40, CONV3_ROW32N_RELU_POOL_F1A = 41,
  // SuperPoint's heads with their post-processing in the epilogue (conv_mfma.hip, Cfg::EPI): convPb + softmax + depth-to-space
  // (96 couts per workgroup), convDb + L2 norm + NHWC (256 couts per workgroup); 8 x 16 cells per workgroup
  CONV1_BLK16S_SOFTMAX_D2S = 42, CONV1_BLK16S_L2NORM_NHWC = 43,
};

// Canonical K segmentation of a YOLOv5s convolution (conv_mfma.hip, Cfg::SEG; oracle/sp_oracle.cpp orc_conv2d_ex follows the
// same rule): a function of the layer's shape only.  3x3 stride 1: 4 segments from 256 input channels, 2 from 128; 3x3
// stride 2: 2 from 128 (four one-row workgroup tiles of a stride-2 layer re-stage three input rows per output row: measured
// slower than the unsplit kernel); 1x1: 2 from 512.
inline int conv_kseg(int cin, int k, int stride) {
  if (k == 3 && stride == 1) return cin >= 256 ? 4 : (cin >= 128 ? 2 : 1);
  if (k == 3) return cin >= 128 ? 2 : 1;
  if (k == 1) return cin >= 512 ? 2 : 1;
  return 1;
}

// tensor views for one conv launch: tensors may be channel slices of larger (concat) buffers
struct ConvIO {
  const float* in = nullptr; float* out = nullptr; const float* res = nullptr;
  int H = 0, W = 0;                      // conv output dims (before fused pooling)
  int iHp = 0, iWp = 0, ipady = PADY, oHp = 0, oWp = 0, opady = PADY;
  size_t in_fs = 0, out_fs = 0, res_fs = 0;
  // Cfg::F1A variants only: the u8 image (device) the layer's input is computed from, and conv1a's raw weights / bias
  const uint8_t* img = nullptr; int img_stride = 0; size_t img_fs = 0;
  const float* w1a = nullptr; const float* b1a = nullptr;
};

struct ConvLayerPlan {
  ConvVariant variant;
  int Cin, Cout, ksize;
  int CinPad = 0;        // Cin rounded up to the chunk size (the YOLO stem pads 3 -> 4 with a zero plane)
  int cout_tiles;        // ceil(Cout / couts-per-workgroup)
  int CC;                // input channels per LDS chunk
  float* w_packed = nullptr;  // device, [cout_tile][chunk][cp][tap][cc][64]
  float* bias_packed = nullptr;  // device, cout_tiles * 64
  size_t w_packed_floats = 0;
};

// host-side packing of OIHW weights into the kernel layout
void pack_conv_weights(const float* w_oihw, const float* bias, int Cin, int Cout, int ksize, int CC, int NCO,
                       float* w_packed, float* bias_packed);
inline int conv_variant_cc(ConvVariant v) {
  return (v == CONV1_BLK16_LINEAR || v == CONV1_ROW32_SILU || v == CONV1_BLK16_SILU || v == CONV1_BLK16S_LINEAR ||
          v == CONV1_BLK16S_SILU || v == CONV1_ROW32N_SILU || v == CONV1_R4_SILU || v == CONV1_R4_LINEAR ||
          v == CONV1_BLK16S_SILU_K2 || v == CONV1_R4_SILU_K2 || v == CONV1_R4_LINEAR_K2 || v == CONV1_F40_SILU ||
          v == CONV1_F40_LINEAR || v == CONV1_F20_SILU_K2 || v == CONV1_BLK16S_SOFTMAX_D2S || v == CONV1_BLK16S_L2NORM_NHWC) ? 32 : ((v == CONV6S2_ROW32_SILU || v == CONV6S2_ROW32N_SILU) ? 4 : 8);
}
inline int conv_variant_kseg(ConvVariant v) {
  switch (v) {
    case CONV3_BLK16S_SILU_K2: case CONV3S2_BLK16S_SILU_K2: case CONV3S2_R4_SILU_K2: case CONV1_BLK16S_SILU_K2: case CONV1_R4_SILU_K2: case CONV1_R4_LINEAR_K2:
    case CONV3_F40_SILU_K2: case CONV3S2_F40_SILU_K2: case CONV3S2_F20_SILU_K2: case CONV1_F20_SILU_K2: return 2;
    case CONV3_R4_SILU_K4: case CONV3_F20_SILU_K4: return 4;
    default: return 1;
  }
}
inline int conv_variant_nco(ConvVariant v) {   // couts per workgroup
  if (v == CONV1_BLK16S_SOFTMAX_D2S) return 96;
  if (v == CONV1_BLK16S_L2NORM_NHWC) return 256;
  return (v >= CONV3_BLK16S_RELU && v != CONV3_ROW32_RELU_POOL_F1A) ? 32 : 64;
}
// the small-grid twin of a 16x16-tile variant (or the variant itself)
inline ConvVariant conv_small_variant(ConvVariant v) {
  switch (v) {
    case CONV3_BLK16_RELU: return CONV3_BLK16S_RELU;
    case CONV1_BLK16_LINEAR: return CONV1_BLK16S_LINEAR;
    case CONV3_BLK16_SILU: return CONV3_BLK16S_SILU;
    case CONV3S2_BLK16_SILU: return CONV3S2_BLK16S_SILU;
    case CONV1_BLK16_SILU: return CONV1_BLK16S_SILU;
    default: return v;
  }
}

// ---- kernel launchers (all asynchronous on `s`) ------------------------------------------------
// conv1a: u8 image -> 64-channel padded planar, fused /255 + 3x3 conv + bias + ReLU (VALU kernel).
void launch_conv1a(const uint8_t* img, int B, int H, int W, int stride, size_t img_frame_stride, const float* w /*64*9*/,
                   const float* bias /*64*/, const ActDesc& out, hipStream_t s);
// false = shape outside the kernel's 32-bit addressing contract: nothing was launched (DANI_ERR_UNSUPPORTED upstream)
bool launch_conv_mfma(const ConvLayerPlan& L, const ActDesc& in, const ActDesc& out, int B, hipStream_t s, int reserve_lds = 0);
// conv1b computing conv1a on the fly (L.variant is an F1A variant): `in` gives the geometry only, the data comes from the image
bool launch_conv_mfma_f1a(const ConvLayerPlan& L, const uint8_t* img, int img_stride, size_t img_fs, const float* w1a, const float* b1a,
                          const ActDesc& in, const ActDesc& out, int B, hipStream_t s, int reserve_lds = 0);
bool launch_conv_io(const ConvLayerPlan& L, const ConvIO& io, int B, hipStream_t s, int reserve_lds = 0);
// SuperPoint's heads finished in the epilogue (CONV1_BLK16S_SOFTMAX_D2S -> prob [B][8Hc][8Wc], CONV1_BLK16S_L2NORM_NHWC -> [B][Hc*Wc][256])
bool launch_conv_head(const ConvLayerPlan& L, const ActDesc& in, float* dst, size_t dst_fs, int B, hipStream_t s, int reserve_lds = 0);
const char* conv_kernel_name(ConvVariant v);

// semi [B][65(+pad)][..] padded planar -> prob [B][PH][PW] dense
void launch_softmax_d2s(const ActDesc& semi, float* prob, int B, hipStream_t s);
// desc raw padded planar [B][256][..] -> normalised NHWC [B][Hc*Wc][256] (+ optional dense NCHW copy for tests)
void launch_desc_norm(const ActDesc& desc, float* desc_nhwc, float* desc_nchw_opt, int B, hipStream_t s);

struct CellRect { int iniX, maxX, iniY, maxY, offX, offY; };
constexpr int CELL_MAX_DIM = 64;  // a cell is at most 64 x 64 pixels (one lane per column)
// per-cell threshold + NMS2 (Q4 semantics) + dynamic-box cull -> per-cell slots + counts
void launch_cell_nms(const float* prob, int PH, int PW, int B, const CellRect* cells, int ncells, float iniTh,
                     float minTh, const int32_t* boxes, const int32_t* box_offsets, int minBorder, float scale,
                     float* slot_xyr, int32_t* slot_count, int slot_stride, hipStream_t s);
// ordered compaction of the per-cell slots -> cand [B][cand_cap][3], cand_count[B]
void launch_cell_compact(const float* slot_xyr, const int32_t* slot_count, int ncells, int slot_stride, int B,
                         float* cand_xyr, int32_t* cand_count, int cand_cap, hipStream_t s);
// bilinear sampling (grid_sampler, align_corners=false) + per-channel norm over keypoints (Q7)
// keypoints [koff[b], koff[b] + nk[b]) of frame b (one pyramid level); max_n = max over b of nk[b]
void launch_descriptors(const float* desc_nhwc, int Hc, int Wc, int PH, int PW, const float* kxy /*[B][cap][2]*/,
                        const int32_t* koff /*[B] device*/, const int32_t* nk /*[B] device*/, int B, int cap, int max_n,
                        float* sampled /*[B][cap][256]*/, float* out /*[B][cap][256]*/, hipStream_t s);
// cv::resize(8UC1, INTER_LINEAR) of B frames (pyramid levels, src/SPextractor.cpp:644)
void launch_resize_u8c1(const uint8_t* src, int B, int sw, int sh, int sstride, size_t sframe, uint8_t* dst, int dw, int dh,
                        hipStream_t s);
void launch_copy_layer_dense(const ActDesc& a, int b, float* out_dense, hipStream_t s);
// L2 descriptor distance + best / second-best scan over candidate lists (CSR)
void launch_bow_descend(const float* feats, int n, const int32_t* child_off, const int32_t* child_ids, const float* node_desc,
                        int nid_level, int32_t* leaf_out, int32_t* nid_out, hipStream_t s);
void launch_desc_mean(const float* rows, const int32_t* off, const int32_t* idx, int nq, float* out, hipStream_t s);
void launch_desc_best2(const float* q, const float* t, const int32_t* t_level, const int32_t* cand_off,
                       const int32_t* cand_idx, int nq, float* best_dist, float* best_dist2, int32_t* best_idx,
                       int32_t* best_level, int32_t* best_level2, float* cand_dist, hipStream_t s);

// SURVEY 8(f) rank 2 on the device (handoff_kernels.hip)
void launch_grid_assign(const float* kxy, int n, float min_x, float min_y, float inv_w, float inv_h, int cols, int rows,
                        int32_t* cell_of, int32_t* cell_offsets, int32_t* cell_idx, hipStream_t s);
void launch_undistort(const float* xy, int n, const float K4[4], const float* dist, int ndist, float* out, hipStream_t s);
void launch_area_count(const float* kxy, const int32_t* octave, float min_x, float min_y, float inv_w, float inv_h, int cols, int rows,
                       const int32_t* cell_offsets, const int32_t* cell_idx, const float* qx, const float* qy, const float* qr,
                       const int32_t* min_level, const int32_t* max_level, int nq, int32_t* counts, int32_t* cand_offsets, hipStream_t s);
void launch_area_fill(const float* kxy, const int32_t* octave, float min_x, float min_y, float inv_w, float inv_h, int cols, int rows,
                      const int32_t* cell_offsets, const int32_t* cell_idx, const float* qx, const float* qy, const float* qr,
                      const int32_t* min_level, const int32_t* max_level, int nq, const int32_t* cand_offsets, int32_t* cand_idx, hipStream_t s);

// ---- LightGlue kernels (lg_kernels.hip) ------------------------------------------------------------
// One keypoint set of a LightGlue batch.  Token layout of the transformer stage (all activations [features][ld]):
//   main region  [tok_off, tok_off + round_up(nmain, 32))   keypoints 0 .. nmain-1 (+ zero padding)
//   tail tokens  [tail_off, tail_off + n - nmain)            keypoints nmain .. n-1, packed after ALL main regions
// A set whose count is a little over a multiple of 32 (n % 32 <= LG_TAIL_MAX, the quadtree's usual overshoot: 1024
// requested -> 1025 returned) keeps only the multiple in its main region, so that one stray keypoint does not cost a
// whole 32-query block / 32-key tile / 128-token GEMM tile per set; tail queries and keys are handled by VALU paths.
// The assignment stage works on a dense copy (dense_off, padded to 32).  kp / desc point at the set's token-major
// inputs ([n][2], [n][256]) so that device-resident SuperPoint outputs are read in place.
constexpr int LG_TAIL_MAX = 4;
struct LgSet {
  int tok_off, n, nmain, tail_off;
  int partner_off, partner_n, partner_nmain, partner_tail_off;
  int dense_off;
  float size_w, size_h;
  const float* kp; const float* desc;
};
// bf16x3 variant (lg_x3.hip): same function and layout, contraction on the bf16 matrix cores at f32 accuracy.  w_x3 is
// the weight split by pack_linear_x3 (x3_packed_u16(out, in) 16-bit words); OUT, K0, K1 multiples of 64.
bool launch_linear_x3(const float* x0, int K0, const float* x1, int K1, const void* w_x3, const float* bias_packed,
                      float* y, int OUT, int T, int ld, const float* res, const float* cosb, const float* sinb,
                      int rot_rows, float alpha, hipStream_t s);
size_t x3_packed_u16(int out, int in);
void pack_linear_x3(const float* w_row_major, int out, int in, uint16_t* dst);
// w_rs: the same weights in pack_linear_rs layout (linear_rs_kernel, lin_rs.h); nullptr = linear_fm_kernel only
bool launch_linear_fm(const float* x0, int K0, const float* x1, int K1, const float* w_packed, const float* bias_packed,
                      float* y, int OUT, int T, int ld, const float* res, const float* cosb, const float* sinb,
                      int rot_rows, float alpha, hipStream_t s, const float* w_rs = nullptr);
void pack_linear_rs_host(const float* w_row_major, int out, int in, float* dst);   // [out][in] -> lin_rs.h layout (out * in floats)
void launch_ln_gelu(float* hbuf, const float* gamma, const float* beta, int T, int ld, hipStream_t s);
// ffn_chain.h: ffn.0 -> LayerNorm + GELU -> ffn.3 (+ residual) -> the next block's projection for the 16-token blocks [0, nblocks) in
// one launch, activations resident in registers.  p0 / p3 / pd: pack_chain_weights_host with role 1 / 2 / 3; biases unpadded.
// pd = nullptr: no projection.  false = shape outside the kernel's contract (nothing launched).
bool launch_ffn_chain_fm(float* x, const float* ctx, const float* p0, const float* b0, const float* gamma, const float* beta,
                         const float* p3, const float* b3, const float* pd, const float* bd, int OUTD, float* y, const float* cosb,
                         const float* sinb, int rot_rows, float alpha, int nblocks, int ld, hipStream_t s);
constexpr int LG_CHAIN_SLACK_FLOATS = 16 * 256;   // zero floats behind every chain-packed matrix (= ffn_chain.h FC_SLACK_FLOATS)
void pack_chain_weights_host(const float* w_row_major, int out, int in, int role, float* dst);   // role 1 / 2 / 3, see lg_engine.cpp lg_pack_raw
void launch_posenc(const LgSet* sets, int nsets, int max_span, const float* wr, float* cosb, float* sinb, int ld, hipStream_t s);
void launch_desc_to_fm(const LgSet* sets, int nsets, int max_span, float* x, int ld, hipStream_t s);
void launch_attention(const float* Q, const float* K, const float* V, float* O, int ld, const LgSet* sets, int nsets,
                      int max_mp, int max_tail, int cross, float scale, hipStream_t s, int x3 = 0);   // x3: attn_x3_kernel (bf16x3 split contractions)
// matchability logit -> logsigmoid and the dense copy of md (set s at [dense_off, dense_off + n)) in one launch
void launch_match_densify(const float* x, const float* w, float b, const float* md, const LgSet* sets, int nsets, int max_n, float* md_dense,
                          float* logsig_dense, int ld, hipStream_t s);
void launch_sim(const float* md, int ld, const LgSet* sets, int npairs, int max_np, float* sim, int sim_ld,
                size_t sim_pair_stride, hipStream_t s);
void launch_assignment_stats(const float* sim, int sim_ld, size_t pstride, const LgSet* sets, int npairs, int max_n,
                             const float* logsig, float* lse_row, float* lse_col, int* m0, float* max0, int* m1, int ld,
                             hipStream_t s);

// ---- YOLO kernels (yolo_kernels.hip) ---------------------------------------------------------------
void launch_yolo_preproc(const uint8_t* bgr, int B, int sw, int sh, int sstride, size_t sframe, const ActDesc& out,
                         hipStream_t s);
void launch_sppf_pool(float* cat, int C, int H, int W, int Hp, int Wp, size_t fs, int B, hipStream_t s);
void launch_upsample2(const float* src, int sHp, int sWp, size_t sfs, float* dst, int C, int Hd, int Wd, int dHp, int dWp,
                      size_t dfs, int B, hipStream_t s);
void launch_yolo_decode(const ActDesc* raws[3], float score_thresh, float* cand, float* cand_fast, int fast_cap, int* cand_count, int cand_cap, int B,
                        hipStream_t s);
// host: greedy class-agnostic NMS of the reference (src/YoloDetect.cpp:95-144) on candidate records of 8 floats
int yolo_nms_host(const float* cand8, int n, float iou_thresh, float* dets6, int cap);
// host: Detect()'s box bookkeeping + RefineDynamicBoxes (src/YoloDetect.cpp:51-80, 148-208)
int yolo_dynamic_areas(const float* dets6, int n, int cols, int rows, const char* const* class_names, int n_classes,
                       const char* const* dynamic_names, int n_dynamic, int32_t* det_rects, int32_t* dyn_boxes, int cap);

// ---- host logic --------------------------------------------------------------------------------
struct HostKey { float x, y, r; };
// DistributeOctTree (reference src/SPextractor.cpp:405-629); returns indices into `keys`, in list order.
// false: the region is one the reference cannot process either (round(width / height) < 1, or a key outside it)
bool distribute_octtree(const HostKey* keys, int n, int minX, int maxX, int minY, int maxY, int N,
                        std::vector<int>& out_idx);
int cell_grid(int width, int height, std::vector<CellRect>& cells);
// hand-off formats (host_logic.cpp)
void assign_features_to_grid(const float* kxy, int n, float min_x, float min_y, float inv_w, float inv_h, int cols, int rows,
                             int32_t* cell_offsets, int32_t* cell_idx);
void features_in_area(const float* kxy, const int32_t* octave, float min_x, float min_y, float inv_w, float inv_h, int cols, int rows,
                      const int32_t* cell_offsets, const int32_t* cell_idx, float x, float y, float r, int min_level, int max_level,
                      std::vector<int32_t>& out);
void undistort_keypoints(const float* xy, int n, const float K[4], const float* dist, int ndist, float* out);
void stereo_from_matches(const float* ul_x, int nl, const float* ur_x, int nr, const int32_t* matches, const float* scores, int k,
                         float mb, float mbf, float* u_right, float* depth);
void features_per_level(int nfeatures, float scaleFactor, int nlevels, std::vector<int>& per_level,
                        std::vector<float>& scales);

}  // namespace dani
