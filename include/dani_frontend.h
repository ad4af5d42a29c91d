/*
 * dani_frontend.h -- C ABI of the MI355X-native front-end perception engine.
 *
 * This is the drop-in boundary for the reference's per-frame front-end.  The
 * reference has no FFI layer: its boundary is three C++ classes built on LibTorch.  Each entry point
 * below names the reference interface it replaces (paths relative to /root/reference); the C++17
 * shims in include/dani_shims.hpp keep those class and method names on top of this ABI, and
 * INTEGRATION.md shows the lines a maintainer changes.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / OpenCV / HIP types cross this boundary;
 *   - every function returns 0 (DANI_OK) or a negative dani_status; dani_last_error() gives text;
 *     no C++ exception crosses the ABI;
 *   - the caller owns and allocates all outputs; a too-small capacity returns DANI_ERR_CAPACITY and
 *     writes the needed count to *n_out;
 *   - an engine is bound to one GPU and may be used from several host threads at once (the reference
 *     runs two extractors concurrently, src/Frame.cc:124-127): weights are shared read-only, every
 *     call takes a private stream + workspace from the engine's pool;
 *   - there is NO CPU fallback: without a usable MI355X, dani_engine_create fails with
 *     DANI_ERR_NO_DEVICE.
 */
#ifndef DANI_FRONTEND_H
#define DANI_FRONTEND_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DANI_API __attribute__((visibility("default")))

typedef enum dani_status {
  DANI_OK = 0,
  DANI_ERR_INVALID = -1,     /* bad argument */
  DANI_ERR_NO_DEVICE = -2,   /* no HIP device / wrong architecture */
  DANI_ERR_HIP = -3,         /* a HIP runtime call failed */
  DANI_ERR_WEIGHTS = -4,     /* missing / mis-shaped weight tensor (reference: std::exit, src/SPextractor.cpp:88-91) */
  DANI_ERR_CAPACITY = -5,    /* output buffer too small, *n_out holds the needed count */
  DANI_ERR_UNSUPPORTED = -6, /* configuration not implemented (e.g. a pyramid level smaller than 80x80) */
  DANI_ERR_EMPTY = -7        /* empty image (reference: operator() returns -1, src/SPextractor.cpp:135-136;
                                Detect() returns false, src/YoloDetect.cpp:32-36) */
} dani_status;

typedef struct dani_engine dani_engine; /* opaque */

typedef struct dani_config {
  int32_t device;      /* HIP device ordinal */
  int32_t max_batch;   /* largest B passed to the *_batch calls (>= 1) */
  int32_t max_width;   /* largest image width  (e.g. 640) */
  int32_t max_height;  /* largest image height (e.g. 480) */
  int32_t n_contexts;  /* stream+workspace pool size = max concurrent calls (0 -> 2) */
} dani_config;

/* ---- life cycle -------------------------------------------------------------------------- */

/* Replaces: SPextractor::SPextractor's torch::load (src/SPextractor.cpp:80-91), LightGlue::LightGlue's
 * jit::load (src/LightGlue.cpp:7-12, paid per construction in the reference, quirk Q12) and
 * YoloDetection::YoloDetection's jit::load (src/YoloDetect.cpp:6-20).  One engine per process and GPU;
 * constructing the C++ shims afterwards is O(1). */
DANI_API int dani_engine_create(const dani_config* cfg, dani_engine** out);
DANI_API void dani_engine_destroy(dani_engine* e);
DANI_API const char* dani_last_error(const dani_engine* e); /* e may be NULL: last create() error */

/* Upload one named f32 tensor.  SuperPoint names/layouts are the reference's own parameter names
 * (src/SuperPoint.cpp:37-52): "conv1a.weight" [64,1,3,3] ... "convDb.bias" [256], OIHW, i.e. what
 * torch::save(model) holds.  tools/convert_weights.py turns a torch archive into these calls / a .dsw
 * file.  `count` is the number of floats and is checked against the expected shape. */
DANI_API int dani_set_tensor(dani_engine* e, const char* name, const float* data, int64_t count);
/* Load every tensor of a .dsw container (tools/convert_weights.py). */
DANI_API int dani_load_weights_file(dani_engine* e, const char* path);
/* Host-only (no GPU, no engine): check a .dsw container against the tensor names and shapes the engine expects --
 * SuperPoint's 24 tensors are required (the reference exits when torch::load fails, src/SPextractor.cpp:86-91); the
 * LightGlue group (public checkpoint names, src/LightGlue.cpp:9) and the YOLOv5s group (fused public names,
 * src/YoloDetect.cpp:9) must each be absent or complete.  Returns DANI_OK or DANI_ERR_WEIGHTS with `msg` naming the first
 * missing / mis-shaped tensor; *groups (optional): bit 0 SuperPoint, bit 1 LightGlue, bit 2 YOLOv5s present.  The same
 * check runs inside dani_finalize_weights.  tools/convert_weights.py calls it on every archive it converts. */
DANI_API int dani_check_weights_file(const char* path, int32_t* groups, char* msg, int32_t msg_cap);
/* Called once after the last dani_set_tensor: packs weights into the kernels' layouts. */
DANI_API int dani_finalize_weights(dani_engine* e);

/* ---- SuperPoint extraction ---------------------------------------------------------------- */

/* The reference's whole extractor configuration (ORBextractor.* YAML keys, src/Tracking.cc:590-602;
 * iniThFAST/minThFAST are reinterpreted as SuperPoint probabilities, src/SPextractor.cpp:271,291). */
typedef struct dani_sp_params {
  int32_t nfeatures;
  float scale_factor;
  int32_t nlevels;
  float ini_th;
  float min_th;
} dani_sp_params;

/* Replaces SPextractor::operator() (src/SPextractor.cpp:130-217) incl. ComputePyramid,
 * ComputeKeyPointsOctTree, the per-cell getKeyPoints/NMS2 loop, the dynamic-box cull, DistributeOctTree
 * and computeDescriptors.
 *   img/stride : 8-bit gray, `stride` bytes per row (a non-contiguous cv::Mat ROI is fine)
 *   boxes      : nboxes x (x, y, w, h) int32 = SPextractor::mvDynamicArea (src/SPextractor.h:69)
 *   lapping    : vLappingArea[2]
 *   kpts_xy    : cap x 2 (x, y) in level-0 pixel coordinates; resp: cap; octave: cap; size: cap
 *   desc       : cap x 256 f32, row i belongs to keypoint i BEFORE the mono/stereo partition (the
 *                reference does not permute descriptors with the keypoints, src/SPextractor.cpp:159 vs
 *                :202-210 -- reproduced)
 *   *n_out     : number of keypoints; *mono_index: operator()'s return value. */
DANI_API int dani_sp_extract(dani_engine* e, const uint8_t* img, int32_t width, int32_t height, int32_t stride,
                             const dani_sp_params* p, const int32_t* boxes, int32_t nboxes, const int32_t lapping[2],
                             float* kpts_xy, float* resp, int32_t* octave, float* size, float* desc, int32_t cap,
                             int32_t* n_out, int32_t* mono_index);

/* Batched variant: B images of identical size.  `imgs` is a HOST pointer to B*height*stride bytes, or,
 * with imgs_on_device != 0, a DEVICE pointer (frames already resident in HBM).  boxes are per frame:
 * box_offsets has B+1 entries indexing into boxes.  Outputs are B slabs of `cap` entries each; n_out and
 * mono_index have B entries. */
DANI_API int dani_sp_extract_batch(dani_engine* e, const uint8_t* imgs, int32_t imgs_on_device, int32_t B,
                                   int32_t width, int32_t height, int32_t stride, const dani_sp_params* p,
                                   const int32_t* boxes, const int32_t* box_offsets, const int32_t lapping[2],
                                   float* kpts_xy, float* resp, int32_t* octave, float* size, float* desc,
                                   int32_t cap, int32_t* n_out, int32_t* mono_index);

/* Test / parity hooks: intermediate results of the LAST dani_sp_extract* call on this thread's context
 * (frame b).  prob: height8 x width8 f32 (SPDetector::mProb, src/SuperPoint.cpp:206); desc_map: 256 x
 * height/8 x width/8 f32 NCHW (mDesc, :207); cand: the ordered vToDistributeKeys list (x, y, response)
 * in border-relative coordinates (src/SPextractor.cpp:306-312).  Any pointer may be NULL. */
DANI_API int dani_sp_debug_fetch(dani_engine* e, int32_t b, float* prob, float* desc_map, float* cand_xyr,
                                 int32_t cand_cap, int32_t* n_cand);
/* Copy one layer's activation (post-ReLU, pre-pool; layer index 0..11 in forward order) of frame b of the
 * last call to host as dense [C][H][W].  Layers 1, 3 and 5 (conv1b/2b/3b) are returned AFTER their fused
 * 2x2 max-pool (the pre-pool tensor never exists in HBM). */
DANI_API int dani_sp_debug_layer(dani_engine* e, int32_t b, int32_t layer, float* out, int64_t count);
/* Which kernel instantiation SuperPoint layer `layer` (0 = conv1a ... 11 = convDb, the order of dani_layer_timings) runs as on
 * this engine -- chosen at dani_finalize_weights from max_batch and the frame size: conv1a inside conv1b (src/SuperPoint.cpp:58-59),
 * small-batch tile twins, the heads' softmax / L2 norm in their last layer's epilogue (:79-93).  Layers 9 / 11 whose epilogue
 * finishes the map report "...+softmax_d2s" / "...+l2norm_nhwc"; the debug hook above then re-runs the plain layer on demand. */
DANI_API int dani_sp_layer_kernel(dani_engine* e, int32_t layer, char* name, int32_t cap);

/* Profiling switch: when on, HIP events are recorded around every conv layer of the next calls (on the
 * context's own stream) and dani_layer_timings() returns the 12 per-layer durations in ms (forward order). */
DANI_API int dani_set_profiling(dani_engine* e, int32_t on);
DANI_API int dani_layer_timings(dani_engine* e, float* ms12);

/* Per-stage device timings (ms) of the last call on this thread's context, measured with HIP events on
 * the context's stream: [0] conv stack, [1] softmax+norm+cells, [2] host quadtree, [3] descriptors,
 * [4] total.  Also used by bench.py for the roofline figure of the dominant kernel. */
DANI_API int dani_last_timings(dani_engine* e, float* ms, int32_t n);

/* ---- LightGlue matching -------------------------------------------------------------------- */

/* Arithmetic of LightGlue's matrix products (linear layers: the dominant kernel of the front-end; attention: the
 * second).  All modes compute the same function to f32 accuracy and meet the 1e-3 score tolerance with the same margin
 * (tests/test_lg_gpu.py):
 *   DANI_LG_GEMM_F32         (default) f32 MFMA, accumulation in the canonical order of the oracle.
 *   DANI_LG_GEMM_BF16X3      linear layers: every f32 operand split into three bf16 terms, six bf16 MFMAs per product
 *                            block, f32 accumulation (error against an f64 contraction within 2x of the f32 mode's);
 *                            the additions are NOT in the canonical order.
 *   DANI_LG_GEMM_BF16X3_ALL  the same for the attention contractions (K Q^T and V^T P^T) as well; softmax stays f32.
 * Applies to the next dani_lg_match* / dani_frontend_batch calls of this engine.  SuperPoint and YOLO are not
 * affected (they are bit-exact against the oracle and stay on the f32 path). */
#define DANI_LG_GEMM_F32 0
#define DANI_LG_GEMM_BF16X3 1
#define DANI_LG_GEMM_BF16X3_ALL 2
DANI_API int dani_lg_set_gemm(dani_engine* e, int32_t mode);
/* Kernel selection of the f32 LightGlue path, process-wide, for A/B tests and measurements (results agree within the
 * tolerances of DESIGN.md; defaults = the shipped choice).  name:
 *   "lin_rs"    -1 = per shape (default), 0 = LDS-tiled linear_fm_kernel only, 1..3 = linear_rs_kernel with that tile,
 *                4 = every token through linear_rs_kernel's K-split 32 x 32 blocks (the form its tail tokens take)
 *   "attn"       0 = per shape (default), 2 = LDS-tiled attn_kernel<4>, 6 / 7 / 8 = attn_rs_kernel with a ring of 8 / 4 / 16 loads
 *   "ffn_chain" -1 = per shape (default), 0 = never, 1 = always: ffn.0 -> LayerNorm + GELU -> ffn.3 -> next projection as ONE
 *                register-resident launch per block (ffn_chain_kernel) instead of four
 * Returns DANI_ERR_INVALID for an unknown name or any other value (the probe-only kernels are not selectable here). */
DANI_API int dani_lg_set_kernel_option(const char* name, int32_t value);
/* How the calling thread waits for the GPU inside every entry point, process-wide (environment: DANI_HOST_WAIT=block|spin|hybrid):
 *   0 = hipStreamSynchronize (which busy-waits on this ROCm build -- one core for the caller and most of one for a runtime thread --
 *       and still comes back a scheduler tick late now and then on a loaded host);
 *   1 = spin on hipStreamQuery (default: the same CPU cost, no late wake-ups);
 *   2 = hybrid: sleep in 40 us naps until a marker event the engine records a few hundred microseconds of GPU work ahead of the
 *       point it will wait for, then spin for the rest; waits without a marker spin for at most 150 us, then use mode 0.  About one
 *       core instead of 2.3 per calling thread at the same median latency; a sleeping thread can be scheduled late.
 * dani_get_host_wait returns the mode in force. */
DANI_API int dani_set_host_wait(int32_t mode);
DANI_API int dani_get_host_wait(void);

/* Replaces LightGlue::matchDescriptors + normalizeKeypoints + filterMatches (src/LightGlue.cpp:18-122), as
 * called from Frame::ComputeStereoMatches (src/Frame.cc:860), LGmatcher::SearchByBoW (src/LGmatcher.cpp:314) and
 * LGmatcher::SearchForTriangulation (src/LGmatcher.cpp:983).  The torch::Tensor arguments of the reference become
 * pointer + count (north_star: no LibTorch at run time).
 *   kpts0/1 : n x 2 (x, y) = cv::KeyPoint::pt;  desc0/1 : n x 256 f32 (rows of the extractor's descriptor Mat)
 *   size0/1 : (w, h) as the reference's size1/size2 tensors, or NULL -> 1 + max - min of the keypoints
 *             (the size.numel()==0 branch, src/LightGlue.cpp:34-36)
 *   threshold : filterMatches threshold (default 0.05 in include/LightGlue.h:23)
 *   matches : cap x 2 int32 (queryIdx, trainIdx);  scores : cap f32 (= cv::DMatch::distance);  *k_out : count.
 * Weight tensors use the public LightGlue checkpoint names (posenc.Wr.weight, transformers.{i}.self_attn.Wqkv.weight,
 * ..., log_assignment.{i}.final_proj.weight) through dani_set_tensor / the .dsw file. */
DANI_API int dani_lg_match(dani_engine* e, const float* kpts0, const float* desc0, int32_t n0, const float* kpts1,
                           const float* desc1, int32_t n1, const float* size0, const float* size1, float threshold,
                           int32_t* matches, float* scores, int32_t cap, int32_t* k_out);

/* Batched variant: P independent pairs.  Inputs are P slabs of `in_cap` keypoints (the layout
 * dani_sp_extract_batch writes: kpts [P][in_cap][2], desc [P][in_cap][256]); n0/n1 have P entries.  With
 * on_device != 0 the kpts/desc pointers are DEVICE pointers.  size0/size1: (w, h) applied to every pair, or NULL.
 * Outputs: matches [P][cap][2], scores [P][cap], k_out [P]. */
DANI_API int dani_lg_match_batch(dani_engine* e, int32_t P, int32_t in_cap, int32_t on_device, const float* kpts0,
                                 const float* desc0, const int32_t* n0, const float* kpts1, const float* desc1,
                                 const int32_t* n1, const float* size0, const float* size1, float threshold,
                                 int32_t* matches, float* scores, int32_t cap, int32_t* k_out);

/* Grouped variant for the per-vocabulary-node call shapes: G independent LightGlue problems over
 * SUBSETS of two frames' features in ONE forward pass.  Replaces
 *   - LGmatcher::SearchForTriangulation's loop, one matchDescriptors call per shared vocabulary node on the few
 *     keypoints of that node without a map point (src/LGmatcher.cpp:905-983: hundreds of calls of 3-40 keypoints per
 *     keyframe pair, each paying ~170 kernel launches), and
 *   - the row-by-row gathers of LGmatcher::SearchByBoW (cv::Mat::push_back per keypoint, src/LGmatcher.cpp:257-288;
 *     one group holding every gathered keypoint reproduces its single call at :314).
 * Frame f in {0, 1}: kpts_f [n_f][2], desc_f [n_f][256] (host pointers: the keyframe's mvKeysUn / mDescriptors).
 * Group g matches rows idx0[off0[g] .. off0[g+1]) of frame 0 (queries, in that order) against rows
 * idx1[off1[g] .. off1[g+1]) of frame 1 -- exactly the kpsKF1 / descKF1 and kpsKF2 / descKF2 the reference builds per
 * node.  size0 / size1 / threshold as dani_lg_match.  Outputs, CSR over the groups: match_offsets [G+1];
 * matches [cap][2] = (queryIdx, trainIdx) LOCAL to the group's lists (the reference indexes f1it->second[m.queryIdx]);
 * scores [cap].  A group with an empty side yields no matches (the reference skips the call, :970).  Returns
 * DANI_ERR_CAPACITY with match_offsets[G] = needed count when cap is too small. */
DANI_API int dani_lg_match_groups(dani_engine* e, const float* kpts0, const float* desc0, int32_t n0, const float* kpts1,
                                  const float* desc1, int32_t n1, int32_t G, const int32_t* off0, const int32_t* idx0,
                                  const int32_t* off1, const int32_t* idx1, const float* size0, const float* size1,
                                  float threshold, int32_t* matches, float* scores, int32_t* match_offsets, int32_t cap);

/* With dani_set_profiling(e, 1): summed duration (HIP events on the engine's stream) and algorithmic FLOP of the 18
 * attention-kernel launches of the last LightGlue call on this thread (bench.py roofline of the dominant kernel). */
DANI_API int dani_lg_attention_profile(dani_engine* e, float* total_ms, double* total_flop, int32_t* launches);
/* The same with both FLOP counts: `executed` as above (a cross launch forms K Q^T once per direction: 4 N M d MAC per pair and layer),
 * `algorithmic` = SURVEY.md section 8d's count, in which a pair's cross block needs its score matrix once (3 N M d MAC).  bench.py
 * quotes roofline fractions on the algorithmic count and reports the executed one beside it. */
DANI_API int dani_lg_attention_profile_ex(dani_engine* e, float* total_ms, double* executed_flop, double* algorithmic_flop, int32_t* launches);
/* same for the feature-major linear layers of the forward pass (55 launches: 6 per layer + final_proj) */
DANI_API int dani_lg_linear_profile(dani_engine* e, float* total_ms, double* total_flop, int32_t* launches);

/* Test hook: final descriptors (after the 9 layers) of keypoint set `set` (2*pair + side) of the last LightGlue
 * call on this thread, as [n][256] f32, and the raw row results (best column and log-score per query row). */
DANI_API int dani_lg_debug_fetch(dani_engine* e, int32_t set, float* desc_out, int32_t* m0, float* max0, int32_t cap);

/* ---- YOLO dynamic-object detector ---------------------------------------------------------- */

/* Replaces the GPU-worthy part of YoloDetection::Detect (src/YoloDetect.cpp:27-82): resize to 640x380 +
 * BGR->RGB + /255 (:39-46), the YOLOv5s forward (:49, absent TorchScript blob -> public v6 architecture) and
 * non_max_suppression (:84-146, score_thresh 0.4 / iou_thresh 0.5 at the call site :50).
 *   bgr/stride : 8-bit BGR image (cv::Mat mRGB), `stride` bytes per row
 *   dets6      : cap x 6 (x1, y1, x2, y2, score, class) in the network's 640x380 input coordinates, in the
 *                reference's keep order;  *n_out : count
 * Weight tensors: public fused YOLOv5s names (model.0.conv.weight/.bias ... model.24.m.2.weight/.bias). */
DANI_API int dani_yolo_detect(dani_engine* e, const uint8_t* bgr, int32_t width, int32_t height, int32_t stride,
                              float score_thresh, float iou_thresh, float* dets6, int32_t cap, int32_t* n_out);
/* Batched: B images of identical size, host or (imgs_on_device != 0) device pointer; outputs B slabs of cap. */
DANI_API int dani_yolo_detect_batch(dani_engine* e, const uint8_t* bgr, int32_t imgs_on_device, int32_t B, int32_t width,
                                    int32_t height, int32_t stride, float score_thresh, float iou_thresh, float* dets6,
                                    int32_t cap, int32_t* n_out);
/* The rest of Detect() (:51-80) + RefineDynamicBoxes (:148-208), pure host integer logic: boxes rescaled by
 * cols/640 and rows/384 (quirk Q10), truncated to int rects, split by class name into static / dynamic
 * (mvDynamicNames), placeholder (1,1,1,1) when detections exist but none is dynamic, static-box subtraction in
 * std::map (class-name) order.  det_rects: n x 4 (may be NULL); dyn_boxes: cap x 4 = mvDynamicArea.
 * Returns the number of dynamic boxes (or a negative status). */
DANI_API int dani_host_dynamic_areas(const float* dets6, int32_t n, int32_t cols, int32_t rows,
                                     const char* const* class_names, int32_t n_classes,
                                     const char* const* dynamic_names, int32_t n_dynamic, int32_t* det_rects,
                                     int32_t* dyn_boxes, int32_t cap);
/* Test hook: raw Detect-layer output [255][ny][nx] of level 0..2 for frame b of the last YOLO call. */
DANI_API int dani_yolo_debug_raw(dani_engine* e, int32_t b, int32_t level, float* out, int64_t count);

/* ---- fused per-frame front-end (sequence mode) ---------------------------------------------- */

/* The class list the reference reads from `coco.names` in the CWD (src/YoloDetect.cpp:12-16) and its dynamic class
 * names (mvDynamicNames, :17-18); needed by dani_frontend_batch to turn detections into mvDynamicArea. */
DANI_API int dani_yolo_set_classes(dani_engine* e, const char* const* class_names, int32_t n_classes,
                                   const char* const* dynamic_names, int32_t n_dynamic);

typedef struct dani_frontend_params {
  dani_sp_params sp;
  int32_t use_yolo;        /* 1: RGB-D path (Tracking::GrabImageRGBD, src/Tracking.cc:1534-1545): YOLO boxes cull SuperPoint */
  float yolo_score_thresh; /* 0.4 at the reference's call site (src/YoloDetect.cpp:50) */
  float yolo_iou_thresh;   /* 0.5 */
  int32_t match_previous;  /* 1: LightGlue-match every frame against its predecessor in the sequence */
  float lg_threshold;      /* 0.05 (include/LightGlue.h:23) */
  int32_t lapping[2];      /* vLappingArea */
  int32_t pair_mode;       /* with match_previous: 0 = sequence (frame b against frame b-1, frame 0 against the previous call's
                              last frame); 1 = stereo: frames (2k, 2k+1) of the call are the left / right images of one
                              timestamp and LightGlue matches left against right (Frame::Frame's two ExtractORB calls +
                              Frame::ComputeStereoMatches, src/Frame.cc:124-143, :813-860); B must be even, nothing is
                              cached across calls; matches / scores / k_out of pair k are delivered in the slots of frame
                              2k+1 as (index in left, index in right), the slots of frame 2k report k_out = 0 */
  int32_t defer_desc;      /* split form only (match_previous = 0 followed by dani_frontend_match_resident, the N > 1 sharding of
                              BASELINE config #4): 1 = the call returns as soon as keypoints and counts are on the host; the
                              descriptor read-back into `desc` keeps running and is completed by dani_frontend_match_resident
                              (under its LightGlue pass), by dani_frontend_reset or by the next dani_frontend_batch of the
                              sequence.  `desc` must stay valid and unread until then.  0 = `desc` is complete on return */
} dani_frontend_params;

/* `sequence` selects one of any number of independent image sequences (cameras): each has its own stream, workspace
 * and cached predecessor frame, so calls for DIFFERENT sequences may run concurrently from different host threads
 * (their kernels interleave on the GPU and fill each other's tail / occupancy gaps).
 * One call = the whole per-frame front-end of the reference for B consecutive frames of ONE sequence, chained on the
 * device: YoloDetection::Detect -> mvDynamicArea hand-off (src/Tracking.cc:1535-1539) -> SPextractor::operator()
 * -> LightGlue::matchDescriptors(previous frame, this frame).  Frame 0 of the call is matched against the last
 * frame of the previous call (features stay resident in HBM); the very first frame has no predecessor (k_out = 0).
 *   gray : B x height x gray_stride u8 (what Tracking passes to the Frame constructor after cvtColor)
 *   bgr  : B x height x bgr_stride u8, 3 channels (what Tracking passes to YoloDetection::GetImage); NULL if !use_yolo
 *   imgs_on_device : both image pointers are device pointers (frames already resident in HBM)
 * Outputs (host): keypoints / responses / descriptors as dani_sp_extract_batch (desc may be NULL: descriptors then
 * stay on the device only), dyn_boxes [B][box_cap][4] + n_boxes [B] (= mvDynamicArea per frame),
 * matches [B][match_cap][2] = (index in previous frame, index in this frame), scores, k_out [B]. */
DANI_API int dani_frontend_batch(dani_engine* e, int32_t sequence, const uint8_t* gray, int32_t gray_stride, const uint8_t* bgr,
                                 int32_t bgr_stride, int32_t imgs_on_device, int32_t B, int32_t width, int32_t height,
                                 const dani_frontend_params* p, float* kpts_xy, float* resp, float* desc, int32_t cap,
                                 int32_t* n_out, int32_t* mono_index, int32_t* dyn_boxes, int32_t box_cap,
                                 int32_t* n_boxes, int32_t* matches, float* scores, int32_t match_cap, int32_t* k_out);
/* Split form of the fused call for batches that are sharded over several GPUs (BASELINE config #4: ONE 64-frame batch,
 * 8 frames per GPU): the pair (last frame of rank r-1, first frame of rank r) needs features that live on another GPU.
 *   1. dani_frontend_batch with match_previous = 0 extracts and keeps the B frames' features resident; the cached
 *      predecessor of the sequence is left as it is;
 *   2. dani_frontend_export_features packs one resident frame (normally the last) into a caller-owned slab --
 *      dst[0] = count, then kpts [cap][2], then desc [cap][256] (cap * 258 + 1 floats; host pointer, or DEVICE pointer with
 *      dst_on_device != 0: the buffer the RCCL all-gather runs on, dani_slam_amd/sharding.py);
 *   3. dani_frontend_match_resident runs LightGlue over the resident frames -- frame b against frame b-1, frame 0 against
 *      `pred` (a slab in the layout of step 2, from the neighbouring rank; NULL = the sequence's own cached predecessor, as
 *      the fused call uses) -- and then makes the last resident frame the sequence's predecessor.
 * Outputs of step 3 as the match outputs of dani_frontend_batch.  Steps 1 + 3 with pred = NULL compute exactly what one
 * fused call with match_previous = 1 computes. */
DANI_API int dani_frontend_export_features(dani_engine* e, int32_t sequence, int32_t frame, float* dst, int32_t dst_on_device,
                                           int32_t cap, int32_t* n_out);
DANI_API int dani_frontend_match_resident(dani_engine* e, int32_t sequence, const float* pred, int32_t pred_on_device, int32_t pred_cap,
                                          float lg_threshold, int32_t* matches, float* scores, int32_t match_cap, int32_t* k_out);
/* ---- the sharded batch's one collective, native: RCCL all-gather over xGMI (no reference counterpart) ---------
 *
 * One engine per process and GPU, `world` processes on one node.  The pair (last frame of rank r-1, first frame of rank r) of a
 * batch of consecutive frames sharded over the GPUs needs features that live on another GPU (LightGlue pair matching as at
 * src/Frame.cc:813-860 / src/LGmatcher.cpp:314):
 *   once:      rank 0: dani_comm_unique_id(id)  -> the 128 bytes reach the other ranks by any means (MPI, a file, a socket,
 *              torch.distributed's store ...) -> every rank: dani_comm_create(engine, id, 128, rank, world)    [collective]
 *   per batch: dani_frontend_batch(match_previous = 0)                       YOLO + SuperPoint, features stay in HBM
 *              dani_frontend_exchange_boundary(seq, B - 1, cap, NULL, ..)    pack kernel -> ncclAllGather on the sequence's stream
 *              dani_frontend_match_exchanged(seq, ...)                       LightGlue, frame 0 against rank r-1's slab (rank 0:
 *                                                                            the last rank's slab of the previous batch)
 * Nothing is synchronised with the host between the three calls: the collective and its consumer are ordered by the sequence's
 * stream.  librccl is loaded (dlopen) when the first id / communicator is created; DANI_ERR_UNSUPPORTED if it cannot be. */
#define DANI_COMM_ID_BYTES 128
DANI_API int dani_comm_available(void);                                          /* DANI_OK when librccl loads with every symbol the engine uses (no communicator, no id): lets every rank check BEFORE the collective calls below */
DANI_API int dani_comm_unique_id(void* id, int32_t id_bytes);                    /* ncclGetUniqueId; id_bytes >= DANI_COMM_ID_BYTES */
DANI_API int dani_comm_create(dani_engine* e, const void* id, int32_t id_bytes, int32_t rank, int32_t world);   /* ncclCommInitRank on the engine's GPU */
DANI_API int dani_comm_destroy(dani_engine* e);                                  /* also done by dani_engine_destroy */
/* rank / world / RCCL version code / number of all-gathers enqueued so far (any pointer may be NULL) */
DANI_API int dani_comm_info(dani_engine* e, int32_t* rank, int32_t* world, int32_t* rccl_version, int64_t* collectives);
/* Pack resident frame `frame` of `sequence` (slab layout of dani_frontend_export_features: count | kpts [cap][2] | desc [cap][256],
 * unused rows zeroed) and all-gather it: every rank receives [world][1 + cap * 258] floats.  gathered_dev = NULL: into an
 * engine-owned device buffer of the sequence (what dani_frontend_match_exchanged consumes; *gathered_out, optional, receives its
 * address -- valid until the sequence's next exchange); otherwise into the caller's DEVICE buffer of that size.  Returns as soon as
 * the work is enqueued on the sequence's stream.  Every rank must call it, in the same order. */
DANI_API int dani_frontend_exchange_boundary(dani_engine* e, int32_t sequence, int32_t frame, int32_t cap, float* gathered_dev,
                                             const float** gathered_out, int32_t* n_out);
/* dani_frontend_match_resident with the predecessor taken from the sequence's last exchange: rank r > 0 the slab of rank r-1; rank
 * 0 the LAST rank's slab of the previous exchange (kept on the device across calls and sequences; none before the first one, or
 * after dani_comm_reset_carry: frame 0 then has no predecessor).  Outputs as dani_frontend_match_resident. */
DANI_API int dani_frontend_match_exchanged(dani_engine* e, int32_t sequence, float lg_threshold, int32_t* matches, float* scores,
                                           int32_t match_cap, int32_t* k_out);
DANI_API int dani_comm_reset_carry(dani_engine* e);
/* Forget the cached previous frame (start of a new sequence). */
DANI_API int dani_frontend_reset(dani_engine* e, int32_t sequence);
/* Complete a descriptor read-back that a defer_desc call left running (no-op otherwise); the cached predecessor is kept.
 * Call it before the `desc` array of that call is freed or reused while no dani_frontend_match_resident follows.
 * dani_engine_destroy CANCELS a pending read-back instead (nothing is written into `desc` after it returns). */
DANI_API int dani_frontend_finish_pending(dani_engine* e, int32_t sequence);
/* Stage wall times (ms) of the last dani_frontend_batch on this thread: [0] yolo, [1] superpoint, [2] lightglue, [3] total. */
DANI_API int dani_frontend_timings(dani_engine* e, float* ms4);

/* ---- projection-search descriptor scans -------------------------- */

/* Replaces LGmatcher::DescriptorDistance (src/LGmatcher.cpp:2009-2013: (float) cv::norm(a, b, NORM_L2) over 256
 * floats) and the best / second-best bookkeeping inside LGmatcher::SearchByProjection (src/LGmatcher.cpp:84-120; the
 * same scan appears at :444-464, :1694-1719, :1901-1916).  The caller (host, unchanged geometry code) builds, for each
 * of the nq query descriptors (map points), the candidate list Frame::GetFeaturesInArea returned minus the entries its
 * own checks reject, as CSR (cand_offsets [nq+1], cand_idx); the engine visits each list in order with strict `<`:
 *   bestDist = bestDist2 = 256, levels = -1, bestIdx = -1;  dist < bestDist -> shift best to second; else
 *   dist < bestDist2 -> update second.  t_level = octave of every train keypoint (may be NULL -> 0).
 * All pointers are HOST pointers (descriptors are uploaded once per call); outputs have nq entries. */
DANI_API int dani_desc_best2(dani_engine* e, const float* q_desc, int32_t nq, const float* t_desc, int32_t nt,
                             const int32_t* t_level, const int32_t* cand_offsets, const int32_t* cand_idx,
                             float* best_dist, float* best_dist2, int32_t* best_idx, int32_t* best_level,
                             int32_t* best_level2);

/* Same scan, and additionally the distance of EVERY candidate (cand_dist[cand_offsets[nq]], aligned with cand_idx).
 * SearchByProjection is sequential across map points: an accepted match claims its keypoint (F.mvpMapPoints[bestIdx] =
 * pMP, src/LGmatcher.cpp:124) and later map points skip claimed keypoints (:92-94).  A caller that must reproduce that
 * exactly replays the short selection loop on the host over these distances; the 256-float
 * L2 norms -- the cost of the search -- are computed here once. */
DANI_API int dani_desc_scan(dani_engine* e, const float* q_desc, int32_t nq, const float* t_desc, int32_t nt,
                            const int32_t* t_level, const int32_t* cand_offsets, const int32_t* cand_idx,
                            float* best_dist, float* best_dist2, int32_t* best_idx, int32_t* best_level,
                            int32_t* best_level2, float* cand_dist);

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:330-379) for nq map points at
 * once: mean_out[q] = (sum of rows[row_idx[c]] for c in [offsets[q], offsets[q+1]), accumulated in f32 in list order)
 * / (float)count -- the reference's cv::Mat += / /= sequence; a map point with no observation gets zeros.  rows is the
 * pool of observing keyframes' descriptor rows [nrows][256]; all pointers are HOST pointers. */
DANI_API int dani_desc_mean(dani_engine* e, const float* rows, int32_t nrows, const int32_t* offsets, const int32_t* row_idx,
                            int32_t nq, float* mean_out);

/* ---- DBoW3 transform of float descriptors ------------------------- */

/* A DBoW3 vocabulary tree resident on the device.  Node 0 is the root; node i's children are
 * child_ids[child_offsets[i] .. child_offsets[i+1]) in DBoW3's stored order (a leaf has none; child ids are larger than
 * their parent's, as DBoW3's builder produces them); node_desc [n_nodes][256] f32 (the root's row is unused),
 * node_weight / node_word_id as Vocabulary::Node::weight / word_id (Thirdparty/DBow3/src/Vocabulary.h:311-342).  L =
 * depth levels (Vocabulary::m_L).  The reference loads the tree with Vocabulary::load; the SuperPoint vocabulary itself
 * is withheld upstream, so only synthetic trees have been through this path. */
typedef struct dani_vocab dani_vocab;
DANI_API int dani_vocab_create(dani_engine* e, int32_t n_nodes, int32_t L, const int32_t* child_offsets, const int32_t* child_ids,
                               const float* node_desc, const double* node_weight, const int32_t* node_word_id, dani_vocab** out);
DANI_API void dani_vocab_destroy(dani_vocab* v);
/* Vocabulary::transform(features, BowVector&, FeatureVector&, levelsup) (Vocabulary.cpp:752-824; called with levelsup = 4
 * by Frame::ComputeBoW, src/Frame.cc:740-747).  The per-feature tree descent (squared-L2 distances to every child,
 * DescManip.cpp:120-129) runs on the GPU; the std::map bookkeeping, in feature order and in double, on the host.
 * weighting: 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY (Vocabulary::m_weighting); norm: 0 none, 1 L1, 2 L2 (the scoring object's
 * mustNormalize).  Outputs: BowVector as word ids in ascending order + values (capacity n); FeatureVector as node ids in
 * ascending order + CSR lists of feature indices (fv_offsets capacity n + 1). */
DANI_API int dani_bow_transform(dani_engine* e, dani_vocab* v, const float* feats, int32_t n, int32_t levelsup, int32_t weighting,
                                int32_t norm, int32_t* word_ids, double* word_values, int32_t* n_words, int32_t* fv_node_ids,
                                int32_t* fv_offsets, int32_t* fv_features, int32_t* n_fv_nodes);

/* ---- device-resident consumers ("descriptors already resident on GPU after extraction") ------- */

/* Frame's keypoint grid: FRAME_GRID_COLS x FRAME_GRID_ROWS (64 x 48, include/Frame.h:52-53) over the undistorted
 * image bounds; inv_w = cols / (mnMaxX - mnMinX), inv_h = rows / (mnMaxY - mnMinY) as floats (src/Frame.cc:253-254). */
typedef struct dani_grid { float min_x, min_y, inv_w, inv_h; int32_t cols, rows; } dani_grid;

/* One frame's features kept in HBM for as long as the caller's Frame / KeyFrame object lives: what the reference holds as
 * mvKeys(Un) + mDescriptors (include/Frame.h:236-250).  Nothing of it is re-uploaded by the calls below. */
typedef struct dani_features dani_features;
/* Keep frame `frame` of the LAST dani_frontend_batch call of `sequence` (device-to-device copy of keypoints, octaves and
 * descriptors).  kpts_un: optional n x 2 HOST array of undistorted keypoints (Frame::UndistortKeyPoints, src/Frame.cc:749-782,
 * dani_host_undistort_keypoints) used for the grid and the window tests instead of the extractor's coordinates (the
 * reference's k1 == 0 calibrations copy them).  grid: optional; builds Frame::AssignFeaturesToGrid's lists on the device
 * (src/Frame.cc:387-418).  octave: the n octaves as dani_sp_extract returned them (host; NULL = all 0). */
DANI_API int dani_frontend_retain(dani_engine* e, int32_t sequence, int32_t frame, const float* kpts_un, const int32_t* octave,
                                  const dani_grid* grid, dani_features** out);
/* Frame::UndistortKeyPoints (src/Frame.cc:749-782) on the retained coordinates, on the device: mvKeysUn replaces the
 * coordinates the grid and the window tests use (dist[0] == 0 or ndist == 0: unchanged, as the reference copies them);
 * same arithmetic as dani_host_undistort_keypoints, bit for bit.  grid (optional) is (re)built from the new coordinates;
 * kpts_un_out (optional, host, n x 2) receives mvKeysUn. */
DANI_API int dani_features_undistort(dani_engine* e, dani_features* f, const float* K4, const float* dist, int32_t ndist,
                                     const dani_grid* grid, float* kpts_un_out);
DANI_API void dani_features_release(dani_features* f);
DANI_API int dani_features_count(const dani_features* f);
/* Test hook: the device-built grid as CSR (cell = col * rows + row), as dani_host_assign_features_to_grid returns it. */
DANI_API int dani_features_fetch_grid(dani_features* f, int32_t* cell_offsets, int32_t* cell_idx);

/* dani_desc_scan with the train descriptors (and their octaves) taken from a retained frame: no descriptor upload. */
DANI_API int dani_desc_scan_resident(dani_engine* e, const float* q_desc, int32_t nq, const dani_features* train,
                                     const int32_t* cand_offsets, const int32_t* cand_idx, float* best_dist, float* best_dist2,
                                     int32_t* best_idx, int32_t* best_level, int32_t* best_level2, float* cand_dist);

/* The windowed search of LGmatcher::SearchByProjection in one call on a retained frame (needs its grid): for each of the nq
 * map points, Frame::GetFeaturesInArea(qx, qy, qr, min_level, max_level) (src/Frame.cc:659-725; NULL level arrays = -1) builds
 * the candidate list ON THE DEVICE, then the descriptor scan of dani_desc_best2 runs over it -- no candidate list and no
 * train descriptor crosses PCIe.  Outputs as dani_desc_best2; optionally (cand_cap > 0) the lists themselves and every
 * candidate's distance for callers that replay the sequential claim logic (cand_offsets [nq+1], cand_idx / cand_dist
 * [cand_cap]; DANI_ERR_CAPACITY with cand_offsets[nq] = needed size when they do not fit). */
DANI_API int dani_search_windows_resident(dani_engine* e, const dani_features* train, const float* q_desc, int32_t nq, const float* qx,
                                          const float* qy, const float* qr, const int32_t* min_level, const int32_t* max_level,
                                          float* best_dist, float* best_dist2, int32_t* best_idx, int32_t* best_level,
                                          int32_t* best_level2, int32_t* cand_offsets, int32_t* cand_idx, float* cand_dist,
                                          int32_t cand_cap);

/* dani_bow_transform on a retained frame's descriptors (Frame::ComputeBoW, src/Frame.cc:740-747): no descriptor upload. */
DANI_API int dani_bow_transform_resident(dani_engine* e, dani_vocab* v, const dani_features* f, int32_t levelsup, int32_t weighting,
                                         int32_t norm, int32_t* word_ids, double* word_values, int32_t* n_words,
                                         int32_t* fv_node_ids, int32_t* fv_offsets, int32_t* fv_features, int32_t* n_fv_nodes);

/* ---- host logic exposed for CPU-side tests (no GPU needed) ---------------------------------- */

/* The product's quadtree (DistributeOctTree, src/SPextractor.cpp:405-629) on host memory.
 * in_xyr: n x (x, y, response) border-relative; out_idx: indices into the input, in the reference's
 * std::list order; returns the count (<= cap) or a negative status. */
DANI_API int dani_host_distribute_octtree(const float* in_xyr, int32_t n, int32_t min_x, int32_t max_x, int32_t min_y,
                                          int32_t max_y, int32_t n_features, int32_t* out_idx, int32_t cap);
/* The cell grid of ComputeKeyPointsOctTree (src/SPextractor.cpp:232-266) for an image size: writes up to
 * cap cells as (iniX, maxX, iniY, maxY, offX, offY) and returns the number of non-skipped cells. */
DANI_API int dani_host_cell_grid(int32_t width, int32_t height, int32_t* cells6, int32_t cap);

/* ---- feature hand-off formats (host, order-defining) ---------------- */


/* Frame::AssignFeaturesToGrid + PosInGrid (src/Frame.cc:387-418,727-737; monocular / RGB-D case): CSR over the cells in
 * mGrid[col][row] order (cell = col * rows + row): cell_offsets[cols*rows + 1], cell_idx[<= n] = keypoint indices in
 * ascending order per cell; keypoints that round outside the grid are dropped.  Returns the number of indices written. */
DANI_API int dani_host_assign_features_to_grid(const float* kpts_xy, int32_t n, const dani_grid* g, int32_t* cell_offsets,
                                               int32_t* cell_idx);
/* Frame::GetFeaturesInArea (src/Frame.cc:659-725) for nq windows (x, y, r, minLevel, maxLevel; NULL levels = -1) at once:
 * CSR candidate lists in the reference's order (cells ix outer / iy inner, cell lists in index order) -- the input of
 * dani_desc_best2 / dani_desc_scan.  Returns the total number of candidates, DANI_ERR_CAPACITY if it exceeds cap. */
DANI_API int dani_host_features_in_area(const float* kpts_xy, const int32_t* octave, int32_t n, const dani_grid* g,
                                        const int32_t* cell_offsets, const int32_t* cell_idx, const float* qx, const float* qy,
                                        const float* qr, const int32_t* min_level, const int32_t* max_level, int32_t nq,
                                        int32_t* cand_offsets, int32_t* cand_idx, int32_t cap);
/* Frame::ComputeStereoMatches after its LightGlue call (src/Frame.cc:862-914): mvuRight / mvDepth (nl entries, -1 =
 * none) from the (left, right) matches and scores of dani_lg_match: disparity gate [0, mbf / mb), depth = mbf /
 * disparity, then every match whose (1 - score) reaches 1.5 x the median is cleared again. */
DANI_API int dani_host_stereo_from_matches(const float* ul_x, int32_t nl, const float* ur_x, int32_t nr, const int32_t* matches,
                                           const float* scores, int32_t k, float mb, float mbf, float* u_right, float* depth);
/* mnFeaturesPerLevel / mvScaleFactor tables (src/SPextractor.cpp:97-126). */
/* Frame::UndistortKeyPoints (src/Frame.cc:749-782): mvKeysUn from mvKeys.  K4 = (fx, fy, cx, cy); dist = (k1, k2, p1, p2
 * [, k3]) as in mDistCoef.  dist[0] == 0 (or ndist == 0) copies the input, as the reference does; otherwise the five
 * fixed-point iterations of cv::undistortPoints(..., K, distCoef, noArray(), K) in double precision (OpenCV itself is an
 * external dependency of the reference: restated from its published algorithm, see csrc/host_logic.cpp).  out_xy may
 * alias kpts_xy. */
DANI_API int dani_host_undistort_keypoints(const float* kpts_xy, int32_t n, const float* K4, const float* dist, int32_t ndist,
                                           float* out_xy);
DANI_API int dani_host_features_per_level(int32_t nfeatures, float scale_factor, int32_t nlevels, int32_t* per_level,
                                          float* scale_factors);

DANI_API const char* dani_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DANI_FRONTEND_H */
