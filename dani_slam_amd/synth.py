"""Seeded synthetic inputs for the front-end path (frames, SuperPoint weights).

The reference ships no weights and no dataset (/sly_slam/superpoint_new.pt is a
hard-coded path to a file that was never committed, /root/reference/src/SPextractor.cpp:86), so every
test and benchmark in this repo runs on the deterministic synthetic inputs defined here
(SURVEY.md section 8d).  Pure numpy, no torch: the same bytes are produced here and on the GPU box.

Weight tensors use the reference's own parameter names and OIHW layout
(`conv1a.weight` ... `convDb.bias`, /root/reference/src/SuperPoint.cpp:16-52) so that real weights
converted with tools/convert_weights.py drop into the same loader.
"""
from __future__ import annotations

import numpy as np

# (name, Cin, Cout, ksize) in forward order -- /root/reference/src/SuperPoint.cpp:16-33
SP_LAYERS = [
    ("conv1a", 1, 64, 3), ("conv1b", 64, 64, 3),
    ("conv2a", 64, 64, 3), ("conv2b", 64, 64, 3),
    ("conv3a", 64, 128, 3), ("conv3b", 128, 128, 3),
    ("conv4a", 128, 128, 3), ("conv4b", 128, 128, 3),
    ("convPa", 128, 256, 3), ("convPb", 256, 65, 1),
    ("convDa", 128, 256, 3), ("convDb", 256, 256, 1),
]


def superpoint_weights(seed: int = 0, pb_gain: float = 3.0, dustbin_bias: float = 9.5, centered_desc: bool = False) -> dict:
    """He-uniform weights (variance preserving through the ReLU stack) + small biases.

    `pb_gain` / `dustbin_bias` shape the detector head so that the 65-way softmax is peaky enough for
    roughly 1-3 k pixels of a synthetic 640x480 frame to exceed the reference's iniTh=0.015
    (Examples/RGB-D/TUM3.yaml:42-55); with an untrained flat head every pixel would sit at 1/65=0.0154.

    `centered_desc` subtracts from convDb.bias the per-channel mean of the raw descriptor head output over synthetic
    frames 0..3 (a stored calibration constant, tools/make_desc_center.py): random weights give every descriptor the
    same large offset, so that different keypoints have cosine 0.85 and LightGlue matches almost nothing between two
    views of one scene; centred, it finds hundreds (bench.py's moving-camera sequence).  Keypoints are unaffected.
    """
    rng = np.random.RandomState(seed)
    w = {}
    for name, cin, cout, k in SP_LAYERS:
        fan_in = cin * k * k
        bound = np.sqrt(6.0 / fan_in)
        w[name + ".weight"] = rng.uniform(-bound, bound, size=(cout, cin, k, k)).astype(np.float32)
        w[name + ".bias"] = rng.uniform(-0.05, 0.05, size=(cout,)).astype(np.float32)
    w["convPb.weight"] = (w["convPb.weight"] * np.float32(pb_gain)).astype(np.float32)
    w["convPb.bias"][64] += np.float32(dustbin_bias)
    if centered_desc:
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", f"sp_desc_center_seed{seed}.npy")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path}: run tools/make_desc_center.py (only seed 0 is shipped)")
        w["convDb.bias"] = (w["convDb.bias"] - np.load(path)).astype(np.float32)
    return w


def _blur_axis(a: np.ndarray, k: np.ndarray, axis: int) -> np.ndarray:
    r = len(k) // 2
    pad = [(0, 0), (0, 0)]
    pad[axis] = (r, r)
    ap = np.pad(a, pad, mode="reflect")
    out = np.zeros_like(a)
    n = a.shape[axis]
    for i, kv in enumerate(k):          # fixed summation order -> identical bits everywhere
        sl = [slice(None), slice(None)]
        sl[axis] = slice(i, i + n)
        out = out + kv * ap[tuple(sl)]
    return out


def frame(i: int, width: int = 640, height: int = 480) -> np.ndarray:
    """Synthetic grayscale frame i (uint8 [H, W]): band-limited noise (sigma=2 px Gaussian blur of
    uniform noise, contrast-stretched) plus 200 random axis-aligned rectangles."""
    rng = np.random.RandomState(1234 + i)
    noise = rng.uniform(0.0, 1.0, size=(height, width))
    xs = np.arange(-6, 7, dtype=np.float64)
    k = np.exp(-(xs * xs) / (2.0 * 2.0 * 2.0))
    k = k / k.sum()
    img = _blur_axis(_blur_axis(noise, k, 0), k, 1)
    img = (img - img.min()) / (img.max() - img.min())
    img = 0.25 + 0.5 * img
    for _ in range(200):
        w = int(rng.randint(6, 60)); h = int(rng.randint(6, 60))
        x = int(rng.randint(0, width - w)); y = int(rng.randint(0, height - h))
        v = float(rng.uniform(0.0, 1.0))
        img[y:y + h, x:x + w] = 0.6 * v + 0.4 * img[y:y + h, x:x + w]
    return np.clip(np.rint(img * 255.0), 0, 255).astype(np.uint8)


def frame_bgr(i: int, width: int = 640, height: int = 480) -> np.ndarray:
    """BGR variant for the YOLO detector: the same pattern with a per-channel gain."""
    g = frame(i, width, height).astype(np.float64)
    gains = (0.9, 1.0, 0.8)
    return np.stack([np.clip(np.rint(g * s), 0, 255) for s in gains], axis=-1).astype(np.uint8)


# ---------------------------------------------------------------------------------------------------------
# Moving-camera sequence ("frame i vs the same frame warped by a fixed affine, 2 degree rotation,
# 8-px shift"): frame i is a 640x480 window of one large synthetic canvas, rotated by 2 degrees per frame and moved
# 8 px per frame along a circle, bilinearly resampled.  Consecutive frames are therefore related by exactly that warp,
# whatever i is, and a shard of consecutive frames is a piece of one camera trajectory.
# ---------------------------------------------------------------------------------------------------------
_CANVAS = {}
SEQ_CANVAS = 1408     # the window's half diagonal (400) + the path's diameter (2 x 229) + margin, either side


def _canvas(seed: int = 0) -> np.ndarray:
    if seed not in _CANVAS:
        _CANVAS[seed] = frame(10_000 + seed, SEQ_CANVAS, SEQ_CANVAS).astype(np.float64)
    return _CANVAS[seed]


def sequence_frame(i: int, width: int = 640, height: int = 480, rot_deg: float = 2.0, step_px: float = 8.0, seed: int = 0) -> np.ndarray:
    """Frame i of the synthetic camera trajectory (uint8 [H, W])."""
    base = _canvas(seed)
    S = base.shape[0]
    th = np.deg2rad(rot_deg) * i
    R = step_px / np.deg2rad(rot_deg)            # consecutive window centres are step_px apart (arc length)
    cx = S / 2.0 + R * np.sin(th); cy = S / 2.0 - R * np.cos(th)
    ys, xs = np.mgrid[0:height, 0:width].astype(np.float64)
    u = xs - (width - 1) / 2.0; v = ys - (height - 1) / 2.0
    c, s = np.cos(th), np.sin(th)
    X = cx + c * u - s * v; Y = cy + s * u + c * v
    x0 = np.floor(X); y0 = np.floor(Y)
    fx = X - x0; fy = Y - y0
    x0 = np.clip(x0.astype(np.int64), 0, S - 2); y0 = np.clip(y0.astype(np.int64), 0, S - 2)
    out = (base[y0, x0] * (1 - fx) * (1 - fy) + base[y0, x0 + 1] * fx * (1 - fy)
           + base[y0 + 1, x0] * (1 - fx) * fy + base[y0 + 1, x0 + 1] * fx * fy)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def sequence_frame_bgr(i: int, width: int = 640, height: int = 480, **kw) -> np.ndarray:
    """BGR variant of sequence_frame for the YOLO detector (same per-channel gains as frame_bgr)."""
    g = sequence_frame(i, width, height, **kw).astype(np.float64)
    return np.stack([np.clip(np.rint(g * s), 0, 255) for s in (0.9, 1.0, 0.8)], axis=-1).astype(np.uint8)


# ---------------------------------------------------------------------------------------------------------
# LightGlue: seeded synthetic weights under the public checkpoint's tensor names (the reference's network is the
# absent TorchScript blob /sly_slam/lightglue_scripted_v2.pt, /root/reference/src/LightGlue.cpp:9).
# ---------------------------------------------------------------------------------------------------------
LG_LAYERS, LG_DIM, LG_HEADS = 9, 256, 4


def lightglue_weights(seed: int = 0, block_gain: float = 0.5, proj_gain: float = 24.0) -> dict:
    """Random transformer blocks (variance-preserving, residual branches scaled by `block_gain`) and an
    assignment head close to a scaled identity, so that descriptors of corresponding points stay similar through
    the 9 layers and the log-double-softmax produces a realistic number of confident mutual matches."""
    rng = np.random.RandomState(1000 + seed)
    d = LG_DIM
    w = {}

    def lin(name, out, inp, gain=1.0, bias=0.02):
        w[name + ".weight"] = (rng.standard_normal((out, inp)) * (gain / np.sqrt(inp))).astype(np.float32)
        w[name + ".bias"] = (rng.standard_normal(out) * bias).astype(np.float32)

    w["posenc.Wr.weight"] = rng.standard_normal((32, 2)).astype(np.float32)
    for i in range(LG_LAYERS):
        for blk in ("self_attn", "cross_attn"):
            p = f"transformers.{i}.{blk}"
            if blk == "self_attn":
                lin(p + ".Wqkv", 3 * d, d, gain=2.0)
                lin(p + ".out_proj", d, d)
            else:
                lin(p + ".to_qk", d, d, gain=2.0)
                lin(p + ".to_v", d, d)
                lin(p + ".to_out", d, d)
            lin(p + ".ffn.0", 2 * d, 2 * d)
            w[p + ".ffn.1.weight"] = (1.0 + 0.1 * rng.standard_normal(2 * d)).astype(np.float32)
            w[p + ".ffn.1.bias"] = (0.05 * rng.standard_normal(2 * d)).astype(np.float32)
            lin(p + ".ffn.3", d, 2 * d, gain=block_gain * 0.06)
        p = f"log_assignment.{i}"
        w[p + ".final_proj.weight"] = (proj_gain * np.eye(d) + 0.5 * rng.standard_normal((d, d))).astype(np.float32)
        w[p + ".final_proj.bias"] = (0.02 * rng.standard_normal(d)).astype(np.float32)
        w[p + ".matchability.weight"] = (0.5 * rng.standard_normal((1, d))).astype(np.float32)
        w[p + ".matchability.bias"] = np.array([2.0], np.float32)
    return w


def matching_pair(seed: int, n0: int = 1024, n1: int = 1024, overlap: float = 0.8, noise: float = 0.08,
                  width: int = 640, height: int = 480):
    """Synthetic keypoint/descriptor sets with known correspondences, in SuperPoint's output scale (each
    descriptor column has unit norm over the keypoints, quirk Q7, so rows have norm ~ sqrt(256/n)).
    Returns kpts0 [n0,2], desc0 [n0,256], kpts1 [n1,2], desc1 [n1,256], gt (index in set 1 for each of set 0, or -1)."""
    rng = np.random.RandomState(777 + seed)
    k0 = np.stack([rng.randint(16, width - 16, n0), rng.randint(16, height - 16, n0)], 1).astype(np.float32)
    d0 = rng.standard_normal((n0, 256))
    n_common = int(min(n0, n1) * overlap)
    perm = rng.permutation(n1)
    src = rng.permutation(n0)[:n_common]
    k1 = np.stack([rng.randint(16, width - 16, n1), rng.randint(16, height - 16, n1)], 1).astype(np.float32)
    d1 = rng.standard_normal((n1, 256))
    gt = -np.ones(n0, np.int64)
    for a, b in zip(src, perm[:n_common]):
        d1[b] = d0[a] + noise * rng.standard_normal(256)
        k1[b] = k0[a] + np.array([8.0, 0.0]) + rng.randint(-2, 3, 2)
        gt[a] = b
    d0 = d0 / np.linalg.norm(d0, axis=0, keepdims=True)
    d1 = d1 / np.linalg.norm(d1, axis=0, keepdims=True)
    return k0, d0.astype(np.float32), k1, d1.astype(np.float32), gt


# ---------------------------------------------------------------------------------------------------------
# YOLOv5s: seeded synthetic weights under the public fused state_dict names (the reference's network is the absent
# TorchScript blob yolov5s.torchscript.pt, /root/reference/src/YoloDetect.cpp:9).
# ---------------------------------------------------------------------------------------------------------
def _yolo_conv_list():
    """(name, cin, cout, k) for every convolution of YOLOv5s v6 (width 0.5, depth 0.33), BN folded."""
    L = []

    def conv(name, cin, cout, k):
        L.append((name + ".conv", cin, cout, k))

    def c3(name, c1, c2, n):
        c_ = c2 // 2
        conv(f"{name}.cv1", c1, c_, 1); conv(f"{name}.cv2", c1, c_, 1); conv(f"{name}.cv3", 2 * c_, c2, 1)
        for j in range(n):
            conv(f"{name}.m.{j}.cv1", c_, c_, 1); conv(f"{name}.m.{j}.cv2", c_, c_, 3)

    conv("model.0", 3, 32, 6); conv("model.1", 32, 64, 3); c3("model.2", 64, 64, 1)
    conv("model.3", 64, 128, 3); c3("model.4", 128, 128, 2)
    conv("model.5", 128, 256, 3); c3("model.6", 256, 256, 3)
    conv("model.7", 256, 512, 3); c3("model.8", 512, 512, 1)
    conv("model.9.cv1", 512, 256, 1); conv("model.9.cv2", 1024, 512, 1)
    conv("model.10", 512, 256, 1); c3("model.13", 512, 256, 1)
    conv("model.14", 256, 128, 1); c3("model.17", 256, 128, 1)
    conv("model.18", 128, 128, 3); c3("model.20", 256, 256, 1)
    conv("model.21", 256, 256, 3); c3("model.23", 512, 512, 1)
    return L


YOLO_CONVS = _yolo_conv_list()
YOLO_DETECT_CH = (128, 256, 512)


def yolo_weights(seed: int = 0, obj_bias: float = -3.0, head_gain: float = 1.0, gain: float = 1.5) -> dict:
    """Variance-preserving random convolutions (SiLU gain) and a Detect head whose objectness logits are biased so
    that a synthetic frame yields 0-50 candidates (a realistic count for an indoor scene) above the reference's 0.4 score threshold (gain 1.5 sits at the
    edge where the SiLU stack neither dies out nor blows up for this seed)."""
    rng = np.random.RandomState(2000 + seed)
    w = {}
    for name, cin, cout, k in YOLO_CONVS:
        fan_in = cin * k * k
        bound = np.sqrt(3.0) * gain / np.sqrt(fan_in)
        w[name + ".weight"] = rng.uniform(-bound, bound, size=(cout, cin, k, k)).astype(np.float32)
        w[name + ".bias"] = rng.uniform(-0.1, 0.1, size=(cout,)).astype(np.float32)
    for i, ch in enumerate(YOLO_DETECT_CH):
        bound = np.sqrt(3.0) * head_gain / np.sqrt(ch)
        w[f"model.24.m.{i}.weight"] = rng.uniform(-bound, bound, size=(255, ch, 1, 1)).astype(np.float32)
        b = rng.uniform(-0.2, 0.2, size=(3, 85)).astype(np.float32)
        b[:, 4] += np.float32(obj_bias)
        w[f"model.24.m.{i}.bias"] = b.reshape(-1)
    return w


# Class lists for the dynamic-object detector.  The reference reads `coco.names` from the CWD (src/YoloDetect.cpp:12-16;
# the file is not shipped) -- this is the standard 80-class COCO list of the YOLOv5 release -- and hard-codes its
# dynamic classes (src/YoloDetect.cpp:17-18, sic "motorbike", "crow").
COCO_NAMES = ["person", "bicycle", "car", "motorcycle", "airplane", "bus", "train", "truck", "boat", "traffic light",
              "fire hydrant", "stop sign", "parking meter", "bench", "bird", "cat", "dog", "horse", "sheep", "cow",
              "elephant", "bear", "zebra", "giraffe", "backpack", "umbrella", "handbag", "tie", "suitcase", "frisbee",
              "skis", "snowboard", "sports ball", "kite", "baseball bat", "baseball glove", "skateboard", "surfboard",
              "tennis racket", "bottle", "wine glass", "cup", "fork", "knife", "spoon", "bowl", "banana", "apple",
              "sandwich", "orange", "broccoli", "carrot", "hot dog", "pizza", "donut", "cake", "chair", "couch",
              "potted plant", "bed", "dining table", "toilet", "tv", "laptop", "mouse", "remote", "keyboard",
              "cell phone", "microwave", "oven", "toaster", "sink", "refrigerator", "book", "clock", "vase",
              "scissors", "teddy bear", "hair drier", "toothbrush"]
DYNAMIC_NAMES = ["person", "car", "motorbike", "bus", "train", "truck", "boat", "bird", "cat",
                 "dog", "horse", "sheep", "crow", "bear"]


# ---------------------------------------------------------------- DBoW3 vocabulary
def vocab_tree(seed, k=10, L=3, dim=256, stop_fraction=0.05):
    """A synthetic k-ary vocabulary of depth L in DBoW3's node order (children appended as their parent is expanded,
    breadth first): returns dict(child_offsets, child_ids, desc, weight, word_id, L)."""
    rng = np.random.default_rng(seed)
    children = [[]]
    desc = [np.zeros(dim, np.float32)]
    level = [0]
    frontier = [0]
    for lv in range(1, L + 1):
        nxt = []
        for p in frontier:
            kk = k if rng.random() > 0.1 else max(2, k // 2)        # some nodes have fewer children
            for _ in range(kk):
                nid = len(children)
                children.append([]); level.append(lv)
                desc.append((desc[p] + rng.standard_normal(dim).astype(np.float32) * np.float32(1.0 / lv)).astype(np.float32))
                children[p].append(nid); nxt.append(nid)
        frontier = nxt
    n = len(children)
    weight = np.zeros(n, np.float64); word_id = np.zeros(n, np.int32)
    w = 0
    for i in range(n):
        if not children[i]:
            word_id[i] = w; w += 1
            weight[i] = 0.0 if rng.random() < stop_fraction else float(rng.uniform(0.1, 8.0))     # idf; 0 = stopped word
    offs = np.zeros(n + 1, np.int32)
    offs[1:] = np.cumsum([len(c) for c in children])
    ids = np.array([c for ch in children for c in ch], np.int32)
    return dict(child_offsets=offs, child_ids=ids, desc=np.stack(desc), weight=weight, word_id=word_id, L=L)


# ---------------------------------------------------------------- bench.py's workload (shared with tests/test_bench_shape_gpu.py)
def bench_weights(seed: int = 0) -> dict:
    """Weights of the bench workload: SuperPoint with the centred descriptor head (see superpoint_weights), LightGlue, YOLOv5s."""
    w = dict(superpoint_weights(seed, centered_desc=True))
    w.update(lightglue_weights(seed)); w.update(yolo_weights(seed))
    return w


def bench_frames(indices):
    """(gray [n, 480, 640] u8, bgr [n, 480, 640, 3] u8) for frames `indices` of the moving-camera sequence."""
    idx = list(indices)
    return (np.stack([sequence_frame(i) for i in idx]), np.stack([sequence_frame_bgr(i) for i in idx]))
