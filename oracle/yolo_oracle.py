"""oracle/yolo_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the reference's dynamic-object detector
    YoloDetection::Detect            /root/reference/src/YoloDetect.cpp:27-82
    non_max_suppression              src/YoloDetect.cpp:84-146
    RefineDynamicBoxes               src/YoloDetect.cpp:148-208
    ctor class lists                 src/YoloDetect.cpp:6-20
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

PARITY UNPINNED for the network: the reference runs the TorchScript blob `yolov5s.torchscript.pt`
(src/YoloDetect.cpp:9), which is not in the repository (upstream ultralytics YOLOv5 "s", version unrecorded; the
wrapper is dated 2022-03, src/YoloDetect.cpp:2, i.e. the v6.x architecture with the 6x6 stem).  The network is
therefore restated from the published YOLOv5s v6 model (BatchNorm folded into the convolutions, as the export
does): Conv = conv + SiLU; C3; SPPF(k=5); PAN head; Detect with the standard anchors; tensor names follow the
public fused state_dict (model.0.conv.weight, model.2.cv1.conv.weight, model.2.m.0.cv1.conv.weight, ...,
model.24.m.0.weight).  Everything AFTER the network (score filter, xywh->xyxy, class-agnostic greedy NMS, box
rescale with the reference's 640/384 constants (quirk Q10), int truncation, placeholder box, static-box
subtraction) is a literal restatement of the cited lines.  The convolutions use the repo's canonical f32 chain
(oracle/sp_oracle.cpp: orc_conv2d_ex), SiLU = x / (1 + dani_expf(-x)).
What IS pinned is the arithmetic of the network restatement: tests/golden/make_golden_nets.py writes the same
architecture with ATen operators (conv2d, silu, max_pool2d, interpolate) and tests/test_oracle_golden_nets.py holds
the three raw head tensors to the committed vectors (tests/golden/yolo_g8.npz).
"""
from __future__ import annotations

import numpy as np

from . import oracle as O

f32 = np.float32

ANCHORS = np.array([[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]], f32).reshape(3, 3, 2)
STRIDES = (8, 16, 32)
NC = 80

# src/YoloDetect.cpp:17-18 (sic: "motorbike", "crow")
DYNAMIC_NAMES = ["person", "car", "motorbike", "bus", "train", "truck", "boat", "bird", "cat",
                 "dog", "horse", "sheep", "crow", "bear"]

# The file `coco.names` is read from the CWD by the reference (src/YoloDetect.cpp:12-16) and is not shipped; this
# is the standard 80-class COCO list the YOLOv5 release uses.
COCO_NAMES = ["person", "bicycle", "car", "motorcycle", "airplane", "bus", "train", "truck", "boat", "traffic light",
              "fire hydrant", "stop sign", "parking meter", "bench", "bird", "cat", "dog", "horse", "sheep", "cow",
              "elephant", "bear", "zebra", "giraffe", "backpack", "umbrella", "handbag", "tie", "suitcase", "frisbee",
              "skis", "snowboard", "sports ball", "kite", "baseball bat", "baseball glove", "skateboard", "surfboard",
              "tennis racket", "bottle", "wine glass", "cup", "fork", "knife", "spoon", "bowl", "banana", "apple",
              "sandwich", "orange", "broccoli", "carrot", "hot dog", "pizza", "donut", "cake", "chair", "couch",
              "potted plant", "bed", "dining table", "toilet", "tv", "laptop", "mouse", "remote", "keyboard",
              "cell phone", "microwave", "oven", "toaster", "sink", "refrigerator", "book", "clock", "vase",
              "scissors", "teddy bear", "hair drier", "toothbrush"]


# ---- network -------------------------------------------------------------------------------------------------
def _conv(w, name, x, k, s, act=2):
    p = {1: 0, 3: 1, 6: 2}[k]
    return O.conv2d_ex(x, w[name + ".weight"], w[name + ".bias"], s, p, act)


def _c3(w, name, x, n, shortcut):
    y = _conv(w, f"{name}.cv1.conv", x, 1, 1)
    for j in range(n):
        t = _conv(w, f"{name}.m.{j}.cv2.conv", _conv(w, f"{name}.m.{j}.cv1.conv", y, 1, 1), 3, 1)
        y = (y + t).astype(f32) if shortcut else t
    y2 = _conv(w, f"{name}.cv2.conv", x, 1, 1)
    return _conv(w, f"{name}.cv3.conv", np.concatenate([y, y2], 0), 1, 1)


def _sppf(w, name, x):
    x = _conv(w, f"{name}.cv1.conv", x, 1, 1)
    y1 = O.maxpool5(x); y2 = O.maxpool5(y1); y3 = O.maxpool5(y2)
    return _conv(w, f"{name}.cv2.conv", np.concatenate([x, y1, y2, y3], 0), 1, 1)


def _up(x):
    return np.repeat(np.repeat(x, 2, axis=1), 2, axis=2)


def _sigmoid(x):
    return O.sigmoid(x)                                  # canonical 1 / (1 + dani_expf(-x))


def network(w, x, want_taps=False):
    """x [3, H, W] f32 (RGB / 255).  Returns preds [N, 85] in the Detect layer's order."""
    taps = {}
    x0 = _conv(w, "model.0.conv", x, 6, 2)
    x1 = _conv(w, "model.1.conv", x0, 3, 2)
    x2 = _c3(w, "model.2", x1, 1, True)
    x3 = _conv(w, "model.3.conv", x2, 3, 2)
    x4 = _c3(w, "model.4", x3, 2, True)
    x5 = _conv(w, "model.5.conv", x4, 3, 2)
    x6 = _c3(w, "model.6", x5, 3, True)
    x7 = _conv(w, "model.7.conv", x6, 3, 2)
    x8 = _c3(w, "model.8", x7, 1, True)
    x9 = _sppf(w, "model.9", x8)
    x10 = _conv(w, "model.10.conv", x9, 1, 1)
    x13 = _c3(w, "model.13", np.concatenate([_up(x10), x6], 0), 1, False)
    x14 = _conv(w, "model.14.conv", x13, 1, 1)
    x17 = _c3(w, "model.17", np.concatenate([_up(x14), x4], 0), 1, False)
    x18 = _conv(w, "model.18.conv", x17, 3, 2)
    x20 = _c3(w, "model.20", np.concatenate([x18, x14], 0), 1, False)
    x21 = _conv(w, "model.21.conv", x20, 3, 2)
    x23 = _c3(w, "model.23", np.concatenate([x21, x10], 0), 1, False)
    outs = []
    for i, f in enumerate((x17, x20, x23)):
        r = O.conv2d_ex(f, w[f"model.24.m.{i}.weight"], w[f"model.24.m.{i}.bias"], 1, 0, 0)   # [255, ny, nx]
        ny, nx = r.shape[1:]
        y = _sigmoid(r.reshape(3, 85, ny, nx).transpose(0, 2, 3, 1))                             # [3, ny, nx, 85]
        gy, gx = np.meshgrid(np.arange(ny, dtype=f32), np.arange(nx, dtype=f32), indexing="ij")
        grid = np.stack([gx, gy], -1)[None]
        xy = (y[..., 0:2] * f32(2) - f32(0.5) + grid) * f32(STRIDES[i])
        wh = (y[..., 2:4] * f32(2)) ** 2 * ANCHORS[i][:, None, None, :]
        outs.append(np.concatenate([xy, wh, y[..., 4:]], -1).reshape(-1, 85).astype(f32))
        if want_taps:
            taps[f"raw{i}"] = r
    if want_taps:
        taps.update(x0=x0, x2=x2, x4=x4, x6=x6, x9=x9, x13=x13, x17=x17, x20=x20, x23=x23)
        return np.concatenate(outs, 0), taps
    return np.concatenate(outs, 0)


# ---- pre / post processing ---------------------------------------------------------------------------------------
def preprocess(bgr):
    """YoloDetect.cpp:39-46: resize to 640x380 (INTER_LINEAR), BGR->RGB, HWC u8 -> CHW f32 / 255."""
    img = O.resize_bilinear_u8c3(bgr, 640, 380)
    rgb = img[:, :, ::-1]
    return (rgb.transpose(2, 0, 1).astype(f32) / f32(255)).astype(f32)


def non_max_suppression(pred, score_thresh=0.4, iou_thresh=0.5):
    """YoloDetect.cpp:84-146 for one image: pred [N, 85] -> dets [k, 6] (x1, y1, x2, y2, score, cls)."""
    pred = np.array(pred, f32, copy=True)
    scores = pred[:, 4] * pred[:, 5:].max(1)
    pred = pred[scores > f32(score_thresh)]
    if pred.shape[0] == 0:
        return np.zeros((0, 6), f32)
    pred[:, 0] = pred[:, 0] - pred[:, 2] / f32(2)
    pred[:, 1] = pred[:, 1] - pred[:, 3] / f32(2)
    pred[:, 2] = pred[:, 0] + pred[:, 2]
    pred[:, 3] = pred[:, 1] + pred[:, 3]
    cls = pred[:, 5:].argmax(1)
    pred[:, 4] = pred[:, 4] * pred[:, 5:].max(1)
    pred[:, 5] = cls.astype(f32)
    dets = pred[:, :6]
    areas = (dets[:, 3] - dets[:, 1]) * (dets[:, 2] - dets[:, 0])
    order = np.argsort(-dets[:, 4], kind="stable")            # torch.sort(descending=True); ties: first wins
    keep = []
    idx = order
    while idx.size > 0:
        i0 = idx[0]
        keep.append(i0)
        rest = idx[1:]
        if rest.size == 0:
            break
        lefts = np.maximum(dets[i0, 0], dets[rest, 0]); tops = np.maximum(dets[i0, 1], dets[rest, 1])
        rights = np.minimum(dets[i0, 2], dets[rest, 2]); bottoms = np.minimum(dets[i0, 3], dets[rest, 3])
        widths = np.maximum(f32(0), rights - lefts); heights = np.maximum(f32(0), bottoms - tops)
        overlaps = widths * heights
        ious = overlaps / (areas[i0] + areas[rest] - overlaps)
        idx = rest[ious <= f32(iou_thresh)]
    return dets[np.array(keep, np.int64)]


def _rect_and(a, b):
    x = max(a[0], b[0]); y = max(a[1], b[1])
    w = min(a[0] + a[2], b[0] + b[2]) - x; h = min(a[1] + a[3], b[1] + b[3]) - y
    return (x, y, w, h) if (w > 0 and h > 0) else (0, 0, 0, 0)    # cv::Rect operator&: empty -> Rect()


def refine_dynamic_boxes(dynamic, detect_map, dynamic_names):
    """YoloDetect.cpp:148-208.  detect_map: dict class name -> list of rects; std::map iterates in key order."""
    out = []
    for D in dynamic:
        sub = [tuple(D)]
        for name in sorted(detect_map.keys()):
            if name in dynamic_names:
                continue
            for S in detect_map[name]:
                new = []
                for region in sub:
                    inter = _rect_and(region, S)
                    if inter[2] * inter[3] > 0:
                        if inter[1] > region[1]:
                            new.append((region[0], region[1], region[2], inter[1] - region[1]))
                        if inter[1] + inter[3] < region[1] + region[3]:
                            new.append((region[0], inter[1] + inter[3], region[2], region[1] + region[3] - (inter[1] + inter[3])))
                        if inter[0] > region[0]:
                            new.append((region[0], region[1], inter[0] - region[0], region[3]))
                        if inter[0] + inter[2] < region[0] + region[2]:
                            new.append((inter[0] + inter[2], region[1], region[0] + region[2] - (inter[0] + inter[2]), region[3]))
                    else:
                        new.append(region)
                sub = new
        if sub:
            best = sub[0]
            for r in sub[1:]:                                   # std::max_element: first maximum
                if best[2] * best[3] < r[2] * r[3]:
                    best = r
            out.append(best)
    return out


def boxes_from_dets(dets, cols, rows, class_names=COCO_NAMES, dynamic_names=DYNAMIC_NAMES):
    """YoloDetect.cpp:51-80: rescale (x by cols/640, y by rows/384 -- quirk Q10), truncate to int Rects, split into
    mmDetectMap / mvDynamicArea, placeholder (1,1,1,1), RefineDynamicBoxes.  Returns (dynamic_area, detect_map)."""
    detect_map, dynamic = {}, []
    if len(dets) == 0:
        return dynamic, detect_map
    for d in dets:
        left = f32(d[0]) * f32(cols) / f32(640); top = f32(d[1]) * f32(rows) / f32(384)
        right = f32(d[2]) * f32(cols) / f32(640); bottom = f32(d[3]) * f32(rows) / f32(384)
        cid = int(d[5])
        rect = (int(left), int(top), int(f32(right - left)), int(f32(bottom - top)))    # float -> int truncation
        name = class_names[cid]
        detect_map.setdefault(name, []).append(rect)
        if name in dynamic_names:
            dynamic.append(rect)
    if len(dynamic) == 0:
        dynamic.append((1, 1, 1, 1))
    return refine_dynamic_boxes(dynamic, detect_map, dynamic_names), detect_map


def detect(w, bgr, score_thresh=0.4, iou_thresh=0.5, class_names=COCO_NAMES, dynamic_names=DYNAMIC_NAMES):
    """YoloDetection::Detect.  Returns (ok, dets [k,6], dynamic_area list, detect_map)."""
    if bgr is None or bgr.size == 0:
        return False, np.zeros((0, 6), f32), [], {}
    x = preprocess(bgr)
    preds = network(w, x)
    dets = non_max_suppression(preds, score_thresh, iou_thresh)
    dyn, dmap = boxes_from_dets(dets, bgr.shape[1], bgr.shape[0], class_names, dynamic_names)
    return True, dets, dyn, dmap
