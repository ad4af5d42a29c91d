#!/usr/bin/env python3
"""Golden vectors G3-G6, G8, G9 of SURVEY.md section 8(c) for the host-side / order-defining logic.

The reference holds no fixture for any of these functions and cannot be built here (no OpenCV), so these vectors are
produced by the repo's own literal restatement (oracle/) and FREEZE it: the specification stays the cited source lines
(src/SuperPoint.cpp:211-363, src/SPextractor.cpp:219-629, src/LightGlue.cpp:28-51,98-122, src/YoloDetect.cpp:84-208),
the fixture guards the restatement -- and the product's independent implementations, which tests/test_golden_host.py
holds to the same vectors -- against silent change.  The SuperPoint inputs are the ATen-generated prob map of
tests/golden/sp_g2_frame0.npz, so nothing here depends on the oracle's own convolutions.

Run:  python tests/golden/make_golden_host.py        (writes tests/golden/host_g3_g9.npz)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from dani_slam_amd import synth  # noqa: E402
from oracle import lg_oracle, oracle, yolo_oracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def synthetic_preds(seed):
    """A network-free [15120, 85] prediction tensor: mostly background, ~60 boxes in overlapping clusters, a few
    exact score ties and boxes that straddle the image border."""
    rng = np.random.default_rng(seed)
    p = np.zeros((15120, 85), np.float32)
    p[:, 4] = rng.uniform(0.0, 0.2, 15120)
    p[:, 5:] = rng.uniform(0.0, 0.3, (15120, 80))
    p[:, 0] = rng.uniform(0, 640, 15120); p[:, 1] = rng.uniform(0, 380, 15120)
    p[:, 2] = rng.uniform(4, 80, 15120); p[:, 3] = rng.uniform(4, 80, 15120)
    classes = [0, 0, 0, 2, 56, 62, 7, 0, 16, 63, 5, 60]            # person, car, chair, tv, truck, dog, laptop, bus, table
    for c in range(12):
        cx, cy = rng.uniform(20, 620), rng.uniform(20, 360)
        w, h = rng.uniform(40, 260), rng.uniform(40, 220)
        for k in range(5):
            i = rng.integers(0, 15120)
            p[i, 0] = cx + rng.normal(0, 6); p[i, 1] = cy + rng.normal(0, 6)
            p[i, 2] = w * rng.uniform(0.85, 1.15); p[i, 3] = h * rng.uniform(0.85, 1.15)
            p[i, 4] = rng.uniform(0.55, 0.99)
            p[i, 5 + classes[c]] = rng.uniform(0.75, 0.99)
    tie = np.nonzero(p[:, 4] > 0.5)[0][:4]
    p[tie[1], 4] = p[tie[0], 4]; p[tie[1], 5:] = p[tie[0], 5:]       # exact score tie
    return p


def main():
    out = {}
    g2 = np.load(os.path.join(HERE, "sp_g2_frame0.npz"))
    prob = g2["prob"]
    # G3: getKeyPoints + NMS2 of six cells (first, interior, right edge, bottom edge, corner) at both thresholds
    cells = oracle_cells = []
    W, H = 640, 480
    minB, maxBX, maxBY = 16, W - 19 + 3, H - 19 + 3
    width, height = maxBX - minB, maxBY - minB
    ncols, nrows = width // 35, height // 35
    wcell, hcell = int(np.ceil(width / ncols)), int(np.ceil(height / nrows))
    for (i, j) in [(0, 0), (5, 7), (0, ncols - 1), (nrows - 1, 3), (nrows - 1, ncols - 1), (6, ncols - 1)]:
        iniY, iniX = minB + i * hcell, minB + j * wcell
        maxY, maxX = min(iniY + hcell + 6, maxBY), min(iniX + wcell + 6, maxBX)
        for th in (0.015, 0.007):
            k = oracle.cell_keypoints(prob, th, iniX, maxX, iniY, maxY)
            out[f"g3_cell_{i}_{j}_{th}"] = k
        oracle_cells.append((i, j, iniX, maxX, iniY, maxY))
    out["g3_cells"] = np.array(cells, np.int32)
    # G4: all candidates of the frame (no boxes) -> DistributeOctTree(N = 1024), ordered output
    cand = oracle.collect_candidates(prob, W, H, 0.015, 0.007)
    out["g4_candidates"] = cand
    sel = oracle.distribute_octtree(cand, minB, maxBX, minB, maxBY, 1024)
    out["g4_selected"] = sel
    for N in (100, 500):
        out[f"g4_selected_{N}"] = oracle.distribute_octtree(cand, minB, maxBX, minB, maxBY, N)
    # G6: dynamic-box cull at scale 1.0 and 1.2 (the injected box of SURVEY 8d + a YOLO-style pair)
    boxes = [(200, 120, 160, 240), (30, 300, 90, 120)]
    out["g6_boxes"] = np.array(boxes, np.int32)
    out["g6_scale1"] = oracle.collect_candidates(prob, W, H, 0.015, 0.007, boxes, 1.0)
    out["g6_scale12"] = oracle.collect_candidates(prob, W, H, 0.015, 0.007, boxes, 1.2)
    # G8: YOLO decode-free post-processing on a synthetic prediction tensor
    preds = synthetic_preds(11)
    out["g8_preds_idx"] = np.nonzero(preds[:, 4] > 0.3)[0].astype(np.int32)      # the tensor is regenerated from the seed
    dets = yolo_oracle.non_max_suppression(preds, 0.4, 0.5)
    out["g8_dets"] = dets
    dyn, dmap = yolo_oracle.boxes_from_dets(dets, 640, 480)
    out["g8_dynamic"] = np.array(dyn, np.int32).reshape(-1, 4)
    out["g8_detmap_counts"] = np.array([len(dmap.get(n, [])) for n in yolo_oracle.COCO_NAMES], np.int32)
    # G9: normalizeKeypoints (explicit size and size-from-keypoints) and filterMatches known answers
    k = np.array([[0, 0], [639, 479], [320, 240], [12.5, 400.25]], np.float32)
    out["g9_kpts"] = k
    out["g9_norm_size"] = lg_oracle.normalize_keypoints(k, (640, 480))
    out["g9_norm_auto"] = lg_oracle.normalize_keypoints(k, None)
    rng = np.random.default_rng(5)
    sc = np.log(rng.uniform(1e-4, 1.0, (7, 9))).astype(np.float32)
    sc[2, 3] = sc[2, 5] = np.float32(-0.01)          # row tie: first maximum wins
    sc[4, 1] = np.float32(-0.02); sc[6, 1] = np.float32(-0.02)
    out["g9_scores"] = sc
    m, s = lg_oracle.filter_matches(sc, 0.1)
    out["g9_matches"] = m.astype(np.int32); out["g9_match_scores"] = s
    np.savez_compressed(os.path.join(HERE, "host_g3_g9.npz"), **out)
    print("wrote host_g3_g9.npz:", {k_: v.shape for k_, v in out.items()})


if __name__ == "__main__":
    main()
