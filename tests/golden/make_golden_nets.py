#!/usr/bin/env python3
"""Golden vectors for the LightGlue and YOLOv5s oracles, generated with ATen CPU operators (torch 2.10).

Both networks are absent TorchScript blobs in the reference (src/LightGlue.cpp:9, src/YoloDetect.cpp:9), so the
ARCHITECTURE the oracles restate is the published one and stays "parity unpinned".  What this script pins is the
ARITHMETIC: the same architecture is written here a second time with torch ops (F.conv2d, F.silu, F.max_pool2d,
F.interpolate, F.scaled_dot_product_attention, F.layer_norm, F.gelu, log_softmax, logsigmoid), run on the seeded
synthetic weights of dani_slam_amd/synth.py, and the results are committed for tests/test_oracle_golden_nets.py.

Run:  python tests/golden/make_golden_nets.py        (writes tests/golden/lg_g7.npz, yolo_g8.npz)
Nothing under /root/reference is read or executed.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from dani_slam_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)


from oracle.aten_ref import lg_forward, yolo_forward  # noqa: E402  (the ATen restatements live in oracle/aten_ref.py)


def main():
    with torch.no_grad():
        w = synth.lightglue_weights(0)
        k0, d0, k1, d1, _ = synth.matching_pair(21, n0=96, n1=80)
        from oracle import lg_oracle
        k0n, k1n = lg_oracle.normalize_keypoints(k0, (640, 480)), lg_oracle.normalize_keypoints(k1, (640, 480))
        x0, x1, scores = lg_forward(w, k0n, k1n, d0, d1)
        np.savez_compressed(os.path.join(HERE, "lg_g7.npz"), k0n=k0n, k1n=k1n, d0=d0, d1=d1, x0=x0, x1=x1, scores=scores)

        yw = synth.yolo_weights(0)
        from oracle import yolo_oracle
        x = yolo_oracle.preprocess(synth.frame_bgr(1))
        raws = yolo_forward(yw, torch.from_numpy(x)[None])
        np.savez_compressed(os.path.join(HERE, "yolo_g8.npz"), raw0=raws[0][:, ::3, ::3], raw1=raws[1][:, ::2, ::2], raw2=raws[2])
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
