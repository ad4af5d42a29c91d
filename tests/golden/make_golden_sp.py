#!/usr/bin/env python3
"""Generate the SuperPoint golden vectors with the ATen CPU operators the reference calls.

The reference's network is pure LibTorch (/root/reference/src/SuperPoint.cpp:56-99, :246-274); its
arithmetic therefore lives in the third-party dependency `torch` (LibTorch; the reference pins 2.4.1 /
1.9.1 only in Dockerfile comments, Dockerfile:32,53-56).  This script replays the reference's op
sequence -- conv2d+relu, max_pool2d(2,2), norm/div, softmax(dim=1), slice/permute/view depth-to-space,
grid_sampler(bilinear, zeros, align_corners=False), norm over keypoints -- with torch 2.10 CPU in this
container, on the seeded synthetic weights/frames of dani_slam_amd/synth.py, and stores the results as
small fixtures.  tests/test_oracle_golden.py then checks oracle/sp_oracle.cpp against them.

Run:  python tests/golden/make_golden_sp.py      (needs torch; writes tests/golden/sp_*.npz)
Nothing under /root/reference is read or executed by this script.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from dani_slam_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)


from oracle.aten_ref import sp_compute_descriptors as compute_descriptors  # noqa: E402  (the ATen op sequence lives in
from oracle.aten_ref import sp_forward as forward  # noqa: E402                      oracle/aten_ref.py, shared with bench.py)


def main():
    w = synth.superpoint_weights(0)
    with torch.no_grad():
        # G1: 64x96 crop -> every layer (strided samples) + full prob/desc
        img = synth.frame(0)[100:164, 200:296].copy()
        x = torch.from_numpy(img)[None, None].to(torch.float32) / 255
        prob, desc, taps = forward(x, w)
        out = {"img": img, "prob": prob, "desc": desc}
        for i, tp in enumerate(taps):
            flat = tp.reshape(-1)
            out[f"tap{i}_sample"] = flat[::53].copy()
            out[f"tap{i}_shape"] = np.array(tp.shape)
        np.savez_compressed(os.path.join(HERE, "sp_g1_crop.npz"), **out)

        # G2: full 480x640 frame 0 -> prob (full), desc map samples, descriptors at 300 fixed keypoints
        img = synth.frame(0)
        x = torch.from_numpy(img)[None, None].to(torch.float32) / 255
        prob, desc, _ = forward(x, w)
        rng = np.random.RandomState(7)
        kxy = np.stack([rng.randint(16, 624, 300), rng.randint(16, 464, 300)], 1).astype(np.float32)
        d = compute_descriptors(desc, kxy, 480, 640)
        np.savez_compressed(os.path.join(HERE, "sp_g2_frame0.npz"), prob=prob.astype(np.float32),
                            desc_sample=desc.reshape(-1)[::37].copy(), kxy=kxy, descriptors=d)
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
