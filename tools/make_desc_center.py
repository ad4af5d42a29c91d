#!/usr/bin/env python3
"""Calibration constant of the synthetic SuperPoint weights' `centered_desc` variant (dani_slam_amd/synth.py).

With He-uniform random weights the raw descriptor head output convDb(relu(convDa(x))) has a large position-independent
component per channel (mean |mean| 0.55 against a spatial standard deviation of 0.20 on the synthetic frames): after the
per-pixel L2 normalisation every descriptor points almost the same way (cosine 0.85 between DIFFERENT keypoints) and
LightGlue finds a handful of matches between two views of the same scene.  A trained network has no such offset.  This
script measures the per-channel mean of the raw convDb output over synthetic frames 0..3 with the ATen CPU operators and
stores it; `superpoint_weights(seed, centered_desc=True)` subtracts it from convDb.bias (cosine between different
keypoints 0.10, ~400 matches for an 8-px shifted view, ~100 for the 2-degree / 8-px warp of SURVEY.md section 8d).
The detector head is untouched, so keypoints are identical in both variants.

Run:  python tools/make_desc_center.py     (needs torch; writes dani_slam_amd/data/sp_desc_center_seed0.npy, 1 KiB)
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from dani_slam_amd import synth  # noqa: E402


def raw_desc(img, t):
    x = torch.from_numpy(img)[None, None].to(torch.float32) / 255
    c = lambda n, x, p: F.conv2d(x, t[n + ".weight"], t[n + ".bias"], padding=p)
    for a, b in (("conv1a", "conv1b"), ("conv2a", "conv2b"), ("conv3a", "conv3b")):
        x = F.max_pool2d(torch.relu(c(b, torch.relu(c(a, x, 1)), 1)), 2, 2)
    x = torch.relu(c("conv4b", torch.relu(c("conv4a", x, 1)), 1))
    return c("convDb", torch.relu(c("convDa", x, 1)), 0)[0]


def main(seed=0):
    torch.set_num_threads(8)
    w = synth.superpoint_weights(seed)
    t = {k: torch.from_numpy(v) for k, v in w.items()}
    acc = np.zeros(256, np.float64)
    with torch.no_grad():
        for i in range(4):
            acc += raw_desc(synth.frame(i), t).to(torch.float64).mean(dim=(1, 2)).numpy()
    out = os.path.join(ROOT, "dani_slam_amd", "data", f"sp_desc_center_seed{seed}.npy")
    np.save(out, (acc / 4).astype(np.float32))
    print("wrote", out, "mean |centre|", float(np.abs(acc / 4).mean()))


if __name__ == "__main__":
    main()
