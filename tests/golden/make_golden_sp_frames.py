#!/usr/bin/env python3
"""Multi-frame SuperPoint golden vectors (G10) and the 96x128 crop G1b, generated with the ATen CPU operators.

SURVEY.md section 7 ("hard parts") asks for the count of threshold-margin flips between an independent f32
implementation and LibTorch-CPU over MANY frames, and for the keypoint-set differences they cause.  For synthetic frames
0..15 (dani_slam_amd/synth.py, seeded weights) this script stores, per frame,
  * the bit masks `prob > 0.015` and `prob > 0.007` of ATen's prob map (np.packbits, 38 KiB each before compression),
  * every pixel of ATen's prob map within 1e-4 of either threshold (flat index + value): the only pixels whose decision
    may legitimately differ under another f32 summation order,
  * the keypoints the reference's host logic (per-cell getKeyPoints + NMS2, DistributeOctTree; literal restatement in
    oracle/sp_oracle.cpp) selects FROM ATEN'S prob map, N = 1024: what LibTorch-CPU + the reference host code would return,
  * for frames 0..3, ATen's descriptors (grid_sampler + norm over keypoints, quirk Q7) at every 4th of those keypoints.
tests/test_oracle_golden.py holds the oracle to them on the CPU, tests/test_aten_fixtures_gpu.py the HIP path on MI355X.

G1b is G1 (every layer of the network, strided samples) on a 96x128 crop: the engine refuses inputs under 80x80, so the
64x96 crop of G1 cannot be run on the GPU.

Run:  python tests/golden/make_golden_sp_frames.py    (needs torch; writes sp_g10_frames.npz, sp_g1b_crop.npz)
Nothing under /root/reference is read or executed.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
from dani_slam_amd import synth  # noqa: E402
from make_golden_sp import compute_descriptors, forward  # noqa: E402
from oracle import oracle  # noqa: E402

NFRAMES = 16
W, H, N = 640, 480, 1024
MARGIN = 1e-4


def host_keypoints(prob):
    """ComputeKeyPointsOctTree's host part on a given prob map: candidates of the 204 cells -> quadtree -> + border."""
    cand = oracle.collect_candidates(prob, W, H, 0.015, 0.007)
    sel = oracle.distribute_octtree(cand, 16, W - 16, 16, H - 16, N)
    return (sel[:, :2] + np.float32(16)).astype(np.float32), sel[:, 2].copy(), len(cand)


def main():
    w = synth.superpoint_weights(0)
    out = {"nframes": np.array(NFRAMES), "margin": np.array(MARGIN)}
    with torch.no_grad():
        for i in range(NFRAMES):
            img = synth.frame(i)
            x = torch.from_numpy(img)[None, None].to(torch.float32) / 255
            prob, desc, _ = forward(x, w)
            prob = np.ascontiguousarray(prob, np.float32)
            for tag, th in (("015", 0.015), ("007", 0.007)):
                out[f"f{i}_mask{tag}"] = np.packbits(prob > np.float32(th))
                near = np.nonzero(np.abs(prob.astype(np.float64) - th).reshape(-1) < MARGIN)[0].astype(np.int32)
                out[f"f{i}_near{tag}_idx"] = near
                out[f"f{i}_near{tag}_val"] = prob.reshape(-1)[near].copy()
            kp, resp, ncand = host_keypoints(prob)
            out[f"f{i}_kpts"] = kp
            out[f"f{i}_resp"] = resp
            out[f"f{i}_ncand"] = np.array(ncand)
            if i < 4:
                d = compute_descriptors(np.ascontiguousarray(desc), kp, H, W)
                out[f"f{i}_desc_rows4"] = d[::4].copy()
            print(f"frame {i}: {int((prob > 0.015).sum())} px > 0.015, {ncand} candidates, {len(kp)} keypoints, "
                  f"{len(out[f'f{i}_near015_idx'])}+{len(out[f'f{i}_near007_idx'])} px within {MARGIN} of a threshold", flush=True)
        np.savez_compressed(os.path.join(HERE, "sp_g10_frames.npz"), **out)

        img = synth.frame(0)[96:96 + 96, 160:160 + 128].copy()
        x = torch.from_numpy(img)[None, None].to(torch.float32) / 255
        prob, desc, taps = forward(x, w)
        g = {"img": img, "prob": prob, "desc": desc}
        for i, tp in enumerate(taps):
            g[f"tap{i}_sample"] = tp.reshape(-1)[::53].copy()
            g[f"tap{i}_shape"] = np.array(tp.shape)
        np.savez_compressed(os.path.join(HERE, "sp_g1b_crop.npz"), **g)
    for f in ("sp_g10_frames.npz", "sp_g1b_crop.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
