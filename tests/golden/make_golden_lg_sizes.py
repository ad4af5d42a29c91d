#!/usr/bin/env python3
"""LightGlue golden vectors G7 at N = M in {7, 128, 1024}, generated with ATen CPU operators.

tests/golden/lg_g7.npz (make_golden_nets.py) pins the oracle's arithmetic at one size (96 x 80).  This script runs the
same ATen restatement of the public architecture (lg_forward of make_golden_nets.py: F.linear,
scaled_dot_product_attention, layer_norm, gelu, log_softmax, logsigmoid) on synth.matching_pair(100 + N, N, N) and keeps,
per size: the final descriptors of every 8th keypoint (all of them for N = 7), the row-wise best log-score and its
column, the column-wise best row, and a strided sample of the score matrix.  Inputs are regenerated from the seed.
The network itself is the absent TorchScript blob of the reference (src/LightGlue.cpp:9): architecture parity unpinned.

Run:  python tests/golden/make_golden_lg_sizes.py     (needs torch; writes tests/golden/lg_g7_sizes.npz)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
from dani_slam_amd import synth  # noqa: E402
from make_golden_nets import lg_forward  # noqa: E402
from oracle import lg_oracle  # noqa: E402

SIZES = (7, 128, 1024)


def main():
    w = synth.lightglue_weights(0)
    out = {"sizes": np.array(SIZES)}
    with torch.no_grad():
        for n in SIZES:
            k0, d0, k1, d1, _ = synth.matching_pair(100 + n, n0=n, n1=n)
            k0n, k1n = lg_oracle.normalize_keypoints(k0, (640, 480)), lg_oracle.normalize_keypoints(k1, (640, 480))
            x0, x1, sc = lg_forward(w, k0n, k1n, d0, d1)
            st = 1 if n < 64 else 8
            out[f"n{n}_x0"] = x0[::st].copy(); out[f"n{n}_x1"] = x1[::st].copy()
            out[f"n{n}_rowmax"] = sc.max(1); out[f"n{n}_rowarg"] = sc.argmax(1).astype(np.int32)
            out[f"n{n}_colarg"] = sc.argmax(0).astype(np.int32)
            out[f"n{n}_scores"] = sc[::st, ::st].copy()
            print(n, "row max range", float(sc.max(1).min()), float(sc.max(1).max()))
    p = os.path.join(HERE, "lg_g7_sizes.npz")
    np.savez_compressed(p, **out)
    print("wrote", p, os.path.getsize(p), "bytes")


if __name__ == "__main__":
    main()
