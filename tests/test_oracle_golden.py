"""CPU: pin oracle/sp_oracle.cpp against the ATen-generated golden vectors (tests/golden/make_golden_sp.py).

The oracle evaluates in the repo's canonical f32 order; ATen's order is opaque, so
agreement is to f32 rounding (tolerances written below), not bitwise.
"""
import os

import numpy as np

from dani_slam_amd import synth
from oracle import oracle


def test_expf_accuracy():
    xs = np.concatenate([np.linspace(-87, 0, 20001), -np.logspace(-8, 1.9, 2000)]).astype(np.float32)
    got = np.array([oracle.lib().orc_expf(float(x)) for x in xs], np.float32)
    ref = np.exp(xs.astype(np.float64))
    rel = np.abs(got - ref) / ref
    assert rel.max() < 2.5e-7, rel.max()          # ~2 ulp
    assert oracle.lib().orc_expf(0.0) == 1.0
    assert oracle.lib().orc_expf(-1000.0) > 0.0   # clamped, stays normal


def test_g1_crop_all_layers(sp_weights, golden_dir):
    g = np.load(os.path.join(golden_dir, "sp_g1_crop.npz"))
    prob, desc, taps = oracle.superpoint_forward(g["img"], sp_weights, want_taps=True)
    for i, tp in enumerate(taps):
        assert tuple(g[f"tap{i}_shape"]) == tp.shape
        ref = g[f"tap{i}_sample"]
        got = tp.reshape(-1)[::53]
        # f32 accumulation-order tolerance: K<=1152 products of O(1) terms
        np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-5, err_msg=f"layer {i}")
    np.testing.assert_allclose(prob, g["prob"], rtol=1e-3, atol=1e-7)
    np.testing.assert_allclose(desc, g["desc"], rtol=0, atol=1e-5)


def test_g2_full_frame(sp_weights, golden_dir):
    g = np.load(os.path.join(golden_dir, "sp_g2_frame0.npz"))
    img = synth.frame(0)
    prob, desc = oracle.superpoint_forward(img, sp_weights)
    assert prob.shape == (480, 640) and desc.shape == (256, 60, 80)
    np.testing.assert_allclose(prob, g["prob"], rtol=2e-3, atol=1e-7)
    np.testing.assert_allclose(desc.reshape(-1)[::37], g["desc_sample"], rtol=0, atol=2e-5)
    # threshold decisions: only pixels within f32 noise of the threshold may differ from ATen
    for th in (0.015, 0.007):
        a, b = prob > th, g["prob"] > th
        diff = a != b
        assert diff.sum() <= 8, diff.sum()
        assert np.all(np.abs(g["prob"][diff] - th) < 2e-5)
    # descriptors (grid_sampler + norm over keypoints, Q7), tolerance 1e-3 per north_star -- we hold 1e-5
    d = oracle.compute_descriptors(desc, g["kxy"], 480, 640)
    np.testing.assert_allclose(d, g["descriptors"], rtol=0, atol=1e-5)
    # Q7: columns (channels) have unit norm over the keypoints, rows do not
    np.testing.assert_allclose(np.linalg.norm(d.astype(np.float64), axis=0), 1.0, atol=1e-5)
