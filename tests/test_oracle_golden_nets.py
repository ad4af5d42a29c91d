"""CPU: pin the ARITHMETIC of the LightGlue (numpy) and YOLOv5s (C conv + numpy) oracles against golden vectors
generated with ATen operators (tests/golden/make_golden_nets.py).  The architecture itself is the published one and
stays "parity unpinned": both networks are absent TorchScript blobs in the reference."""
import os

import numpy as np

from dani_slam_amd import synth
from oracle import lg_oracle, yolo_oracle


def test_lightglue_oracle_vs_aten(golden_dir):
    g = np.load(os.path.join(golden_dir, "lg_g7.npz"))
    w = synth.lightglue_weights(0)
    m, s, ex = lg_oracle.forward(w, g["k0n"], g["k1n"], g["d0"], g["d1"], want_intermediates=True)
    np.testing.assert_allclose(ex["d0"], g["x0"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(ex["d1"], g["x1"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(ex["scores"], g["scores"], rtol=0, atol=2e-3)      # log-scores down to about -100
    assert np.array_equal(ex["scores"].argmax(1), g["scores"].argmax(1))


def test_yolo_oracle_vs_aten(golden_dir):
    g = np.load(os.path.join(golden_dir, "yolo_g8.npz"))
    yw = synth.yolo_weights(0)
    x = yolo_oracle.preprocess(synth.frame_bgr(1))
    preds, taps = yolo_oracle.network(yw, x, want_taps=True)
    for lvl, sl in ((0, (slice(None), slice(None, None, 3), slice(None, None, 3))),
                    (1, (slice(None), slice(None, None, 2), slice(None, None, 2))), (2, (slice(None),) * 3)):
        ref = g[f"raw{lvl}"]
        got = taps[f"raw{lvl}"][sl]
        assert got.shape == ref.shape
        np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-3)
