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


def test_lightglue_oracle_vs_aten_three_sizes(golden_dir):
    """G7 at N = M in {7, 128, 1024}: final descriptors, best log-score per row and both arg-max
    directions against the ATen restatement (tests/golden/make_golden_lg_sizes.py)."""
    g = np.load(os.path.join(golden_dir, "lg_g7_sizes.npz"))
    w = synth.lightglue_weights(0)
    for n in g["sizes"].tolist():
        k0, d0, k1, d1, _ = synth.matching_pair(100 + n, n0=n, n1=n)
        k0n, k1n = lg_oracle.normalize_keypoints(k0, (640, 480)), lg_oracle.normalize_keypoints(k1, (640, 480))
        m, s, ex = lg_oracle.forward(w, k0n, k1n, d0, d1, want_intermediates=True)
        st = 1 if n < 64 else 8
        np.testing.assert_allclose(ex["d0"][::st], g[f"n{n}_x0"], rtol=0, atol=5e-5)
        np.testing.assert_allclose(ex["d1"][::st], g[f"n{n}_x1"], rtol=0, atol=5e-5)
        sc = ex["scores"]
        tol = 2e-3 * max(1.0, float(np.abs(g[f"n{n}_rowmax"]).max()) / 50.0)
        np.testing.assert_allclose(sc.max(1), g[f"n{n}_rowmax"], rtol=0, atol=tol)
        np.testing.assert_allclose(sc[::st, ::st], g[f"n{n}_scores"], rtol=2e-5, atol=tol)
        # arg-max may differ only where ATen's own top-2 margin is inside the tolerance
        for got, ref, axis in ((sc.argmax(1), g[f"n{n}_rowarg"], 1), (sc.argmax(0), g[f"n{n}_colarg"], 0)):
            for i in np.nonzero(got != ref)[0]:
                line = sc[i] if axis == 1 else sc[:, i]
                top = np.sort(line)[::-1]
                assert top[0] - top[1] < 2 * tol, (n, axis, int(i))
