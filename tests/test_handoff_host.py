"""CPU: SURVEY 8(f) rank 2 -- the product's host restatements of Frame's hand-off formats (dani_host_assign_features_to_grid,
dani_host_features_in_area, dani_host_stereo_from_matches) against oracle/handoff_oracle.py (literal Python restatement of
src/Frame.cc:387-418, 659-737, 862-914)."""
import numpy as np
import pytest

from dani_slam_amd import frontend
from oracle import handoff_oracle as ho

MINX, MINY, MAXX, MAXY = 0.0, 0.0, 640.0, 480.0
INVW, INVH = np.float32(64) / np.float32(MAXX - MINX), np.float32(48) / np.float32(MAXY - MINY)
GRID = (MINX, MINY, float(INVW), float(INVH), 64, 48)


def _keypoints(rng, n):
    k = np.stack([rng.uniform(-8, 648, n), rng.uniform(-8, 488, n)], 1).astype(np.float32)
    k[:20] = np.round(k[:20] / 5) * 5 + 0.0                 # cell-boundary values: round() half cases (x * 0.1 = *.5)
    k[20:30, 0] = (np.arange(10) * 10 + 5).astype(np.float32)   # (x - 0) * 0.1 = k + 0.5 exactly
    return k


def test_assign_features_to_grid_matches_oracle():
    rng = np.random.RandomState(1)
    for n in (0, 1, 300, 1025):
        k = _keypoints(rng, n) if n >= 30 else rng.uniform(0, 400, (n, 2)).astype(np.float32)
        offs, idx = frontend.host_assign_features_to_grid(k, *GRID)
        grid = ho.assign_features_to_grid(k, MINX, MINY, INVW, INVH)
        assert offs[0] == 0 and offs[-1] == len(idx)
        for c in range(64):
            for r in range(48):
                assert idx[offs[c * 48 + r]:offs[c * 48 + r + 1]].tolist() == grid[c][r], (n, c, r)
        inside = sum(len(grid[c][r]) for c in range(64) for r in range(48))
        assert inside == len(idx) and (n < 30 or inside < n)       # some keypoints round outside the grid and are dropped


def test_features_in_area_matches_oracle():
    rng = np.random.RandomState(2)
    k = _keypoints(rng, 1025)
    octave = rng.randint(0, 4, len(k)).astype(np.int32)
    offs, idx = frontend.host_assign_features_to_grid(k, *GRID)
    grid = ho.assign_features_to_grid(k, MINX, MINY, INVW, INVH)
    nq = 400
    qx = rng.uniform(-30, 670, nq).astype(np.float32); qy = rng.uniform(-30, 510, nq).astype(np.float32)
    qr = rng.choice([2.5, 4.0, 15.0, 40.0], nq).astype(np.float32)
    lvl = rng.randint(0, 4, nq).astype(np.int32)
    for mn, mx in ((None, None), (lvl - 1, lvl), (lvl, np.full(nq, -1, np.int32))):
        got = frontend.host_features_in_area(k, octave, GRID, offs, idx, qx, qy, qr, mn, mx)
        for q in range(nq):
            ref = ho.get_features_in_area(k, octave, grid, MINX, MINY, INVW, INVH, qx[q], qy[q], qr[q],
                                          -1 if mn is None else int(mn[q]), -1 if mx is None else int(mx[q]))
            assert got[q].tolist() == ref, q
    assert any(len(g) > 5 for g in got) and any(len(g) == 0 for g in got)


def test_stereo_from_matches_matches_oracle():
    rng = np.random.RandomState(3)
    nl, nr = 500, 480
    ul = rng.uniform(0, 640, nl).astype(np.float32); ur = rng.uniform(0, 640, nr).astype(np.float32)
    for k in (0, 1, 2, 200):
        il = rng.permutation(nl)[:k]; ir = rng.permutation(nr)[:k]
        if k >= 4:
            ur[ir[:k // 2]] = ul[il[:k // 2]] - rng.uniform(0, 60, k // 2).astype(np.float32)    # plausible disparities
            ur[ir[0]] = ul[il[0]]                                                                 # zero disparity -> 0.01
            ur[ir[1]] = ul[il[1]] + 3                                                             # negative: rejected
        m = np.stack([il, ir], 1).astype(np.int32)
        s = rng.uniform(0.05, 1.0, k).astype(np.float32)
        if k >= 4:
            s[2] = s[3]                                                                           # tie in the sort key
        u, d = frontend.host_stereo_from_matches(ul, ur, m, s, 0.1, 40.0)
        ru, rd = ho.stereo_from_matches(ul, ur, m, s, 0.1, 40.0)
        assert np.array_equal(u, ru) and np.array_equal(d, rd), k
        if k == 200:
            assert (d > 0).sum() > 20 and (d[il[1]] == -1) and (u[il[1]] == -1)


# ---- Frame::UndistortKeyPoints (src/Frame.cc:749-782) ----------------------------------------------------------------
TUM1_K = (517.306408, 516.469215, 318.643040, 255.313989)
TUM1_DIST = (0.262383, -0.953104, -0.005358, 0.002628, 1.163314)      # the reference's Examples/RGB-D/TUM1.yaml camera


def _distort(xy_un, K4, dist):
    """Forward radial-tangential model (what undistortion inverts), float64."""
    fx, fy, cx, cy = K4
    k1, k2, p1, p2, k3 = (list(dist) + [0.0] * 5)[:5]
    x = (xy_un[:, 0] - cx) / fx; y = (xy_un[:, 1] - cy) / fy
    r2 = x * x + y * y
    cd = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
    xd = x * cd + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * cd + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return np.stack([fx * xd + cx, fy * yd + cy], 1)


@pytest.mark.parametrize("dist", [TUM1_DIST, TUM1_DIST[:4], (-0.28, 0.07, 0.0002, 0.00002), (1e-3,)])
def test_undistort_keypoints_matches_oracle(dist):
    rng = np.random.default_rng(5)
    k = np.stack([rng.uniform(0, 640, 1025), rng.uniform(0, 480, 1025)], 1).astype(np.float32)
    got = frontend.host_undistort_keypoints(k, TUM1_K, dist)
    ref = ho.undistort_keypoints(k, TUM1_K, dist)
    assert got.dtype == np.float32 and got.shape == k.shape
    assert np.array_equal(got, ref)                                   # same doubles, same order: bit-exact
    if abs(dist[0]) > 0.05:
        assert np.abs(got - k).max() > 0.5                            # it does move points


def test_undistort_keypoints_inverts_the_distortion_model():
    rng = np.random.default_rng(6)
    un = np.stack([rng.uniform(60, 580, 500), rng.uniform(40, 440, 500)], 1)
    dist = (-0.12, 0.03, 0.0004, -0.0003)
    k = _distort(un, TUM1_K, dist).astype(np.float32)
    got = frontend.host_undistort_keypoints(k, TUM1_K, dist)
    assert np.abs(got - un).max() < 2e-2                              # five fixed-point iterations, mild distortion


def test_undistort_keypoints_zero_k1_copies_and_bad_arguments():
    k = np.array([[10.5, 20.25], [600.0, 470.0]], np.float32)
    assert np.array_equal(frontend.host_undistort_keypoints(k, TUM1_K, (0.0, 0.3, 0.1, 0.1)), k)   # the reference's early return
    assert np.array_equal(frontend.host_undistort_keypoints(k, TUM1_K, ()), k)
    assert frontend.host_undistort_keypoints(np.zeros((0, 2), np.float32), TUM1_K, TUM1_DIST).shape == (0, 2)
    with pytest.raises(frontend.DaniError):
        frontend.host_undistort_keypoints(k, (0.0, 500.0, 320.0, 240.0), TUM1_DIST)
    with pytest.raises(frontend.DaniError):
        frontend.host_undistort_keypoints(k, TUM1_K, (0.1,) * 6)
