"""GPU: SURVEY.md section 8(f) without the PCIe bounce -- a frame's features stay in HBM after extraction
(dani_frontend_retain) and the consumers run on them there:
  * Frame::AssignFeaturesToGrid built on the device          vs oracle/handoff_oracle.py (src/Frame.cc:387-418)
  * Frame::GetFeaturesInArea + the best / second-best scan    vs handoff_oracle + match_oracle (src/Frame.cc:659-725,
    (dani_search_windows_resident)                               src/LGmatcher.cpp:84-120, 2009-2013)
  * the scan over host-built lists, resident train set        vs match_oracle
  * Frame::ComputeBoW on the resident descriptors             vs oracle/bow_oracle.py (src/Frame.cc:740-747)"""
import numpy as np
import pytest

from dani_slam_amd import synth
from oracle import bow_oracle as bo
from oracle import handoff_oracle as ho
from oracle import match_oracle as mo

pytestmark = pytest.mark.gpu
GRID = (0.0, 0.0, 64.0 / 640.0, 48.0 / 480.0, 64, 48)      # mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv, cols, rows


def _csr(grid):
    """oracle grid[col][row] lists -> CSR over cell = col * rows + row (the layout of the product's grid)."""
    offs = [0]; idx = []
    for col in grid:
        for cell in col:
            idx += cell; offs.append(len(idx))
    return np.array(offs, np.int32), np.array(idx, np.int32)


@pytest.fixture(scope="module")
def state():
    from dani_slam_amd import frontend
    w = synth.bench_weights()
    e = frontend.Engine(device=0, max_batch=2, weights=w)
    gray, _ = synth.bench_frames([0, 1])
    out = e.frontend_batch(gray, None, nfeatures=1000, nlevels=3, scaleFactor=1.2, use_yolo=False, match_previous=False)
    out = [{k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in o.items()} for o in out]
    octs = e.sp_extract_batch(gray, nfeatures=1000, nlevels=3, scaleFactor=1.2)       # the octaves (the fused call does not return them)
    for o, x in zip(out, octs):
        assert np.array_equal(o["kpts"], x["kpts"])
        o["octave"] = x["octave"]
    yield e, out
    e.close()


def test_grid_built_on_the_device(state):
    e, out = state
    f = e.frontend_retain(1, octave=out[1]["octave"], grid=GRID)
    assert e.features_count(f) == len(out[1]["kpts"])
    offs, idx = e.features_fetch_grid(f, 64 * 48, len(out[1]["kpts"]))
    ro, ri = _csr(ho.assign_features_to_grid(out[1]["kpts"], *GRID[:4], 64, 48))
    assert np.array_equal(offs, ro) and np.array_equal(idx, ri)
    e.features_release(f)
    # undistorted coordinates override the extractor's; keypoints pushed outside the grid are dropped
    un = out[0]["kpts"] + np.float32(37.25)
    f = e.frontend_retain(0, kpts_un=un, grid=GRID)
    offs, idx = e.features_fetch_grid(f, 64 * 48, len(un))
    ro, ri = _csr(ho.assign_features_to_grid(un, *GRID[:4], 64, 48))
    assert np.array_equal(offs, ro) and np.array_equal(idx, ri) and len(idx) < len(un)
    e.features_release(f)


def test_undistort_on_the_device_equals_the_host_function(state):
    """Frame::UndistortKeyPoints (src/Frame.cc:749-782) on the retained coordinates: bit-identical to
    dani_host_undistort_keypoints (five fixed-point iterations in double), the grid rebuilt from mvKeysUn; the k1 == 0
    calibration copies, as the reference does."""
    from dani_slam_amd import frontend
    e, out = state
    K4 = (517.3, 516.5, 318.6, 255.3); dist = (0.2624, -0.9531, -0.0054, 0.0026, 1.1633)      # TUM1-like
    f = e.frontend_retain(0, octave=out[0]["octave"])
    un = e.features_undistort(f, K4, dist, grid=GRID)
    ref = frontend.host_undistort_keypoints(out[0]["kpts"], K4, dist)
    assert np.array_equal(un, ref) and np.abs(un - out[0]["kpts"]).max() > 1.0
    offs, idx = e.features_fetch_grid(f, 64 * 48, len(un))
    ro, ri = _csr(ho.assign_features_to_grid(ref, *GRID[:4], 64, 48))
    assert np.array_equal(offs, ro) and np.array_equal(idx, ri)
    e.features_release(f)
    f = e.frontend_retain(0)
    assert np.array_equal(e.features_undistort(f, K4, (0.0, 0.3, 0.0, 0.0)), out[0]["kpts"])     # k1 == 0: copied
    assert np.array_equal(e.features_undistort(f, K4, dist[:4]), frontend.host_undistort_keypoints(out[0]["kpts"], K4, dist[:4]))
    e.features_release(f)


def test_window_search_and_scan_on_resident_features(state):
    """Map points projected into frame 1: windows around (perturbed) keypoints of frame 0's scene, radii as
    SearchByProjection computes them, level ranges as its predicted-octave test (src/LGmatcher.cpp:60-82)."""
    e, out = state
    kp, desc, octv = out[1]["kpts"], out[1]["desc"], out[1]["octave"]
    f = e.frontend_retain(1, octave=octv, grid=GRID)
    rng = np.random.RandomState(4)
    nq = 1500
    src = rng.randint(0, len(kp), nq)
    qx = (kp[src, 0] + rng.uniform(-6, 6, nq)).astype(np.float32); qy = (kp[src, 1] + rng.uniform(-6, 6, nq)).astype(np.float32)
    qx[:20] = rng.uniform(-50, 700, 20); qy[:20] = rng.uniform(-50, 530, 20)          # windows partly or wholly off the image
    qr = rng.choice([4.0, 7.5, 15.0, 40.0], nq).astype(np.float32)
    minl = rng.choice([-1, 0, 1], nq).astype(np.int32); maxl = np.where(rng.rand(nq) < 0.5, -1, minl + 1).astype(np.int32)
    qd = (desc[src] + rng.standard_normal((nq, 256)).astype(np.float32) * np.float32(0.01)).astype(np.float32)
    best, lists, dists = e.search_windows_resident(f, qd, qx, qy, qr, minl, maxl)
    grid = ho.assign_features_to_grid(kp, *GRID[:4], 64, 48)
    want = [ho.get_features_in_area(kp, octv, grid, *GRID[:4], float(qx[i]), float(qy[i]), float(qr[i]), int(minl[i]), int(maxl[i])) for i in range(nq)]
    assert sum(len(x) for x in want) > 5000
    for i in range(nq):
        assert lists[i].tolist() == list(want[i]), i                                   # same candidates, same order
    ref = mo.best2_scan(qd, desc, octv, want)
    for i, r in enumerate(ref):
        assert (best[0][i], best[1][i], best[2][i], best[3][i], best[4][i]) == (r[0], r[1], r[2], r[3], r[4]), i
    for i in range(0, nq, 41):
        for k, idx in enumerate(want[i]):
            assert dists[i][k] == mo.descriptor_distance(qd[i], desc[idx])
    # the same scan over host-built lists with the resident train set, and without level arrays
    best2, dist2 = e.desc_scan_resident(qd, f, want)
    assert all(np.array_equal(a, b) for a, b in zip(best, best2))
    best3, lists3, _ = e.search_windows_resident(f, qd[:50], qx[:50], qy[:50], qr[:50])
    want3 = [ho.get_features_in_area(kp, octv, grid, *GRID[:4], float(qx[i]), float(qy[i]), float(qr[i])) for i in range(50)]
    assert [l.tolist() for l in lists3] == [list(x) for x in want3]
    from dani_slam_amd import frontend
    with pytest.raises(frontend.DaniError):
        e.search_windows_resident(f, qd, qx, qy, qr, minl, maxl, cand_cap=100)         # the lists do not fit: capacity error
    e.features_release(f)
    g = e.frontend_retain(1)                                                            # no grid -> the window search is refused
    with pytest.raises(frontend.DaniError):
        e.search_windows_resident(g, qd[:2], qx[:2], qy[:2], qr[:2])
    e.features_release(g)


def test_bow_transform_on_resident_descriptors(state):
    e, out = state
    tree = synth.vocab_tree(21, k=10, L=3)
    v = e.vocab_create(tree["L"], tree["child_offsets"], tree["child_ids"], tree["desc"], tree["weight"], tree["word_id"])
    f = e.frontend_retain(0)
    bow, fv = e.bow_transform_resident(v, f, 4, 0, 1)
    rbow, rfv = bo.transform(tree, out[0]["desc"], 4, 0, 1)
    assert list(bow.keys()) == list(rbow.keys()) and fv == rfv and all(bow[w] == rbow[w] for w in bow)
    assert (bow, fv) == e.bow_transform(v, out[0]["desc"], 4, 0, 1)
    e.features_release(f); e.vocab_destroy(v)
