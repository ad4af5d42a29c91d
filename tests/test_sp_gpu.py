"""GPU parity tests for the SuperPoint path: HIP engine (through the C ABI) vs the CPU oracle on the same
seeded inputs.  Bar (BASELINE.json north_star): keypoint indices bit-exact, descriptors within 1e-3.
Because the kernels evaluate the canonical f32 order with v_mfma_f32_32x32x2_f32 (an exact fmaf chain),
we additionally assert BIT-EXACT prob maps, activations and descriptors (np.array_equal; +0 == -0)."""
import numpy as np
import pytest

from dani_slam_amd import synth
from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine(sp_weights):
    from dani_slam_amd import frontend
    e = frontend.Engine(device=0, max_batch=4, max_width=640, max_height=480, weights=sp_weights)
    yield e
    e.close()


@pytest.fixture(scope="module")
def oracle_frames(sp_weights):
    out = {}
    for i in (0, 1, 2):
        img = synth.frame(i)
        out[i] = (img, oracle.sp_extract(img, sp_weights, want_maps=True))
    return out


def test_crop_every_layer_bit_exact(engine, sp_weights):
    img = synth.frame(0)[96:96 + 96, 160:160 + 128].copy()      # 96 x 128 crop
    prob_o, desc_o, taps = oracle.superpoint_forward(img, sp_weights, want_taps=True)
    engine.sp_extract_batch(img[None])
    for li, tp in enumerate(taps):
        ref = oracle.maxpool2(tp) if li in (1, 3, 5) else tp    # pooled layers: fused pool output
        got = engine.sp_debug_layer(0, li, ref.shape)
        assert np.array_equal(got, ref), f"layer {li}: max abs diff {np.abs(got - ref).max()}"
    prob, dmap, _ = engine.sp_debug_fetch(0, 96, 128)
    assert np.array_equal(prob, prob_o)
    assert np.array_equal(dmap, desc_o)


def test_full_frame_prob_and_descmap_bit_exact(engine, oracle_frames):
    img, ref = oracle_frames[0]
    engine.sp_extract_batch(img[None])
    prob, dmap, cand = engine.sp_debug_fetch(0, 480, 640)
    assert np.array_equal(prob, ref["prob"])
    assert np.array_equal(dmap, ref["desc_map"])
    cand_ref = oracle.collect_candidates(ref["prob"], 640, 480, 0.015, 0.007)
    assert np.array_equal(cand, cand_ref)


@pytest.mark.parametrize("i", [0, 1, 2])
def test_extract_matches_oracle(engine, oracle_frames, i):
    img, ref = oracle_frames[i]
    r = engine.sp_extract_batch(img[None])[0]
    assert r["kpts"].shape == ref["kpts"].shape
    assert np.array_equal(r["kpts"], ref["kpts"])               # bit-exact keypoint coordinates and order
    assert np.array_equal(r["resp"], ref["resp"])
    assert r["mono_index"] == ref["mono_index"]
    assert np.array_equal(r["octave"], ref["octave"]) and np.array_equal(r["size"], ref["size"])
    np.testing.assert_allclose(r["desc"], ref["desc"], rtol=0, atol=1e-3)   # the stated bar
    assert np.array_equal(r["desc"], ref["desc"])                             # what we actually hold


def test_batch_equals_single(engine, oracle_frames):
    imgs = np.stack([oracle_frames[i][0] for i in (0, 1, 2, 0)])
    rs = engine.sp_extract_batch(imgs)
    for b, i in enumerate((0, 1, 2, 0)):
        ref = oracle_frames[i][1]
        assert np.array_equal(rs[b]["kpts"], ref["kpts"])
        assert np.array_equal(rs[b]["desc"], ref["desc"])


def test_dynamic_boxes_and_lapping(engine, sp_weights, oracle_frames):
    img = oracle_frames[1][0]
    boxes = [(200, 120, 160, 240), (1, 1, 1, 1), (500, 300, 90, 120)]   # incl. the reference's placeholder box
    ref = oracle.sp_extract(img, sp_weights, boxes=boxes, lapping=(0, 300))
    r = engine.sp_extract_batch(img[None], boxes=[np.array(boxes)], lapping=(0, 300))[0]
    assert np.array_equal(r["kpts"], ref["kpts"]) and r["mono_index"] == ref["mono_index"]
    assert np.array_equal(r["desc"], ref["desc"])
    inside = (r["kpts"][:, 0] >= 200) & (r["kpts"][:, 0] < 360) & (r["kpts"][:, 1] >= 120) & (r["kpts"][:, 1] < 360)
    assert not inside.any()
    assert 0 < r["mono_index"] < len(r["kpts"])


def test_other_params_and_thresholds(engine, sp_weights, oracle_frames):
    img = oracle_frames[2][0]
    for nf, ini, mn in [(300, 0.015, 0.007), (2000, 0.02, 0.004), (1024, 0.5, 0.3)]:
        ref = oracle.sp_extract(img, sp_weights, nfeatures=nf, iniTh=ini, minTh=mn)
        r = engine.sp_extract_batch(img[None], nfeatures=nf, iniTh=ini, minTh=mn)[0]
        assert np.array_equal(r["kpts"], ref["kpts"]), (nf, ini, mn)
        assert np.array_equal(r["desc"], ref["desc"])


def test_heads_finished_in_the_conv_epilogue_bit_exact(sp_weights, oracle_frames):
    """An engine built for 8 frames runs convPb + softmax + depth-to-space and convDb + L2 norm + NHWC as ONE kernel each
    (conv_mfma.hip, Cfg::EPI; /root/reference/src/SuperPoint.cpp:73-93) and conv1a inside conv1b: the probability map, the
    normalised descriptor map, every layer the debug hook can rebuild, and the extraction of a whole batch against the oracle."""
    from dani_slam_amd import frontend
    e = frontend.Engine(device=0, max_batch=8, max_width=640, max_height=480, weights=sp_weights)
    try:
        assert e.sp_layer_kernel(9).endswith("+softmax_d2s") and e.sp_layer_kernel(11).endswith("+l2norm_nhwc"), \
            (e.sp_layer_kernel(9), e.sp_layer_kernel(11))
        assert e.sp_layer_kernel(1).endswith("+conv1a") and e.sp_layer_kernel(0) == "(computed inside conv1b)"
        # a 96 x 128 crop (partial tiles on both axes of the 12 x 16 cell grid): every layer, including the two the heads consumed
        img = synth.frame(0)[96:96 + 96, 160:160 + 128].copy()
        prob_o, desc_o, taps = oracle.superpoint_forward(img, sp_weights, want_taps=True)
        e.sp_extract_batch(img[None])
        prob, dmap, _ = e.sp_debug_fetch(0, 96, 128)
        assert np.array_equal(prob, prob_o) and np.array_equal(dmap, desc_o)
        for li, tp in enumerate(taps):
            ref = oracle.maxpool2(tp) if li in (1, 3, 5) else tp
            assert np.array_equal(e.sp_debug_layer(0, li, ref.shape), ref), f"layer {li}"
        # full frames, batch of 8 (frames 0, 1, 2 repeated): maps of two of them, keypoints and descriptors of all
        order = (0, 1, 2, 0, 1, 2, 0, 1)
        rs = e.sp_extract_batch(np.stack([oracle_frames[i][0] for i in order]))
        for b in (1, 7):
            prob, dmap, cand = e.sp_debug_fetch(b, 480, 640)
            ref = oracle_frames[order[b]][1]
            assert np.array_equal(prob, ref["prob"]) and np.array_equal(dmap, ref["desc_map"])
        for b, i in enumerate(order):
            ref = oracle_frames[i][1]
            assert np.array_equal(rs[b]["kpts"], ref["kpts"]) and np.array_equal(rs[b]["desc"], ref["desc"]), b
    finally:
        e.close()


def test_fused_heads_on_small_engines_pyramids_and_odd_sizes(sp_weights, oracle_frames, monkeypatch):
    """DANI_SP_FUSE_HEADS=1 puts the epilogue-fused heads on engines the size rule would leave alone: a four-level pyramid (odd
    level sizes, partial 8 x 16 cell tiles on both axes, boxes and a lapping area) on a one-frame engine, and a 328 x 248 image
    (41 x 31 cells) -- keypoints, octaves, responses and descriptors against the oracle."""
    from dani_slam_amd import frontend
    monkeypatch.setenv("DANI_SP_FUSE_HEADS", "1")
    e = frontend.Engine(device=0, max_batch=1, max_width=640, max_height=480, weights=sp_weights)
    try:
        assert e.sp_layer_kernel(9).endswith("+softmax_d2s") and e.sp_layer_kernel(11).endswith("+l2norm_nhwc")
        img = oracle_frames[1][0]
        boxes = [(200, 120, 160, 240)]
        ref = oracle.sp_extract(img, sp_weights, nfeatures=1000, nlevels=4, boxes=boxes, lapping=(0, 300))
        r = e.sp_extract_batch(img[None], nfeatures=1000, nlevels=4, boxes=[np.array(boxes, np.int32)], lapping=(0, 300))[0]
        for k in ("kpts", "octave", "size", "resp", "desc"):
            assert np.array_equal(r[k], ref[k]), k
        assert r["mono_index"] == ref["mono_index"] and len(np.unique(ref["octave"])) == 4
        small = synth.frame(2)[100:100 + 248, 200:200 + 328].copy()
        ref = oracle.sp_extract(small, sp_weights, nfeatures=400)
        r = e.sp_extract_batch(small[None], nfeatures=400)[0]
        assert np.array_equal(r["kpts"], ref["kpts"]) and np.array_equal(r["desc"], ref["desc"])
    finally:
        e.close()


def test_dense_frames_have_no_candidate_limit(sp_weights):
    """A frame on which EVERY pixel of every cell is a candidate -- a detector head with zero weights gives the flat probability
    map 1/65 = 0.0154 > iniTh = 0.015 -- i.e. 204 cells x up to 1848 pixels with the 6-px overlap strips counted twice (quirk
    Q5): 327 248 candidates where rounds 1-3 answered DANI_ERR_CAPACITY above 32 768.  The reference processes any count
    (vToDistributeKeys grows, /root/reference/src/SPextractor.cpp:248-312); so does the engine, with boxes and in a batch next
    to a second frame."""
    from dani_slam_amd import frontend
    w = dict(sp_weights)
    w["convPb.weight"] = np.zeros_like(w["convPb.weight"]); w["convPb.bias"] = np.zeros_like(w["convPb.bias"])
    e = frontend.Engine(device=0, max_batch=2, max_width=640, max_height=480, weights=w)
    img = synth.frame(1)
    ref = oracle.sp_extract(img, w, nfeatures=1024)
    r = e.sp_extract_batch(img[None], nfeatures=1024)[0]
    prob, _, cand = e.sp_debug_fetch(0, 480, 640, cand_cap=1 << 19)
    assert np.all(prob == np.float32(1.0) / np.float32(65.0))
    # the oracle's own candidate list on that map: every cell pixel except the ones NMS2's out-of-buffer reads suppress at a
    # cell's last rows (quirk Q4: reads past the confidence buffer are 0.0 < 1/65) -- 327 248 of the 361 856 cell pixels
    cells = frontend.host_cell_grid(640, 480)      # (iniX, maxX, iniY, maxY, offX, offY) of the non-skipped cells
    assert int(((cells[:, 1] - cells[:, 0]) * (cells[:, 3] - cells[:, 2])).sum()) == 361856
    want = oracle.collect_candidates(prob, 640, 480, 0.015, 0.007)
    assert len(cand) == len(want) > 300_000 and np.array_equal(cand, want)
    assert np.array_equal(r["kpts"], ref["kpts"]) and np.array_equal(r["resp"], ref["resp"])
    assert np.array_equal(r["desc"], ref["desc"])
    boxes = [(100, 60, 300, 200), (500, 300, 90, 120)]
    refb = oracle.sp_extract(img, w, nfeatures=2000, boxes=boxes)
    ref2 = oracle.sp_extract(synth.frame(2), w, nfeatures=2000)
    two = e.sp_extract_batch(np.stack([img, synth.frame(2)]), nfeatures=2000,
                             boxes=[np.array(boxes, np.int32), np.zeros((0, 4), np.int32)])
    assert np.array_equal(two[0]["kpts"], refb["kpts"]) and np.array_equal(two[0]["desc"], refb["desc"])
    assert np.array_equal(two[1]["kpts"], ref2["kpts"]) and np.array_equal(two[1]["desc"], ref2["desc"])
    assert len({tuple(k) for k in refb["kpts"].tolist()} & {(float(x), float(y)) for x in range(100, 400) for y in range(60, 260)}) == 0
    # the fused call's host-side cull sees the same dense list (single level: candidates are fetched without the boxes)
    wf = dict(w); wf.update(synth.lightglue_weights(0)); wf.update(synth.yolo_weights(0))
    e2 = frontend.Engine(device=0, max_batch=1, weights=wf)
    e2.yolo_set_classes(synth.COCO_NAMES, synth.DYNAMIC_NAMES)
    from oracle import yolo_oracle as yo
    bgr = synth.frame_bgr(1)
    o = e2.frontend_batch(img[None], bgr[None], nfeatures=1024, match_previous=False)[0]
    dyn = yo.detect(wf, bgr)[2]
    refd = oracle.sp_extract(img, w, nfeatures=1024, boxes=dyn)
    assert np.array_equal(o["kpts"], refd["kpts"]) and np.array_equal(o["desc"], refd["desc"])
    e.close(); e2.close()


def test_non_contiguous_stride_and_smaller_image(sp_weights):
    from dani_slam_amd import frontend
    e = frontend.Engine(device=0, max_batch=1, max_width=640, max_height=480, weights=sp_weights)
    big = synth.frame(3)
    roi = big[40:40 + 240, 100:100 + 320]                      # non-contiguous ROI view, stride 640
    ref = oracle.sp_extract(np.ascontiguousarray(roi), sp_weights, nfeatures=500)
    lib = e.lib
    import ctypes as C
    cap = 600
    k = np.empty((cap, 2), np.float32); rr = np.empty(cap, np.float32); o = np.empty(cap, np.int32)
    sz = np.empty(cap, np.float32); d = np.empty((cap, 256), np.float32)
    n = np.zeros(1, np.int32); mono = np.zeros(1, np.int32)
    p = frontend._SpParams(500, 1.2, 1, 0.015, 0.007)
    lap = np.zeros(2, np.int32)
    ptr = C.c_void_p(big.ctypes.data + 40 * 640 + 100)
    rc = lib.dani_sp_extract_batch(e.h, ptr, 0, 1, 320, 240, 640, C.byref(p), None, None, frontend._ip(lap),
                                   frontend._fp(k), frontend._fp(rr), frontend._ip(o), frontend._fp(sz),
                                   frontend._fp(d), cap, frontend._ip(n), frontend._ip(mono))
    assert rc == 0, lib.dani_last_error(e.h)
    assert np.array_equal(k[:n[0]], ref["kpts"]) and np.array_equal(d[:n[0]], ref["desc"])
    e.close()


def test_error_behaviour(engine, sp_weights):
    from dani_slam_amd import frontend
    img = synth.frame(0)
    with pytest.raises(frontend.DaniError) as ei:
        engine.sp_extract_batch(img[None], nlevels=0)
    assert ei.value.code == frontend.DANI_ERR_INVALID
    with pytest.raises(frontend.DaniError) as ei:
        engine.sp_extract_batch(img[None], nlevels=12)          # level 11 would be 86x64: loud error, no fallback
    assert ei.value.code == frontend.DANI_ERR_UNSUPPORTED
    with pytest.raises(frontend.DaniError) as ei:
        engine.sp_extract_batch(img[None], cap=10)
    assert ei.value.code == frontend.DANI_ERR_CAPACITY
    ex = frontend.SPextractor(1024, 1.2, 1, 0.015, 0.007, engine=engine)
    mono, kps, desc = ex(np.zeros((0, 0), np.uint8))
    assert mono == -1 and kps == [] and desc.shape == (0, 256)  # SPextractor.cpp:135-136
    e2 = frontend.Engine(device=0)
    with pytest.raises(frontend.DaniError) as ei:
        e2.sp_extract_batch(img[None])                          # weights never set
    assert ei.value.code == frontend.DANI_ERR_WEIGHTS
    bad = dict(sp_weights); bad["conv3a.weight"] = bad["conv3a.weight"][:, :32]
    with pytest.raises(frontend.DaniError) as ei:
        e2.set_weights(bad)
    assert ei.value.code == frontend.DANI_ERR_WEIGHTS
    e2.close()


@pytest.mark.parametrize("nlevels,nf,boxes,lap", [(4, 1000, [(200, 120, 160, 240)], (0, 300)), (8, 1000, [], (0, 0)),
                                                    (2, 1250, [(10, 10, 300, 200), (400, 300, 100, 100)], (0, 1000))])
def test_pyramid_levels_match_oracle(engine, sp_weights, oracle_frames, nlevels, nf, boxes, lap):
    """nLevels > 1: ComputePyramid (fixed-point cv::resize chain), per-level network / cells / quadtree / descriptors,
    the scaled dynamic-box cull with its f32 drift (Q6), octave / size fields and the cross-level mono/stereo partition."""
    img = oracle_frames[1][0]
    ref = oracle.sp_extract(img, sp_weights, nfeatures=nf, nlevels=nlevels, boxes=boxes, lapping=lap)
    r = engine.sp_extract_batch(img[None], nfeatures=nf, nlevels=nlevels, boxes=[np.array(boxes, np.int32).reshape(-1, 4)],
                                lapping=lap)[0]
    assert len(r["kpts"]) == len(ref["kpts"]) and len(ref["kpts"]) > nf // 2
    assert np.array_equal(r["octave"], ref["octave"]) and len(np.unique(ref["octave"])) == nlevels
    assert np.array_equal(r["size"], ref["size"])
    assert np.array_equal(r["kpts"], ref["kpts"])
    assert np.array_equal(r["resp"], ref["resp"])
    assert r["mono_index"] == ref["mono_index"]
    assert np.array_equal(r["desc"], ref["desc"])


def test_reference_style_wrapper(engine, oracle_frames):
    from dani_slam_amd import frontend
    img, ref = oracle_frames[0]
    ex = frontend.SPextractor(1024, 1.2, 1, 0.015, 0.007, engine=engine)
    assert ex.GetLevels() == 1 and abs(ex.GetScaleFactor() - 1.2) < 1e-6
    mono, kps, desc = ex(img, None, (0, 0))
    assert mono == ref["mono_index"] and len(kps) == len(ref["kpts"])
    assert all(k.angle == -1.0 and k.size == 31.0 and k.octave == 0 for k in kps)   # quirk Q15
    assert np.array_equal(np.array([[k.x, k.y] for k in kps], np.float32), ref["kpts"])


def test_flat_image_no_crash(engine, sp_weights):
    img = np.full((480, 640), 128, np.uint8)
    ref = oracle.sp_extract(img, sp_weights)
    r = engine.sp_extract_batch(img[None])[0]
    assert np.array_equal(r["kpts"], ref["kpts"])


def test_two_threads_like_the_stereo_frame_constructor(sp_weights, oracle_frames):
    """Frame::Frame(imL, imR) runs two extractors concurrently from two std::threads (src/Frame.cc:124-127): the ABI is
    re-entrant per engine (private stream + workspace per call)."""
    import threading
    from dani_slam_amd import frontend
    e = frontend.Engine(device=0, max_batch=1, n_contexts=2, weights=sp_weights)
    left = frontend.SPextractor(1024, 1.2, 1, 0.015, 0.007, engine=e)
    right = frontend.SPextractor(1024, 1.2, 1, 0.015, 0.007, engine=e)
    out = {}

    def run(name, ex, img):
        for _ in range(3):
            out[name] = ex(img, None, (0, 0))
    ts = [threading.Thread(target=run, args=("L", left, oracle_frames[0][0])),
          threading.Thread(target=run, args=("R", right, oracle_frames[1][0]))]
    for t in ts: t.start()
    for t in ts: t.join()
    for name, i in (("L", 0), ("R", 1)):
        mono, kps, desc = out[name]
        ref = oracle_frames[i][1]
        assert np.array_equal(np.array([[k.x, k.y] for k in kps], np.float32), ref["kpts"])
        assert np.array_equal(desc, ref["desc"])
    e.close()


@pytest.mark.parametrize("w,h", [(752, 480), (1280, 720), (328, 248)])
def test_other_image_sizes(sp_weights, w, h):
    """EuRoC (752x480), 720p and a small non-multiple-of-32 size."""
    from dani_slam_amd import frontend
    rng = np.random.RandomState(w)
    base = synth.frame(5)
    img = np.ascontiguousarray(np.tile(base, (2, 2))[:h, :w])
    img = (img.astype(np.int32) + rng.randint(-3, 4, img.shape)).clip(0, 255).astype(np.uint8)
    e = frontend.Engine(device=0, max_batch=1, max_width=w, max_height=h, weights=sp_weights)
    ref = oracle.sp_extract(img, sp_weights, nfeatures=1500)
    r = e.sp_extract_batch(img[None], nfeatures=1500)[0]
    assert np.array_equal(r["kpts"], ref["kpts"]) and np.array_equal(r["desc"], ref["desc"])
    e.close()


def test_more_callers_than_contexts_wait_instead_of_failing(sp_weights):
    """Four host threads on an engine with two contexts (the reference runs two extractors concurrently,
    src/Frame.cc:124-127, and its mapping threads construct matchers next to Tracking): a call that finds every context
    taken waits for one, it no longer fails with 'all contexts busy'; every thread gets the single-threaded result."""
    import threading
    from dani_slam_amd import frontend
    e = frontend.Engine(device=0, max_batch=1, n_contexts=2, weights=sp_weights)
    imgs = [synth.frame(i) for i in range(4)]
    want = [e.sp_extract_batch(im[None])[0] for im in imgs]
    got, errs = [None] * 4, []

    def worker(i):
        try:
            for _ in range(6):
                got[i] = e.sp_extract_batch(imgs[i][None])[0]
        except Exception as ex:        # surfaces in the main thread below
            errs.append(ex)
    th = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in th: t.start()
    for t in th: t.join()
    assert not errs, errs
    for a, b in zip(got, want):
        assert np.array_equal(a["kpts"], b["kpts"]) and np.array_equal(a["desc"], b["desc"])
    e.close()
