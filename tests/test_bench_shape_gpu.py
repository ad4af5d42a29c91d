"""GPU: the exact workload bench.py times -- dani_frontend_batch over B = 8 device-resident 640x480 frames of the
moving-camera sequence, nFeatures = 1024, YOLO on, two consecutive calls -- against the composition of the three
oracles.  This is the shape at which the single-level host cull (SpCall::host_cull), the 257th token tile of the linear
layers (linear_rs_kernel's tail workgroups), the tail-token paths and the 128-query attention kernels (attn_rs_kernel<8>,
attn_x3w_kernel) run in situ; the
smaller fused-call tests (tests/test_frontend_gpu.py, nFeatures = 300, B = 2) never reach them."""
import numpy as np
import pytest

from dani_slam_amd import synth
from oracle import lg_oracle as lg
from oracle import oracle
from oracle import yolo_oracle as yo

pytestmark = pytest.mark.gpu
B, NF = 8, 1024
SIZE = (640, 480)
CHECK_PAIRS = (1, 5, 8, 9, 15)      # frame index of the pair's second frame; 8 = first frame of call 2 vs last of call 1


def _match_dict(o):
    return {tuple(x): float(v) for x, v in zip(np.asarray(o["matches"]).tolist(), np.asarray(o["scores"]))}


def _check_matches(got, mo, so, tag):
    want = {tuple(x): float(v) for x, v in zip(mo.tolist(), so)}
    assert len(want) >= 40, (tag, len(want))                                   # the comparison is not vacuous
    assert len(set(got) ^ set(want)) <= max(2, len(want) // 50), (tag, len(got), len(want), len(set(got) ^ set(want)))
    common = set(got) & set(want)
    assert max(abs(got[k] - want[k]) for k in common) < 1e-3, tag


def test_bench_workload_two_calls_match_oracle_composition():
    import torch
    from dani_slam_amd import frontend
    w = synth.bench_weights()
    gray, bgr = synth.bench_frames(range(2 * B))
    d_gray = torch.from_numpy(gray).cuda(); d_bgr = torch.from_numpy(bgr).cuda()
    e = frontend.Engine(device=0, max_batch=B, weights=w)
    e.yolo_set_classes(synth.COCO_NAMES, synth.DYNAMIC_NAMES)

    def two_calls():
        outs = []
        for c in range(2):
            o = e.frontend_batch(None, None, nfeatures=NF, device_ptrs=(d_gray[c * B:].data_ptr(), d_bgr[c * B:].data_ptr()),
                                 shape=(B, 480, 640), want_desc=True, copy=True)
            outs += o
        return outs

    out = two_calls()
    refs = []
    for i in range(2 * B):
        ok, dets, dyn, _ = yo.detect(w, bgr[i])
        assert [tuple(int(v) for v in r) for r in out[i]["boxes"]] == [tuple(r) for r in dyn], i
        ref = oracle.sp_extract(gray[i], w, nfeatures=NF, boxes=dyn)
        assert np.array_equal(out[i]["kpts"], ref["kpts"]), i                   # bit-exact keypoints, in order
        assert np.array_equal(out[i]["resp"], ref["resp"]), i
        assert np.array_equal(out[i]["desc"], ref["desc"]), i                   # bit-exact descriptors
        refs.append(ref)
    assert len(out[0]["matches"]) == 0                                          # no predecessor
    assert min(len(o["matches"]) for o in out[1:]) >= 40, [len(o["matches"]) for o in out]
    oracle_matches = {}
    for i in CHECK_PAIRS:
        mo, so = lg.match_descriptors(w, refs[i - 1]["kpts"], refs[i - 1]["desc"], refs[i]["kpts"], refs[i]["desc"], SIZE, SIZE, 0.05)
        oracle_matches[i] = (mo, so)
        _check_matches(_match_dict(out[i]), mo, so, f"f32 pair {i - 1}->{i}")

    # the opt-in bf16x3 arithmetic on the same workload: against the oracle and against the f32 mode
    e.lg_set_gemm(e.LG_GEMM_BF16X3_ALL)
    e.frontend_reset()
    out3 = two_calls()
    e.lg_set_gemm(e.LG_GEMM_F32)
    for i in range(1, 2 * B):
        a, b = _match_dict(out[i]), _match_dict(out3[i])
        assert len(set(a) ^ set(b)) <= 2, (i, len(a), len(b))
        assert max(abs(a[k] - b[k]) for k in set(a) & set(b)) < 1e-4, i
        assert np.array_equal(out3[i]["kpts"], out[i]["kpts"]) and np.array_equal(out3[i]["desc"], out[i]["desc"])
    for i in CHECK_PAIRS:
        _check_matches(_match_dict(out3[i]), *oracle_matches[i], f"bf16x3_all pair {i - 1}->{i}")
    e.close()


def test_bench_shape_with_culling_boxes():
    """Same shape on the stand-alone synthetic frames: frame 1 carries 18 dynamic boxes that cover candidates (the
    moving-camera frames of the bench yield only degenerate boxes with the synthetic YOLO weights), so the host-side
    dynamic-box cull of the fused single-level call removes keypoints before the quadtree (src/SPextractor.cpp:314-352)."""
    from dani_slam_amd import frontend
    w = synth.bench_weights()
    gray = np.stack([synth.frame(i) for i in range(B)]); bgr = np.stack([synth.frame_bgr(i) for i in range(B)])
    e = frontend.Engine(device=0, max_batch=B, weights=w)
    e.yolo_set_classes(synth.COCO_NAMES, synth.DYNAMIC_NAMES)
    out = e.frontend_batch(gray, bgr, nfeatures=NF)
    for i in range(B):
        ok, dets, dyn, _ = yo.detect(w, bgr[i])
        assert [tuple(int(v) for v in r) for r in out[i]["boxes"]] == [tuple(r) for r in dyn], i
        ref = oracle.sp_extract(gray[i], w, nfeatures=NF, boxes=dyn)
        assert np.array_equal(out[i]["kpts"], ref["kpts"]) and np.array_equal(out[i]["desc"], ref["desc"]), i
        if i == 1:      # the cull is effective here: without the boxes the keypoint set differs
            nobox = oracle.sp_extract(gray[i], w, nfeatures=NF)
            assert len(dyn) >= 10 and len({tuple(x) for x in nobox["kpts"].tolist()} ^ {tuple(x) for x in ref["kpts"].tolist()}) >= 8
    e.close()
