"""GPU: the engine configurations behind the side numbers of the bench line, each against the oracle composition.

bench.py quotes `latency_b1_ms.config2 / config3` from an engine built with max_batch = 2 (bench.py: "an engine built for
that use": small-batch conv tile twins, SuperPoint head fork, attn_k4_kernel, ln_gelu16_kernel, linear_fm_kernel's tail K split,
all in situ at nFeatures = 1024) and `--frames-per-gpu 64` from a max_batch = 64 engine (BASELINE config #4's whole batch on
one GPU).  The other fused-call tests run nFeatures = 300 on max_batch 2 / 4 engines or B = 8; the compositions measured there
are checked here at the measured shapes.
  config #2: SPextractor::operator() on one frame            /root/reference/src/Frame.cc:420-427
  config #3: two extractions + Frame::ComputeStereoMatches   src/Frame.cc:124-143, 813-860
  config #4: YOLO boxes -> extract -> LightGlue vs previous  src/Tracking.cc:1534-1545
"""
import numpy as np
import pytest

from dani_slam_amd import synth
from oracle import lg_oracle as lg
from oracle import oracle
from oracle import yolo_oracle as yo

pytestmark = pytest.mark.gpu
NF = 1024
SIZE = (640, 480)


def _match_dict(o):
    return {tuple(x): float(v) for x, v in zip(np.asarray(o["matches"]).tolist(), np.asarray(o["scores"]))}


def _check_matches(got, mo, so, tag, min_matches=40):
    want = {tuple(x): float(v) for x, v in zip(mo.tolist(), so)}
    assert len(want) >= min_matches, (tag, len(want))                          # the comparison is not vacuous
    assert len(set(got) ^ set(want)) <= max(2, len(want) // 50), (tag, len(got), len(want), len(set(got) ^ set(want)))
    common = set(got) & set(want)
    assert max(abs(got[k] - want[k]) for k in common) < 1e-3, tag              # north_star: match scores within 1e-3


def _bench_engine(max_batch):
    """Built exactly as bench.py builds its engines (bench.py main(): Engine(device, max_batch, 640, 480, weights))."""
    from dani_slam_amd import frontend
    w = synth.bench_weights()
    e = frontend.Engine(device=0, max_batch=max_batch, max_width=640, max_height=480, weights=w)
    e.yolo_set_classes(synth.COCO_NAMES, synth.DYNAMIC_NAMES)
    return e, w


def test_config2_and_config3_on_the_latency_engine():
    """bench.py's `latency_b1_ms`: mode 2 = frontend_batch(use_yolo=False, match_previous=False) on ONE frame, mode 3 =
    frontend_batch(use_yolo=False, stereo_pairs=True) on two frames, both on a max_batch = 2 engine with device-resident
    frames, nFeatures = 1024; f32 and the opt-in bf16x3 arithmetic."""
    import torch
    e, w = _bench_engine(2)
    gray, _ = synth.bench_frames([0, 1])
    d_gray = torch.from_numpy(gray).cuda()
    refs = [oracle.sp_extract(g, w, nfeatures=NF) for g in gray]
    assert all(len(r["kpts"]) >= NF for r in refs)

    # ---- config #2: one frame per call, extract only (both frames, one after the other, as a tracker would)
    for i in range(2):
        o = e.frontend_batch(None, None, nfeatures=NF, use_yolo=False, match_previous=False,
                             device_ptrs=(d_gray[i:].data_ptr(), 0), shape=(1, 480, 640))[0]
        assert np.array_equal(o["kpts"], refs[i]["kpts"]), i                   # bit-exact keypoints, in order
        assert np.array_equal(o["resp"], refs[i]["resp"]), i
        assert np.array_equal(o["desc"], refs[i]["desc"]), i                   # bit-exact descriptors
        assert o["mono_index"] == refs[i]["mono_index"] and len(o["matches"]) == 0

    # ---- config #3: the stereo frame (left = frame 0, right = frame 1)
    mo, so = lg.match_descriptors(w, refs[0]["kpts"], refs[0]["desc"], refs[1]["kpts"], refs[1]["desc"], SIZE, SIZE, 0.05)

    def stereo_call():
        return e.frontend_batch(None, None, nfeatures=NF, use_yolo=False, stereo_pairs=True,
                                device_ptrs=(d_gray.data_ptr(), 0), shape=(2, 480, 640))
    out = stereo_call()
    for i in range(2):
        assert np.array_equal(out[i]["kpts"], refs[i]["kpts"]) and np.array_equal(out[i]["desc"], refs[i]["desc"]), i
    assert len(out[0]["matches"]) == 0
    _check_matches(_match_dict(out[1]), mo, so, "config 3, f32")
    again = stereo_call()                                                      # nothing is carried from call to call
    assert np.array_equal(again[1]["matches"], out[1]["matches"]) and np.array_equal(again[1]["scores"], out[1]["scores"])
    for mode, name in ((e.LG_GEMM_BF16X3, "bf16x3"), (e.LG_GEMM_BF16X3_ALL, "bf16x3_all")):
        e.lg_set_gemm(mode)
        o3 = stereo_call()
        _check_matches(_match_dict(o3[1]), mo, so, f"config 3, {name}")
        a, b = _match_dict(out[1]), _match_dict(o3[1])
        assert len(set(a) ^ set(b)) <= 2 and max(abs(a[k] - b[k]) for k in set(a) & set(b)) < 1e-4, name
        assert np.array_equal(o3[1]["kpts"], out[1]["kpts"]) and np.array_equal(o3[1]["desc"], out[1]["desc"])
    e.lg_set_gemm(e.LG_GEMM_F32)
    e.close()


def test_config4_whole_batch_of_64_frames_on_one_engine():
    """`bench.py --frames-per-gpu 64`: BASELINE config #4's 64-frame batch in ONE dani_frontend_batch call on a
    max_batch = 64 engine.  Every frame's keypoint count and boxes, the full extraction of the frames of four checked pairs
    (first / middle / last of the batch and the pair behind a 32-frame boundary) and those pairs' matches."""
    import torch
    B = 64
    e, w = _bench_engine(B)
    gray, bgr = synth.bench_frames(range(B))
    d_gray = torch.from_numpy(gray).cuda(); d_bgr = torch.from_numpy(bgr).cuda()
    out = e.frontend_batch(None, None, nfeatures=NF, device_ptrs=(d_gray.data_ptr(), d_bgr.data_ptr()), shape=(B, 480, 640))
    assert len(out) == B and len(out[0]["matches"]) == 0
    assert min(len(o["matches"]) for o in out[1:]) >= 40, [len(o["matches"]) for o in out]
    pairs = (1, 17, 32, 63)                       # index of the pair's second frame
    need = sorted({i for p in pairs for i in (p - 1, p)})
    refs = {}
    for i in need:
        ok, dets, dyn, _ = yo.detect(w, bgr[i])
        assert [tuple(int(v) for v in r) for r in out[i]["boxes"]] == [tuple(r) for r in dyn], i
        ref = oracle.sp_extract(gray[i], w, nfeatures=NF, boxes=dyn)
        assert np.array_equal(out[i]["kpts"], ref["kpts"]), i
        assert np.array_equal(out[i]["resp"], ref["resp"]), i
        assert np.array_equal(out[i]["desc"], ref["desc"]), i
        refs[i] = ref
    for p in pairs:
        mo, so = lg.match_descriptors(w, refs[p - 1]["kpts"], refs[p - 1]["desc"], refs[p]["kpts"], refs[p]["desc"], SIZE, SIZE, 0.05)
        _check_matches(_match_dict(out[p]), mo, so, f"pair {p - 1}->{p} of the 64-frame call")
    # the frames the oracle did not see: the same binary on an 8-frame engine shape is covered by test_bench_shape_gpu.py; here
    # the 64-frame call must at least agree with eight 8-frame calls of the same engine (same kernels, other grids)
    e.frontend_reset()
    small = []
    for c in range(0, B, 8):
        small += e.frontend_batch(None, None, nfeatures=NF, device_ptrs=(d_gray[c:].data_ptr(), d_bgr[c:].data_ptr()), shape=(8, 480, 640))
    for i in range(B):
        assert np.array_equal(small[i]["kpts"], out[i]["kpts"]) and np.array_equal(small[i]["desc"], out[i]["desc"]), i
        assert np.array_equal(small[i]["boxes"], out[i]["boxes"]), i
        a, b = _match_dict(small[i]), _match_dict(out[i])
        assert len(set(a) ^ set(b)) <= 2, (i, len(a), len(b))                 # other batch composition: the linear tail tokens' order differs
        if a and b:
            assert max(abs(a[k] - b[k]) for k in set(a) & set(b)) < 1e-4, i
    e.close()
