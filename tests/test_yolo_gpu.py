"""GPU parity tests for the YOLOv5s dynamic-object detector: HIP engine (C ABI) vs oracle/yolo_oracle.py.
The network is an absent TorchScript blob in the reference (parity unpinned; public v6 architecture restated);
post-processing is a literal restatement of src/YoloDetect.cpp.  Convolutions share the canonical f32 chain, so
raw head outputs, detections and boxes are expected to be bit-identical."""
import numpy as np
import pytest

from dani_slam_amd import synth
from oracle import yolo_oracle as yo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def yw():
    return synth.yolo_weights(0)


@pytest.fixture(scope="module")
def engine(sp_weights, yw):
    from dani_slam_amd import frontend
    w = dict(sp_weights); w.update(yw)
    e = frontend.Engine(device=0, max_batch=3, weights=w)
    yield e
    e.close()


def test_raw_head_and_dets_match_oracle(engine, yw):
    img = synth.frame_bgr(1)
    x = yo.preprocess(img)
    preds, taps = yo.network(yw, x, want_taps=True)
    dets_ref = yo.non_max_suppression(preds, 0.4, 0.5)
    dets = engine.yolo_detect_batch(img)[0]
    for lvl in range(3):
        raw = engine.yolo_debug_raw(0, lvl)
        ref = taps[f"raw{lvl}"]
        assert raw.shape == ref.shape
        assert np.array_equal(raw, ref), f"level {lvl}: max diff {np.abs(raw - ref).max()}"
    assert dets.shape == dets_ref.shape and len(dets) > 5
    assert np.array_equal(dets, dets_ref)


@pytest.mark.parametrize("i", [4, 2, 6])
def test_detect_and_dynamic_areas(engine, yw, i):
    from dani_slam_amd import frontend
    img = synth.frame_bgr(i)
    ok, dets_ref, dyn_ref, dmap_ref = yo.detect(yw, img)
    det = frontend.YoloDetection(engine=engine, class_names=yo.COCO_NAMES)
    det.GetImage(img)
    assert det.Detect() is True
    assert [tuple(r) for r in det.mvDynamicArea] == [tuple(r) for r in dyn_ref]
    assert {k: [tuple(r) for r in v] for k, v in det.mmDetectMap.items()} == {k: [tuple(r) for r in v] for k, v in dmap_ref.items()}
    assert det.mvDynamicNames == yo.DYNAMIC_NAMES


def test_batch_and_other_thresholds(engine, yw):
    imgs = np.stack([synth.frame_bgr(i) for i in (4, 5, 1)])
    res = engine.yolo_detect_batch(imgs, 0.3, 0.45)
    for b in range(3):
        preds = yo.network(yw, yo.preprocess(imgs[b]))
        ref = yo.non_max_suppression(preds, 0.3, 0.45)
        assert np.array_equal(res[b], ref)


def test_empty_image_and_missing_weights(engine, sp_weights):
    from dani_slam_amd import frontend
    det = frontend.YoloDetection(engine=engine, class_names=yo.COCO_NAMES)
    det.GetImage(np.zeros((0, 0, 3), np.uint8))
    assert det.Detect() is False                                  # src/YoloDetect.cpp:32-36
    e2 = frontend.Engine(device=0, weights=sp_weights)
    with pytest.raises(frontend.DaniError) as ei:
        e2.yolo_detect_batch(synth.frame_bgr(0))
    assert ei.value.code == frontend.DANI_ERR_WEIGHTS
    e2.close()
