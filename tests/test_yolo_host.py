"""CPU: YOLO host logic of the product (dani_host_dynamic_areas) vs the oracle's literal restatement of
Detect()'s box bookkeeping + RefineDynamicBoxes (src/YoloDetect.cpp:51-80,148-208)."""
import numpy as np

from dani_slam_amd import frontend
from oracle import yolo_oracle as yo


def _random_dets(rng, n):
    x1 = rng.uniform(-20, 600, n); y1 = rng.uniform(-20, 350, n)
    w = rng.uniform(0, 300, n); h = rng.uniform(0, 250, n)
    cls = rng.choice([0, 2, 5, 7, 13, 16, 56, 60, 62, 72], n)     # persons/cars/... + static chair/table/tv/fridge
    return np.stack([x1, y1, x1 + w, y1 + h, rng.uniform(0.4, 1, n), cls], 1).astype(np.float32)


def test_dynamic_areas_match_oracle():
    rng = np.random.RandomState(3)
    for n in [0, 1, 2, 5, 12, 40]:
        for _ in range(5):
            dets = _random_dets(rng, n)
            ref_dyn, ref_map = yo.boxes_from_dets(dets, 640, 480)
            dyn, rects = frontend.host_dynamic_areas(dets, 640, 480, yo.COCO_NAMES, yo.DYNAMIC_NAMES)
            assert [tuple(int(v) for v in r) for r in dyn] == [tuple(r) for r in ref_dyn]


def test_placeholder_box_when_only_static():
    dets = np.array([[10, 10, 100, 100, 0.9, 56]], np.float32)     # a chair: static
    dyn, rects = frontend.host_dynamic_areas(dets, 640, 480, yo.COCO_NAMES, yo.DYNAMIC_NAMES)
    ref_dyn, _ = yo.boxes_from_dets(dets, 640, 480)
    assert [tuple(r) for r in ref_dyn] == [tuple(int(v) for v in r) for r in dyn]
    assert len(dyn) == 1                                            # the (1,1,1,1) placeholder, refined
    assert rects[0].tolist() == [10, 12, 90, 112]                   # x*640/640, y*480/384 (quirk Q10), truncated


def test_static_subtraction_keeps_largest_strip():
    dets = np.array([[100, 100, 300, 300, 0.9, 0],                  # person (dynamic)
                     [150, 80, 250, 200, 0.8, 62]], np.float32)     # tv (static) overlapping the top middle
    dyn, _ = frontend.host_dynamic_areas(dets, 640, 384, yo.COCO_NAMES, yo.DYNAMIC_NAMES)
    ref_dyn, _ = yo.boxes_from_dets(dets, 640, 384)
    assert [tuple(int(v) for v in r) for r in dyn] == [tuple(r) for r in ref_dyn]
