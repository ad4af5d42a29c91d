"""GPU: include/dani_shims.hpp -- the C++17 shims that keep SPextractor::operator(), LightGlue::matchDescriptors() and
YoloDetection::Detect() -- compiled as C++ and driven end to end through the C ABI from a C++ program
(tests/cpp/shim_smoke.cpp).  OpenCV is absent from this image, so the shim header is compiled against the mock core types
of tests/cpp/mock_opencv (test scaffolding, nothing of the reference is built); results must equal the ctypes path."""
import os
import shutil
import struct
import subprocess
import sys

import numpy as np
import pytest

from dani_slam_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_shims_match_ctypes_path(tmp_path, sp_weights):
    from dani_slam_amd import frontend
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import convert_weights
    cxx = shutil.which("g++") or shutil.which("hipcc")
    assert cxx, "no C++ compiler"
    exe = str(tmp_path / "shim_smoke")
    libdir = os.path.dirname(frontend.LIB_PATH)
    subprocess.check_call([cxx, "-std=c++17", "-O1", "-I", os.path.join(ROOT, "tests/cpp/mock_opencv"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests/cpp/shim_smoke.cpp"), "-o", exe, "-L", libdir, "-ldani_frontend",
                           "-Wl,-rpath," + libdir, "-lpthread"])
    w = dict(sp_weights); w.update(synth.lightglue_weights(0)); w.update(synth.yolo_weights(0))
    convert_weights.write_dsw(str(tmp_path / "w.dsw"), w)
    (tmp_path / "coco.names").write_text("\n".join(synth.COCO_NAMES) + "\n")
    g0, g1, c0 = synth.frame(1), synth.frame(2), synth.frame_bgr(1)
    for name, a in (("g0.bin", g0), ("g1.bin", g1), ("c0.bin", c0)):
        a.tofile(str(tmp_path / name))
    env = dict(os.environ, DANI_WEIGHTS=str(tmp_path / "w.dsw"))
    out = subprocess.run([exe, "g0.bin", "g1.bin", "c0.bin", "out.bin"], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "shim_smoke ok" in out.stdout
    buf = (tmp_path / "out.bin").read_bytes()
    pos = 0

    def geti():
        nonlocal pos
        v = struct.unpack_from("<i", buf, pos)[0]; pos += 4
        return v

    def getf(n):
        nonlocal pos
        v = np.frombuffer(buf, np.float32, n, pos).copy(); pos += 4 * n
        return v
    ok, nb = geti(), geti()
    boxes = [tuple(geti() for _ in range(4)) for _ in range(nb)]
    kps = []
    for _ in range(2):
        n = geti()
        rec = []
        for _ in range(n):
            x, y, r, s = getf(4)
            rec.append((x, y, r, s, geti()))
        kps.append(rec)
    mono0, mono1 = geti(), geti()
    d0 = getf(len(kps[0]) * 256).reshape(-1, 256); d1 = getf(len(kps[1]) * 256).reshape(-1, 256)
    nm = geti()
    matches = [(geti(), geti(), float(getf(1)[0])) for _ in range(nm)]
    empty_rc = geti()
    st8, mono8, n8, rows8 = geti(), geti(), geti(), geti()
    k8 = []
    for _ in range(n8):
        x, y = getf(2)
        k8.append((x, y, geti()))
    dropped_before, tiny_rc, tiny_status, dropped_after, tiny_n = geti(), geti(), geti(), geti(), geti()
    assert pos == len(buf) and ok == 1 and empty_rc == -1
    # an engine error on one frame: -1 and no keypoints (as for an empty image), counted, never silent
    assert (dropped_before, tiny_rc, tiny_status, dropped_after, tiny_n) == (0, -1, frontend.DANI_ERR_UNSUPPORTED, 1, 0)
    assert "frame dropped (1 so far" in out.stderr

    # the same calls through the ctypes mirror
    e = frontend.Engine(device=0, max_batch=8, max_width=1280, max_height=1024, weights=w)
    dets = e.yolo_detect_batch(c0)[0]
    dyn, _ = frontend.host_dynamic_areas(dets, 640, 480, synth.COCO_NAMES, synth.DYNAMIC_NAMES)
    assert boxes == [tuple(int(v) for v in r) for r in dyn] and len(boxes) >= 1
    r0 = e.sp_extract_batch(g0, nfeatures=300, boxes=[np.array(boxes, np.int32).reshape(-1, 4)])[0]
    r1 = e.sp_extract_batch(g1, nfeatures=300)[0]
    for rec, ref, d in ((kps[0], r0, d0), (kps[1], r1, d1)):
        assert np.array_equal(np.array([[k[0], k[1]] for k in rec], np.float32), ref["kpts"])
        assert np.array_equal(np.array([k[2] for k in rec], np.float32), ref["resp"])
        assert all(k[3] == 31.0 and k[4] == 0 for k in rec)
        assert np.array_equal(d, ref["desc"])
    assert (mono0, mono1) == (r0["mono_index"], r1["mono_index"])
    m, s = e.lg_match(r0["kpts"], r0["desc"], r1["kpts"], r1["desc"], (640.0, 480.0), (640.0, 480.0), 0.05)
    assert [(a, b) for a, b, _ in matches] == [tuple(x) for x in m.tolist()]
    np.testing.assert_allclose([c for _, _, c in matches], s, rtol=0, atol=1e-6)
    # the 8-level call: status OK (capacity covers the per-level overshoot), same keypoints as the ctypes path
    r8 = e.sp_extract_batch(g0, nfeatures=1250, nlevels=8)[0]
    assert st8 == 0 and n8 == rows8 == len(r8["kpts"]) and n8 >= 1250 and mono8 == r8["mono_index"]
    assert np.array_equal(np.array([[k[0], k[1]] for k in k8], np.float32), r8["kpts"])
    assert [k[2] for k in k8] == r8["octave"].tolist() and max(k[2] for k in k8) == 7
    e.close()
