"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE -- see oracle/sp_oracle.cpp header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "liboracle.so")
_lib = None

SP_LAYER_NAMES = ["conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b",
                  "conv4a", "conv4b", "convPa", "convPb", "convDa", "convDb"]

_f = C.POINTER(C.c_float)
_i32 = C.POINTER(C.c_int32)
_u8 = C.POINTER(C.c_uint8)


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_expf.restype = C.c_float
        _lib.orc_expf.argtypes = [C.c_float]
    return _lib


def _fp(a):
    return a.ctypes.data_as(_f)


def _weights_array(weights: dict):
    keep = []
    arr = (_f * 24)()
    for i, n in enumerate(SP_LAYER_NAMES):
        for j, suffix in enumerate((".weight", ".bias")):
            a = np.ascontiguousarray(weights[n + suffix], dtype=np.float32)
            keep.append(a)
            arr[2 * i + j] = _fp(a)
    return arr, keep


def conv2d(x, w, b, relu):
    x = np.ascontiguousarray(x, np.float32); w = np.ascontiguousarray(w, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    cin, h, wd = x.shape
    cout, _, k, _ = w.shape
    out = np.empty((cout, h, wd), np.float32)
    rc = lib().orc_conv2d(_fp(x), cin, h, wd, _fp(w), _fp(b), cout, k, int(relu), _fp(out))
    assert rc == 0
    return out


def maxpool2(x):
    x = np.ascontiguousarray(x, np.float32)
    c, h, w = x.shape
    out = np.empty((c, h // 2, w // 2), np.float32)
    lib().orc_maxpool2(_fp(x), c, h, w, _fp(out))
    return out


def softmax_d2s(semi):
    semi = np.ascontiguousarray(semi, np.float32)
    _, hc, wc = semi.shape
    prob = np.empty((hc * 8, wc * 8), np.float32)
    lib().orc_softmax_d2s(_fp(semi), hc, wc, _fp(prob))
    return prob


def desc_l2norm(desc):
    d = np.array(desc, np.float32, order="C", copy=True)
    c, h, w = d.shape
    lib().orc_desc_l2norm(_fp(d), c, h * w)
    return d


def superpoint_forward(img, weights, want_taps=False):
    """img uint8 [H, W] -> (prob [H8, W8], desc [256, H/8, W/8][, taps list of 12 arrays])."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    warr, keep = _weights_array(weights)
    prob = np.empty(((h // 8) * 8, (w // 8) * 8), np.float32)
    desc = np.empty((256, h // 8, w // 8), np.float32)
    taps = None
    tap_arr = None
    if want_taps:
        from dani_slam_amd.synth import SP_LAYERS
        taps = []
        tap_arr = (_f * 12)()
        hh, ww = h, w
        for i, (_, _, cout, _) in enumerate(SP_LAYERS):
            t = np.empty((cout, hh, ww), np.float32)
            taps.append(t)
            tap_arr[i] = _fp(t)
            if i in (1, 3, 5):
                hh //= 2; ww //= 2
    rc = lib().orc_superpoint_forward(img.ctypes.data_as(_u8), h, w, img.strides[0], warr, _fp(prob), _fp(desc), tap_arr)
    assert rc == 0, rc
    return (prob, desc, taps) if want_taps else (prob, desc)


def cell_keypoints(prob, th, iniX, maxX, iniY, maxY):
    prob = np.ascontiguousarray(prob, np.float32)
    ph, pw = prob.shape
    cap = max(1, (maxX - iniX) * (maxY - iniY))
    out = np.empty((cap, 3), np.float32)
    n = lib().orc_cell_keypoints(_fp(prob), ph, pw, C.c_float(th), iniX, maxX, iniY, maxY, _fp(out), cap)
    return out[:n].copy()


def _boxes(boxes):
    b = np.ascontiguousarray(np.asarray(boxes, np.int32).reshape(-1, 4))
    return b, b.ctypes.data_as(_i32), b.shape[0]


def collect_candidates(prob, cols, rows, iniTh, minTh, boxes=(), scale=1.0):
    prob = np.ascontiguousarray(prob, np.float32)
    ph, pw = prob.shape
    b, bp, nb = _boxes(boxes)
    cap = ph * pw * 2
    out = np.empty((cap, 3), np.float32)
    n = lib().orc_collect_candidates(_fp(prob), ph, pw, cols, rows, C.c_float(iniTh), C.c_float(minTh), bp, nb,
                                     C.c_float(scale), _fp(out), cap)
    return out[:n].copy()


def distribute_octtree(xyr, minX, maxX, minY, maxY, N):
    xyr = np.ascontiguousarray(xyr, np.float32).reshape(-1, 3)
    n = xyr.shape[0]
    cap = max(n, 1)
    out = np.empty((cap, 3), np.float32)
    m = lib().orc_distribute_octtree(_fp(xyr), n, minX, maxX, minY, maxY, N, _fp(out), cap)
    return out[:m].copy()


def compute_descriptors(desc_map, kxy, PH, PW):
    desc_map = np.ascontiguousarray(desc_map, np.float32)
    kxy = np.ascontiguousarray(kxy, np.float32).reshape(-1, 2)
    c, hc, wc = desc_map.shape
    n = kxy.shape[0]
    out = np.empty((n, c), np.float32)
    if n:
        lib().orc_compute_descriptors(_fp(desc_map), c, hc, wc, PH, PW, _fp(kxy), n, _fp(out))
    return out


def features_per_level(nfeatures, scaleFactor, nlevels):
    out = np.zeros(nlevels, np.int32)
    sc = np.zeros(nlevels, np.float32)
    lib().orc_features_per_level(nfeatures, C.c_float(scaleFactor), nlevels, out.ctypes.data_as(_i32), _fp(sc))
    return out, sc


def sp_extract(img, weights, nfeatures=1024, scaleFactor=1.2, nlevels=1, iniTh=0.015, minTh=0.007, boxes=(),
               lapping=(0, 0), cap=8192, want_maps=False):
    """Whole SPextractor::operator().  Returns dict(kpts[n,2], resp[n], octave[n], size[n], desc[n,256], mono_index)."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    warr, keep = _weights_array(weights)
    b, bp, nb = _boxes(boxes)
    kpts = np.empty((cap, 2), np.float32); resp = np.empty(cap, np.float32)
    octave = np.empty(cap, np.int32); size = np.empty(cap, np.float32)
    desc = np.empty((cap, 256), np.float32)
    mono = C.c_int(0)
    prob = np.empty(((h // 8) * 8, (w // 8) * 8), np.float32) if want_maps else None
    dmap = np.empty((256, h // 8, w // 8), np.float32) if want_maps else None
    n = lib().orc_sp_extract(img.ctypes.data_as(_u8), w, h, img.strides[0], warr, nfeatures, C.c_float(scaleFactor),
                             nlevels, C.c_float(iniTh), C.c_float(minTh), bp, nb, int(lapping[0]), int(lapping[1]),
                             _fp(kpts), _fp(resp), octave.ctypes.data_as(_i32), _fp(size), _fp(desc), cap,
                             C.byref(mono), _fp(prob) if want_maps else None, _fp(dmap) if want_maps else None)
    if n < 0:
        raise RuntimeError(f"orc_sp_extract failed: {n}")
    r = dict(kpts=kpts[:n].copy(), resp=resp[:n].copy(), octave=octave[:n].copy(), size=size[:n].copy(),
             desc=desc[:n].copy(), mono_index=mono.value)
    if want_maps:
        r["prob"] = prob; r["desc_map"] = dmap
    return r


# ---- general conv / pooling / resize used by oracle/yolo_oracle.py ---------------------------------------
def conv2d_ex(x, w, b, stride, pad, act):
    """act: 0 none, 1 ReLU, 2 SiLU.  x [Cin,H,W], w [Cout,Cin,k,k]."""
    x = np.ascontiguousarray(x, np.float32); w = np.ascontiguousarray(w, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    cin, h, wd = x.shape
    cout, cin2, k, _ = w.shape
    assert cin == cin2
    ho, wo = (h + 2 * pad - k) // stride + 1, (wd + 2 * pad - k) // stride + 1
    out = np.empty((cout, ho, wo), np.float32)
    rc = lib().orc_conv2d_ex(_fp(x), cin, h, wd, _fp(w), _fp(b), cout, k, stride, pad, act, _fp(out))
    assert rc == 0
    return out


def maxpool5(x):
    x = np.ascontiguousarray(x, np.float32)
    c, h, w = x.shape
    out = np.empty_like(x)
    lib().orc_maxpool5(_fp(x), c, h, w, _fp(out))
    return out


def resize_bilinear_u8c3(img, dw, dh):
    img = np.ascontiguousarray(img, np.uint8)
    sh, sw, _ = img.shape
    out = np.empty((dh, dw, 3), np.uint8)
    lib().orc_resize_bilinear_u8c3(img.ctypes.data_as(_u8), sw, sh, img.strides[0], out.ctypes.data_as(_u8), dw, dh)
    return out


def resize_bilinear_u8c1(img, dw, dh):
    img = np.ascontiguousarray(img, np.uint8)
    sh, sw = img.shape
    out = np.empty((dh, dw), np.uint8)
    lib().orc_resize_bilinear_u8c1(img.ctypes.data_as(_u8), sw, sh, img.strides[0], out.ctypes.data_as(_u8), dw, dh)
    return out


def sigmoid(x):
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    lib().orc_sigmoid(_fp(x), _fp(out), C.c_long(x.size))
    return out
