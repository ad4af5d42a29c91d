"""Python host side of the front-end engine: a thin ctypes binding over the C ABI (include/dani_frontend.h)
plus mirrors of the reference's C++ classes for this path, so that parity tests read like calls into the
reference:

    SPextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)   /root/reference/include/SPextractor.h:24-96
        .__call__(image, mask, vLappingArea) -> (monoIndex, keypoints, descriptors)   src/SPextractor.cpp:130-217
        .mvDynamicArea, GetLevels(), GetScaleFactor(), GetScaleFactors(), ...          include/SPextractor.h:39-69

All compute happens in libdani_frontend.so (hand-written HIP for gfx950).  There is NO fallback: if the
library is missing or no MI355X is visible, construction raises.  PyTorch is not used here at all.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdani_frontend.so")

_f = C.POINTER(C.c_float)
_i32 = C.POINTER(C.c_int32)
_u8 = C.POINTER(C.c_uint8)

DANI_OK = 0
DANI_ERR_INVALID, DANI_ERR_NO_DEVICE, DANI_ERR_HIP, DANI_ERR_WEIGHTS = -1, -2, -3, -4
DANI_ERR_CAPACITY, DANI_ERR_UNSUPPORTED, DANI_ERR_EMPTY = -5, -6, -7


class DaniError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"dani error {code}: {msg}")
        self.code = code


class _Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("max_batch", C.c_int32), ("max_width", C.c_int32),
                ("max_height", C.c_int32), ("n_contexts", C.c_int32)]


class _SpParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th", C.c_float), ("min_th", C.c_float)]


_lib = None


def load_library():
    """Load libdani_frontend.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.dani_version.restype = C.c_char_p
    lib.dani_last_error.restype = C.c_char_p
    lib.dani_last_error.argtypes = [C.c_void_p]
    lib.dani_engine_create.argtypes = [C.POINTER(_Config), C.POINTER(C.c_void_p)]
    lib.dani_engine_destroy.argtypes = [C.c_void_p]
    lib.dani_engine_destroy.restype = None
    lib.dani_set_tensor.argtypes = [C.c_void_p, C.c_char_p, _f, C.c_int64]
    lib.dani_load_weights_file.argtypes = [C.c_void_p, C.c_char_p]
    lib.dani_finalize_weights.argtypes = [C.c_void_p]
    lib.dani_set_profiling.argtypes = [C.c_void_p, C.c_int32]
    lib.dani_layer_timings.argtypes = [C.c_void_p, _f]
    lib.dani_last_timings.argtypes = [C.c_void_p, _f, C.c_int32]
    lib.dani_sp_extract_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_int32, C.POINTER(_SpParams), _i32, _i32, _i32, _f, _f, _i32, _f, _f,
                                          C.c_int32, _i32, _i32]
    lib.dani_sp_debug_fetch.argtypes = [C.c_void_p, C.c_int32, _f, _f, _f, C.c_int32, _i32]
    lib.dani_sp_debug_layer.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _f, C.c_int64]
    lib.dani_host_distribute_octtree.argtypes = [_f, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                                 C.c_int32, _i32, C.c_int32]
    lib.dani_host_cell_grid.argtypes = [C.c_int32, C.c_int32, _i32, C.c_int32]
    lib.dani_host_features_per_level.argtypes = [C.c_int32, C.c_float, C.c_int32, _i32, _f]
    _lib = lib
    return lib


def _fp(a):
    return a.ctypes.data_as(_f)


def _ip(a):
    return a.ctypes.data_as(_i32)


# ---- host logic (no GPU needed) -----------------------------------------------------------------
def host_distribute_octtree(xyr, minX, maxX, minY, maxY, N):
    lib = load_library()
    xyr = np.ascontiguousarray(xyr, np.float32).reshape(-1, 3)
    n = xyr.shape[0]
    out = np.empty(max(n, 1), np.int32)
    m = lib.dani_host_distribute_octtree(_fp(xyr), n, minX, maxX, minY, maxY, N, _ip(out), out.size)
    if m < 0:
        raise DaniError(m, "dani_host_distribute_octtree")
    return out[:m].copy()


def host_cell_grid(width, height):
    lib = load_library()
    out = np.empty((4096, 6), np.int32)
    n = lib.dani_host_cell_grid(width, height, _ip(out), 4096)
    return out[:n].copy()


def host_features_per_level(nfeatures, scaleFactor, nlevels):
    lib = load_library()
    per = np.zeros(nlevels, np.int32)
    sc = np.zeros(nlevels, np.float32)
    rc = lib.dani_host_features_per_level(nfeatures, C.c_float(scaleFactor), nlevels, _ip(per), _fp(sc))
    if rc:
        raise DaniError(rc, "dani_host_features_per_level")
    return per, sc


# ---- engine -------------------------------------------------------------------------------------
class Engine:
    """One engine per process and GPU (weights resident, stream/workspace pool)."""

    def __init__(self, device=0, max_batch=1, max_width=640, max_height=480, n_contexts=2, weights=None):
        self.lib = load_library()
        cfg = _Config(device, max_batch, max_width, max_height, n_contexts)
        h = C.c_void_p()
        rc = self.lib.dani_engine_create(C.byref(cfg), C.byref(h))
        if rc:
            raise DaniError(rc, self.lib.dani_last_error(None).decode())
        self.h = h
        self.max_batch = max_batch
        if weights is not None:
            self.set_weights(weights)

    def _check(self, rc):
        if rc:
            raise DaniError(rc, self.lib.dani_last_error(self.h).decode())

    def set_weights(self, weights: dict):
        for name, arr in weights.items():
            a = np.ascontiguousarray(arr, np.float32)
            self._check(self.lib.dani_set_tensor(self.h, name.encode(), _fp(a), a.size))
        self._check(self.lib.dani_finalize_weights(self.h))

    def load_weights_file(self, path):
        self._check(self.lib.dani_load_weights_file(self.h, path.encode()))
        self._check(self.lib.dani_finalize_weights(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.lib.dani_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- SuperPoint ------------------------------------------------------------------------------
    def sp_extract_batch(self, imgs, nfeatures=1024, scaleFactor=1.2, nlevels=1, iniTh=0.015, minTh=0.007,
                         boxes=None, lapping=(0, 0), cap=None, device_ptr=None, shape=None):
        """imgs: uint8 [B, H, W] host array (or device_ptr + shape=(B, H, W, stride) for frames resident in HBM).
        boxes: list (len B) of [k, 4] int arrays (x, y, w, h).  Returns a list of dicts per frame."""
        if device_ptr is None:
            imgs = np.ascontiguousarray(imgs, np.uint8)
            if imgs.ndim == 2:
                imgs = imgs[None]
            B, H, W = imgs.shape
            stride = imgs.strides[1]
            ptr, on_dev = imgs.ctypes.data_as(C.c_void_p), 0
        else:
            B, H, W, stride = shape
            ptr, on_dev = C.c_void_p(device_ptr), 1
        cap = cap or (nfeatures + 8)
        p = _SpParams(nfeatures, scaleFactor, nlevels, iniTh, minTh)
        if boxes is None:
            boxes = [np.zeros((0, 4), np.int32)] * B
        offs = np.zeros(B + 1, np.int32)
        for b in range(B):
            offs[b + 1] = offs[b] + len(np.asarray(boxes[b]).reshape(-1, 4))
        allb = (np.concatenate([np.asarray(x, np.int32).reshape(-1, 4) for x in boxes], 0)
                if offs[B] else np.zeros((1, 4), np.int32))
        allb = np.ascontiguousarray(allb, np.int32)
        lap = np.asarray(lapping, np.int32)
        k = np.empty((B, cap, 2), np.float32); r = np.empty((B, cap), np.float32)
        o = np.empty((B, cap), np.int32); sz = np.empty((B, cap), np.float32)
        d = np.empty((B, cap, 256), np.float32)
        n = np.zeros(B, np.int32); mono = np.zeros(B, np.int32)
        rc = self.lib.dani_sp_extract_batch(self.h, ptr, on_dev, B, W, H, stride, C.byref(p), _ip(allb), _ip(offs),
                                            _ip(lap), _fp(k), _fp(r), _ip(o), _fp(sz), _fp(d), cap, _ip(n), _ip(mono))
        self._check(rc)
        return [dict(kpts=k[b, :n[b]].copy(), resp=r[b, :n[b]].copy(), octave=o[b, :n[b]].copy(),
                     size=sz[b, :n[b]].copy(), desc=d[b, :n[b]].copy(), mono_index=int(mono[b])) for b in range(B)]

    def sp_debug_fetch(self, b, H, W, cand_cap=32768):
        prob = np.empty(((H // 8) * 8, (W // 8) * 8), np.float32)
        dmap = np.empty((256, H // 8, W // 8), np.float32)
        cand = np.empty((cand_cap, 3), np.float32)
        nc = C.c_int32(0)
        self._check(self.lib.dani_sp_debug_fetch(self.h, b, _fp(prob), _fp(dmap), _fp(cand), cand_cap, C.byref(nc)))
        return prob, dmap, cand[:nc.value].copy()

    def sp_debug_layer(self, b, layer, shape):
        out = np.empty(shape, np.float32)
        self._check(self.lib.dani_sp_debug_layer(self.h, b, layer, _fp(out), out.size))
        return out

    def set_profiling(self, on=True):
        self._check(self.lib.dani_set_profiling(self.h, int(on)))

    def layer_timings(self):
        ms = np.zeros(12, np.float32)
        self._check(self.lib.dani_layer_timings(self.h, _fp(ms)))
        return ms

    def last_timings(self):
        ms = np.zeros(5, np.float32)
        self._check(self.lib.dani_last_timings(self.h, _fp(ms), 5))
        return ms


_default_engine = None


def default_engine(**kw) -> Engine:
    """Process-wide engine singleton (construction of the reference-style wrappers is then O(1); the reference
    re-loads its TorchScript blobs per construction, quirk Q12)."""
    global _default_engine
    if _default_engine is None:
        _default_engine = Engine(**kw)
    return _default_engine


@dataclass
class KeyPoint:
    """The cv::KeyPoint fields the path fills (src/SuperPoint.cpp:220,346; src/SPextractor.cpp:365-374)."""
    x: float
    y: float
    size: float
    angle: float
    response: float
    octave: int


class SPextractor:
    """Mirror of ORB_SLAM3::SPextractor (include/SPextractor.h:24-96) on top of the HIP engine."""

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, engine: Engine | None = None):
        self.engine = engine or default_engine()
        self.nfeatures, self.scaleFactor, self.nlevels = int(nfeatures), float(scaleFactor), int(nlevels)
        self.iniThFAST, self.minThFAST = float(iniThFAST), float(minThFAST)
        per, sc = host_features_per_level(self.nfeatures, self.scaleFactor, self.nlevels)
        self.mnFeaturesPerLevel = per
        self.mvScaleFactor = sc
        self.mvLevelSigma2 = (sc * sc).astype(np.float32)
        self.mvInvScaleFactor = (np.float32(1.0) / sc).astype(np.float32)
        self.mvInvLevelSigma2 = (np.float32(1.0) / self.mvLevelSigma2).astype(np.float32)
        self.mvDynamicArea = []   # list of (x, y, w, h), written by the caller (Tracking.cc:1539)

    def GetLevels(self): return self.nlevels
    def GetScaleFactor(self): return self.scaleFactor
    def GetScaleFactors(self): return self.mvScaleFactor
    def GetInverseScaleFactors(self): return self.mvInvScaleFactor
    def GetScaleSigmaSquares(self): return self.mvLevelSigma2
    def GetInverseScaleSigmaSquares(self): return self.mvInvLevelSigma2

    def __call__(self, image, mask=None, vLappingArea=(0, 0)):
        """Returns (monoIndex, keypoints [n] of KeyPoint, descriptors [n, 256] f32); monoIndex = -1 and empty
        outputs for an empty image, as SPextractor::operator() (src/SPextractor.cpp:135-136)."""
        if image is None or np.asarray(image).size == 0:
            return -1, [], np.zeros((0, 256), np.float32)
        boxes = [np.asarray(self.mvDynamicArea, np.int32).reshape(-1, 4)]
        r = self.engine.sp_extract_batch(np.asarray(image), self.nfeatures, self.scaleFactor, self.nlevels,
                                         self.iniThFAST, self.minThFAST, boxes=boxes, lapping=vLappingArea)[0]
        kps = [KeyPoint(float(x), float(y), float(s), -1.0, float(rr), int(o))
               for (x, y), s, rr, o in zip(r["kpts"], r["size"], r["resp"], r["octave"])]
        return r["mono_index"], kps, r["desc"]
