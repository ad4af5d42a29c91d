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
LIB_PATH = os.environ.get("DANI_FRONTEND_LIB", os.path.join(_HERE, "lib", "libdani_frontend.so"))   # override: kernel A/B experiments

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


class _FrontendParams(C.Structure):
    pass


class _SpParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th", C.c_float), ("min_th", C.c_float)]


_FrontendParams._fields_ = [("sp", _SpParams), ("use_yolo", C.c_int32), ("yolo_score_thresh", C.c_float),
                            ("yolo_iou_thresh", C.c_float), ("match_previous", C.c_int32), ("lg_threshold", C.c_float),
                            ("lapping", C.c_int32 * 2), ("pair_mode", C.c_int32), ("defer_desc", C.c_int32)]

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
    lib.dani_lg_set_gemm.argtypes = [C.c_void_p, C.c_int32]
    lib.dani_lg_set_kernel_option.argtypes = [C.c_char_p, C.c_int32]
    lib.dani_set_host_wait.argtypes = [C.c_int32]
    lib.dani_get_host_wait.argtypes = []
    lib.dani_layer_timings.argtypes = [C.c_void_p, _f]
    lib.dani_last_timings.argtypes = [C.c_void_p, _f, C.c_int32]
    lib.dani_sp_extract_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_int32, C.POINTER(_SpParams), _i32, _i32, _i32, _f, _f, _i32, _f, _f,
                                          C.c_int32, _i32, _i32]
    lib.dani_sp_debug_fetch.argtypes = [C.c_void_p, C.c_int32, _f, _f, _f, C.c_int32, _i32]
    lib.dani_sp_debug_layer.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _f, C.c_int64]
    lib.dani_sp_layer_kernel.argtypes = [C.c_void_p, C.c_int32, C.c_char_p, C.c_int32]
    lib.dani_lg_match_batch.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, _i32,
                                        C.c_void_p, C.c_void_p, _i32, _f, _f, C.c_float, _i32, _f, C.c_int32, _i32]
    lib.dani_frontend_retain.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _f, _i32, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.dani_features_undistort.argtypes = [C.c_void_p, C.c_void_p, _f, _f, C.c_int32, C.c_void_p, _f]
    lib.dani_features_release.argtypes = [C.c_void_p]
    lib.dani_features_release.restype = None
    lib.dani_features_count.argtypes = [C.c_void_p]
    lib.dani_features_fetch_grid.argtypes = [C.c_void_p, _i32, _i32]
    lib.dani_desc_scan_resident.argtypes = [C.c_void_p, _f, C.c_int32, C.c_void_p, _i32, _i32, _f, _f, _i32, _i32, _i32, _f]
    lib.dani_search_windows_resident.argtypes = [C.c_void_p, C.c_void_p, _f, C.c_int32, _f, _f, _f, _i32, _i32, _f, _f, _i32, _i32, _i32,
                                                 _i32, _i32, _f, C.c_int32]
    lib.dani_bow_transform_resident.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, _i32, C.POINTER(C.c_double),
                                                _i32, _i32, _i32, _i32, _i32]
    lib.dani_lg_match_groups.argtypes = [C.c_void_p, _f, _f, C.c_int32, _f, _f, C.c_int32, C.c_int32, _i32, _i32, _i32, _i32, _f, _f,
                                         C.c_float, _i32, _f, _i32, C.c_int32]
    lib.dani_lg_attention_profile.argtypes = [C.c_void_p, _f, C.POINTER(C.c_double), _i32]
    lib.dani_lg_attention_profile_ex.argtypes = [C.c_void_p, _f, C.POINTER(C.c_double), C.POINTER(C.c_double), _i32]
    lib.dani_vocab_create.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _i32, _i32, _f, C.POINTER(C.c_double), _i32, C.POINTER(C.c_void_p)]
    lib.dani_vocab_destroy.argtypes = [C.c_void_p]
    lib.dani_vocab_destroy.restype = None
    lib.dani_bow_transform.argtypes = [C.c_void_p, C.c_void_p, _f, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _i32, C.POINTER(C.c_double), _i32, _i32, _i32, _i32, _i32]
    lib.dani_desc_mean.argtypes = [C.c_void_p, _f, C.c_int32, _i32, _i32, C.c_int32, _f]
    lib.dani_desc_scan.argtypes = [C.c_void_p, _f, C.c_int32, _f, C.c_int32, _i32, _i32, _i32, _f, _f, _i32, _i32, _i32, _f]
    lib.dani_lg_linear_profile.argtypes = [C.c_void_p, _f, C.POINTER(C.c_double), _i32]
    lib.dani_lg_debug_fetch.argtypes = [C.c_void_p, C.c_int32, _f, _i32, _f, C.c_int32]
    lib.dani_yolo_detect_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                           C.c_float, C.c_float, _f, C.c_int32, _i32]
    lib.dani_host_dynamic_areas.argtypes = [_f, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_char_p), C.c_int32,
                                            C.POINTER(C.c_char_p), C.c_int32, _i32, _i32, C.c_int32]
    lib.dani_yolo_debug_raw.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _f, C.c_int64]
    lib.dani_yolo_set_classes.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_int32, C.POINTER(C.c_char_p), C.c_int32]
    lib.dani_frontend_batch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_int32, C.POINTER(_FrontendParams), _f, _f, _f, C.c_int32, _i32, _i32,
                                        _i32, C.c_int32, _i32, _i32, _f, C.c_int32, _i32]
    lib.dani_frontend_reset.argtypes = [C.c_void_p, C.c_int32]
    lib.dani_frontend_finish_pending.argtypes = [C.c_void_p, C.c_int32]
    lib.dani_frontend_export_features.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, _i32]
    lib.dani_frontend_match_resident.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_float, _i32, _f, C.c_int32, _i32]
    lib.dani_frontend_timings.argtypes = [C.c_void_p, _f]
    lib.dani_comm_available.argtypes = []
    lib.dani_comm_unique_id.argtypes = [C.c_void_p, C.c_int32]
    lib.dani_comm_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    lib.dani_comm_destroy.argtypes = [C.c_void_p]
    lib.dani_comm_reset_carry.argtypes = [C.c_void_p]
    lib.dani_comm_info.argtypes = [C.c_void_p, _i32, _i32, _i32, C.POINTER(C.c_int64)]
    lib.dani_frontend_exchange_boundary.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p), _i32]
    lib.dani_frontend_match_exchanged.argtypes = [C.c_void_p, C.c_int32, C.c_float, _i32, _f, C.c_int32, _i32]
    lib.dani_desc_best2.argtypes = [C.c_void_p, _f, C.c_int32, _f, C.c_int32, _i32, _i32, _i32, _f, _f, _i32, _i32, _i32]
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


class _Grid(C.Structure):
    _fields_ = [("min_x", C.c_float), ("min_y", C.c_float), ("inv_w", C.c_float), ("inv_h", C.c_float), ("cols", C.c_int32), ("rows", C.c_int32)]


def host_assign_features_to_grid(kxy, min_x, min_y, inv_w, inv_h, cols=64, rows=48):
    """Frame::AssignFeaturesToGrid (src/Frame.cc:387-418): CSR (cell_offsets [cols*rows+1], cell_idx), cell = col*rows + row."""
    lib = load_library()
    kxy = np.ascontiguousarray(kxy, np.float32).reshape(-1, 2)
    g = _Grid(min_x, min_y, inv_w, inv_h, cols, rows)
    offs = np.zeros(cols * rows + 1, np.int32); idx = np.zeros(max(len(kxy), 1), np.int32)
    m = lib.dani_host_assign_features_to_grid(_fp(kxy), len(kxy), C.byref(g), _ip(offs), _ip(idx))
    if m < 0:
        raise DaniError(m, "dani_host_assign_features_to_grid")
    return offs, idx[:m].copy()


def host_features_in_area(kxy, octave, grid_params, cell_offsets, cell_idx, qx, qy, qr, min_level=None, max_level=None, cap=1 << 22):
    """Frame::GetFeaturesInArea (src/Frame.cc:659-725) for many windows: returns a list of candidate index arrays."""
    lib = load_library()
    kxy = np.ascontiguousarray(kxy, np.float32).reshape(-1, 2)
    g = _Grid(*grid_params)
    qx = np.ascontiguousarray(qx, np.float32); qy = np.ascontiguousarray(qy, np.float32); qr = np.ascontiguousarray(qr, np.float32)
    nq = len(qx)
    octv = np.ascontiguousarray(octave, np.int32) if octave is not None else None
    mn = np.ascontiguousarray(min_level, np.int32) if min_level is not None else None
    mx = np.ascontiguousarray(max_level, np.int32) if max_level is not None else None
    offs = np.zeros(nq + 1, np.int32); idx = np.zeros(cap, np.int32)
    cell_offsets = np.ascontiguousarray(cell_offsets, np.int32); cell_idx = np.ascontiguousarray(cell_idx, np.int32)
    m = lib.dani_host_features_in_area(_fp(kxy), _ip(octv) if octv is not None else None, len(kxy), C.byref(g), _ip(cell_offsets),
                                       _ip(cell_idx) if len(cell_idx) else None, _fp(qx), _fp(qy), _fp(qr),
                                       _ip(mn) if mn is not None else None, _ip(mx) if mx is not None else None, nq, _ip(offs), _ip(idx), cap)
    if m < 0:
        raise DaniError(m, "dani_host_features_in_area")
    return [idx[offs[i]:offs[i + 1]].copy() for i in range(nq)]


def host_undistort_keypoints(kpts_xy, K4, dist):
    """Frame::UndistortKeyPoints (src/Frame.cc:749-782): undistorted pixel coordinates [n, 2] float32."""
    lib = load_library()
    k = np.ascontiguousarray(kpts_xy, np.float32).reshape(-1, 2)
    K = np.ascontiguousarray(K4, np.float32); d = np.ascontiguousarray(dist, np.float32)
    out = np.empty((max(len(k), 1), 2), np.float32)
    lib.dani_host_undistort_keypoints.argtypes = [_f, C.c_int32, _f, _f, C.c_int32, _f]
    rc = lib.dani_host_undistort_keypoints(_fp(k) if len(k) else None, len(k), _fp(K), _fp(d) if len(d) else None, len(d), _fp(out))
    if rc:
        raise DaniError(rc, "dani_host_undistort_keypoints")
    return out[:len(k)]


def host_stereo_from_matches(ul_x, ur_x, matches, scores, mb, mbf):
    """Frame::ComputeStereoMatches post-filter (src/Frame.cc:862-914): returns (mvuRight, mvDepth)."""
    lib = load_library()
    ul = np.ascontiguousarray(ul_x, np.float32); ur = np.ascontiguousarray(ur_x, np.float32)
    m = np.ascontiguousarray(matches, np.int32).reshape(-1, 2); s = np.ascontiguousarray(scores, np.float32)
    u_right = np.empty(max(len(ul), 1), np.float32); depth = np.empty(max(len(ul), 1), np.float32)
    rc = lib.dani_host_stereo_from_matches(_fp(ul), len(ul), _fp(ur), len(ur), _ip(m) if len(m) else None, _fp(s) if len(m) else None,
                                           len(m), C.c_float(mb), C.c_float(mbf), _fp(u_right), _fp(depth))
    if rc:
        raise DaniError(rc, "dani_host_stereo_from_matches")
    return u_right[:len(ul)], depth[:len(ul)]


def host_dynamic_areas(dets6, cols, rows, class_names, dynamic_names):
    """Detect()'s box bookkeeping + RefineDynamicBoxes (src/YoloDetect.cpp:51-80,148-208).
    Returns (dynamic boxes [k,4] int32, det rects [n,4] int32)."""
    lib = load_library()
    dets6 = np.ascontiguousarray(dets6, np.float32).reshape(-1, 6)
    n = len(dets6)
    cn = (C.c_char_p * len(class_names))(*[x.encode() for x in class_names])
    dn = (C.c_char_p * len(dynamic_names))(*[x.encode() for x in dynamic_names])
    rects = np.zeros((max(n, 1), 4), np.int32)
    cap = max(n, 1) + 1
    dyn = np.zeros((cap, 4), np.int32)
    k = lib.dani_host_dynamic_areas(_fp(dets6), n, cols, rows, cn, len(class_names), dn, len(dynamic_names), _ip(rects),
                                    _ip(dyn), cap)
    if k < 0:
        raise DaniError(k, "dani_host_dynamic_areas")
    return dyn[:k].copy(), rects[:n].copy()


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
        cap = cap or (nfeatures + 8 + 4 * nlevels)
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

    def sp_layer_kernel(self, layer):
        """Name of the kernel instantiation SuperPoint layer 0..11 runs as on this engine (dani_sp_layer_kernel)."""
        buf = C.create_string_buffer(256)
        self._check(self.lib.dani_sp_layer_kernel(self.h, layer, buf, 256))
        return buf.value.decode()

    def sp_debug_layer(self, b, layer, shape):
        out = np.empty(shape, np.float32)
        self._check(self.lib.dani_sp_debug_layer(self.h, b, layer, _fp(out), out.size))
        return out

    # -- LightGlue -------------------------------------------------------------------------------
    def lg_match_batch(self, kpts0, desc0, n0, kpts1, desc1, n1, size0=None, size1=None, threshold=0.05, cap=None,
                       device_ptrs=None, in_cap=None):
        """P pairs.  Host arrays kpts [P, in_cap, 2], desc [P, in_cap, 256], counts n0/n1 [P]; or
        device_ptrs=(kpts0, desc0, kpts1, desc1) raw device addresses with in_cap given.
        Returns a list of (matches [K,2] int32, scores [K] f32)."""
        n0 = np.ascontiguousarray(n0, np.int32).reshape(-1); n1 = np.ascontiguousarray(n1, np.int32).reshape(-1)
        P = len(n0)
        if device_ptrs is None:
            kpts0 = np.ascontiguousarray(kpts0, np.float32); kpts1 = np.ascontiguousarray(kpts1, np.float32)
            desc0 = np.ascontiguousarray(desc0, np.float32); desc1 = np.ascontiguousarray(desc1, np.float32)
            in_cap = kpts0.shape[1]
            assert kpts1.shape[1] == in_cap and desc0.shape[1] == in_cap and desc1.shape[1] == in_cap
            ptrs = [a.ctypes.data_as(C.c_void_p) for a in (kpts0, desc0, kpts1, desc1)]
            on_dev = 0
        else:
            ptrs = [C.c_void_p(p) for p in device_ptrs]
            on_dev = 1
        cap = cap or max(1, int(max(n0.max(), 1)))
        m = np.empty((P, cap, 2), np.int32); sc = np.empty((P, cap), np.float32); k = np.zeros(P, np.int32)
        s0 = np.asarray(size0, np.float32) if size0 is not None else None
        s1 = np.asarray(size1, np.float32) if size1 is not None else None
        rc = self.lib.dani_lg_match_batch(self.h, P, in_cap, on_dev, ptrs[0], ptrs[1], _ip(n0), ptrs[2], ptrs[3], _ip(n1),
                                          _fp(s0) if s0 is not None else None, _fp(s1) if s1 is not None else None,
                                          C.c_float(threshold), _ip(m), _fp(sc), cap, _ip(k))
        self._check(rc)
        return [(m[p, :k[p]].copy(), sc[p, :k[p]].copy()) for p in range(P)]

    def lg_match(self, kpts0, desc0, kpts1, desc1, size0=None, size1=None, threshold=0.05):
        kpts0 = np.asarray(kpts0, np.float32).reshape(-1, 2); kpts1 = np.asarray(kpts1, np.float32).reshape(-1, 2)
        n0, n1 = len(kpts0), len(kpts1)
        if n0 == 0 or n1 == 0:
            return np.zeros((0, 2), np.int32), np.zeros(0, np.float32)
        cap = max(n0, n1)
        def pad(a, w):
            out = np.zeros((1, cap, w), np.float32); out[0, :len(a)] = np.asarray(a, np.float32).reshape(-1, w); return out
        return self.lg_match_batch(pad(kpts0, 2), pad(desc0, 256), [n0], pad(kpts1, 2), pad(desc1, 256), [n1],
                                   size0, size1, threshold)[0]

    def lg_match_groups(self, kpts0, desc0, kpts1, desc1, groups0, groups1, size0=None, size1=None, threshold=0.05):
        """dani_lg_match_groups: groups0[g] / groups1[g] are the row lists of frame 0 / frame 1 matched in group g (one
        vocabulary node in LGmatcher::SearchForTriangulation).  Returns a list (len G) of (matches [k,2] local to the
        group's lists, scores [k])."""
        k0 = np.ascontiguousarray(kpts0, np.float32).reshape(-1, 2); d0 = np.ascontiguousarray(desc0, np.float32).reshape(-1, 256)
        k1 = np.ascontiguousarray(kpts1, np.float32).reshape(-1, 2); d1 = np.ascontiguousarray(desc1, np.float32).reshape(-1, 256)
        G = len(groups0)
        assert len(groups1) == G
        off0 = np.zeros(G + 1, np.int32); off1 = np.zeros(G + 1, np.int32)
        off0[1:] = np.cumsum([len(g) for g in groups0]); off1[1:] = np.cumsum([len(g) for g in groups1])
        cat = lambda gs, n: np.ascontiguousarray(np.concatenate([np.asarray(g, np.int32).reshape(-1) for g in gs]) if n else np.zeros(1, np.int32), np.int32)
        i0, i1 = cat(groups0, off0[G]), cat(groups1, off1[G])
        cap = max(1, int(sum(min(len(a), len(b)) for a, b in zip(groups0, groups1))))
        m = np.empty((cap, 2), np.int32); sc = np.empty(cap, np.float32); mo = np.zeros(G + 1, np.int32)
        s0 = np.asarray(size0, np.float32) if size0 is not None else None
        s1 = np.asarray(size1, np.float32) if size1 is not None else None
        self._check(self.lib.dani_lg_match_groups(self.h, _fp(k0), _fp(d0), len(k0), _fp(k1), _fp(d1), len(k1), G, _ip(off0), _ip(i0),
                                                  _ip(off1), _ip(i1), _fp(s0) if s0 is not None else None,
                                                  _fp(s1) if s1 is not None else None, C.c_float(threshold), _ip(m), _fp(sc), _ip(mo), cap))
        return [(m[mo[g]:mo[g + 1]].copy(), sc[mo[g]:mo[g + 1]].copy()) for g in range(G)]

    HOST_WAIT = {"block": 0, "spin": 1, "hybrid": 2}

    def set_host_wait(self, mode):
        """Process-wide (dani_set_host_wait): "block" = hipStreamSynchronize, "spin" (the library's default) = poll hipStreamQuery,
        "hybrid" = nap until the engine's near-end marker, then spin.  True / False = spin / block (older callers)."""
        if isinstance(mode, bool):
            mode = "spin" if mode else "block"
        self._check(self.lib.dani_set_host_wait(self.HOST_WAIT[mode]))

    def host_wait(self):
        """The wait mode in force (dani_get_host_wait) as its name."""
        v = self.lib.dani_get_host_wait()
        return {m: k for k, m in self.HOST_WAIT.items()}[v]

    def lg_set_kernel_option(self, name, value):
        """Process-wide kernel selection of the f32 LightGlue path (dani_lg_set_kernel_option): 'lin_rs', 'attn'."""
        self._check(self.lib.dani_lg_set_kernel_option(name.encode(), int(value)))

    def lg_attention_profile(self):
        """(total ms, EXECUTED flop, launches, ALGORITHMIC flop) of the attention launches of the last LightGlue call (profiling on):
        a cross launch computes K Q^T once per direction; SURVEY.md section 8d counts a pair's cross scores once."""
        ms = C.c_float(0); fl = C.c_double(0); fa = C.c_double(0); n = C.c_int32(0)
        self._check(self.lib.dani_lg_attention_profile_ex(self.h, C.byref(ms), C.byref(fl), C.byref(fa), C.byref(n)))
        return ms.value, fl.value, n.value, fa.value

    def lg_linear_profile(self):
        """(total ms, total algorithmic flop, launches) of the linear_fm launches in the last LightGlue call (profiling on)."""
        ms = C.c_float(0); fl = C.c_double(0); n = C.c_int32(0)
        self._check(self.lib.dani_lg_linear_profile(self.h, C.byref(ms), C.byref(fl), C.byref(n)))
        return ms.value, fl.value, n.value

    def lg_debug_fetch(self, set_index, n):
        d = np.empty((n, 256), np.float32); m0 = np.empty(n, np.int32); mx = np.empty(n, np.float32)
        self._check(self.lib.dani_lg_debug_fetch(self.h, set_index, _fp(d), _ip(m0), _fp(mx), n))
        return d, m0, mx

    # -- YOLO ------------------------------------------------------------------------------------
    def yolo_detect_batch(self, bgr, score_thresh=0.4, iou_thresh=0.5, cap=2048, device_ptr=None, shape=None):
        """bgr: uint8 [B, H, W, 3] host array (or device_ptr + shape=(B, H, W, stride)).  Returns a list of dets [k, 6]."""
        if device_ptr is None:
            bgr = np.ascontiguousarray(bgr, np.uint8)
            if bgr.ndim == 3:
                bgr = bgr[None]
            B, H, W, _ = bgr.shape
            stride = bgr.strides[1]
            ptr, on_dev = bgr.ctypes.data_as(C.c_void_p), 0
        else:
            B, H, W, stride = shape
            ptr, on_dev = C.c_void_p(device_ptr), 1
        d = np.empty((B, cap, 6), np.float32); n = np.zeros(B, np.int32)
        self._check(self.lib.dani_yolo_detect_batch(self.h, ptr, on_dev, B, W, H, stride, C.c_float(score_thresh),
                                                    C.c_float(iou_thresh), _fp(d), cap, _ip(n)))
        return [d[b, :n[b]].copy() for b in range(B)]

    def yolo_debug_raw(self, b, level):
        ny, nx = [(48, 80), (24, 40), (12, 20)][level]
        out = np.empty((255, ny, nx), np.float32)
        self._check(self.lib.dani_yolo_debug_raw(self.h, b, level, _fp(out), out.size))
        return out

    # -- fused front-end (sequence mode) ------------------------------------------------------------
    def yolo_set_classes(self, class_names, dynamic_names):
        cn = (C.c_char_p * len(class_names))(*[x.encode() for x in class_names])
        dn = (C.c_char_p * len(dynamic_names))(*[x.encode() for x in dynamic_names])
        self._check(self.lib.dani_yolo_set_classes(self.h, cn, len(class_names), dn, len(dynamic_names)))

    def frontend_reset(self, sequence=0):
        self._check(self.lib.dani_frontend_reset(self.h, sequence))

    def frontend_finish_pending(self, sequence=0):
        """Complete a descriptor read-back left running by frontend_batch(defer_desc=True); keeps the cached predecessor."""
        self._check(self.lib.dani_frontend_finish_pending(self.h, sequence))

    def frontend_batch(self, gray, bgr=None, nfeatures=1024, iniTh=0.015, minTh=0.007, use_yolo=True, match_previous=True,
                       lg_threshold=0.05, lapping=(0, 0), want_desc=True, device_ptrs=None, shape=None, box_cap=512, copy=True,
                       sequence=0, scaleFactor=1.2, nlevels=1, stereo_pairs=False, defer_desc=False, match_cap=None):
        """gray [B,H,W] u8 and bgr [B,H,W,3] u8 host arrays, or device_ptrs=(gray_ptr, bgr_ptr) + shape=(B,H,W).
        Returns a list of per-frame dicts: kpts, resp, desc (or None), mono_index, boxes [k,4], matches [K,2], scores.
        copy=False returns views into buffers that the next call overwrites.
        stereo_pairs: frames (2k, 2k+1) are left / right images; frame 2k+1 carries the left-vs-right matches
        (dani_frontend_params.pair_mode = 1, Frame::ComputeStereoMatches).
        match_cap: per-frame capacity of the match outputs (default: the keypoint capacity)."""
        if defer_desc and copy:
            raise ValueError("defer_desc needs copy=False: the descriptor rows are complete only after frontend_match_resident")
        if device_ptrs is None:
            gray = np.ascontiguousarray(gray, np.uint8)
            B, H, W = gray.shape
            gp, gs = gray.ctypes.data_as(C.c_void_p), gray.strides[1]
            if use_yolo:
                bgr = np.ascontiguousarray(bgr, np.uint8)
                bp, bs = bgr.ctypes.data_as(C.c_void_p), bgr.strides[1]
            else:
                bp, bs = None, 0
            on_dev = 0
        else:
            B, H, W = shape
            gp, gs = C.c_void_p(device_ptrs[0]), W
            bp, bs = (C.c_void_p(device_ptrs[1]), 3 * W) if use_yolo else (None, 0)
            on_dev = 1
        cap = nfeatures + 8 + 4 * nlevels      # DistributeOctTree may exceed its share by up to 3 per level
        # the parameter block and the output pointers are built once per distinct call shape (this wrapper sits inside bench.py's
        # timed region: 0.07 ms per call when everything was converted anew every time)
        pkey = (nfeatures, scaleFactor, nlevels, iniTh, minTh, bool(use_yolo), bool(match_previous), lg_threshold, int(lapping[0]), int(lapping[1]),
                bool(stereo_pairs), bool(defer_desc))
        if not hasattr(self, "_fe_params"):
            self._fe_params = {}
        p = self._fe_params.get(pkey)
        if p is None:
            p = _FrontendParams()
            p.sp = _SpParams(nfeatures, scaleFactor, nlevels, iniTh, minTh)
            p.use_yolo = int(use_yolo); p.yolo_score_thresh = 0.4; p.yolo_iou_thresh = 0.5
            p.match_previous = int(match_previous); p.lg_threshold = lg_threshold
            p.lapping[0], p.lapping[1] = int(lapping[0]), int(lapping[1])
            p.pair_mode = 1 if stereo_pairs else 0
            # split form (match_previous=False, then frontend_match_resident): the descriptor rows of the returned views are
            # complete only after frontend_match_resident / frontend_reset / the next frontend_batch of the sequence
            p.defer_desc = 1 if (defer_desc and not match_previous) else 0
            self._fe_params[pkey] = p
        if not hasattr(self, "_fe_bufs_by_seq"):
            self._fe_bufs_by_seq = {}
        bufs = self._fe_bufs_by_seq.get(sequence)
        mcap = int(match_cap) if match_cap else cap
        key = (B, cap, box_cap, want_desc, mcap)
        if bufs is None or bufs[0] != key:
            if bufs is not None:
                # the engine may still hold a pointer into the old descriptor array (defer_desc): finish that read-back
                # before the array is dropped
                self._check(self.lib.dani_frontend_finish_pending(self.h, sequence))
            guard = 64   # canary words behind the match arrays (tests check that a capacity error wrote nothing past them)
            mflat = np.full(B * mcap * 2 + guard, -7777, np.int32); scflat = np.full(B * mcap + guard, -7777.0, np.float32)
            bufs = (key, dict(k=np.empty((B, cap, 2), np.float32), r=np.empty((B, cap), np.float32),
                              d=np.empty((B, cap, 256), np.float32) if want_desc else None,
                              n=np.zeros(B, np.int32), mono=np.zeros(B, np.int32),
                              boxes=np.zeros((B, box_cap, 4), np.int32), nb=np.zeros(B, np.int32),
                              m=mflat[:B * mcap * 2].reshape(B, mcap, 2), sc=scflat[:B * mcap].reshape(B, mcap),
                              m_guard=mflat[B * mcap * 2:], sc_guard=scflat[B * mcap:], ko=np.zeros(B, np.int32)))
            bufs[1]["ptrs"] = (_fp(bufs[1]["k"]), _fp(bufs[1]["r"]), _fp(bufs[1]["d"]) if want_desc else None, _ip(bufs[1]["n"]), _ip(bufs[1]["mono"]),
                               _ip(bufs[1]["boxes"]), _ip(bufs[1]["nb"]), _ip(bufs[1]["m"]), _fp(bufs[1]["sc"]), _ip(bufs[1]["ko"]))
            self._fe_bufs_by_seq[sequence] = bufs
        o = bufs[1]
        pk, pr, pd, pn, pmono, pboxes, pnb, pm, psc, pko = o["ptrs"]
        rc = self.lib.dani_frontend_batch(self.h, sequence, gp, gs, bp, bs, on_dev, B, W, H, C.byref(p), pk, pr, pd, cap, pn, pmono,
                                          pboxes, box_cap, pnb, pm, psc, mcap, pko)
        if rc:
            self._check(rc)
        ns, nbs, ks, monos = o["n"].tolist(), o["nb"].tolist(), (o["ko"].tolist() if match_previous else [0] * B), o["mono"].tolist()
        ok, orr, od, oboxes, om, osc = o["k"], o["r"], o["d"], o["boxes"], o["m"], o["sc"]
        res = [dict(kpts=ok[b, :ns[b]], resp=orr[b, :ns[b]], desc=od[b, :ns[b]] if want_desc else None, mono_index=monos[b],
                    boxes=oboxes[b, :nbs[b]], matches=om[b, :ks[b]], scores=osc[b, :ks[b]]) for b in range(B)]
        if copy:
            res = [{k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in r.items()} for r in res]
        return res

    @staticmethod
    def feature_slab_floats(cap):
        """Size of the packed feature slab of dani_frontend_export_features: count, kpts [cap][2], desc [cap][256]."""
        return 1 + cap * 258

    def frontend_export_features(self, frame, cap, sequence=0, device_ptr=None):
        """Pack resident frame `frame` of the last frontend_batch call of `sequence` into a slab (feature_slab_floats(cap)
        floats): into the device buffer at `device_ptr` (returns the keypoint count), or into a new host array (returned)."""
        n = C.c_int32(0)
        if device_ptr is not None:
            self._check(self.lib.dani_frontend_export_features(self.h, sequence, frame, C.c_void_p(device_ptr), 1, cap, C.byref(n)))
            return n.value
        slab = np.zeros(self.feature_slab_floats(cap), np.float32)
        self._check(self.lib.dani_frontend_export_features(self.h, sequence, frame, slab.ctypes.data_as(C.c_void_p), 0, cap, C.byref(n)))
        return slab

    def frontend_match_resident(self, nframes, pred=None, pred_cap=0, pred_device_ptr=None, lg_threshold=0.05, cap=None, sequence=0):
        """LightGlue over the frames the last frontend_batch(match_previous=False) call left resident: frame b against b-1,
        frame 0 against `pred` (host slab array, or pred_device_ptr + pred_cap; None = the sequence's cached predecessor).
        Returns a list of (matches [k,2], scores [k]) per frame."""
        cap = cap or 4096
        m = np.zeros((nframes, cap, 2), np.int32); sc = np.zeros((nframes, cap), np.float32); k = np.zeros(nframes, np.int32)
        if pred_device_ptr is not None:
            pp, on_dev = C.c_void_p(pred_device_ptr), 1
        elif pred is not None:
            pred = np.ascontiguousarray(pred, np.float32)
            pred_cap = (pred.size - 1) // 258
            pp, on_dev = pred.ctypes.data_as(C.c_void_p), 0
        else:
            pp, on_dev, pred_cap = None, 0, 0
        self._check(self.lib.dani_frontend_match_resident(self.h, sequence, pp, on_dev, pred_cap, C.c_float(lg_threshold), _ip(m), _fp(sc), cap, _ip(k)))
        return [(m[b, :k[b]].copy(), sc[b, :k[b]].copy()) for b in range(nframes)]

    # -- the sharded batch's collective, native (RCCL all-gather behind the C ABI; include/dani_frontend.h) -------------
    COMM_ID_BYTES = 128

    @staticmethod
    def comm_available() -> bool:
        """True when librccl loads with every symbol the engine uses (dani_comm_available); no communicator, no id."""
        return load_library().dani_comm_available() == 0

    @staticmethod
    def comm_unique_id() -> bytes:
        """ncclGetUniqueId through the engine's library (rank 0); hand the 128 bytes to the other ranks by any means."""
        lib = load_library()
        buf = C.create_string_buffer(Engine.COMM_ID_BYTES)
        rc = lib.dani_comm_unique_id(buf, Engine.COMM_ID_BYTES)
        if rc:
            raise DaniError(rc, lib.dani_last_error(None).decode())
        return buf.raw

    def comm_create(self, unique_id: bytes, rank: int, world: int):
        """ncclCommInitRank on the engine's GPU: a collective, every rank calls it with the same id."""
        assert len(unique_id) >= self.COMM_ID_BYTES
        buf = C.create_string_buffer(bytes(unique_id), len(unique_id))
        self._check(self.lib.dani_comm_create(self.h, buf, len(unique_id), rank, world))

    def comm_destroy(self):
        self._check(self.lib.dani_comm_destroy(self.h))

    def comm_reset_carry(self):
        self._check(self.lib.dani_comm_reset_carry(self.h))

    def comm_info(self):
        """dict(rank, world, rccl_version, collectives) of the engine's communicator."""
        r = C.c_int32(0); w = C.c_int32(0); v = C.c_int32(0); n = C.c_int64(0)
        self._check(self.lib.dani_comm_info(self.h, C.byref(r), C.byref(w), C.byref(v), C.byref(n)))
        return dict(rank=r.value, world=w.value, rccl_version=v.value, collectives=n.value)

    def frontend_exchange_boundary(self, frame, cap, sequence=0, device_ptr=None):
        """Pack resident frame `frame` and all-gather it over the communicator (enqueued on the sequence's stream, no host
        wait).  Returns (keypoint count of the own frame, device address of the gathered [world][1 + cap * 258] floats)."""
        n = C.c_int32(0); out = C.c_void_p()
        self._check(self.lib.dani_frontend_exchange_boundary(self.h, sequence, frame, cap, C.c_void_p(device_ptr) if device_ptr else None,
                                                             C.byref(out), C.byref(n)))
        return n.value, out.value

    def frontend_match_exchanged(self, nframes, cap=None, lg_threshold=0.05, sequence=0):
        """LightGlue over the resident frames, frame 0 against the neighbouring rank's slab of the sequence's last exchange
        (rank 0: the last rank's slab of the previous exchange).  Returns a list of (matches [k,2], scores [k]) per frame."""
        cap = cap or 4096
        m = np.zeros((nframes, cap, 2), np.int32); sc = np.zeros((nframes, cap), np.float32); k = np.zeros(nframes, np.int32)
        self._check(self.lib.dani_frontend_match_exchanged(self.h, sequence, C.c_float(lg_threshold), _ip(m), _fp(sc), cap, _ip(k)))
        return [(m[b, :k[b]].copy(), sc[b, :k[b]].copy()) for b in range(nframes)]

    def frontend_timings(self):
        ms = np.zeros(4, np.float32)
        self._check(self.lib.dani_frontend_timings(self.h, _fp(ms)))
        return ms

    # -- projection-search descriptor scans -----------------------------------------
    def desc_best2(self, q_desc, t_desc, t_level, cand_lists):
        """LGmatcher::DescriptorDistance + best/second-best scan (src/LGmatcher.cpp:84-120, 2009-2013).
        cand_lists: list (len nq) of index lists into t_desc.  Returns arrays (bestDist, bestDist2, bestIdx, bestLevel, bestLevel2)."""
        q = np.ascontiguousarray(q_desc, np.float32).reshape(-1, 256); t = np.ascontiguousarray(t_desc, np.float32).reshape(-1, 256)
        nq, nt = len(q), len(t)
        offs = np.zeros(nq + 1, np.int32)
        for i, cl in enumerate(cand_lists):
            offs[i + 1] = offs[i] + len(cl)
        idx = np.ascontiguousarray(np.concatenate([np.asarray(cl, np.int32) for cl in cand_lists]) if offs[nq] else np.zeros(1, np.int32), np.int32)
        lvl = np.ascontiguousarray(t_level, np.int32) if t_level is not None else None
        bd = np.empty(nq, np.float32); bd2 = np.empty(nq, np.float32)
        bi = np.empty(nq, np.int32); bl = np.empty(nq, np.int32); bl2 = np.empty(nq, np.int32)
        self._check(self.lib.dani_desc_best2(self.h, _fp(q), nq, _fp(t), nt, _ip(lvl) if lvl is not None else None, _ip(offs), _ip(idx),
                                             _fp(bd), _fp(bd2), _ip(bi), _ip(bl), _ip(bl2)))
        return bd, bd2, bi, bl, bl2

    def desc_scan(self, q_desc, t_desc, t_level, cand_lists):
        """desc_best2 plus the distance of every candidate (list of arrays aligned with cand_lists): what a host replay of
        SearchByProjection's sequential claim logic needs (dani_desc_scan)."""
        q = np.ascontiguousarray(q_desc, np.float32).reshape(-1, 256); t = np.ascontiguousarray(t_desc, np.float32).reshape(-1, 256)
        nq, nt = len(q), len(t)
        offs = np.zeros(nq + 1, np.int32)
        offs[1:] = np.cumsum([len(cl) for cl in cand_lists])
        idx = np.ascontiguousarray(np.concatenate([np.asarray(cl, np.int32) for cl in cand_lists]) if offs[nq] else np.zeros(1, np.int32), np.int32)
        lvl = np.ascontiguousarray(t_level, np.int32) if t_level is not None else None
        bd = np.empty(nq, np.float32); bd2 = np.empty(nq, np.float32)
        bi = np.empty(nq, np.int32); bl = np.empty(nq, np.int32); bl2 = np.empty(nq, np.int32)
        cd = np.empty(max(int(offs[nq]), 1), np.float32)
        self._check(self.lib.dani_desc_scan(self.h, _fp(q), nq, _fp(t), nt, _ip(lvl) if lvl is not None else None, _ip(offs), _ip(idx),
                                            _fp(bd), _fp(bd2), _ip(bi), _ip(bl), _ip(bl2), _fp(cd)))
        return (bd, bd2, bi, bl, bl2), [cd[offs[i]:offs[i + 1]] for i in range(nq)]

    def desc_mean(self, rows, obs_lists):
        """MapPoint::ComputeDistinctiveDescriptors for many map points (src/MapPoint.cc:330-379): mean of rows[obs] in list order."""
        rows = np.ascontiguousarray(rows, np.float32).reshape(-1, 256)
        nq = len(obs_lists)
        offs = np.zeros(nq + 1, np.int32)
        offs[1:] = np.cumsum([len(o) for o in obs_lists])
        idx = np.ascontiguousarray(np.concatenate([np.asarray(o, np.int32) for o in obs_lists]) if offs[nq] else np.zeros(1, np.int32), np.int32)
        out = np.empty((max(nq, 1), 256), np.float32)
        self._check(self.lib.dani_desc_mean(self.h, _fp(rows), len(rows), _ip(offs), _ip(idx), nq, _fp(out)))
        return out[:nq]

    # -- DBoW3 transform -----------------------------------------------------------
    def vocab_create(self, L, child_offsets, child_ids, node_desc, node_weight, node_word_id):
        """A DBoW3 vocabulary tree on the device (dani_vocab_create); returns an opaque handle for bow_transform / vocab_destroy."""
        co = np.ascontiguousarray(child_offsets, np.int32); ci = np.ascontiguousarray(child_ids, np.int32)
        nd = np.ascontiguousarray(node_desc, np.float32).reshape(-1, 256)
        nw = np.ascontiguousarray(node_weight, np.float64); wi = np.ascontiguousarray(node_word_id, np.int32)
        h = C.c_void_p()
        self._check(self.lib.dani_vocab_create(self.h, len(nd), int(L), _ip(co), _ip(ci), _fp(nd), nw.ctypes.data_as(C.POINTER(C.c_double)),
                                               _ip(wi), C.byref(h)))
        return h

    def vocab_destroy(self, vocab):
        self.lib.dani_vocab_destroy(vocab)

    def bow_transform(self, vocab, feats, levelsup=4, weighting=0, norm=1):
        """Vocabulary::transform(features, BowVector, FeatureVector, levelsup): returns (bow dict word -> value,
        feature-vector dict node -> list of feature indices), both in ascending key order."""
        f = np.ascontiguousarray(feats, np.float32).reshape(-1, 256)
        n = len(f)
        wid = np.zeros(max(n, 1), np.int32); wv = np.zeros(max(n, 1), np.float64); nw = C.c_int32(0)
        fn = np.zeros(max(n, 1), np.int32); fo = np.zeros(n + 1, np.int32); ff = np.zeros(max(n, 1), np.int32); nf = C.c_int32(0)
        self._check(self.lib.dani_bow_transform(self.h, vocab, _fp(f), n, levelsup, weighting, norm, _ip(wid), wv.ctypes.data_as(C.POINTER(C.c_double)),
                                                C.byref(nw), _ip(fn), _ip(fo), _ip(ff), C.byref(nf)))
        bow = {int(wid[i]): float(wv[i]) for i in range(nw.value)}
        fv = {int(fn[i]): ff[fo[i]:fo[i + 1]].tolist() for i in range(nf.value)}
        return bow, fv

    # -- device-resident consumers (SURVEY 8f without the PCIe bounce) ----------------------------------
    def frontend_retain(self, frame, sequence=0, kpts_un=None, octave=None, grid=None):
        """Keep frame `frame` of the last frontend_batch call in HBM (dani_frontend_retain).  grid = (min_x, min_y, inv_w,
        inv_h, cols, rows) builds Frame::AssignFeaturesToGrid's lists on the device.  Returns an opaque handle."""
        h = C.c_void_p()
        ku = np.ascontiguousarray(kpts_un, np.float32) if kpts_un is not None else None
        oc = np.ascontiguousarray(octave, np.int32) if octave is not None else None
        g = _Grid(*grid) if grid is not None else None
        self._check(self.lib.dani_frontend_retain(self.h, sequence, frame, _fp(ku) if ku is not None else None,
                                                  _ip(oc) if oc is not None else None, C.byref(g) if g is not None else None, C.byref(h)))
        return h

    def features_undistort(self, f, K4, dist, grid=None):
        """Frame::UndistortKeyPoints on the retained coordinates, on the device; returns mvKeysUn [n, 2] (host copy)."""
        K = np.ascontiguousarray(K4, np.float32); d = np.ascontiguousarray(dist, np.float32)
        n = self.features_count(f)
        out = np.zeros((max(n, 1), 2), np.float32)
        g = _Grid(*grid) if grid is not None else None
        self._check(self.lib.dani_features_undistort(self.h, f, _fp(K), _fp(d) if len(d) else None, len(d), C.byref(g) if g is not None else None, _fp(out)))
        return out[:n]

    def features_release(self, f):
        self.lib.dani_features_release(f)

    def features_count(self, f):
        return self.lib.dani_features_count(f)

    def features_fetch_grid(self, f, ncells, n):
        offs = np.zeros(ncells + 1, np.int32); idx = np.zeros(max(n, 1), np.int32)
        m = self.lib.dani_features_fetch_grid(f, _ip(offs), _ip(idx))
        if m < 0:
            raise DaniError(m, "dani_features_fetch_grid")
        return offs, idx[:m].copy()

    def desc_scan_resident(self, q_desc, train, cand_lists):
        """desc_scan with the train descriptors / octaves of a retained frame (no descriptor upload)."""
        q = np.ascontiguousarray(q_desc, np.float32).reshape(-1, 256)
        nq = len(q)
        offs = np.zeros(nq + 1, np.int32)
        offs[1:] = np.cumsum([len(cl) for cl in cand_lists])
        idx = np.ascontiguousarray(np.concatenate([np.asarray(cl, np.int32) for cl in cand_lists]) if offs[nq] else np.zeros(1, np.int32), np.int32)
        bd = np.empty(nq, np.float32); bd2 = np.empty(nq, np.float32)
        bi = np.empty(nq, np.int32); bl = np.empty(nq, np.int32); bl2 = np.empty(nq, np.int32)
        cd = np.empty(max(int(offs[nq]), 1), np.float32)
        self._check(self.lib.dani_desc_scan_resident(self.h, _fp(q), nq, train, _ip(offs), _ip(idx), _fp(bd), _fp(bd2), _ip(bi), _ip(bl), _ip(bl2), _fp(cd)))
        return (bd, bd2, bi, bl, bl2), [cd[offs[i]:offs[i + 1]] for i in range(nq)]

    def search_windows_resident(self, train, q_desc, qx, qy, qr, min_level=None, max_level=None, want_lists=True, cand_cap=1 << 20):
        """Frame::GetFeaturesInArea + the best / second-best descriptor scan for nq map points on a retained frame, all on
        the device.  Returns ((bestDist, bestDist2, bestIdx, bestLevel, bestLevel2), candidate lists, candidate distances)."""
        q = np.ascontiguousarray(q_desc, np.float32).reshape(-1, 256)
        nq = len(q)
        qx = np.ascontiguousarray(qx, np.float32); qy = np.ascontiguousarray(qy, np.float32); qr = np.ascontiguousarray(qr, np.float32)
        mn = np.ascontiguousarray(min_level, np.int32) if min_level is not None else None
        mx = np.ascontiguousarray(max_level, np.int32) if max_level is not None else None
        bd = np.empty(nq, np.float32); bd2 = np.empty(nq, np.float32)
        bi = np.empty(nq, np.int32); bl = np.empty(nq, np.int32); bl2 = np.empty(nq, np.int32)
        offs = np.zeros(nq + 1, np.int32)
        ci = np.zeros(cand_cap if want_lists else 1, np.int32); cd = np.zeros(cand_cap if want_lists else 1, np.float32)
        self._check(self.lib.dani_search_windows_resident(self.h, train, _fp(q), nq, _fp(qx), _fp(qy), _fp(qr), _ip(mn) if mn is not None else None,
                                                          _ip(mx) if mx is not None else None, _fp(bd), _fp(bd2), _ip(bi), _ip(bl), _ip(bl2),
                                                          _ip(offs), _ip(ci), _fp(cd), cand_cap if want_lists else 0))
        lists = [ci[offs[i]:offs[i + 1]].copy() for i in range(nq)] if want_lists else None
        dists = [cd[offs[i]:offs[i + 1]].copy() for i in range(nq)] if want_lists else None
        return (bd, bd2, bi, bl, bl2), lists, dists

    def bow_transform_resident(self, vocab, train, levelsup=4, weighting=0, norm=1):
        n = self.features_count(train)
        wid = np.zeros(max(n, 1), np.int32); wv = np.zeros(max(n, 1), np.float64); nw = C.c_int32(0)
        fn = np.zeros(max(n, 1), np.int32); fo = np.zeros(n + 1, np.int32); ff = np.zeros(max(n, 1), np.int32); nf = C.c_int32(0)
        self._check(self.lib.dani_bow_transform_resident(self.h, vocab, train, levelsup, weighting, norm, _ip(wid), wv.ctypes.data_as(C.POINTER(C.c_double)),
                                                         C.byref(nw), _ip(fn), _ip(fo), _ip(ff), C.byref(nf)))
        bow = {int(wid[i]): float(wv[i]) for i in range(nw.value)}
        fv = {int(fn[i]): ff[fo[i]:fo[i + 1]].tolist() for i in range(nf.value)}
        return bow, fv

    LG_GEMM_F32, LG_GEMM_BF16X3, LG_GEMM_BF16X3_ALL = 0, 1, 2

    def lg_set_gemm(self, mode):
        """Arithmetic of LightGlue's linear layers: LG_GEMM_F32 (default, canonical order), LG_GEMM_BF16X3 (linear layers) or LG_GEMM_BF16X3_ALL (+ attention)."""
        self._check(self.lib.dani_lg_set_gemm(self.h, int(mode)))

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


@dataclass
class DMatch:
    """cv::DMatch as LightGlue::filterMatches fills it (src/LightGlue.cpp:114-119): distance = match confidence."""
    queryIdx: int
    trainIdx: int
    distance: float


class LightGlue:
    """Mirror of ORB_SLAM3::LightGlue (include/LightGlue.h:11-36).  Construction is O(1): the weights live in the
    process-wide engine (the reference jit::loads the blob in every constructor, quirk Q12)."""

    def __init__(self, engine: Engine | None = None):
        self.engine = engine or default_engine()

    def matchDescriptors(self, keypoints1, descriptors1, keypoints2, descriptors2, size1=None, size2=None,
                         threshold=0.05):
        """keypoints: sequences of KeyPoint (or [n,2] arrays); descriptors [n,256]; size = (w, h) or None/empty.
        Returns a list of DMatch, in increasing queryIdx order like the reference."""
        def xy(k):
            if len(k) and hasattr(k[0], "x"):
                return np.array([[p.x, p.y] for p in k], np.float32)
            return np.asarray(k, np.float32).reshape(-1, 2)
        s1 = None if size1 is None or len(size1) == 0 else size1
        s2 = None if size2 is None or len(size2) == 0 else size2
        m, s = self.engine.lg_match(xy(keypoints1), descriptors1, xy(keypoints2), descriptors2, s1, s2, threshold)
        return [DMatch(int(a), int(b), float(c)) for (a, b), c in zip(m, s)]


COCO_NAMES_DEFAULT = None   # set by YoloDetection from `coco.names` (CWD, as the reference) or the ctor argument


class YoloDetection:
    """Mirror of ::YoloDetection (include/YoloDetect.h:16-41): GetImage / Detect / ClearImage / ClearArea and the
    public result fields mvDynamicArea, mmDetectMap, mClassnames, mvDynamicNames."""

    def __init__(self, engine: Engine | None = None, class_names=None):
        self.engine = engine or default_engine()
        if class_names is None:
            if not os.path.exists("coco.names"):
                raise FileNotFoundError("coco.names (read from the CWD, src/YoloDetect.cpp:12) not found")
            class_names = [l.rstrip("\n") for l in open("coco.names")]
        self.mClassnames = list(class_names)
        self.mvDynamicNames = ["person", "car", "motorbike", "bus", "train", "truck", "boat", "bird", "cat",
                               "dog", "horse", "sheep", "crow", "bear"]     # src/YoloDetect.cpp:17-18
        self.mvDynamicArea = []
        self.mmDetectMap = {}
        self.mvPersonArea = []
        self.mRGB = None

    def GetImage(self, rgb): self.mRGB = rgb
    def ClearImage(self): self.mRGB = None
    def ClearArea(self): self.mvPersonArea = []

    def Detect(self) -> bool:
        if self.mRGB is None or np.asarray(self.mRGB).size == 0:
            print("Read RGB failed!")
            return False
        img = np.asarray(self.mRGB)
        dets = self.engine.yolo_detect_batch(img, 0.4, 0.5)[0]
        if len(dets) > 0:
            dyn, rects = host_dynamic_areas(dets, img.shape[1], img.shape[0], self.mClassnames, self.mvDynamicNames)
            for d, r in zip(dets, rects):
                self.mmDetectMap.setdefault(self.mClassnames[int(d[5])], []).append(tuple(int(v) for v in r))
            self.mvDynamicArea = [tuple(int(v) for v in r) for r in dyn]
        return True
