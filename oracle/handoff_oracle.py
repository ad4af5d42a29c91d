"""oracle/handoff_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (plain Python loops; the inputs are ~10^3 keypoints) of the feature hand-off formats right behind the
front-end:
    Frame::PosInGrid / AssignFeaturesToGrid      /root/reference/src/Frame.cc:727-737, 387-418   (Nleft == -1)
    Frame::GetFeaturesInArea                      src/Frame.cc:659-725
    Frame::ComputeStereoMatches (post-filter)     src/Frame.cc:862-914
    Frame::UndistortKeyPoints                     src/Frame.cc:749-782  (cv::undistortPoints of OpenCV 4.2, an external
                                                  dependency: restated from its published algorithm)
PARITY UNPINNED by any reference fixture (none exist); literal restatements of the cited lines.  Only tests/ may import
this module.
"""
import math

import numpy as np

f32 = np.float32


def _round_half_away(v):
    """C round(): half away from zero (numpy rounds half to even)."""
    v = float(v)
    return int(math.floor(v + 0.5)) if v >= 0 else -int(math.floor(-v + 0.5))


def assign_features_to_grid(kxy, min_x, min_y, inv_w, inv_h, cols=64, rows=48):
    """Returns grid[col][row] = list of keypoint indices (push_back order = ascending index)."""
    grid = [[[] for _ in range(rows)] for _ in range(cols)]
    for i, (x, y) in enumerate(np.asarray(kxy, f32).reshape(-1, 2)):
        px = _round_half_away(f32(f32(x - f32(min_x)) * f32(inv_w)))
        py = _round_half_away(f32(f32(y - f32(min_y)) * f32(inv_h)))
        if px < 0 or px >= cols or py < 0 or py >= rows:
            continue
        grid[px][py].append(i)
    return grid


def get_features_in_area(kxy, octave, grid, min_x, min_y, inv_w, inv_h, x, y, r, min_level=-1, max_level=-1):
    cols, rows = len(grid), len(grid[0])
    x, y, r = f32(x), f32(y), f32(r)
    out = []
    n_min_x = max(0, int(math.floor(f32(f32(x - f32(min_x) - r) * f32(inv_w)))))
    if n_min_x >= cols:
        return out
    n_max_x = min(cols - 1, int(math.ceil(f32(f32(x - f32(min_x) + r) * f32(inv_w)))))
    if n_max_x < 0:
        return out
    n_min_y = max(0, int(math.floor(f32(f32(y - f32(min_y) - r) * f32(inv_h)))))
    if n_min_y >= rows:
        return out
    n_max_y = min(rows - 1, int(math.ceil(f32(f32(y - f32(min_y) + r) * f32(inv_h)))))
    if n_max_y < 0:
        return out
    check = (min_level > 0) or (max_level >= 0)
    kxy = np.asarray(kxy, f32).reshape(-1, 2)
    for ix in range(n_min_x, n_max_x + 1):
        for iy in range(n_min_y, n_max_y + 1):
            for k in grid[ix][iy]:
                if check:
                    o = int(octave[k]) if octave is not None else 0
                    if o < min_level:
                        continue
                    if max_level >= 0 and o > max_level:
                        continue
                if abs(f32(kxy[k, 0] - x)) < r and abs(f32(kxy[k, 1] - y)) < r:
                    out.append(k)
    return out


def undistort_keypoints(kpts_xy, K4, dist):
    """Frame::UndistortKeyPoints (/root/reference/src/Frame.cc:749-782).  The reference delegates to
    cv::undistortPoints(mat, mat, K, mDistCoef, cv::Mat(), mK); OpenCV (4.2, found by the reference's CMake) is not in
    the reference tree, so this follows the published algorithm of calib3d for R = I, P = K and the default five
    iterations -- parity for this function is UNPINNED (no OpenCV here, no vectors in the reference)."""
    k = np.asarray(kpts_xy, f32).reshape(-1, 2)
    d = np.asarray(dist, f32)
    if len(d) == 0 or d[0] == 0.0:
        return k.copy()
    c = np.zeros(5, np.float64); c[:min(len(d), 5)] = d[:5].astype(np.float64)
    k1, k2, p1, p2, k3 = c
    fx, fy, cx, cy = [np.float64(v) for v in np.asarray(K4, f32)]
    ifx, ify = 1.0 / fx, 1.0 / fy
    out = np.empty_like(k)
    for i in range(len(k)):
        x = (np.float64(k[i, 0]) - cx) * ifx; y = (np.float64(k[i, 1]) - cy) * ify
        x0, y0 = x, y
        for _ in range(5):
            r2 = x * x + y * y
            icdist = 1.0 / (1.0 + ((k3 * r2 + k2) * r2 + k1) * r2)
            if icdist < 0:
                x, y = x0, y0
                break
            dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
            dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
            x = (x0 - dx) * icdist; y = (y0 - dy) * icdist
        out[i, 0] = f32(fx * x + cx); out[i, 1] = f32(fy * y + cy)
    return out


def stereo_from_matches(ul_x, ur_x, matches, scores, mb, mbf):
    """Returns (mvuRight, mvDepth) float32 arrays of len(ul_x)."""
    nl = len(ul_x)
    u_right = np.full(nl, -1, f32); depth = np.full(nl, -1, f32)
    mb, mbf = f32(mb), f32(mbf)
    max_d = f32(mbf / mb)
    disp = []
    for (il, ir), s in zip(np.asarray(matches).reshape(-1, 2), scores):
        il, ir = int(il), int(ir)
        if il >= nl or ir >= len(ur_x):
            continue
        ul, ur = f32(ul_x[il]), f32(ur_x[ir])
        d = f32(ul - ur)
        if d >= 0 and d < max_d:
            if d <= 0:
                d = f32(0.01)
            depth[il] = f32(mbf / d)
            u_right[il] = ur
            disp.append((f32(f32(1.0) - f32(s)), il))
    if not disp:
        return u_right, depth
    disp.sort()                                    # std::sort of pair<float, int>: by distance, then by index
    th = f32(f32(1.5) * disp[len(disp) // 2][0])
    for dist, il in reversed(disp):
        if dist < th:
            break
        u_right[il] = -1; depth[il] = -1
    return u_right, depth
