"""oracle/match_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the descriptor-distance scans of the reference's projection searches:
    LGmatcher::DescriptorDistance     /root/reference/src/LGmatcher.cpp:2009-2013   (float) cv::norm(a, b, NORM_L2)
    best / second-best bookkeeping    src/LGmatcher.cpp:84-120 (SearchByProjection; same scan at :444-464, :1694-1719, :1901-1916)
cv::norm(NORM_L2) on CV_32F: the difference is formed in float, squared and accumulated in double, sqrt, cast to
float (OpenCV normL2Sqr<float, double>; the library itself is absent from this image).  PARITY UNPINNED by any
reference fixture (none exist); literal restatement of the cited lines.
"""
import numpy as np


def descriptor_distance(a, b):
    d = (np.asarray(a, np.float32) - np.asarray(b, np.float32)).astype(np.float64)
    return np.float32(np.sqrt(np.sum(d * d)))


def best2_scan(q_desc, t_desc, t_level, cand_lists):
    """Per query: (bestDist, bestDist2, bestIdx, bestLevel, bestLevel2) over its candidate list, in list order."""
    out = []
    for qi, cands in enumerate(cand_lists):
        bd, bd2, bl, bl2, bi = np.float32(256), np.float32(256), -1, -1, -1
        for idx in cands:
            dist = descriptor_distance(q_desc[qi], t_desc[idx])
            lvl = int(t_level[idx]) if t_level is not None else 0
            if dist < bd:
                bd2, bd, bl2, bl, bi = bd, dist, bl, lvl, int(idx)
            elif dist < bd2:
                bl2, bd2 = lvl, dist
        out.append((bd, bd2, bi, bl, bl2))
    return out
