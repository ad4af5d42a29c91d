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


TH_HIGH = np.float32(0.9)      # src/LGmatcher.cpp:35
TH_LOW = np.float32(0.4)       # :36


def search_by_projection(q_desc, t_desc, t_level, cand_lists, nn_ratio=0.8, claimed=None, dist=None):
    """The selection loop of LGmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, ...) for the monocular / RGB-D
    case (src/LGmatcher.cpp:42-135, F.Nleft == -1, no right-image check): map points in order; candidates of
    GetFeaturesInArea in list order; keypoints already holding a map point with observations are skipped (:92-94);
    best / second-best with strict `<` (:104-120); accept if bestDist <= TH_HIGH and (levels differ or
    bestDist <= mfNNratio * bestDist2) (:124-129); an accepted match CLAIMS its keypoint for all later map points.
    `dist`: optional precomputed per-candidate distances (list of arrays aligned with cand_lists) -- what
    dani_desc_scan returns; None -> computed here.  Returns the list of (map point index, keypoint index)."""
    claimed = np.zeros(len(t_desc), bool) if claimed is None else np.array(claimed, bool)
    ratio = np.float32(nn_ratio)
    matches = []
    for qi, cands in enumerate(cand_lists):
        bd, bd2, bl, bl2, bi = np.float32(256), np.float32(256), -1, -1, -1
        for k, idx in enumerate(cands):
            if claimed[idx]:
                continue
            d = np.float32(dist[qi][k]) if dist is not None else descriptor_distance(q_desc[qi], t_desc[idx])
            lvl = int(t_level[idx]) if t_level is not None else 0
            if d < bd:
                bd2, bd, bl2, bl, bi = bd, d, bl, lvl, int(idx)
            elif d < bd2:
                bl2, bd2 = lvl, d
        if bi >= 0 and bd <= TH_HIGH:
            if bl == bl2 and bd > ratio * bd2:
                continue
            claimed[bi] = True
            matches.append((qi, bi))
    return matches


def distinctive_descriptor(rows, obs):
    """MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:330-379): meanDescriptor = zeros; += every observation's
    descriptor row in order (f32); /= (float)count.  None when there is no observation (the reference returns early)."""
    if len(obs) == 0:
        return None
    acc = np.zeros(256, np.float32)
    for i in obs:
        acc = (acc + np.asarray(rows[i], np.float32)).astype(np.float32)
    return (acc / np.float32(len(obs))).astype(np.float32)
