"""oracle/bow_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of DBoW3's transform of float descriptors, as Frame::ComputeBoW uses it
(/root/reference/src/Frame.cc:740-747 -> Vocabulary::transform(features, BowVector&, FeatureVector&, levelsup = 4)):
    Vocabulary::transform (tree descent)     Thirdparty/DBow3/src/Vocabulary.cpp:836-873
    Vocabulary::transform (bookkeeping)      Vocabulary.cpp:752-824
    DescManip::distance (CV_32F branch)      DescManip.cpp:120-129   float difference, float product, double accumulation
    BowVector::addWeight / addIfNotExist / normalize    BowVector.cpp:34-84
    FeatureVector::addFeature                FeatureVector.cpp:31-45
DBoW3 is vendored in the reference tree (Thirdparty/DBow3), but the SuperPoint vocabulary file is withheld upstream
 and the library needs OpenCV to build: PARITY UNPINNED by any reference fixture; literal restatement of the
cited lines on synthetic trees.  Only tests/ may import this module.
"""
import math

import numpy as np


def distance(a, b):
    d = (np.asarray(a, np.float32) - np.asarray(b, np.float32)).astype(np.float32)
    p = (d * d).astype(np.float32)                       # float product
    return float(np.cumsum(p.astype(np.float64))[-1])    # sequential double accumulation (cumsum is sequential)


def descend(tree, feature, levelsup):
    offs, ids, desc = tree["child_offsets"], tree["child_ids"], tree["desc"]
    nid_level = tree["L"] - levelsup
    nid = 0
    node, level = 0, 0
    while offs[node + 1] > offs[node]:
        level += 1
        best, best_id = float("inf"), None
        for c in ids[offs[node]:offs[node + 1]]:
            d = distance(feature, desc[c])
            if d < best:
                best, best_id = d, int(c)
        node = best_id
        if level == nid_level:
            nid = node
    return node, nid


def transform(tree, feats, levelsup=4, weighting=0, norm=1):
    """Returns (bow: dict word -> value in ascending key order, fv: dict node -> feature indices)."""
    bow, fv = {}, {}
    tf = weighting <= 1
    for i, f in enumerate(np.asarray(feats, np.float32).reshape(-1, 256)):
        leaf, nid = descend(tree, f, levelsup)
        w = float(tree["weight"][leaf])
        if not w > 0:
            continue
        wid = int(tree["word_id"][leaf])
        if tf:
            bow[wid] = bow.get(wid, 0.0) + w
        elif wid not in bow:
            bow[wid] = w
        fv.setdefault(nid, []).append(i)
    bow = dict(sorted(bow.items())); fv = dict(sorted(fv.items()))
    if tf and bow and norm == 0:
        nd = float(len(bow))
        bow = {k: v / nd for k, v in bow.items()}
    if norm:
        nn = 0.0
        if norm == 1:
            for v in bow.values():
                nn += abs(v)
        else:
            for v in bow.values():
                nn += v * v
            nn = math.sqrt(nn)
        if nn > 0:
            bow = {k: v / nn for k, v in bow.items()}
    return bow, fv
