"""GPU parity for SURVEY 8(f) rank 1: LGmatcher::DescriptorDistance + the best/second-best scan of SearchByProjection."""
import numpy as np
import pytest

from dani_slam_amd import synth
from oracle import match_oracle as mo

pytestmark = pytest.mark.gpu


def test_best2_scan_matches_oracle(sp_weights):
    from dani_slam_amd import frontend
    e = frontend.Engine(device=0, weights=sp_weights)
    rng = np.random.RandomState(11)
    k0, d0, k1, d1, gt = synth.matching_pair(3, n0=600, n1=900)
    lvl = rng.randint(0, 4, len(d1)).astype(np.int32)
    cands = []
    for i in range(len(d0)):
        n = int(rng.choice([0, 1, 2, 7, 30, 64]))
        c = rng.randint(0, len(d1), n)
        if gt[i] >= 0 and n:
            c[rng.randint(0, n)] = gt[i]                      # the true match somewhere in the list
        if n >= 2 and i % 7 == 0:
            c[-1] = c[0]                                        # duplicate candidate -> exact distance tie (strict <)
        cands.append(c.tolist())
    bd, bd2, bi, bl, bl2 = e.desc_best2(d0, d1, lvl, cands)
    ref = mo.best2_scan(d0, d1, lvl, cands)
    for i, r in enumerate(ref):
        assert (bd[i], bd2[i], bi[i], bl[i], bl2[i]) == (r[0], r[1], r[2], r[3], r[4]), i
    # empty query set / no level array
    bd, bd2, bi, bl, bl2 = e.desc_best2(d0[:3], d1, None, [[], [5], [5, 6]])
    assert bi.tolist() == [-1, 5, mo.best2_scan(d0[:3], d1, None, [[], [5], [5, 6]])[2][2]] and bd[0] == 256.0
    with pytest.raises(frontend.DaniError):
        e.desc_best2(d0[:1], d1, None, [[len(d1)]])            # out-of-range candidate index is refused
    e.close()


def test_search_by_projection_replay_over_gpu_distances(sp_weights):
    """SearchByProjection is sequential across map points (accepted matches claim keypoints, src/LGmatcher.cpp:92-94,
    124): the engine returns every candidate's distance (dani_desc_scan) and the short selection loop is replayed on the
    host -- identical matches to the literal restatement that computes each distance itself."""
    from dani_slam_amd import frontend
    e = frontend.Engine(device=0, weights=sp_weights)
    rng = np.random.RandomState(5)
    k0, d0, k1, d1, gt = synth.matching_pair(8, n0=700, n1=800)
    lvl = rng.randint(0, 3, len(d1)).astype(np.int32)
    cands = []
    for i in range(len(d0)):
        n = int(rng.choice([0, 3, 12, 40]))
        c = rng.randint(0, len(d1), n)
        if n:
            c[rng.randint(0, n)] = gt[i] if gt[i] >= 0 else c[0]
            if i % 5 == 0:
                c[rng.randint(0, n)] = gt[(i + 1) % len(d0)] if gt[(i + 1) % len(d0)] >= 0 else c[0]   # contested keypoints
        cands.append(c.tolist())
    # the synthetic descriptors are unit-norm rows: scale the distances into the reference's TH_HIGH range
    q = d0 * np.float32(0.45); t = d1 * np.float32(0.45)
    best, dist = e.desc_scan(q, t, lvl, cands)
    for qi in range(0, len(q), 37):
        for k, idx in enumerate(cands[qi]):
            assert dist[qi][k] == mo.descriptor_distance(q[qi], t[idx])
    ref = mo.search_by_projection(q, t, lvl, cands, 0.8)
    got = mo.search_by_projection(q, t, lvl, cands, 0.8, dist=dist)
    assert got == ref and len(ref) > 300
    claimed = {bi for _, bi in ref}
    assert len(claimed) == len(ref)                             # every keypoint is claimed at most once
    # without the claim rule some keypoints would be taken twice: the replay is not optional
    bd, bd2, bi, bl, bl2 = best
    free = [int(b) for b, d_ in zip(bi, bd) if b >= 0 and d_ <= mo.TH_HIGH]
    assert len(free) > len(set(free))
    e.close()
