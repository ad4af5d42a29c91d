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
