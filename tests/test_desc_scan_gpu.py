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


def test_projection_search_pipeline(sp_weights):
    """Rank 1 + 2 chained as a tracking thread would: Frame's keypoint grid -> GetFeaturesInArea windows around the
    projected map points -> descriptor distances on the GPU -> sequential selection replay; against the oracle chain."""
    from dani_slam_amd import frontend
    from oracle import handoff_oracle as ho
    e = frontend.Engine(device=0, weights=sp_weights)
    rng = np.random.RandomState(9)
    k0, d0, k1, d1, gt = synth.matching_pair(12, n0=900, n1=1025)
    octave = np.zeros(len(k1), np.int32)
    inv_w, inv_h = np.float32(64) / np.float32(640), np.float32(48) / np.float32(480)
    grid_params = (0.0, 0.0, float(inv_w), float(inv_h), 64, 48)
    offs, idx = frontend.host_assign_features_to_grid(k1, *grid_params)
    # "projections" of the map points: their true keypoint position in the frame plus noise, or a random place
    proj = np.where(gt[:, None] >= 0, k1[np.maximum(gt, 0)] + rng.normal(0, 2.0, (len(k0), 2)), rng.uniform(0, 640, (len(k0), 2))).astype(np.float32)
    r = np.full(len(k0), 15.0, np.float32)
    lists = frontend.host_features_in_area(k1, octave, grid_params, offs, idx, proj[:, 0], proj[:, 1], r, np.full(len(k0), -1, np.int32), np.zeros(len(k0), np.int32))
    grid = ho.assign_features_to_grid(k1, 0.0, 0.0, inv_w, inv_h)
    ref_lists = [ho.get_features_in_area(k1, octave, grid, 0.0, 0.0, inv_w, inv_h, proj[i, 0], proj[i, 1], r[i], -1, 0) for i in range(len(k0))]
    assert [l.tolist() for l in lists] == ref_lists
    q = d0 * np.float32(0.45); t = d1 * np.float32(0.45)
    best, dist = e.desc_scan(q, t, octave, [l.tolist() for l in lists])
    got = mo.search_by_projection(q, t, octave, ref_lists, 0.8, dist=dist)
    ref = mo.search_by_projection(q, t, octave, ref_lists, 0.8)
    assert got == ref
    good = sum(1 for qi, ki in ref if gt[qi] == ki)
    assert good > 0.9 * len(ref) and len(ref) > 500
    e.close()


def test_distinctive_descriptors_match_oracle(sp_weights):
    """batched MapPoint::ComputeDistinctiveDescriptors, bit-identical to the sequential f32 host loop."""
    from dani_slam_amd import frontend
    e = frontend.Engine(device=0, weights=sp_weights)
    rng = np.random.RandomState(4)
    rows = rng.standard_normal((3000, 256)).astype(np.float32)
    obs = [rng.randint(0, len(rows), int(rng.choice([0, 1, 2, 3, 4, 5, 9, 40]))).tolist() for _ in range(500)]
    got = e.desc_mean(rows, obs)
    for q, o in enumerate(obs):
        ref = mo.distinctive_descriptor(rows, o)
        if ref is None:
            assert not got[q].any()
        else:
            assert np.array_equal(got[q], ref), q
    with pytest.raises(frontend.DaniError):
        e.desc_mean(rows, [[len(rows)]])
    e.close()
