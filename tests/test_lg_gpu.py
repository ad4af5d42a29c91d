"""GPU parity tests for LightGlue: HIP engine (C ABI) vs oracle/lg_oracle.py (numpy f32 restatement of the public
model -- the reference's network is an absent TorchScript blob, parity unpinned, see the oracle header).
Bar (BASELINE.json north_star): match scores within 1e-3.  Accumulation orders differ (flash attention, MFMA
chains vs BLAS), so this path is tolerance-checked, not bit-exact."""
import numpy as np
import pytest

from dani_slam_amd import synth
from oracle import lg_oracle as lg

pytestmark = pytest.mark.gpu
SIZE = (640.0, 480.0)


@pytest.fixture(scope="module")
def lg_weights():
    return synth.lightglue_weights(0)


@pytest.fixture(scope="module")
def engine(sp_weights, lg_weights):
    from dani_slam_amd import frontend
    w = dict(sp_weights); w.update(lg_weights)
    e = frontend.Engine(device=0, max_batch=4, weights=w)
    yield e
    e.close()


def _check_pair(engine, w, k0, d0, k1, d1, size0=SIZE, size1=SIZE, thr=0.05):
    k0n = lg.normalize_keypoints(k0, size0); k1n = lg.normalize_keypoints(k1, size1)
    mo, so, ex = lg.forward(w, k0n, k1n, d0, d1, want_intermediates=True)
    keep = so > np.float32(thr)
    mo, so = mo[keep].astype(np.int32), so[keep]
    m, s = engine.lg_match(k0, d0, k1, d1, size0, size1, thr)
    # final descriptors after the 9 layers
    g0, m0, mx0 = engine.lg_debug_fetch(0, len(k0))
    g1, _, _ = engine.lg_debug_fetch(1, len(k1))
    assert np.abs(g0 - ex["d0"]).max() < 2e-4, np.abs(g0 - ex["d0"]).max()
    assert np.abs(g1 - ex["d1"]).max() < 2e-4
    # per-row best log-score vs the oracle's score matrix
    ref_row_max = ex["scores"].max(1)
    assert np.abs(mx0 - ref_row_max).max() < 1e-3 * max(1.0, np.abs(ref_row_max).max())
    # matches: identical unless the oracle's own top-2 margin is inside float noise
    sc = ex["scores"]
    if len(mo) == len(m) and np.array_equal(mo, m):
        np.testing.assert_allclose(s, so, rtol=0, atol=1e-3)      # the stated bar
    else:
        a = {tuple(x) for x in mo.tolist()}; b = {tuple(x) for x in m.tolist()}
        for (i, j) in a ^ b:
            row = np.sort(sc[i])[::-1]; col = np.sort(sc[:, j])[::-1]
            near_tie = (row[0] - row[1] < 1e-3) or (col[0] - col[1] < 1e-3)
            near_thr = abs(np.exp(sc[i].max()) - thr) < 1e-3 or abs(np.exp(sc[i].max()) - 0.1) < 1e-3
            assert near_tie or near_thr, f"match ({i},{j}) differs without a near-tie"
        common = sorted(a & b)
        so_d = {tuple(x): v for x, v in zip(mo.tolist(), so)}; s_d = {tuple(x): v for x, v in zip(m.tolist(), s)}
        assert all(abs(so_d[c] - s_d[c]) < 1e-3 for c in common)
    return m, s, mo, so


@pytest.mark.parametrize("n0,n1", [(1024, 1024), (128, 128), (7, 7), (1025, 1000), (33, 700), (1025, 1025), (40, 1030), (64, 70), (1027, 1028), (35, 36), (2049, 1999)])
def test_pair_matches_oracle(engine, lg_weights, n0, n1):
    k0, d0, k1, d1, gt = synth.matching_pair(n0 + n1, n0=n0, n1=n1)
    m, s, mo, so = _check_pair(engine, lg_weights, k0, d0, k1, d1)
    if 1000 <= min(n0, n1) <= 1100:                               # (denser sets crowd the synthetic head)
        assert len(m) > 0.5 * (gt >= 0).sum()                     # the synthetic head finds most true matches
        assert (gt[m[:, 0]] == m[:, 1]).mean() > 0.99
    if min(n0, n1) > 1100:
        assert len(m) > 100 and (gt[m[:, 0]] == m[:, 1]).mean() > 0.95
    assert np.all(np.diff(m[:, 0]) > 0) if len(m) > 1 else True   # increasing queryIdx order


@pytest.mark.parametrize("mode", ["LG_GEMM_BF16X3", "LG_GEMM_BF16X3_ALL"])
@pytest.mark.parametrize("n0,n1", [(1024, 1024), (1025, 1025), (7, 7), (40, 1030), (2049, 1999), (35, 36), (64, 70)])
def test_bf16x3_modes_match_oracle_and_f32_mode(engine, lg_weights, n0, n1, mode):
    """dani_lg_set_gemm(DANI_LG_GEMM_BF16X3 / _ALL): the linear layers (and the attention contractions) on the bf16 matrix
    cores with three-term operand splits (csrc/lg_x3.hip, attn_x3_kernel).  Same bar against the oracle as the default mode (_check_pair: final descriptors 2e-4, scores
    1e-3), and within 1e-4 of the f32-MFMA mode on every score -- both are f32-accurate, only the order of the
    additions differs."""
    k0, d0, k1, d1, _ = synth.matching_pair(n0 + n1, n0=n0, n1=n1)
    m32, s32 = engine.lg_match(k0, d0, k1, d1, SIZE, SIZE, 0.05)
    g32, _, mx32 = engine.lg_debug_fetch(0, n0)
    engine.lg_set_gemm(getattr(engine, mode))
    try:
        m3, s3, _, _ = _check_pair(engine, lg_weights, k0, d0, k1, d1)
        g3, _, mx3 = engine.lg_debug_fetch(0, n0)
    finally:
        engine.lg_set_gemm(engine.LG_GEMM_F32)
    assert np.abs(g3 - g32).max() < 1e-4, np.abs(g3 - g32).max()
    assert np.abs(mx3 - mx32).max() < 1e-4 * max(1.0, np.abs(mx32).max())
    a = {tuple(x): v for x, v in zip(m32.tolist(), s32)}; b = {tuple(x): v for x, v in zip(m3.tolist(), s3)}
    common = set(a) & set(b)
    assert len(common) >= max(len(a), len(b)) - 2          # a near-tie may flip, nothing more
    assert all(abs(a[c] - b[c]) < 1e-4 for c in common)


def test_unknown_gemm_mode_is_rejected(engine):
    with pytest.raises(Exception):
        engine.lg_set_gemm(7)


def test_size_from_keypoints_and_threshold(engine, lg_weights):
    k0, d0, k1, d1, _ = synth.matching_pair(5, n0=300, n1=280)
    _check_pair(engine, lg_weights, k0, d0, k1, d1, size0=None, size1=None, thr=0.3)   # size.numel()==0 branch


def test_empty_sets(engine):
    k0, d0, k1, d1, _ = synth.matching_pair(6, n0=50, n1=50)
    m, s = engine.lg_match(k0[:0], d0[:0], k1, d1, SIZE, SIZE)
    assert m.shape == (0, 2) and s.shape == (0,)
    m, s = engine.lg_match(k0, d0, k1[:0], d1[:0], SIZE, SIZE)
    assert m.shape == (0, 2)


def test_batch_equals_single(engine, lg_weights):
    pairs = [synth.matching_pair(40 + i, n0=n0, n1=n1) for i, (n0, n1) in enumerate([(512, 500), (1024, 1024), (96, 130)])]
    cap = 1024
    K0 = np.zeros((3, cap, 2), np.float32); D0 = np.zeros((3, cap, 256), np.float32)
    K1 = np.zeros((3, cap, 2), np.float32); D1 = np.zeros((3, cap, 256), np.float32)
    n0 = []; n1 = []
    for p, (k0, d0, k1, d1, _) in enumerate(pairs):
        K0[p, :len(k0)] = k0; D0[p, :len(k0)] = d0; K1[p, :len(k1)] = k1; D1[p, :len(k1)] = d1
        n0.append(len(k0)); n1.append(len(k1))
    res = engine.lg_match_batch(K0, D0, n0, K1, D1, n1, SIZE, SIZE, 0.05)
    for p, (k0, d0, k1, d1, _) in enumerate(pairs):
        m, s = engine.lg_match(k0, d0, k1, d1, SIZE, SIZE, 0.05)
        assert np.array_equal(res[p][0], m)
        # the batch of three takes attn_ks_kernel, a single pair attn_k4_kernel: two key splits, two summation orders
        np.testing.assert_allclose(res[p][1], s, rtol=0, atol=1e-4)


def test_large_batch_kernel_with_ragged_sets(lg_weights, sp_weights):
    """32 keypoint sets of ~1024 keypoints take the 128-query-tile attention kernel (512 workgroups and more: the
    register-streaming attn_rs_kernel<8> since round 3; attn_kernel<4> with dani_lg_set_kernel_option("attn", 2)) instead of
    the key-split ones: its tail-key / tail-query paths (sets of 1025..1028 keypoints) must give what the single-pair path gives."""
    from dani_slam_amd import frontend
    w = dict(sp_weights); w.update(lg_weights)
    e = frontend.Engine(device=0, weights=w, max_batch=16)
    sizes = [(1025, 1027), (1024, 1028), (1026, 1025), (1000, 1025)] * 4
    pairs = [synth.matching_pair(70 + i, n0=a, n1=b) for i, (a, b) in enumerate(sizes)]
    cap = 1028
    K0 = np.zeros((16, cap, 2), np.float32); D0 = np.zeros((16, cap, 256), np.float32)
    K1 = np.zeros((16, cap, 2), np.float32); D1 = np.zeros((16, cap, 256), np.float32)
    for p, (k0, d0, k1, d1, _) in enumerate(pairs):
        K0[p, :len(k0)] = k0; D0[p, :len(k0)] = d0; K1[p, :len(k1)] = k1; D1[p, :len(k1)] = d1
    res = e.lg_match_batch(K0, D0, [a for a, _ in sizes], K1, D1, [b for _, b in sizes], SIZE, SIZE, 0.05)
    for p in (0, 1, 2, 3, 9, 15):
        k0, d0, k1, d1, _ = pairs[p]
        m, s = e.lg_match(k0, d0, k1, d1, SIZE, SIZE, 0.05)
        assert np.array_equal(res[p][0], m), p
        np.testing.assert_allclose(res[p][1], s, rtol=0, atol=1e-4)      # two kernels, two summation orders
    # the bf16x3 mode takes the 128-query kernel of ITS pair here (attn_x3w_kernel) and the key-split one for single pairs
    e.lg_set_gemm(e.LG_GEMM_BF16X3_ALL)
    res3 = e.lg_match_batch(K0, D0, [a for a, _ in sizes], K1, D1, [b for _, b in sizes], SIZE, SIZE, 0.05)
    for p in range(16):
        a = {tuple(x): v for x, v in zip(res[p][0].tolist(), res[p][1])}; b = {tuple(x): v for x, v in zip(res3[p][0].tolist(), res3[p][1])}
        common = set(a) & set(b)
        assert len(common) >= max(len(a), len(b)) - 2, p               # a near-tie may flip, nothing more
        assert all(abs(a[c] - b[c]) < 1e-4 for c in common), p
    for p in (0, 3, 15):
        k0, d0, k1, d1, _ = pairs[p]
        m, s = e.lg_match(k0, d0, k1, d1, SIZE, SIZE, 0.05)
        assert np.array_equal(res3[p][0], m), p
        np.testing.assert_allclose(res3[p][1], s, rtol=0, atol=1e-4)
    # ... and both 128-query kernels DIRECTLY against the oracle (not only against the other HIP kernel): four pairs of
    # the batched results, every ragged size class, f32 (attn_kernel<4>) and bf16x3 (attn_x3w_kernel)
    for p in (0, 1, 2, 3):
        k0, d0, k1, d1, _ = pairs[p]
        mo, so = lg.match_descriptors(lg_weights, k0, d0, k1, d1, SIZE, SIZE, 0.05)
        want = {tuple(x): float(v) for x, v in zip(mo.tolist(), so)}
        assert len(want) > 300
        for tag, r in (("f32", res), ("bf16x3_all", res3)):
            got = {tuple(x): float(v) for x, v in zip(r[p][0].tolist(), r[p][1])}
            assert len(set(got) ^ set(want)) <= max(2, len(want) // 100), (tag, p, len(got), len(want))
            assert max(abs(got[k] - want[k]) for k in set(got) & set(want)) < 1e-3, (tag, p)      # the stated bar
    e.close()


def test_reference_style_wrapper(engine, lg_weights):
    from dani_slam_amd import frontend
    k0, d0, k1, d1, _ = synth.matching_pair(9, n0=200, n1=210)
    matcher = frontend.LightGlue(engine=engine)
    kp0 = [frontend.KeyPoint(float(x), float(y), 31.0, -1.0, 0.5, 0) for x, y in k0]
    kp1 = [frontend.KeyPoint(float(x), float(y), 31.0, -1.0, 0.5, 0) for x, y in k1]
    dm = matcher.matchDescriptors(kp0, d0, kp1, d1, SIZE, SIZE)
    mo, so = lg.match_descriptors(lg_weights, k0, d0, k1, d1, SIZE, SIZE, 0.05)
    assert [(d.queryIdx, d.trainIdx) for d in dm] == [tuple(x) for x in mo.tolist()]
    assert all(d.distance > 0.05 for d in dm)


def test_missing_lg_weights_is_loud(sp_weights):
    from dani_slam_amd import frontend
    e = frontend.Engine(device=0, weights=sp_weights)            # SuperPoint only
    k0, d0, k1, d1, _ = synth.matching_pair(1, n0=40, n1=40)
    with pytest.raises(frontend.DaniError) as ei:
        e.lg_match(k0, d0, k1, d1, SIZE, SIZE)
    assert ei.value.code == frontend.DANI_ERR_WEIGHTS
    e.close()


def test_match_groups_many_small_ragged_sets(lg_weights, sp_weights):
    """dani_lg_match_groups: the per-vocabulary-node call shape of LGmatcher::SearchForTriangulation
    (src/LGmatcher.cpp:905-983) -- a few hundred independent problems of 3-40 keypoints -- in ONE forward pass, every
    group against the oracle run on that group alone; plus empty-sided groups and the one-group SearchByBoW shape."""
    from dani_slam_amd import frontend
    w = dict(sp_weights); w.update(lg_weights)
    e = frontend.Engine(device=0, weights=w, max_batch=2)
    k0, d0, k1, d1, gt = synth.matching_pair(321, n0=4200, n1=4100, overlap=0.7)
    rng = np.random.RandomState(5)
    G = 240
    node0 = rng.randint(0, G, len(k0)); node1 = rng.randint(0, G, len(k1))
    has = gt >= 0
    same = has & (rng.rand(len(k0)) < 0.9)
    node1[gt[same]] = node0[same]                       # corresponding points mostly share their vocabulary node
    groups0 = [np.nonzero(node0 == g)[0][:40] for g in range(G)]
    groups1 = [np.nonzero(node1 == g)[0][:40] for g in range(G)]
    for g in range(0, G, 7):                            # thin some groups down to 3..6 keypoints, shuffle the row order
        groups0[g] = rng.permutation(groups0[g])[:3 + g % 4]; groups1[g] = rng.permutation(groups1[g])[:3 + (g // 7) % 4]
    groups0[11] = groups0[11][:0]; groups1[12] = groups1[12][:0]      # empty sides: no call in the reference (:970)
    sizes = [len(a) for a in groups0] + [len(b) for b in groups1]
    assert min(s for s in sizes if s) == 3 and max(sizes) <= 40 and G >= 200
    res = e.lg_match_groups(k0, d0, k1, d1, groups0, groups1, SIZE, SIZE, 0.05)
    assert len(res) == G and len(res[11][0]) == 0 and len(res[12][0]) == 0
    total = wrong = 0
    for g in range(G):
        a, b = groups0[g], groups1[g]
        if len(a) == 0 or len(b) == 0:
            continue
        mo, so = lg.match_descriptors(lg_weights, k0[a], d0[a], k1[b], d1[b], SIZE, SIZE, 0.05)
        want = {tuple(x): float(v) for x, v in zip(mo.tolist(), so)}
        got = {tuple(x): float(v) for x, v in zip(res[g][0].tolist(), res[g][1])}
        total += len(want); wrong += len(set(want) ^ set(got))
        assert all(abs(got[c] - want[c]) < 1e-3 for c in set(got) & set(want)), g
        assert all(0 <= i < len(a) and 0 <= j < len(b) for i, j in got), g                  # indices are local to the group
    assert total > 500 and wrong <= max(2, total // 100), (total, wrong)                    # near-tie flips only
    # the groups agree with single calls through the plain entry point (other batch geometry; a single small pair takes the
    # 16-token LayerNorm kernel and the four-way key split: other reduction orders, scores within 1e-4)
    for g in (0, 7, 100):
        m, s = e.lg_match(k0[groups0[g]], d0[groups0[g]], k1[groups1[g]], d1[groups1[g]], SIZE, SIZE, 0.05)
        assert np.array_equal(m, res[g][0]) and np.allclose(s, res[g][1], rtol=0, atol=1e-4)
    # SearchByBoW's shape: ONE group holding everything that was gathered (src/LGmatcher.cpp:257-314)
    rows0 = np.nonzero(node0 < 60)[0]; rows1 = np.nonzero(node1 < 60)[0]
    one = e.lg_match_groups(k0, d0, k1, d1, [rows0], [rows1], SIZE, SIZE, 0.05)[0]
    m, s = e.lg_match(k0[rows0], d0[rows0], k1[rows1], d1[rows1], SIZE, SIZE, 0.05)
    assert len(m) > 20 and np.array_equal(one[0], m) and np.allclose(one[1], s, rtol=0, atol=1e-6)
    with pytest.raises(frontend.DaniError):
        e.lg_match_groups(k0, d0, k1, d1, [np.array([len(k0)])], [np.array([0])], SIZE, SIZE)   # row index out of range
    e.close()


def test_register_streaming_kernels_match_the_lds_tiled_ones(lg_weights, sp_weights):
    """Round 3's kernels against the ones they replace, in one process (dani_lg_set_kernel_option): the register-streaming
    linear layers (lin_rs.h, all three tile shapes, with the K-split tail workgroups of a 16 x ~1025-keypoint batch) and the
    register-streaming attention (attn_rs.h, both ring depths).  Same function, other summation orders."""
    from dani_slam_amd import frontend
    w = dict(sp_weights); w.update(lg_weights)
    e = frontend.Engine(device=0, weights=w, max_batch=8)
    sizes = [(1025, 1025), (1026, 1024), (1024, 1025), (1025, 1026), (1000, 1025), (1025, 1001), (1024, 1024), (1027, 1025)]
    # second batch: 7 pairs of 1030-keypoint sets (33 blocks of 32 per set, no tail region): 14 784 tokens = 115 tiles of 128
    # + 64 (two K-split tail blocks per out block), and key sets whose last 128-key super-tile holds 6 keys (the masked tile body)
    sizes_b = [(1030, 1030)] * 6 + [(1030, 1000)]

    # third batch, forced onto the register-streaming kernels although the launcher would not choose them at this size: unequal
    # sets (1025 + 1060 + 700 + 1280 keypoints -> 1024 + 1088 + 704 + 1280 + 32 tail tokens = 4128 = 32 tiles of 128 + 32; with
    # the first pair alone 2144 = 16 tiles + 96), key sets of 5.5 / 8.5 / 10 super-tiles, one tail query and one tail key
    sizes_c = [(1025, 1060), (700, 1280)]

    def batch(sz, seed):
        prs = [synth.matching_pair(seed + i, n0=a, n1=b) for i, (a, b) in enumerate(sz)]
        cap = max(1032, max(max(a, b) for a, b in sz) + 2)
        K0 = np.zeros((len(sz), cap, 2), np.float32); D0 = np.zeros((len(sz), cap, 256), np.float32)
        K1 = np.zeros((len(sz), cap, 2), np.float32); D1 = np.zeros((len(sz), cap, 256), np.float32)
        for p, (k0, d0, k1, d1, _) in enumerate(prs):
            K0[p, :len(k0)] = k0; D0[p, :len(k0)] = d0; K1[p, :len(k1)] = k1; D1[p, :len(k1)] = d1
        return prs, (K0, D0, [a for a, _ in sz], K1, D1, [b for _, b in sz])
    pairs, args_a = batch(sizes, 170)
    _, args_b = batch(sizes_b, 190)
    _, args_c = batch(sizes_c, 230)
    args_c1 = tuple(a[:1] for a in args_c)                             # the first pair alone

    def run(args=args_a, **opts):
        for k, v in {**dict(lin_rs=-1, attn=0), **opts}.items():
            e.lg_set_kernel_option(k, v)
        return e.lg_match_batch(*args, SIZE, SIZE, 0.05)
    try:
        base = run(lin_rs=0, attn=2)                                   # round 2's kernels: linear_fm_kernel, attn_kernel<4>
        assert min(len(m) for m, _ in base) > 300
        variants = dict(default=run(), lin_tile_1=run(lin_rs=1), lin_tile_2=run(lin_rs=2), attn_ring_4=run(attn=7),
                        only_linear=run(attn=2), only_attention=run(lin_rs=0),
                        lin_all_k_split_blocks=run(lin_rs=4))          # every token through linear_rs_tail's 32 x 32 blocks
        base_b = run(args_b, lin_rs=0, attn=2)
        assert min(len(m) for m, _ in base_b) > 300
        variants_b = dict(default=run(args_b), lin_tile_1=run(args_b, lin_rs=1), attn_ring_4=run(args_b, attn=7), attn_ring_16=run(args_b, attn=8))
        base_c = run(args_c, lin_rs=0, attn=2); base_c1 = run(args_c1, lin_rs=0, attn=2)
        assert min(len(m) for m, _ in base_c) > 200
        variants_c = dict(rs_1=run(args_c, lin_rs=1, attn=6), rs_2=run(args_c, lin_rs=2, attn=8), rs_4=run(args_c, lin_rs=4, attn=7))
        variants_c1 = dict(rs_1=run(args_c1, lin_rs=1, attn=6), rs_2=run(args_c1, lin_rs=2, attn=6))
        refs = {"_b": base_b, "_c": base_c, "_c1": base_c1}
        for tag, res in (list(variants.items()) + [(t + "_b", r) for t, r in variants_b.items()] + [(t + "_c", r) for t, r in variants_c.items()]
                         + [(t + "_c1", r) for t, r in variants_c1.items()]):
            ref = next((r for sfx, r in refs.items() if tag.endswith(sfx)), base)
            for p in range(len(ref)):
                a = {tuple(x): v for x, v in zip(ref[p][0].tolist(), ref[p][1])}
                b = {tuple(x): v for x, v in zip(res[p][0].tolist(), res[p][1])}
                common = set(a) & set(b)
                assert len(common) >= max(len(a), len(b)) - 2, (tag, p, len(a), len(b))      # a near-tie may flip, nothing more
                assert max(abs(a[c] - b[c]) for c in common) < 1e-4, (tag, p)
        # ... and the default form against the oracle itself on two ragged pairs
        for p in (0, 3):
            k0, d0, k1, d1, _ = pairs[p]
            mo, so = lg.match_descriptors(lg_weights, k0, d0, k1, d1, SIZE, SIZE, 0.05)
            want = {tuple(x): float(v) for x, v in zip(mo.tolist(), so)}
            got = {tuple(x): float(v) for x, v in zip(variants["default"][p][0].tolist(), variants["default"][p][1])}
            assert len(set(got) ^ set(want)) <= max(2, len(want) // 100), (p, len(got), len(want))
            assert max(abs(got[k] - want[k]) for k in set(got) & set(want)) < 1e-3, p
    finally:
        for k, v in dict(lin_rs=-1, attn=0).items():
            e.lg_set_kernel_option(k, v)
        e.close()



def test_non_finite_inputs_do_not_poison_later_calls(lg_weights, sp_weights):
    """A call whose descriptors hold NaN leaves NaN in its workspace columns.  A later, SMALLER batch ends in front of those
    columns, and the register-streaming attention loads whole 128-key super-tiles: the ragged last tile of the last set reads
    up to 127 columns past the set (masked scores, V operands multiplied by P = 0 -- 0 * NaN unless they are zeroed in the
    register, attn_rs.h).  The smaller batch must give bit for bit what it gave before the NaN call."""
    from dani_slam_amd import frontend
    w = dict(sp_weights); w.update(lg_weights)
    e = frontend.Engine(device=0, weights=w, max_batch=8)

    def batch(sz, seed):
        prs = [synth.matching_pair(seed + i, n0=a, n1=b) for i, (a, b) in enumerate(sz)]
        cap = max(max(a, b) for a, b in sz) + 2
        K0 = np.zeros((len(sz), cap, 2), np.float32); D0 = np.zeros((len(sz), cap, 256), np.float32)
        K1 = np.zeros((len(sz), cap, 2), np.float32); D1 = np.zeros((len(sz), cap, 256), np.float32)
        for p, (k0, d0, k1, d1, _) in enumerate(prs):
            K0[p, :len(k0)] = k0; D0[p, :len(k0)] = d0; K1[p, :len(k1)] = k1; D1[p, :len(k1)] = d1
        return [K0, D0, [a for a, _ in sz], K1, D1, [b for _, b in sz]]
    small = batch([(1030, 1030)] * 6 + [(1030, 1000)], 190)         # key sets whose last super-tile holds 6 / 104 valid keys
    big = batch([(1200, 1200)] * 8, 40)
    before = e.lg_match_batch(*small, SIZE, SIZE, 0.05)
    assert min(len(m) for m, _ in before) > 300
    big[1][:] = np.nan; big[4][:] = np.nan                           # every descriptor of the big batch
    poisoned = e.lg_match_batch(*big, SIZE, SIZE, 0.05)
    assert all(len(m) == 0 for m, _ in poisoned)                     # NaN scores never pass the threshold
    after = e.lg_match_batch(*small, SIZE, SIZE, 0.05)
    for (m0, s0), (m1, s1) in zip(before, after):
        assert np.array_equal(m0, m1) and np.array_equal(s0, s1)
    e.close()


def test_ffn_chain_matches_the_four_launch_path(lg_weights, sp_weights):
    """Round 4: ffn.0 -> LayerNorm + GELU -> ffn.3 (+ residual) -> the next block's projection as ONE register-resident launch per
    block (csrc/ffn_chain.h, 16 tokens per wavefront through v_mfma_f32_16x16x4_f32, accumulators reused as B operands) against
    the four launches it replaces, in one process (dani_lg_set_kernel_option "ffn_chain").  Same function, other summation
    orders: <= 2 near-tie match flips per pair, scores within 1e-4; and against the oracle itself on two pairs."""
    from dani_slam_amd import frontend
    w = dict(sp_weights); w.update(lg_weights)
    e = frontend.Engine(device=0, weights=w, max_batch=8)

    def batch(sz, seed):
        prs = [synth.matching_pair(seed + i, n0=a, n1=b) for i, (a, b) in enumerate(sz)]
        cap = max(1032, max(max(a, b) for a, b in sz) + 2)
        K0 = np.zeros((len(sz), cap, 2), np.float32); D0 = np.zeros((len(sz), cap, 256), np.float32)
        K1 = np.zeros((len(sz), cap, 2), np.float32); D1 = np.zeros((len(sz), cap, 256), np.float32)
        for p, (k0, d0, k1, d1, _) in enumerate(prs):
            K0[p, :len(k0)] = k0; D0[p, :len(k0)] = d0; K1[p, :len(k1)] = k1; D1[p, :len(k1)] = d1
        return prs, (K0, D0, [a for a, _ in sz], K1, D1, [b for _, b in sz])
    # (a) the bench's shape: 16 sets of 1024-1027 keypoints = 16 384 main tokens (1024 blocks: one round) + a tail region that
    #     keeps the four small launches; (b) no tail region at all; (c) forced onto a small ragged batch (any block count works)
    shapes = {"bench": [(1025, 1025), (1026, 1024), (1024, 1025), (1025, 1026), (1027, 1025), (1025, 1024), (1024, 1024), (1026, 1025)],
              "no_tail": [(1024, 1024)] * 8, "small": [(700, 1280), (1025, 333)]}
    try:
        for name, sz in shapes.items():
            pairs, args = batch(sz, 300)
            res = {}
            for mode in (0, 1, -1):
                e.lg_set_kernel_option("ffn_chain", mode)
                res[mode] = e.lg_match_batch(*args, SIZE, SIZE, 0.05)
            assert min(len(m) for m, _ in res[0]) > 100, name
            for mode in (1, -1):
                for p in range(len(sz)):
                    a = {tuple(x): v for x, v in zip(res[0][p][0].tolist(), res[0][p][1])}
                    b = {tuple(x): v for x, v in zip(res[mode][p][0].tolist(), res[mode][p][1])}
                    common = set(a) & set(b)
                    assert len(common) >= max(len(a), len(b)) - 2, (name, mode, p, len(a), len(b))
                    assert max(abs(a[c] - b[c]) for c in common) < 1e-4, (name, mode, p)
            if name == "bench":
                for p in (0, 4):
                    k0, d0, k1, d1, _ = pairs[p]
                    mo, so = lg.match_descriptors(lg_weights, k0, d0, k1, d1, SIZE, SIZE, 0.05)
                    want = {tuple(x): float(v) for x, v in zip(mo.tolist(), so)}
                    got = {tuple(x): float(v) for x, v in zip(res[1][p][0].tolist(), res[1][p][1])}
                    assert len(set(got) ^ set(want)) <= max(2, len(want) // 100), (p, len(got), len(want))
                    assert max(abs(got[k] - want[k]) for k in set(got) & set(want)) < 1e-3, p
    finally:
        e.lg_set_kernel_option("ffn_chain", -1)
        e.close()
