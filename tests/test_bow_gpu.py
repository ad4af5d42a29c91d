"""GPU parity for SURVEY 8(f) rank 3: DBoW3's Vocabulary::transform for float descriptors (tree descent on the GPU,
BowVector / FeatureVector bookkeeping on the host) against oracle/bow_oracle.py, on synthetic vocabularies (the
SuperPoint vocabulary is withheld upstream)."""
import numpy as np
import pytest

from dani_slam_amd import synth
from oracle import bow_oracle as bo

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k,L,levelsup,weighting,norm", [(10, 3, 4, 0, 1), (10, 3, 2, 0, 1), (6, 4, 1, 1, 0), (8, 2, 1, 2, 2), (10, 3, 3, 3, 1)])
def test_transform_matches_oracle(sp_weights, k, L, levelsup, weighting, norm):
    from dani_slam_amd import frontend
    e = frontend.Engine(device=0, weights=sp_weights)
    tree = synth.vocab_tree(7 + k + L, k=k, L=L)
    rng = np.random.default_rng(3)
    leaves = np.nonzero(np.diff(tree["child_offsets"]) == 0)[0]
    # features near leaves (repeated words, incl. stopped ones) and far from everything
    pick = rng.choice(leaves, 300)
    feats = np.concatenate([tree["desc"][pick] + rng.standard_normal((300, 256)).astype(np.float32) * 0.2,
                            rng.standard_normal((60, 256)).astype(np.float32) * 2.0]).astype(np.float32)
    feats[10] = tree["desc"][pick[10]]                                   # exactly on a node
    v = e.vocab_create(tree["L"], tree["child_offsets"], tree["child_ids"], tree["desc"], tree["weight"], tree["word_id"])
    bow, fv = e.bow_transform(v, feats, levelsup, weighting, norm)
    rbow, rfv = bo.transform(tree, feats, levelsup, weighting, norm)
    assert list(bow.keys()) == list(rbow.keys()) and fv == rfv
    assert all(bow[w] == rbow[w] for w in bow)                            # same double arithmetic, same order
    assert len(bow) > 50 and sum(len(x) for x in fv.values()) <= len(feats)
    if norm == 1:
        assert abs(sum(bow.values()) - 1.0) < 1e-12
    bow0, fv0 = e.bow_transform(v, feats[:0], levelsup, weighting, norm)
    assert bow0 == {} and fv0 == {}
    e.vocab_destroy(v)
    e.close()


def test_vocabulary_validation(sp_weights):
    from dani_slam_amd import frontend
    e = frontend.Engine(device=0, weights=sp_weights)
    tree = synth.vocab_tree(1, k=3, L=2)
    bad = tree["child_ids"].copy(); bad[0] = 0                            # a child pointing back at the root: would never terminate
    with pytest.raises(frontend.DaniError):
        e.vocab_create(tree["L"], tree["child_offsets"], bad, tree["desc"], tree["weight"], tree["word_id"])
    e.close()
