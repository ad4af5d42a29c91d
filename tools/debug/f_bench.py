"""SURVEY 8(f) ranks 3 and 4 at tracking-like sizes, for `rocprofv3 --kernel-trace --stats` (kernel durations) ."""
import sys
import numpy as np
sys.path.insert(0, ".")
from dani_slam_amd import frontend, synth

e = frontend.Engine(max_batch=1, weights=synth.superpoint_weights(0))
rng = np.random.default_rng(0)
rows = rng.standard_normal((20000, 256)).astype(np.float32)
obs = [rng.integers(0, len(rows), rng.integers(2, 12)).tolist() for _ in range(4000)]
tree = synth.vocab_tree(3, k=10, L=4)
v = e.vocab_create(tree["L"], tree["child_offsets"], tree["child_ids"], tree["desc"], tree["weight"], tree["word_id"])
feats = rng.standard_normal((1025, 256)).astype(np.float32)
for _ in range(5):
    e.desc_mean(rows, obs)
    e.bow_transform(v, feats, 4, 0, 1)
print("mean: %d map points, %d rows gathered (%.1f MB); bow: %d features, tree %d nodes (k=10, L=4)" %
      (len(obs), sum(len(o) for o in obs), sum(len(o) for o in obs) * 1024 / 1e6, len(feats), len(tree["desc"])))
