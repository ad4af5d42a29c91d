"""time dani_desc_best2 (projection-search descriptor scans) on a tracking-sized problem.
Run under `rocprofv3 --kernel-trace --stats` to get the kernel's own duration; prints the end-to-end call time too."""
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from dani_slam_amd import frontend, synth

rng = np.random.default_rng(0)
nq, nt, avg = int(sys.argv[1]) if len(sys.argv) > 1 else 2000, 1025, 24
q = rng.standard_normal((nq, 256)).astype(np.float32); t = rng.standard_normal((nt, 256)).astype(np.float32)
lvl = np.zeros(nt, np.int32)
cands = [rng.integers(0, nt, rng.integers(0, 2 * avg + 1)).astype(np.int32) for _ in range(nq)]
total = sum(len(c) for c in cands)
eng = frontend.Engine(max_batch=1, weights=synth.superpoint_weights(0))
for _ in range(3):
    eng.desc_best2(q, t, lvl, cands)
t0 = time.perf_counter()
for _ in range(10):
    out = eng.desc_best2(q, t, lvl, cands)
dt = (time.perf_counter() - t0) / 10
print(f"nq={nq} nt={nt} candidates={total}: {dt * 1e3:.3f} ms per call end to end (host lists in, results out); "
      f"algorithmic bytes {(total + nq) * 1024 / 1e6:.1f} MB")
