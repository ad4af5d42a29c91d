"""Where does a step's wall time go outside the engine?  Python wall around frontend_batch against the C call's own clock
(dani_frontend_timings[3]) over 300 steps of the default bench workload, plus the cost of the wrapper's pieces in isolation."""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, ".")
from dani_slam_amd import frontend, synth
B = 8
frames, bgr = synth.bench_frames(range(B))
dg = torch.from_numpy(frames).cuda(); db = torch.from_numpy(bgr).cuda()
eng = frontend.Engine(max_batch=B, weights=synth.bench_weights())
eng.yolo_set_classes(synth.COCO_NAMES, synth.DYNAMIC_NAMES)
kw = dict(nfeatures=1024, device_ptrs=(dg.data_ptr(), db.data_ptr()), shape=(B, 480, 640), copy=False)
for _ in range(30):
    eng.frontend_batch(None, **kw)
wall, inner, gap = [], [], []
t_prev_end = None
for _ in range(300):
    t0 = time.perf_counter()
    eng.frontend_batch(None, **kw)
    t1 = time.perf_counter()
    wall.append((t1 - t0) * 1e3)
    inner.append(float(eng.frontend_timings()[3]))
    if t_prev_end is not None:
        gap.append((t0 - t_prev_end) * 1e3)
    t_prev_end = t1
med = lambda v: float(np.median(v))
print(f"python wall per call {med(wall):.4f} ms, C call's own clock {med(inner):.4f} ms, wrapper overhead {med(wall) - med(inner):.4f} ms, loop gap between calls {med(gap):.4f} ms")
