"""Which phase of the front-end call absorbs a slow step?  200 default-shape calls, host phase timers of every call
(dani_frontend_timings: YOLO wait, SuperPoint post, LightGlue, total), printed for the calls slower than median + 0.3 ms."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from dani_slam_amd import frontend, synth
B = 8
eng = frontend.Engine(max_batch=B, weights=synth.bench_weights())
eng.yolo_set_classes(synth.COCO_NAMES, synth.DYNAMIC_NAMES)
g, b = synth.bench_frames(range(B))
dg = torch.from_numpy(g).cuda(); db = torch.from_numpy(b).cuda()
import gc; gc.collect(); gc.disable()
rows = []
for i in range(460):
    t = time.perf_counter()
    eng.frontend_batch(None, None, device_ptrs=(dg.data_ptr(), db.data_ptr()), shape=(B, 480, 640), nfeatures=1024, copy=False)
    dt = (time.perf_counter() - t) * 1e3
    rows.append([dt] + list(eng.frontend_timings()))
rows = np.array(rows[60:])
med = np.median(rows, axis=0)
print("median: call %.2f  phases %s" % (med[0], np.round(med[1:], 2)))
for i, r in enumerate(rows):
    if r[0] > med[0] + 0.3:
        print("step %3d: call %.2f (+%.2f)  phase deltas %s" % (i, r[0], r[0] - med[0], np.round(r[1:] - med[1:], 2)))
