"""Per-step wall times of the fused call right after start-up (is there a slow transient after another GPU process exits?)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from dani_slam_amd import frontend, synth
B = 8
w = dict(synth.superpoint_weights(0)); w.update(synth.lightglue_weights(0)); w.update(synth.yolo_weights(0))
eng = frontend.Engine(max_batch=B, weights=w)
eng.yolo_set_classes(synth.COCO_NAMES, synth.DYNAMIC_NAMES)
g = torch.from_numpy(np.stack([synth.frame(i) for i in range(B)])).cuda()
c = torch.from_numpy(np.stack([synth.frame_bgr(i) for i in range(B)])).cuda()
import gc, os
if os.environ.get('NOGC'):
    gc.collect(); gc.disable()
ts = []
t_start = time.perf_counter()
for it in range(400):
    t0 = time.perf_counter()
    eng.frontend_batch(None, None, nfeatures=1024, device_ptrs=(g.data_ptr(), c.data_ptr()), shape=(B, 480, 640), copy=False)
    ts.append((time.perf_counter() - t0) * 1e3)
ts = np.array(ts)
print('steps > 13.5 ms:', [(i, round(float(t), 1)) for i, t in enumerate(ts) if t > 13.5][:40])
for k in range(0, 400, 100):
    print(f"steps {k:3d}-{k+99:3d} (t={ts[:k].sum()/1e3:5.2f}s): mean {ts[k:k+100].mean():6.2f} ms  min {ts[k:k+100].min():6.2f}  max {ts[k:k+100].max():6.2f}")
