import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from dani_slam_amd import frontend, synth
B=8
eng = frontend.Engine(max_batch=B, weights=synth.bench_weights())
eng.yolo_set_classes(synth.COCO_NAMES, synth.DYNAMIC_NAMES)
g, b = synth.bench_frames(range(B))
dg = torch.from_numpy(g).cuda(); db = torch.from_numpy(b).cuda()
inner = []
orig = eng.lib.dani_frontend_batch
class W:
    def __call__(self, *a):
        t = time.perf_counter(); r = orig(*a); inner.append(time.perf_counter() - t); return r
eng.lib.dani_frontend_batch = W()
outer = []
import gc; gc.disable()
for i in range(260):
    t = time.perf_counter()
    o = eng.frontend_batch(None, None, device_ptrs=(dg.data_ptr(), db.data_ptr()), shape=(B, 480, 640), nfeatures=1024, copy=False)
    outer.append(time.perf_counter() - t)
inner = np.array(inner[60:]) * 1e3; outer = np.array(outer[60:]) * 1e3
print("C call median %.3f ms, python call median %.3f ms, python overhead median %.3f ms" % (np.median(inner), np.median(outer), np.median(outer - inner)))
