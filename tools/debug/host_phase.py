"""Host-side time of the SuperPoint post stage inside the fused call (quadtree + assembly), per step."""
import sys
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
acc = []
for it in range(30):
    eng.frontend_batch(None, None, nfeatures=1024, device_ptrs=(g.data_ptr(), c.data_ptr()), shape=(B, 480, 640), copy=False)
    if it >= 5:
        acc.append((eng.last_timings()[2], *eng.frontend_timings()))
a = np.array(acc)
print("host quadtree+assembly ms: mean %.3f  min %.3f | stage ms (yolo, sp, lg, total): %s" % (a[:, 0].mean(), a[:, 0].min(), np.round(a[:, 1:].mean(0), 3)))
