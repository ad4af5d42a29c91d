import sys
import numpy as np
sys.path.insert(0, ".")
from dani_slam_amd import frontend, synth
from oracle import yolo_oracle as yo
yw = synth.yolo_weights(0)
w = dict(synth.superpoint_weights(0)); w.update(yw)
img = synth.frame_bgr(0)
x = yo.preprocess(img)
preds, taps = yo.network(yw, x, want_taps=True)
for mb in (1, 2, 3, 8):
    e = frontend.Engine(device=0, max_batch=mb, weights=w)
    dets = e.yolo_detect_batch(img)[0]
    msg = []
    for lvl in range(3):
        raw = e.yolo_debug_raw(0, lvl)
        msg.append(float(np.abs(raw - taps[f"raw{lvl}"]).max()))
    print("max_batch", mb, "raw maxdiff", msg, "ndets", len(dets))
    e.close()
