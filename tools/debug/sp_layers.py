"""SuperPoint conv stack alone (no concurrent YOLO): per-layer device ms over 20 calls of 8 resident 640x480 frames."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from dani_slam_amd import frontend, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
eng = frontend.Engine(max_batch=B, weights=synth.superpoint_weights(0))
frames = np.stack([synth.frame(i) for i in range(B)])
dev = torch.from_numpy(frames).cuda()
eng.set_profiling(True)
acc = np.zeros(12)
for it in range(25):
    eng.sp_extract_batch(None, nfeatures=1024, device_ptr=dev.data_ptr(), shape=(B, 480, 640, 640))
    if it >= 5:
        acc += eng.layer_timings()
acc /= 20
print("layer ms:", " ".join(f"{x:.4f}" for x in acc), " total", f"{acc.sum():.4f}")
