"""Run the YOLO leg alone a few times (for `rocprofv3 --kernel-trace`): per-dispatch durations in launch order."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from dani_slam_amd import frontend, synth

B = 8
eng = frontend.Engine(max_batch=B)
eng.set_weights({**synth.superpoint_weights(0), **synth.yolo_weights(0)})
imgs = np.stack([synth.frame_bgr(i) for i in range(B)])
dev = torch.from_numpy(imgs).cuda()
for it in range(4):
    eng.yolo_detect_batch(None, device_ptr=dev.data_ptr(), shape=(B, 480, 640, 640 * 3))
torch.cuda.synchronize()
print("done")
