"""Debug (library built with `make -C dani_slam_amd/csrc STAMPS=1`): in-kernel clock stamps of workgroup 0 of every conv
launch of ONE YOLO pass alone.  Prints per launch: chunks, effective core clock, and the mean time a chunk spends waiting
(MFMAs issued -> next chunk ready in LDS) and computing (chunk ready -> its MFMAs issued)."""
import ctypes, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from dani_slam_amd import frontend, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
FUSED = len(sys.argv) > 2 and sys.argv[2] == "fused"   # the whole front-end call (SuperPoint || YOLO) instead of YOLO alone
eng = frontend.Engine(max_batch=B)
eng.set_weights({**synth.superpoint_weights(0), **synth.yolo_weights(0)})
eng.yolo_set_classes(synth.COCO_NAMES, synth.DYNAMIC_NAMES)
imgs = np.stack([synth.frame_bgr(i) for i in range(B)])
dev = torch.from_numpy(imgs).cuda()
lib = frontend.load_library()
f = lib.dani_debug_conv_stamps
f.argtypes = [ctypes.c_void_p, ctypes.c_int]; f.restype = ctypes.c_int
gray = torch.from_numpy(np.stack([synth.frame(i) for i in range(B)])).cuda()


def run():
    if FUSED:
        eng.frontend_batch(None, None, device_ptrs=(gray.data_ptr(), dev.data_ptr()), shape=(B, 480, 640), match_previous=False)
    else:
        eng.yolo_detect_batch(None, device_ptr=dev.data_ptr(), shape=(B, 480, 640, 640 * 3))


for it in range(3):
    run()
torch.cuda.synchronize()
f(None, 1)
run()
torch.cuda.synchronize()
buf = np.zeros(256 * 96, np.uint64)
n = f(buf.ctypes.data, 0)
print("launches", n)
for li in range(min(n, 256)):
    r = buf[li * 96:(li + 1) * 96].astype(np.int64)
    nch = int(r[0]); w0, c0, c1, w1 = r[1], r[2], r[94], r[95]
    wall_us = (w1 - w0) / 100.0
    ghz = (c1 - c0) / max(wall_us, 1e-9) / 1e3
    m = min(nch, 45)
    ready = r[3:3 + 2 * m:2]; done = r[4:4 + 2 * m:2]
    comp = (done - ready).mean() if m else 0
    wait = (ready[1:] - done[:-1]).mean() if m > 1 else 0
    print("launch %3d %s grid %5d cfg %4d chunks %3d  wg0 %6.1f us  clock %.2f GHz  first chunk ready after %6d cyc  compute/chunk %6.0f cyc  wait/chunk %6.0f cyc  epilogue %6d cyc"
          % (li, "yolo" if r[93] >= 2000 else "sp  ", r[92], r[93], nch, wall_us, ghz, ready[0] - c0 if m else 0, comp, wait, c1 - done[m - 1] if m else 0))
