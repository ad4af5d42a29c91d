"""What does YOLO cost beside LightGlue, and beside SuperPoint?  Three engines' worth of work on ONE engine, two host threads:
t(A), t(B) alone and t(A || B) for (A, B) = (LightGlue of 8 pairs, YOLO of 8 frames) and (SuperPoint of 8 frames, YOLO of 8 frames).
Decides whether enqueueing the NEXT batch's YOLO under THIS batch's LightGlue (instead of beside its own SuperPoint) would pay."""
import sys, time, threading
import numpy as np
import torch
sys.path.insert(0, ".")
from dani_slam_amd import frontend, synth

B = 8
w = synth.bench_weights()
eng = frontend.Engine(max_batch=B, weights=w, n_contexts=3)
eng.yolo_set_classes(synth.COCO_NAMES, synth.DYNAMIC_NAMES)
gray, bgr = synth.bench_frames(range(B))
dg = torch.from_numpy(gray).cuda(); db = torch.from_numpy(bgr).cuda()
out = eng.frontend_batch(None, None, device_ptrs=(dg.data_ptr(), db.data_ptr()), shape=(B, 480, 640), nfeatures=1024)
n = [len(o["kpts"]) for o in out]
cap = max(n)
K = np.zeros((B, cap, 2), np.float32); D = np.zeros((B, cap, 256), np.float32)
for i, o in enumerate(out):
    K[i, :n[i]] = o["kpts"]; D[i, :n[i]] = o["desc"]
K1 = np.roll(K, -1, 0); D1 = np.roll(D, -1, 0); n1 = n[1:] + n[:1]

def lg():
    eng.lg_match_batch(K, D, n, K1, D1, n1, (640, 480), (640, 480), 0.05)
def yolo():
    eng.yolo_detect_batch(None, device_ptr=db.data_ptr(), shape=(B, 480, 640, 640 * 3))
def sp():
    eng.sp_extract_batch(None, nfeatures=1024, device_ptr=dg.data_ptr(), shape=(B, 480, 640, 640))

def timeit(fns, reps=30):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=f) for f in fns[1:]]
        for t in th: t.start()
        fns[0]()
        for t in th: t.join()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]

for f in (lg, yolo, sp):
    for _ in range(3): f()
print("alone  ms: LightGlue %.3f  YOLO %.3f  SuperPoint %.3f" % (timeit([lg]), timeit([yolo]), timeit([sp])))
print("paired ms: LightGlue || YOLO %.3f   SuperPoint || YOLO %.3f   LightGlue || SuperPoint %.3f" % (timeit([lg, yolo]), timeit([sp, yolo]), timeit([lg, sp])))
