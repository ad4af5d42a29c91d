"""Host timeline of the fused front-end call: median microseconds since the call's start at every trace mark of the engine
(DANI_TRACE_HOST=1), over 200 default-shape calls, next to the device-side end of YOLO's chain and LightGlue's forward pass
(HIP events).  usage: python tools/debug/host_trace.py   (re-runs itself with the engine's stderr in a file)"""
import os, re, subprocess, sys
import numpy as np
if os.environ.get("DANI_TRACE_HOST") != "1":
    env = dict(os.environ, DANI_TRACE_HOST="1")
    with open("gpurun_out/host_trace.err", "w") as f:
        rc = subprocess.call([sys.executable, __file__], env=env, stderr=f)
    rows = {}
    for line in open("gpurun_out/host_trace.err"):
        if not line.startswith("dani-trace:"):
            continue
        for m in re.finditer(r"(\w+)=([\d.]+)", line):
            rows.setdefault(m.group(1), []).append(float(m.group(2)))
    order = sorted(rows, key=lambda k: np.median(rows[k][40:]))
    prev = 0.0
    for k in order:
        v = np.array(rows[k][40:])
        print(f"{k:32s} median {np.median(v):9.1f} us   (+{np.median(v) - prev:7.1f})   p10 {np.percentile(v, 10):9.1f}  p90 {np.percentile(v, 90):9.1f}")
        prev = float(np.median(v))
    sys.exit(rc)
import torch
sys.path.insert(0, ".")
from dani_slam_amd import frontend, synth
B = 8
eng = frontend.Engine(max_batch=B, weights=synth.bench_weights())
eng.yolo_set_classes(synth.COCO_NAMES, synth.DYNAMIC_NAMES)
g, b = synth.bench_frames(range(B))
dg = torch.from_numpy(g).cuda(); db = torch.from_numpy(b).cuda()
for i in range(240):
    eng.frontend_batch(None, None, device_ptrs=(dg.data_ptr(), db.data_ptr()), shape=(B, 480, 640), nfeatures=1024, copy=False)
