#!/usr/bin/env python3
"""bench.py -- front-end throughput on MI355X (BASELINE.json metric: frames/sec at 640x480).

A "step" = one pass of the hot path over one batch of synthetic frames already resident in HBM:
FRAMES_PER_GPU (8 = BASELINE config #4's 64 frames / 8 GPUs) 640x480 frames per GPU through
dani_sp_extract_batch (SuperPoint extraction, 1024 keypoints; the LightGlue and YOLO stages are added to the
step as they land -- `config.workload` always says what was timed).  Frames are batch-sharded over ranks with
no data-path collective (weak scaling).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel (FP32-MFMA conv, conv1b/2b/3b instances) priced against the gfx950
                  FP32 matrix peak, from HIP events recorded on the engine's own stream over the timed steps;
  cpu_baseline -- the CPU oracle ("port" of the reference's LibTorch-CPU path) timed on this host's cores on a
                  bounded sample of the same frames (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FRAMES_PER_GPU = 8
W, H = 640, 480
NFEATURES = 1024
# algorithmic MAC per 640x480 frame and layer: M*N*K of the implicit GEMM
LAYER_GMAC = [0.177, 11.325, 2.831, 2.831, 1.416, 2.831, 0.708, 0.708, 1.416, 0.080, 1.416, 0.315]
SP_GFLOP_PER_FRAME = 52.10
FP32_MATRIX_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
DOMINANT_LAYERS = (1, 3, 5)       # conv1b, conv2b, conv3b: conv_mfma_kernel<Cfg<3,1,32,2,8,true,true>>
DOMINANT_KERNEL = "conv_mfma_kernel<Cfg<3,1,32,2,8,true,true>> (conv1b+conv2b+conv3b)"


def exact_layer_flops(li, B):
    from dani_slam_amd.synth import SP_LAYERS
    dims = [(H, W), (H, W), (H // 2, W // 2), (H // 2, W // 2), (H // 4, W // 4), (H // 4, W // 4)] + [(H // 8, W // 8)] * 6
    _, cin, cout, k = SP_LAYERS[li]
    h, w = dims[li]
    return 2.0 * h * w * cout * cin * k * k * B


def cpu_baseline(frames):
    """Time the oracle (CPU port of the reference path) on a bounded sample: whole-frame extraction."""
    from dani_slam_amd import synth
    from oracle import oracle
    cores = len(os.sched_getaffinity(0))
    w = synth.superpoint_weights(0)
    oracle.lib()
    n = 0
    t0 = time.perf_counter()
    while True:
        oracle.sp_extract(frames[n % len(frames)], w, nfeatures=NFEATURES)
        n += 1
        dt = time.perf_counter() - t0
        if dt > 12.0 or n >= 24:
            break
    return {"value": round(n / dt, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n} synthetic 640x480 frames through oracle.sp_extract (CPU restatement of "
                      f"SPextractor::operator(), OpenMP over {cores} cores), {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import numpy as np
    import torch
    import torch.distributed as dist
    from dani_slam_amd import frontend, synth

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    B = FRAMES_PER_GPU
    # this rank's shard of the synthetic sequence: frames rank*B .. rank*B+B-1
    frames = np.stack([synth.frame(rank * B + i) for i in range(B)])
    d_frames = torch.from_numpy(frames).to(dev)          # inputs resident in HBM before the timed region
    eng = frontend.Engine(device=local_rank, max_batch=B, max_width=W, max_height=H,
                          weights=synth.superpoint_weights(0))

    def step():
        return eng.sp_extract_batch(None, nfeatures=NFEATURES, device_ptr=d_frames.data_ptr(), shape=(B, H, W, W))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    eng.set_profiling(True)
    layer_ms = np.zeros(12)
    stage_ms = np.zeros(5)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
        layer_ms += eng.layer_timings()
        stage_ms += eng.last_timings()
    barrier()
    dt = time.perf_counter() - t0
    eng.set_profiling(False)

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t.item())
    nk = [len(o["kpts"]) for o in out]

    if rank == 0:
        total_frames = B * world * args.steps
        fps = total_frames / dt_max
        layer_ms /= args.steps
        stage_ms /= args.steps
        dom_ms = float(sum(layer_ms[i] for i in DOMINANT_LAYERS))
        dom_flops = sum(exact_layer_flops(i, B) for i in DOMINANT_LAYERS)
        achieved = dom_flops / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")   # PMC-derived HBM bytes per launch (committed)
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        conv_ms = float(stage_ms[0])
        res = {
            "metric": "frames/sec SuperPoint+LightGlue+YOLO front-end, 640x480, 1/2/4/8 MI355X",
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "SuperPoint extract stage of the front-end (BASELINE configs #2/#4): "
                                   f"{B} synthetic 640x480 frames per GPU per step resident in HBM, nFeatures=1024, "
                                   "nLevels=1, iniTh=0.015, minTh=0.007, seeded synthetic weights; LightGlue/YOLO "
                                   "stages not yet in the step",
                       "frames_per_gpu_per_step": B, "keypoints_per_frame": nk[:4],
                       "parallelism": f"frame-sharded x{world}, no data-path collective"},
            "roofline": {"bound": "mfma", "kernel": DOMINANT_KERNEL, "achieved": round(achieved, 3),
                         "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / FP32_MATRIX_PEAK_TFLOPS, 4), "traffic": traffic,
                         "launches_per_step": len(DOMINANT_LAYERS),
                         "avg_launch_ms": round(dom_ms / len(DOMINANT_LAYERS), 4),
                         "algorithmic_gflop_per_step": round(dom_flops / 1e9, 2)},
            "stages_ms_per_step": {"conv_stack": round(conv_ms, 3), "softmax_norm_cells": round(float(stage_ms[1]), 3),
                                   "host_quadtree": round(float(stage_ms[2]), 3), "descriptors": round(float(stage_ms[3]), 3)},
            "conv_stack_tflops": round(SP_GFLOP_PER_FRAME * B / conv_ms, 2) if conv_ms > 0 else None,
            "layer_ms": [round(float(x), 4) for x in layer_ms],
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(frames[:4])
        print(json.dumps(res), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
