#!/usr/bin/env python3
"""bench.py -- front-end throughput on MI355X (BASELINE.json metric: frames/sec at 640x480).

A "step" = one pass of the hot path over one batch of synthetic frames already resident in HBM:
FRAMES_PER_GPU (8 = BASELINE config #4's 64 frames / 8 GPUs) 640x480 frames per GPU through dani_frontend_batch,
i.e. the full front-end of BASELINE config #4: YOLOv5s dynamic-object boxes -> SuperPoint extraction (1024
keypoints, boxes culled) -> LightGlue match of every frame against its predecessor.  The frames are consecutive frames
of ONE synthetic camera trajectory (2 degrees of rotation + 8 px per frame; dani_slam_amd/synth.py
sequence_frame), batch-sharded over ranks with no data-path collective (weak scaling); the
synthetic SuperPoint weights carry the centred descriptor head, so LightGlue finds ~100 matches per consecutive pair.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement) with extra objects:
  roofline      -- the dominant kernel by total time (linear_rs_kernel, LightGlue's linear layers, 55 launches per
                   step) priced against the gfx950 FP32 matrix peak, from HIP events recorded on the engine's own
                   stream around every launch during sampled timed steps; `other_kernels` prices the attention and
                   the pooled conv kernels the same way; `traffic` is the PMC figure committed under profiles/;
  cpu_baseline  -- the reference-faithful CPU path (ATen-CPU operators for the three networks + the restated host logic,
                   oracle/aten_ref.py frontend_frame) timed on this host's cores on a bounded sample of the same frames,
                   single-thread and all cores, per stage (rank 0, N = 1 only);
  step_ms       -- median / p10 / p90 of the per-step wall times of the timed region (rank 0);
  latency_b1_ms -- BASELINE configs #2 (SuperPoint extract, one frame per call) and #3 (SuperPoint x 2 + one LightGlue
                   pair per call), measured after the timed region (N = 1 only); `--config 2|3` makes one of them the run.
A default N = 1 run also adds lg_gemm_bf16x3 / lg_gemm_bf16x3_all (the same workload with LightGlue's matrix products
in the opt-in bf16x3 arithmetic), measured AFTER the timed region and never part of `value`.
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
SP_GFLOP_PER_FRAME = 52.10
FP32_MATRIX_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
POOLED_CONV_LAYERS = (1, 3, 5)    # conv1b, conv2b, conv3b: conv_mfma_kernel<Cfg<3,1,1,32,2,8,RELU,pool>> (third kernel by total time;
POOLED_CONV_KERNEL = "conv_mfma_kernel<Cfg<3,1,1,32,2,8,RELU,pool>> (conv1b+conv2b+conv3b)"   # the dominant one is linear_rs_kernel)


def exact_layer_flops(li, B):
    from dani_slam_amd.synth import SP_LAYERS
    dims = [(H, W), (H, W), (H // 2, W // 2), (H // 2, W // 2), (H // 4, W // 4), (H // 4, W // 4)] + [(H // 8, W // 8)] * 6
    _, cin, cout, k = SP_LAYERS[li]
    h, w = dims[li]
    return 2.0 * h * w * cout * cin * k * k * B


def pct(xs, q):
    import numpy as np
    return round(float(np.percentile(np.asarray(xs, np.float64), q)), 4) if len(xs) else None


def host_cpu_share():
    """Cores this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU box hands a job a share
    of the host's cores; 256 threads on a 16-core quota are throttled to a crawl)."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return n


def cpu_baseline(gray, bgr, weights):
    """The reference's CPU path restated: LibTorch's own ATen-CPU kernels for SuperPoint::forward, the LightGlue and
    YOLOv5s forwards (the reference runs all three on LibTorch-CPU: src/SPextractor.cpp:230, src/LightGlue.cpp:10,
    src/YoloDetect.cpp:9) + the literal restatement of its host logic (per-cell getKeyPoints / NMS2 loop, quadtree, box
    logic), one frame at a time as the reference does (batch 1).  Timed on a bounded sample of the bench frames with one
    thread and with all host cores."""
    import torch
    from oracle import aten_ref, oracle
    oracle.lib()
    cores = host_cpu_share()
    os.environ["OMP_NUM_THREADS"] = str(cores)      # the C host logic of the oracle (OpenMP) gets the same share
    tw = aten_ref.torch_weights(weights)
    runs = {}
    # "all cores": ATen's intra-op pool does not always scale to the whole share on these layer sizes (and a share reported
    # too large is throttled), so the thread count is picked by a one-frame trial among {share, 64, 16}, and stated
    trial = {}
    for nt in sorted({cores, min(cores, 64), min(cores, 16)}):      # ascending; stop as soon as more threads are slower
        torch.set_num_threads(nt)
        aten_ref.frontend_frame(tw, gray[0], bgr[0], None, nfeatures=NFEATURES)      # untimed: first-call set-up of the ATen kernels
        t0 = time.perf_counter()
        aten_ref.frontend_frame(tw, gray[0], bgr[0], None, nfeatures=NFEATURES)
        trial[nt] = time.perf_counter() - t0
        if trial[nt] > min(trial.values()):
            break
    many = min(trial, key=trial.get)
    for label, nthreads, budget_s, max_frames in (("all_cores", many, 12.0, 48), ("one_thread", 1, 12.0, 8)):
        torch.set_num_threads(nthreads)
        aten_ref.frontend_frame(tw, gray[0], bgr[0], None, nfeatures=NFEATURES)
        stages, prev, n = {}, None, 0
        t0 = time.perf_counter()
        while True:
            i = n % len(gray)
            prev = aten_ref.frontend_frame(tw, gray[i], bgr[i], prev, nfeatures=NFEATURES, stages=stages)
            n += 1
            dt = time.perf_counter() - t0
            if dt > budget_s or n >= max_frames:
                break
        runs[label] = {"frames_per_s": round(n / dt, 4), "threads": nthreads, "frames": n, "seconds": round(dt, 2),
                       "matches_last_pair": int(len(prev.get("matches", []))),
                       "stage_ms_per_frame": {k: round(v / n * 1e3, 2) for k, v in stages.items()}}
    runs["all_cores"]["thread_count_trial_s_per_frame"] = {str(k): round(v, 3) for k, v in trial.items()}
    torch.set_num_threads(cores)
    best = runs["all_cores"]
    return {"value": best["frames_per_s"], "unit": "frames/s", "cores": best["threads"], "kind": "port", "host_cpu_share": cores,
            "flavour": f"ATen-CPU op sequence of the reference (torch {torch.__version__}, torch.get_num_threads() = {best['threads']}) + "
                       "restated host logic in C (the reference's own 204-cell loop indexes tensors element by element and is "
                       "slower than this); the reference's sources need OpenCV and cannot be built on this box",
            "sample": f"{best['frames']} consecutive 640x480 frames of the bench sequence, one at a time (batch 1, as the "
                      f"reference): YOLOv5s + NMS + box logic -> SuperPoint forward + 204-cell loop + quadtree + descriptors -> "
                      f"LightGlue vs the previous frame; {best['seconds']} s on {best['threads']} threads, then "
                      f"{runs['one_thread']['frames']} frames in {runs['one_thread']['seconds']} s on 1 thread",
            "one_thread": runs["one_thread"], "all_cores": runs["all_cores"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=4, choices=(2, 3, 4),
                    help="BASELINE config: 4 (default) = full front-end, frames sharded over the GPUs; 2 = SuperPoint extract, ONE "
                         "frame per call; 3 = SuperPoint x 2 + one LightGlue pair per call (a stereo frame).  2 and 3 are the "
                         "batch-1 operating points of a live tracker: a step is one call, the line reports calls' latency")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frames-per-gpu", type=int, default=FRAMES_PER_GPU,
                    help="frames per GPU per step (default 8 = BASELINE config #4's 64 frames / 8 GPUs)")
    ap.add_argument("--host-inputs", action="store_true",
                    help="hand the frames over as HOST buffers (PCIe-inclusive rate for DESIGN.md; never the headline value)")
    ap.add_argument("--sequences", type=int, default=1,
                    help="independent image sequences (cameras) processed concurrently per GPU, one host thread each; "
                         "steps are dealt round-robin to them.  2 gives a few %% more frames/s by filling host-sync gaps, but the "
                         "per-kernel HIP-event durations behind `roofline` then include time shared with the other "
                         "sequence's kernels, so the default keeps one sequence for clean attribution")
    ap.add_argument("--lg-gemm", choices=("f32", "bf16x3", "bf16x3_all"), default="f32",
                    help="arithmetic of LightGlue's linear layers for the TIMED region (dani_lg_set_gemm): f32 = f32 MFMA in the "
                         "oracle's canonical order (default, the headline); bf16x3 = three-term bf16 operand split, f32-accurate. "
                         "A default run also reports the bf16x3 modes, measured after the timed region")
    ap.add_argument("--no-boundary-exchange", action="store_true",
                    help="N > 1: treat every rank's frames as a sequence of its own instead of exchanging the shard-boundary frame "
                         "(rank r's first frame is then matched against rank r's own last frame of the previous step)")
    ap.add_argument("--no-pipelined-exchange", action="store_true",
                    help="N > 1: lock-step form (extract, all-gather, match inside one step) instead of matching step t behind "
                         "the extraction of step t + 1")
    ap.add_argument("--force-boundary-exchange", action="store_true",
                    help="N = 1 only: take the N > 1 code path (extract, export of the last frame's features on the device, "
                         "match against the carried slab) to price the split call on one GPU")
    ap.add_argument("--exchange", choices=("native", "torch"), default="native",
                    help="N > 1: who runs the shard-boundary all-gather -- native = the engine itself (ncclAllGather behind the C ABI on the "
                         "sequence's stream, csrc/comm_engine.cpp; default with the nccl backend), torch = torch.distributed's "
                         "all_gather_into_tensor (the form of rounds 2-3; always used by the gloo rehearsal on one device)")
    ap.add_argument("--no-side-modes", action="store_true", help="skip the extra bf16x3 / batch-1 measurements (profiling runs)")
    ap.add_argument("--host-wait", choices=("library", "hybrid", "spin", "block"), default="library",
                    help="how the calling thread waits for the GPU inside the engine (dani_set_host_wait): library = whatever the library "
                         "does when nobody asks (the headline; today: spin on hipStreamQuery); block = hipStreamSynchronize; hybrid = naps "
                         "until the engine's near-end marker, then spins (one host core instead of 2.3).  A default N = 1 run reports the "
                         "other modes as `host_wait_block` / `host_wait_hybrid` / `host_wait_spin`, each with its host CPU time per step")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import numpy as np
    import torch
    import torch.distributed as dist
    from dani_slam_amd import frontend, synth

    # rehearsal knobs (1-GPU box): DANI_BENCH_BACKEND=gloo DANI_BENCH_ONE_DEVICE=1 runs all ranks on GPU 0 over gloo
    backend = os.environ.get("DANI_BENCH_BACKEND", "nccl")
    if os.environ.get("DANI_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)     # "nccl" is RCCL on ROCm (xGMI on the MI355X node)
        else:
            dist.init_process_group(backend)

    from dani_slam_amd import sharding
    cfg = args.config
    B = {2: 1, 3: 2}.get(cfg, args.frames_per_gpu)
    shard = sharding.shard_frames(B * world, rank, world)   # this rank's block of consecutive frames of the trajectory
    frames, frames_bgr = synth.bench_frames(shard)
    d_frames = torch.from_numpy(frames).to(dev)          # inputs resident in HBM before the timed region
    d_bgr = torch.from_numpy(frames_bgr).to(dev)
    weights = synth.bench_weights()
    eng = frontend.Engine(device=local_rank, max_batch=max(B, 2), max_width=W, max_height=H, weights=weights)
    eng.yolo_set_classes(synth.COCO_NAMES, synth.DYNAMIC_NAMES)   # coco.names is not shipped by the reference
    if args.host_wait != "library":
        eng.set_host_wait(args.host_wait)
    wait_in_force = eng.host_wait()
    if args.lg_gemm != "f32":
        eng.lg_set_gemm(eng.LG_GEMM_BF16X3 if args.lg_gemm == "bf16x3" else eng.LG_GEMM_BF16X3_ALL)

    import threading
    NSEQ = max(1, args.sequences) if cfg == 4 else 1
    STAGGER_MS = float(os.environ.get("DANI_BENCH_STAGGER_MS", "0"))
    PROF_EVERY = 20
    # N > 1: the batch is ONE sequence of consecutive frames sharded over the ranks (BASELINE config #4), so the pair
    # (last frame of rank r-1, first frame of rank r) crosses a GPU boundary: extract -> all-gather of every rank's last
    # frame (features packed on the device, one RCCL all_gather_into_tensor over xGMI) -> LightGlue with that predecessor
    boundary = cfg == 4 and (world > 1 or args.force_boundary_exchange) and NSEQ == 1 and not args.no_boundary_exchange
    CAP = NFEATURES + 12
    carry = {"slab": None}      # rank 0: the last rank's final frame of the previous step
    # N > 1: the LightGlue of step t runs behind the extraction of step t + 1 (two engine sequences, ping-pong), with the
    # all-gather of step t's boundary features posted in between: one extraction of slack, so that a rank that is a few
    # milliseconds late in one step does not stall the others (lock-step, every step costs the MAX over ranks).
    pipelined = boundary and not args.no_pipelined_exchange
    pend = {"job": None, "step": 0}
    xchg = None
    # the collective itself: native RCCL inside the engine when every rank owns a GPU (RCCL refuses two ranks on one device: the
    # gloo / one-device rehearsal keeps torch.distributed's form).  A communicator that cannot be built is reported and replaced
    # by the torch form -- the line's `parallelism` field says which one ran.
    native = None
    exchange_used = "none"
    if boundary:
        exchange_used = "torch.distributed all_gather_into_tensor"
        if args.exchange == "native" and (backend == "nccl" or world == 1) and os.environ.get("DANI_BENCH_ONE_DEVICE") != "1":
            try:
                native = sharding.NativeExchange(eng, CAP, rank, world)
                exchange_used = f"native ncclAllGather behind the C ABI (RCCL {native.info()['rccl_version']}, dani_frontend_exchange_boundary)"
            except Exception as ex:
                print(f"[bench rank {rank}] native RCCL exchange unavailable ({ex}); using torch.distributed", file=sys.stderr, flush=True)
                native = None
        if world > 1:      # all ranks must agree (a rank that failed alone would leave the others waiting in the collective)
            flag = torch.tensor([1 if native is not None else 0], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0 and native is not None:
                native.close(); native = None
                exchange_used = "torch.distributed all_gather_into_tensor (native RCCL failed on another rank)"

    def call(mode, seq=0, nb=None):
        """mode 4: full front-end; 2: SuperPoint extract only; 3: SuperPoint x 2 + LightGlue of the (left, right) pair."""
        nb = nb or B
        kw = dict(nfeatures=NFEATURES, want_desc=True, copy=False, sequence=seq)
        if mode == 2:
            kw.update(use_yolo=False, match_previous=False)
        elif mode == 3:
            kw.update(use_yolo=False, stereo_pairs=True)
        if mode == 4 and pipelined:
            return call_pipelined(kw, nb)
        if mode == 4 and boundary:
            kw.update(match_previous=False, defer_desc=True)   # descriptors reach the host under the LightGlue of match_resident
        if args.host_inputs:
            out_ = eng.frontend_batch(frames[:nb], frames_bgr[:nb] if mode == 4 else None, **kw)
        else:
            out_ = eng.frontend_batch(None, None, device_ptrs=(d_frames.data_ptr(), d_bgr.data_ptr()), shape=(nb, H, W), **kw)
        if mode == 4 and boundary and native is not None:
            native.start(nb - 1, seq)                       # pack + ncclAllGather on the sequence's stream, no host wait
            for o_, (m_, s_) in zip(out_, native.match(nb, seq)):
                o_["matches"], o_["scores"] = m_, s_
        elif mode == 4 and boundary:
            match_boundary(out_, sharding.exchange_boundary_features(eng, nb - 1, CAP, reduce_dev, seq), nb, seq)
        return out_

    def match_boundary(out_, slabs, nb, seq):
        pred = slabs[rank - 1] if rank > 0 else carry["slab"]
        if pred is None:
            eng.frontend_reset(seq)
            res_ = eng.frontend_match_resident(nb, cap=CAP, sequence=seq)
        elif pred.is_cuda:
            res_ = eng.frontend_match_resident(nb, pred_device_ptr=pred.data_ptr(), pred_cap=CAP, cap=CAP, sequence=seq)
        else:
            res_ = eng.frontend_match_resident(nb, pred=pred.numpy(), cap=CAP, sequence=seq)
        if rank == 0:
            carry["slab"] = slabs[world - 1].clone()
        for o_, (m_, s_) in zip(out_, res_):
            o_["matches"], o_["scores"] = m_, s_
        return out_

    def finish_pending():
        job = pend["job"]
        if job is None:
            return None
        pend["job"] = None
        out_, ticket, nb_, seq_ = job
        if native is not None:
            for o_, (m_, s_) in zip(out_, native.match(nb_, seq_)):
                o_["matches"], o_["scores"] = m_, s_
            return out_
        return match_boundary(out_, xchg.wait(ticket), nb_, seq_)

    def call_pipelined(kw, nb):
        nonlocal xchg
        if xchg is None and native is None:
            xchg = sharding.BoundaryExchange(eng, CAP, reduce_dev)
        seq = pend["step"] & 1
        pend["step"] += 1
        kw = dict(kw, match_previous=False, defer_desc=True, sequence=seq)
        if args.host_inputs:
            out_ = eng.frontend_batch(frames[:nb], frames_bgr[:nb], **kw)
        else:
            out_ = eng.frontend_batch(None, None, device_ptrs=(d_frames.data_ptr(), d_bgr.data_ptr()), shape=(nb, H, W), **kw)
        if pend.get("sample"):                    # SuperPoint's layer timers belong to THIS extraction's context: read them now
            pend["lt"], pend["ft"] = eng.layer_timings(), eng.frontend_timings()
        ticket = native.start(nb - 1, seq) if native is not None else xchg.start(nb - 1, seq)
        done = finish_pending()                  # LightGlue of the PREVIOUS step, whose all-gather was posted one extraction ago
        pend["job"] = (out_, ticket, nb, seq)
        return done if done is not None else out_

    def run_steps(total, collect, mode=cfg, nb=None):
        """`total` steps dealt round-robin to NSEQ concurrent sequences (one host thread each).  Returns the last outputs,
        the per-kernel profile averages over the sampled steps and every step's wall time in ms."""
        outs = [None] * NSEQ
        acc = [(np.zeros(12), np.zeros(4), np.zeros(4), np.zeros(3)) for _ in range(NSEQ)]
        errs = []
        nsamp = [0] * NSEQ
        step_ms = [[] for _ in range(NSEQ)]

        def worker(seq):
            try:
                if seq and STAGGER_MS > 0:        # sequence q starts q * STAGGER_MS late: its SuperPoint / YOLO phase then runs beside
                    time.sleep(seq * STAGGER_MS * 1e-3)   # the other sequence's LightGlue instead of beside its SuperPoint / YOLO phase
                for it in range(total // NSEQ + (1 if seq < total % NSEQ else 0)):
                    # per-kernel HIP-event timing costs ~0.4 ms per step (166 events): sample every PROF_EVERY-th step
                    sample = collect and mode == 4 and it % PROF_EVERY == min(PROF_EVERY, total // NSEQ) // 2
                    if collect:
                        eng.set_profiling(sample)
                    pend["sample"] = sample
                    ts = time.perf_counter()
                    outs[seq] = call(mode, seq, nb)
                    step_ms[seq].append((time.perf_counter() - ts) * 1e3)
                    if sample:
                        nsamp[seq] += 1
                        acc[seq][0][:] += pend["lt"] if (pipelined and mode == 4) else eng.layer_timings()
                        acc[seq][1][:] += pend["ft"] if (pipelined and mode == 4) else eng.frontend_timings()
                        acc[seq][2][:] += np.array(eng.lg_attention_profile(), np.float64)
                        acc[seq][3][:] += np.array(eng.lg_linear_profile(), np.float64)
            except Exception as ex:   # surface worker failures in the main thread
                errs.append(ex)
        if NSEQ == 1:
            worker(0)
            if mode == 4 and pipelined:          # the last step's LightGlue belongs to this region
                ts = time.perf_counter()
                last = finish_pending()
                if last is not None:
                    outs[0] = last
                    if step_ms[0]:
                        step_ms[0][-1] += (time.perf_counter() - ts) * 1e3
        else:
            th = [threading.Thread(target=worker, args=(q,)) for q in range(NSEQ)]
            for t_ in th: t_.start()
            for t_ in th: t_.join()
        if errs:
            raise errs[0]
        ns = max(sum(nsamp), 1)   # per-step averages over the sampled steps
        return (outs, sum(a[0] for a in acc) / ns, sum(a[1] for a in acc) / ns, sum(a[2] for a in acc) / ns,
                sum(a[3] for a in acc) / ns, [x for s_ in step_ms for x in s_])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    reduce_dev = dev if backend == "nccl" else None

    # settle phase (untimed, part of set-up): first touches of every workspace, clocks and the host worker pool reach steady
    # state; a process that starts right after another one released the GPU sees a few slow steps
    run_steps(30, False)
    if cfg == 4:   # profiled steps outside the timed region: every context creates its ~170 timing events on first use (7 ms);
        # the pipelined N > 1 form alternates two contexts and runs a step's LightGlue one call later: four calls cover both
        for q in (range(4) if pipelined else range(NSEQ)):
            eng.set_profiling(True); call(cfg, q % max(NSEQ, 1)); eng.set_profiling(False)
    # like timeit: no cyclic-GC pass inside the timed region (a generation-2 collection over the weight / frame objects
    # of this process takes 35-50 ms, four steps' worth, and lands ~0.7 s after start-up).  Collected BEFORE the warm-up
    # steps: the GPU idles while the collector runs and the first step after such a pause is ~2 ms slow (clocks).
    import gc
    gc.collect()
    gc.disable()
    run_steps(max(args.warmup, NSEQ), False)
    barrier()
    def thread_cpu():   # CPU seconds of every thread of this process so far (utime + stime), by thread id
        out, tck = {}, os.sysconf("SC_CLK_TCK")
        for tid in os.listdir("/proc/self/task"):
            try:
                f = open(f"/proc/self/task/{tid}/stat").read()
                name = f[f.index("(") + 1:f.rindex(")")]
                v = f[f.rindex(")") + 2:].split()
                out[tid] = (name, (int(v[11]) + int(v[12])) / tck)
            except (OSError, ValueError):
                pass
        return out
    t0 = time.perf_counter()
    cpu0 = time.process_time()
    tc0 = thread_cpu()
    outs, layer_ms, fe_ms, attn, lin, step_ms = run_steps(args.steps, True)
    cpu1 = time.process_time()
    tc1 = thread_cpu()
    cpu_by_thread = sorted(((round((tc1[t][1] - tc0.get(t, (None, 0.0))[1]) / args.steps * 1e3, 2), tc1[t][0] + ("" if t != str(os.getpid()) else " (caller)")) for t in tc1), reverse=True)[:5]
    barrier()
    dt = time.perf_counter() - t0
    eng.set_profiling(False)
    out = next(o for o in outs if o is not None)
    nk = [len(o["kpts"]) for o in out]
    nmatch = [len(o["matches"]) for o in out]
    nbox = [len(o["boxes"]) for o in out]

    # ---- outside the timed region (N = 1): the opt-in arithmetics and the batch-1 operating points
    side = {}
    SIDE_STEPS = 100
    extras = cfg == 4 and args.lg_gemm == "f32" and world == 1 and NSEQ == 1 and not args.no_side_modes
    if extras:
        ref = [(np.array(o["matches"]).copy(), np.array(o["scores"]).copy()) for o in call(4)]   # f32 mode
        for key, mode, what in (("lg_gemm_bf16x3", eng.LG_GEMM_BF16X3, "linear layers"),
                                ("lg_gemm_bf16x3_all", eng.LG_GEMM_BF16X3_ALL, "linear layers and attention contractions")):
            eng.lg_set_gemm(mode)
            run_steps(20, False)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            outs3, _, fe3, attn3, lin3, sm3 = run_steps(SIDE_STEPS, True)
            torch.cuda.synchronize()
            dt3 = time.perf_counter() - t1
            eng.set_profiling(False)
            worst, flips, nm = 0.0, 0, 0
            for (m_a, s_a), o in zip(ref, outs3[0]):
                a = {tuple(x): float(v) for x, v in zip(m_a.tolist(), s_a)}
                b = {tuple(x): float(v) for x, v in zip(np.array(o["matches"]).tolist(), np.array(o["scores"]))}
                flips += len(set(a) ^ set(b)); nm += len(a)
                worst = max([worst] + [abs(a[k] - b[k]) for k in set(a) & set(b)])
            side[key] = {"value": round(B * SIDE_STEPS / dt3, 3), "unit": "frames/s", "steps": SIDE_STEPS, "ms_per_step": round(dt3 / SIDE_STEPS * 1e3, 4),
                         "ms_per_step_median": pct(sm3, 50), "lightglue_ms_per_step": round(float(fe3[2]), 3),
                         "linear_avg_launch_ms": round(float(lin3[0]) / max(int(round(lin3[2])), 1), 4),
                         "attention_avg_launch_ms": round(float(attn3[0]) / max(int(round(attn3[2])), 1), 4),
                         "matches_compared": nm, "max_abs_score_diff_vs_f32_mode": worst, "matches_not_in_both": flips,
                         "note": f"dani_lg_set_gemm({key[8:].upper()}): {what} as six bf16 MFMAs over three-term operand splits, "
                                 "f32 accumulation; opt-in, not the headline"}
        eng.lg_set_gemm(eng.LG_GEMM_F32)
        for hw in ("block", "hybrid", "spin"):   # the other wait modes, same workload, after the timed region
            if hw == wait_in_force:
                continue
            eng.set_host_wait(hw)
            run_steps(20, False)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            c1 = time.process_time()
            _, _, _, _, _, sms = run_steps(SIDE_STEPS, False)
            c2 = time.process_time()
            torch.cuda.synchronize()
            dts = time.perf_counter() - t1
            eng.set_host_wait(wait_in_force)
            side["host_wait_" + hw] = {"value": round(B * SIDE_STEPS / dts, 3), "unit": "frames/s", "steps": SIDE_STEPS,
                                       "ms_per_step": round(dts / SIDE_STEPS * 1e3, 4), "ms_per_step_median": pct(sms, 50), "ms_per_step_max": pct(sms, 100),
                                       "host_cpu_ms_per_step": round((c2 - c1) / SIDE_STEPS * 1e3, 3),
                                       "note": {"block": "dani_set_host_wait(0): hipStreamSynchronize", "spin": "dani_set_host_wait(1): the calling thread polls hipStreamQuery",
                                                "hybrid": "dani_set_host_wait(2): naps until the engine's near-end marker, then polls"}[hw] + "; not the headline"}
        # BASELINE configs #2 and #3: one call at a time, as a live tracker issues them (src/Frame.cc:420-427, :813-860), on an
        # engine built for that use (max_batch = 2: the engines pick their tile shapes for the batch they are built for)
        lat = {}
        eng_big = eng
        eng = frontend.Engine(device=local_rank, max_batch=2, max_width=W, max_height=H, weights=weights)
        eng.yolo_set_classes(synth.COCO_NAMES, synth.DYNAMIC_NAMES)
        for key, mode, nb, what in (("config2_superpoint_extract_1_frame", 2, 1, "SPextractor::operator() on one 640x480 frame, 1024 keypoints"),
                                    ("config3_stereo_pair_sp_x2_plus_lightglue", 3, 2, "two extractions + one LightGlue pair (Frame::ComputeStereoMatches)")):
            eng.frontend_reset()
            run_steps(30, False, mode, nb)
            torch.cuda.synchronize()
            _, _, _, _, _, sm = run_steps(200, False, mode, nb)
            lat[key] = {"median": pct(sm, 50), "p10": pct(sm, 10), "p90": pct(sm, 90), "calls": len(sm), "frames_per_call": nb,
                        "frames_per_s": round(nb * len(sm) / (sum(sm) * 1e-3), 2), "what": what}
        side["latency_b1_ms"] = lat
        eng.close()
        eng = eng_big
        eng.frontend_reset()
    gc.enable()

    dt_max = sharding.reduce_max_seconds(dt, reduce_dev)
    first_pair = None
    if world > 1 and cfg == 4:   # matches of every rank's FIRST frame (the pair that crosses the shard boundary), for the report
        t_ = torch.tensor([nmatch[0]], dtype=torch.int64, device=reduce_dev or "cpu")
        g_ = torch.empty(world, dtype=torch.int64, device=reduce_dev or "cpu")
        dist.all_gather_into_tensor(g_, t_)
        first_pair = [int(v) for v in g_.cpu().tolist()]

    if rank == 0:
        total_frames = B * world * args.steps
        fps = total_frames / dt_max
        res = {
            "metric": "frames/sec SuperPoint+LightGlue+YOLO front-end, 640x480, 1/2/4/8 MI355X",
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.lg_gemm == "f32" else f"f32 (LightGlue matrix products: {args.lg_gemm} operand split on the bf16 MFMA, f32 accumulate)", "data": "synthetic" + (" (frames handed over as HOST buffers: PCIe-inclusive)" if args.host_inputs else ""),
            "host_wait": wait_in_force + (" (the library's default)" if args.host_wait == "library" else " (asked for with --host-wait)") +
                         {"hybrid": ": naps until the engine's near-end marker, then polls hipStreamQuery", "spin": ": the calling thread polls hipStreamQuery",
                          "block": ": hipStreamSynchronize"}[wait_in_force],
            "host_cpu_ms_per_step": round((cpu1 - cpu0) / args.steps * 1e3, 3),   # CPU time of the whole process (caller, engine pool, runtime threads) inside the timed region
            "host_cpu_ms_per_step_top_threads": [[ms, nm] for ms, nm in cpu_by_thread if ms > 0.0],   # 10 ms clock ticks: coarse
            "step_ms": {"median": pct(step_ms, 50), "p10": pct(step_ms, 10), "p90": pct(step_ms, 90), "min": pct(step_ms, 0), "max": pct(step_ms, 100),
                        "n": len(step_ms), "what": "wall time of every dani_frontend_batch call of the timed region on rank 0 (all sequences)",
                        "slow_steps": [[i, round(x, 2)] for i, x in enumerate(step_ms) if x > pct(step_ms, 50) + 0.3][:24]},
        }
        if cfg == 4:
            conv_dom_ms = float(sum(layer_ms[i] for i in POOLED_CONV_LAYERS))
            conv_dom_flops = sum(exact_layer_flops(i, B) for i in POOLED_CONV_LAYERS)
            conv_achieved = conv_dom_flops / (conv_dom_ms * 1e-3) / 1e12 if conv_dom_ms > 0 else 0.0
            # second kernel by total time: LightGlue attention (18 launches per step)
            # attention: priced on the ALGORITHMIC flop (a pair's cross scores counted once, 3 N M d MAC); the kernel
            # executes K Q^T once per direction (4 N M d), reported beside it
            attn_ms, attn_flop_exec, attn_launches, attn_flop = float(attn[0]), float(attn[1]), int(round(attn[2])), float(attn[3])
            attn_achieved = attn_flop / (attn_ms * 1e-3) / 1e12 if attn_ms > 0 else 0.0
            attn_executed = attn_flop_exec / (attn_ms * 1e-3) / 1e12 if attn_ms > 0 else 0.0
            # dominant kernel of the full front-end by total time (profiles/: linear_rs_kernel, 55 launches per step)
            lin_ms, lin_flop, lin_launches = float(lin[0]), float(lin[1]), int(round(lin[2]))
            achieved = lin_flop / (lin_ms * 1e-3) / 1e12 if lin_ms > 0 else 0.0
            traffic, traffic_src = None, None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")   # PMC-derived HBM bytes per launch (committed)
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_src = "profiles/traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/run_profiles.sh), NOT measured in this run"
                except Exception:
                    traffic = None
            conv_ms = float(layer_ms.sum())
            res["config"] = {"workload": "full front-end (BASELINE config #4 sharded): YOLOv5s boxes -> SuperPoint extract -> "
                                         f"LightGlue vs previous frame, {B} consecutive frames of a synthetic camera trajectory (2 deg + 8 px "
                                         "per frame, 640x480) per GPU per step resident in HBM, nFeatures=1024, nLevels=1, iniTh=0.015, "
                                         "minTh=0.007, LightGlue 9 layers no early exit, seeded synthetic weights (centred descriptor head)",
                             "frames_per_gpu_per_step": B, "keypoints_per_frame": nk[:8], "matches_per_frame": nmatch[:8],
                             "dynamic_boxes_per_frame": nbox[:8], "concurrent_sequences_per_gpu": NSEQ,
                             "first_frame_matches_per_rank": first_pair,
                             "note_matches": "frame 0 of a step is matched against frame B-1 of the previous step (the same 8 frames "
                                             "are replayed every step): a 14-degree jump, few matches; frames 1..B-1 are consecutive",
                             "exchange": exchange_used,
                             "parallelism": (f"frame-sharded x{world}; one all-gather of every rank's last-frame features per step (1 MB per rank, "
                                             "device buffers) for the pair that crosses a shard boundary"
                                             + ("; posted after a step's extraction and waited for one extraction later: the LightGlue of "
                                                "step t runs behind the extraction of step t + 1" if pipelined else "; lock-step") if boundary else
                                             f"frame-sharded x{world}, no data-path collective")}
            chain_active = lin_launches != 55      # lg_forward counts an FFN block that went through ffn_chain_kernel as ONE timed interval
            res["roofline"] = {"bound": "mfma", "kernel": "linear_rs_kernel<NCB,UQ> (LightGlue feature-major linear layers, register-streaming form of lin_rs.h: 6 per layer x 9 + final_proj; linear_fm_kernel<64> where its grid fills the chip better)"
                                       + (" + ffn_chain_kernel (ffn.0 -> LayerNorm + GELU -> ffn.3 -> next projection as one launch per block, ffn_chain.h: taken at this batch size; "
                                          "its LayerNorm + GELU time is inside the denominator, the numerator counts the linear layers' FLOP only)" if chain_active else ""),
                               "achieved": round(achieved, 3), "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(achieved / FP32_MATRIX_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                               "launches_per_step": lin_launches,
                               "avg_launch_ms": round(lin_ms / max(lin_launches, 1), 4),
                               "algorithmic_gflop_per_step": round(lin_flop / 1e9, 2),
                               "other_kernels": {
                                   "attn_rs_kernel (LightGlue f32 flash attention, register-streaming form of attn_rs.h, 9 self + 9 cross launches; attn_ks_kernel / attn_k4_kernel below 512 workgroups)": {
                                       "achieved": round(attn_achieved, 3), "frac": round(attn_achieved / FP32_MATRIX_PEAK_TFLOPS, 4),
                                       "avg_launch_ms": round(attn_ms / max(attn_launches, 1), 4), "launches_per_step": attn_launches,
                                       "algorithmic_gflop_per_step": round(attn_flop / 1e9, 2),
                                       "executed_gflop_per_step": round(attn_flop_exec / 1e9, 2),
                                       "executed_tflops": round(attn_executed, 3), "executed_frac": round(attn_executed / FP32_MATRIX_PEAK_TFLOPS, 4),
                                       "note": "a cross launch forms K Q^T once per direction (S and S^T of a pair): executed = 4/3 of the "
                                               "algorithmic cross work; DESIGN.md section 5.2 says why the flash form cannot share it"},
                                   POOLED_CONV_KERNEL: {"achieved": round(conv_achieved, 3),
                                                     "frac": round(conv_achieved / FP32_MATRIX_PEAK_TFLOPS, 4),
                                                     "avg_launch_ms": round(conv_dom_ms / len(POOLED_CONV_LAYERS), 4),
                                                     "launches_per_step": len(POOLED_CONV_LAYERS)}}}
            # host wall-clock phases of one call (they overlap device work of the neighbouring phases; NOT device time per network)
            res["stages_ms_per_step"] = {
                "phase1_enqueue_sp_network_then_wait_for_yolo_boxes": round(float(fe_ms[0]), 3),
                "phase2_sp_candidates_sync_quadtree_descriptors_enqueue": round(float(fe_ms[1]), 3),
                "phase3_lightglue_enqueue_and_final_sync": round(float(fe_ms[2]), 3),
                "call_total": round(float(fe_ms[3]), 3),
                "sp_conv_stack_device_ms_sum_of_12_layers": round(conv_ms, 3)}
            res["sp_conv_stack_tflops"] = round(SP_GFLOP_PER_FRAME * B / conv_ms, 2) if conv_ms > 0 else None
            res["layer_ms"] = [round(float(x), 4) for x in layer_ms]
        else:
            what = {2: "BASELINE config #2: SuperPoint extract, ONE 640x480 frame per call, 1024 keypoints (SPextractor::operator(), src/Frame.cc:420-427)",
                    3: "BASELINE config #3: SuperPoint x 2 + one LightGlue pair per call, 640x480, 1024 keypoints per frame "
                       "(Frame::Frame's two ExtractORB calls + ComputeStereoMatches, src/Frame.cc:124-143, 813-860)"}[cfg]
            res["config"] = {"workload": what + "; frames of the synthetic camera trajectory resident in HBM, seeded synthetic weights",
                             "frames_per_call": B, "keypoints_per_frame": nk, "matches_per_frame": nmatch,
                             "parallelism": f"replicas x{world}: every rank runs its own calls, no collective"}
            res["latency_ms_per_call"] = res["step_ms"]
            res["roofline"] = {"bound": "latency", "note": "one frame (pair) per call: the chain of dependent launches is launch / "
                               "dependency-latency bound, not MFMA- or HBM-bound; the per-kernel roofline figures are those of the default run"}
        res.update(side)
        try:   # the facts the peaks in `roofline` are priced against (print them next to every roofline number)
            pr = torch.cuda.get_device_properties(dev)
            res["device"] = {"name": pr.name, "arch": getattr(pr, "gcnArchName", None), "compute_units": pr.multi_processor_count,
                             "hbm_gib": round(pr.total_memory / 2**30, 1),
                             "max_engine_clock_mhz": round(getattr(pr, "clock_rate", 0) / 1000) or None,
                             "fp32_matrix_peak_tflops_used": FP32_MATRIX_PEAK_TFLOPS}
        except Exception:
            pass
        if world == 1 and not args.no_cpu_baseline and cfg == 4:
            n = min(len(frames), 8)
            res["cpu_baseline"] = cpu_baseline(frames[:n], frames_bgr[:n], weights)
        print(json.dumps(res), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
