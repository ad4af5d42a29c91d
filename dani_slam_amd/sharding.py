"""Multi-GPU plumbing for the front-end: frames are independent units, so ranks take
disjoint frame blocks with NO data-path collective.  The only exchange the design foresees is one all-gather of
per-frame features (count, keypoints, descriptors) for the multi-camera / batched-sequence LightGlue case, where a
pair's two frames live on different GPUs.  ~1.06 MB per 1024-keypoint frame: one direct all-gather per batch (all 7
xGMI links of a GPU in parallel), not a ring of small messages.

Since round 4 that collective is NATIVE: `ncclAllGather` is called by the engine itself, behind the C ABI
(csrc/comm_engine.cpp: dani_comm_create / dani_frontend_exchange_boundary / dani_frontend_match_exchanged), on the
sequence's own HIP stream -- a C++ host can drive it, nothing is synchronised with the host.  `NativeExchange` below is
the thin Python caller; torch.distributed only carries the 128-byte RCCL id to the ranks (and bench.py's barrier /
max-reduce).  The torch.distributed forms (`exchange_boundary_features`, `BoundaryExchange`) stay for the gloo rehearsals
on CPU / on a one-GPU box, where RCCL cannot run several ranks on one device.
"""
from __future__ import annotations

import numpy as np


def shard_frames(n_frames: int, rank: int, world: int) -> range:
    """Contiguous block of frame indices owned by `rank` (blocks differ by at most one frame)."""
    base, rem = divmod(n_frames, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def reduce_max_seconds(dt: float, device=None) -> float:
    """MAX over ranks of a wall time (bench.py contract).  No-op without an initialised process group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(dt)
    t = torch.tensor([dt], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allgather_features(kpts, resp, desc, cap: int, device=None):
    """All-gather of one frame's features across ranks (RCCL all-gather on GPU tensors, gloo on CPU tensors).
    kpts [n,2], resp [n], desc [n,256] (numpy or torch, n <= cap).  Returns a list (len = world size) of
    (kpts, resp, desc) numpy triples.  Packing: one f32 slab of cap*(2+1+256)+1 floats per rank, count in slot 0,
    so a single collective moves everything."""
    import torch
    import torch.distributed as dist
    kpts = np.asarray(kpts, np.float32).reshape(-1, 2); resp = np.asarray(resp, np.float32).reshape(-1)
    desc = np.asarray(desc, np.float32).reshape(-1, 256)
    n = len(kpts)
    assert n <= cap and len(resp) == n and len(desc) == n
    slab = np.zeros(1 + cap * 259, np.float32)
    slab[0] = n
    slab[1:1 + 2 * n] = kpts.reshape(-1)
    slab[1 + 2 * cap:1 + 2 * cap + n] = resp
    slab[1 + 3 * cap:1 + 3 * cap + 256 * n] = desc.reshape(-1)
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return [(kpts, resp, desc)]
    dev = device or "cpu"
    mine = torch.from_numpy(slab).to(dev)
    out = torch.empty(world * slab.size, dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(out, mine)
    out = out.cpu().numpy().reshape(world, -1)
    res = []
    for r in range(world):
        m = int(out[r, 0])
        res.append((out[r, 1:1 + 2 * m].reshape(m, 2).copy(), out[r, 1 + 2 * cap:1 + 2 * cap + m].copy(),
                    out[r, 1 + 3 * cap:1 + 3 * cap + 256 * m].reshape(m, 256).copy()))
    return res


def exchange_boundary_features(eng, last_frame: int, cap: int, device=None, sequence: int = 0):
    """The one collective of BASELINE config #4 (one batch of consecutive frames sharded over the GPUs): every rank packs
    the features of its LAST resident frame into a slab ON THE DEVICE (dani_frontend_export_features, no host round trip)
    and one all-gather (RCCL over xGMI with backend "nccl": ~1 MB per rank, direct exchange) hands every rank all of
    them; rank r matches its first frame against the slab of rank r-1 (rank 0: against the last rank's slab of the
    previous step -- the caller keeps it).  With a gloo group (CPU rehearsal) the slab is staged through host memory.
    Returns the gathered [world, slab] tensor (on `device` for nccl, on the CPU for gloo)."""
    import torch
    import torch.distributed as dist
    nfl = eng.feature_slab_floats(cap)
    world = dist.get_world_size() if dist.is_initialized() else 1
    on_gpu = device is not None and (not dist.is_initialized() or dist.get_backend() == "nccl")
    if on_gpu:
        mine = torch.empty(nfl, dtype=torch.float32, device=device)
        eng.frontend_export_features(last_frame, cap, sequence, device_ptr=mine.data_ptr())
    else:
        mine = torch.from_numpy(eng.frontend_export_features(last_frame, cap, sequence))
    if world == 1:
        return mine.reshape(1, nfl)
    out = torch.empty(world * nfl, dtype=torch.float32, device=mine.device)
    dist.all_gather_into_tensor(out, mine)
    if on_gpu:
        # the collective is only ENQUEUED on torch's stream when the call returns; the engine reads the slab from its own
        # (non-blocking) HIP stream, so the gathered bytes must have landed before the pointer is handed over
        torch.cuda.current_stream(mine.device).synchronize()
    return out.reshape(world, nfl)


class BoundaryExchange:
    """exchange_boundary_features split into start() and wait(): the all-gather is posted right after a rank's extraction and
    waited for one extraction later (bench.py's N > 1 loop runs the LightGlue of step t behind the extraction of step t + 1, on
    a second engine sequence), so that a rank that was late by a few milliseconds in one step does not stall the other N - 1 --
    every step of the lock-step form costs the group the MAXIMUM of the ranks' step times."""

    def __init__(self, eng, cap: int, device=None):
        import torch.distributed as dist
        self.eng, self.cap, self.device = eng, cap, device
        self.nfl = eng.feature_slab_floats(cap)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.on_gpu = device is not None and (not dist.is_initialized() or dist.get_backend() == "nccl")

    def start(self, last_frame: int, sequence: int = 0):
        """Pack this rank's boundary frame (on the device for nccl) and post the collective.  Returns a ticket for wait()."""
        import torch
        import torch.distributed as dist
        if self.on_gpu:
            mine = torch.empty(self.nfl, dtype=torch.float32, device=self.device)
            self.eng.frontend_export_features(last_frame, self.cap, sequence, device_ptr=mine.data_ptr())
        else:
            mine = torch.from_numpy(self.eng.frontend_export_features(last_frame, self.cap, sequence))
        if self.world == 1:
            return (None, mine.reshape(1, self.nfl), mine)
        out = torch.empty(self.world * self.nfl, dtype=torch.float32, device=mine.device)
        work = dist.all_gather_into_tensor(out, mine, async_op=True)
        return (work, out.reshape(self.world, self.nfl), mine)      # `mine` is kept alive until the collective has read it

    def wait(self, ticket):
        """The gathered [world, slab] tensor, complete (the engine reads it from its own stream)."""
        import torch
        work, out, _mine = ticket
        if work is not None:
            work.wait()
            if self.on_gpu:
                torch.cuda.current_stream(out.device).synchronize()
        return out


class NativeExchange:
    """The shard-boundary exchange through the engine's own RCCL communicator (include/dani_frontend.h, "the sharded batch's
    one collective").  Construction is a collective: rank 0 creates the RCCL id, torch.distributed's process group (any
    backend) carries its 128 bytes to the other ranks, every rank calls dani_comm_create.  start() = pack kernel +
    ncclAllGather enqueued on the sequence's stream (returns at once); match() = LightGlue of the resident frames with the
    neighbouring rank's slab as predecessor, ordered behind the collective by that stream."""

    def __init__(self, eng, cap: int, rank: int = 0, world: int = 1):
        import os
        import sys
        import torch.distributed as dist
        self.eng, self.cap, self.rank, self.world = eng, cap, rank, world
        # RCCL prints a version banner on STDOUT when a communicator is initialised; bench.py's contract is ONE JSON line there
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            # Every step below that can fail on ONE rank is followed by an agreement before anybody enters a collective that the
            # failed rank would never join (a rank waiting alone in a broadcast or in ncclCommInitRank hangs the whole job).
            uid, err = None, None
            try:
                if not eng.comm_available():        # every rank: does librccl load here?
                    raise RuntimeError("librccl could not be loaded")
                if rank == 0:
                    uid = eng.comm_unique_id()
            except Exception as ex:                 # noqa: BLE001 -- reported below, on every rank
                err = f"rank {rank}: {ex}"
            if world > 1:
                import torch
                dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
                okf = torch.tensor([0 if err else 1], dtype=torch.int32, device=dev)
                dist.all_reduce(okf, op=dist.ReduceOp.MIN)
                if int(okf.item()) == 0:
                    raise RuntimeError(err or "librccl could not be loaded on another rank")
                box = [uid if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                uid = box[0]
            elif err:
                raise RuntimeError(err)
            eng.comm_create(uid, rank, world)
        finally:
            os.dup2(saved, 1)
            os.close(saved)

    def start(self, last_frame: int, sequence: int = 0):
        return self.eng.frontend_exchange_boundary(last_frame, self.cap, sequence)

    def match(self, nframes: int, sequence: int = 0, lg_threshold: float = 0.05):
        return self.eng.frontend_match_exchanged(nframes, cap=self.cap, lg_threshold=lg_threshold, sequence=sequence)

    def info(self):
        return self.eng.comm_info()

    def close(self):
        self.eng.comm_destroy()
