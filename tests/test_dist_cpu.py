"""CPU, world_size 2, gloo: the N>1 path of bench.py (frame sharding, max-over-ranks timing) and the feature
all-gather used for the multi-camera LightGlue case."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dani_slam_amd import sharding


def test_shard_frames_partition():
    for n, w in [(64, 8), (64, 1), (10, 4), (3, 8), (17, 2)]:
        got = []
        for r in range(w):
            got += list(sharding.shard_frames(n, r, w))
        assert got == list(range(n))
        sizes = [len(sharding.shard_frames(n, r, w)) for r in range(w)]
        assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.RandomState(100 + rank)
        n = 5 + 3 * rank
        k = rng.rand(n, 2).astype(np.float32); r = rng.rand(n).astype(np.float32); d = rng.rand(n, 256).astype(np.float32)
        allf = sharding.allgather_features(k, r, d, cap=16)
        ok = len(allf) == world
        for rr in range(world):
            g = np.random.RandomState(100 + rr)
            nn = 5 + 3 * rr
            ek = g.rand(nn, 2).astype(np.float32); er = g.rand(nn).astype(np.float32); ed = g.rand(nn, 256).astype(np.float32)
            ok = ok and np.array_equal(allf[rr][0], ek) and np.array_equal(allf[rr][1], er) and np.array_equal(allf[rr][2], ed)
        # the shard-boundary exchange of bench.py (N > 1): every rank packs its last frame, one all-gather, rank r reads r-1
        class FakeEngine:   # stands in for frontend.Engine (no GPU here): same slab layout as dani_frontend_export_features
            @staticmethod
            def feature_slab_floats(cap): return 1 + cap * 258
            def frontend_export_features(self, frame, cap, sequence=0, device_ptr=None):
                assert device_ptr is None and frame == 7
                slab = np.zeros(self.feature_slab_floats(cap), np.float32)
                slab[0] = n; slab[1:1 + 2 * n] = k.reshape(-1); slab[1 + 2 * cap:1 + 2 * cap + 256 * n] = d.reshape(-1)
                return slab
        slabs = sharding.exchange_boundary_features(FakeEngine(), 7, 16, None)
        ok = ok and tuple(slabs.shape) == (world, 1 + 16 * 258)
        for rr in range(world):
            g = np.random.RandomState(100 + rr)
            nn = 5 + 3 * rr
            ek = g.rand(nn, 2).astype(np.float32); g.rand(nn); ed = g.rand(nn, 256).astype(np.float32)
            row = slabs[rr].numpy()
            ok = ok and int(row[0]) == nn and np.array_equal(row[1:1 + 2 * nn].reshape(nn, 2), ek)
            ok = ok and np.array_equal(row[1 + 32:1 + 32 + 256 * nn].reshape(nn, 256), ed)
        # the asynchronous form bench.py uses by default: two collectives posted back to back, waited for in order
        x = sharding.BoundaryExchange(FakeEngine(), 16, None)
        t0 = x.start(7, 0); t1 = x.start(7, 1)
        for t in (t0, t1):
            again = x.wait(t)
            ok = ok and tuple(again.shape) == (world, 1 + 16 * 258) and bool((again == slabs).all())
        mx = sharding.reduce_max_seconds(1.0 + rank)
        frames = list(sharding.shard_frames(16, rank, world))
        q.put((rank, ok, mx, frames))
    finally:
        dist.destroy_process_group()


def test_world2_gloo_allgather_and_max():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps: p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in ps: p.join(timeout=60)
    res.sort()
    assert all(r[1] for r in res)
    assert all(abs(r[2] - 2.0) < 1e-12 for r in res)                 # max over ranks of (1.0, 2.0)
    assert res[0][3] + res[1][3] == list(range(16))


def _native_worker(rank, world, port, q, broken_rank):
    """NativeExchange's construction with a stand-in engine: on `broken_rank` librccl "does not load"."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = []

        class FakeEngine:
            def comm_available(self): return rank != broken_rank
            def comm_unique_id(self): calls.append("id"); return bytes(range(128))
            def comm_create(self, uid, r, w): calls.append(("create", bytes(uid), r, w))
        try:
            sharding.NativeExchange(FakeEngine(), 16, rank, world)
            outcome = "built"
        except RuntimeError as ex:
            outcome = "refused: " + str(ex)
        q.put((rank, outcome, calls))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("broken_rank", [-1, 0, 1])
def test_native_exchange_construction_never_leaves_a_rank_alone_in_a_collective(broken_rank):
    """World size 2 over gloo: when librccl loads everywhere, rank 0's id reaches rank 1 and both call comm_create with it; when it
    does not load on ONE rank (either one), BOTH ranks give up before the id broadcast / ncclCommInitRank -- nobody hangs."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_native_worker, args=(r, world, port, q, broken_rank)) for r in range(world)]
    for p in ps: p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps: p.join(timeout=60)
    if broken_rank < 0:
        assert [r[1] for r in res] == ["built", "built"]
        assert res[0][2] == ["id", ("create", bytes(range(128)), 0, 2)] and res[1][2] == [("create", bytes(range(128)), 1, 2)]
    else:
        assert all(r[1].startswith("refused") for r in res), res
        assert all(not any(isinstance(c, tuple) for c in r[2]) for r in res)      # comm_create was never reached
