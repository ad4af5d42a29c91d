"""GPU: the native shard-boundary exchange -- ncclAllGather called by the engine behind the C ABI (csrc/comm_engine.cpp) -- on
a ONE-rank communicator: the test box has one GPU, and RCCL refuses several ranks on one device.  librccl is really loaded, a
communicator really initialised and ncclAllGather really executed on the engine's stream; with world = 1 the gathered slab is
the rank's own slab and rank 0's predecessor is its own previous batch, so extract -> exchange -> match must reproduce the
fused call (the same composition tests/test_frontend_gpu.py checks against the oracle).  The slab layout across ranks is covered
by the world-size-2 gloo test (tests/test_dist_cpu.py); the N > 1 launch is the driver's."""
import numpy as np
import pytest

from dani_slam_amd import synth

pytestmark = pytest.mark.gpu
NF = 300
CAP = NF + 12


def _same(a, b):
    return np.array_equal(a[0], b["matches"]) and np.allclose(a[1], b["scores"], rtol=0, atol=1e-6)


def test_one_rank_communicator_reproduces_the_fused_call():
    import torch
    from dani_slam_amd import frontend, sharding
    w = synth.bench_weights()
    e = frontend.Engine(device=0, max_batch=4, weights=w)
    e.yolo_set_classes(synth.COCO_NAMES, synth.DYNAMIC_NAMES)
    gray, bgr = synth.bench_frames(range(12))
    fused = []
    for c in range(0, 12, 4):
        fused += e.frontend_batch(gray[c:c + 4], bgr[c:c + 4], nfeatures=NF)
    assert min(len(o["matches"]) for o in fused[1:]) >= 20
    with pytest.raises(frontend.DaniError):
        e.frontend_exchange_boundary(3, CAP)                       # no communicator yet
    x = sharding.NativeExchange(e, CAP, rank=0, world=1)           # ncclGetUniqueId + ncclCommInitRank
    info = x.info()
    assert info["rank"] == 0 and info["world"] == 1 and info["rccl_version"] > 20000 and info["collectives"] == 0
    with pytest.raises(frontend.DaniError):
        e.comm_create(e.comm_unique_id(), 0, 1)                    # one communicator per engine

    # ---- lock-step: extract, exchange, match -- three batches; rank 0's predecessor is carried on the device
    e.frontend_reset()
    got = []
    for c in range(0, 12, 4):
        o = e.frontend_batch(gray[c:c + 4], bgr[c:c + 4], nfeatures=NF, match_previous=False)
        assert all(np.array_equal(a["kpts"], b["kpts"]) for a, b in zip(o, fused[c:c + 4]))
        n, ptr = x.start(3)
        assert n == len(fused[c + 3]["kpts"]) and ptr
        got += x.match(4)
    assert all(_same(m, f) for m, f in zip(got, fused)), [len(m[0]) for m in got]
    assert x.info()["collectives"] == 3

    # ---- the gathered bytes: into a caller-owned device buffer == the host export of the same frame (layout of
    # dani_frontend_export_features, unused rows zeroed)
    d_out = torch.full((e.feature_slab_floats(CAP),), -5.0, dtype=torch.float32, device="cuda")
    n, ptr = e.frontend_exchange_boundary(2, CAP, device_ptr=d_out.data_ptr())
    assert ptr == d_out.data_ptr()
    e.frontend_finish_pending()                                    # (nothing pending; the next call synchronises the stream)
    host = e.frontend_export_features(2, CAP)                      # synchronises the sequence's stream behind the collective
    slab = d_out.cpu().numpy()
    assert int(slab[0]) == n == int(host[0]) and np.array_equal(slab, host)
    with pytest.raises(frontend.DaniError):
        x.match(4)                                                 # that exchange went to the caller's buffer: nothing to consume

    # ---- pipelined, two sequences ping-pong (bench.py's N > 1 loop): the LightGlue of batch t runs behind the extraction and
    # the exchange of batch t + 1; the carried slab crosses from one sequence's stream to the other's
    e.comm_reset_carry(); e.frontend_reset(0); e.frontend_reset(1)
    got, pending = [], None
    for t, c in enumerate(range(0, 12, 4)):
        seq = t & 1
        e.frontend_batch(gray[c:c + 4], bgr[c:c + 4], nfeatures=NF, match_previous=False, sequence=seq, defer_desc=True, copy=False)
        x.start(3, seq)
        if pending is not None:
            got += x.match(4, pending)
        pending = seq
    got += x.match(4, pending)
    assert all(_same(m, f) for m, f in zip(got, fused)), [len(m[0]) for m in got]

    # ---- without the carry rank 0's first frame has no predecessor again
    e.comm_reset_carry()
    e.frontend_batch(gray[4:8], bgr[4:8], nfeatures=NF, match_previous=False)
    x.start(3)
    r = x.match(4)
    assert len(r[0][0]) == 0
    # (another batch composition than the fused call's -- an empty set in front -- moves the tokens' tile boundaries: the linear
    # layers' tail tokens sum in another order, scores agree to 1e-4 instead of bit for bit)
    for m, f in zip(r[1:], fused[5:8]):
        a = {tuple(k): v for k, v in zip(m[0].tolist(), m[1])}; b = {tuple(k): v for k, v in zip(f["matches"].tolist(), f["scores"])}
        assert len(set(a) ^ set(b)) <= 2 and max(abs(a[k] - b[k]) for k in set(a) & set(b)) < 1e-4
    x.close()
    with pytest.raises(frontend.DaniError):
        e.frontend_exchange_boundary(3, CAP)
    x2 = sharding.NativeExchange(e, CAP)                            # a new communicator on the same engine
    e.frontend_batch(gray[:4], bgr[:4], nfeatures=NF, match_previous=False)
    x2.start(3)
    assert all(_same(m, f) for m, f in zip(x2.match(4)[1:], fused[1:4]))
    e.close()                                                      # destroys the communicator with the engine
