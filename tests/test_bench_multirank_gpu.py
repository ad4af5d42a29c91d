"""GPU (one card): the N > 1 code path of bench.py end to end -- two ranks launched exactly as the driver launches them
(python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2), both on GPU 0 with a gloo group
(DANI_BENCH_BACKEND=gloo DANI_BENCH_ONE_DEVICE=1: a rehearsal of the multi-GPU path on the one-GPU test box, not a
measurement).  Checks the contract line and that the shard-boundary pair is matched across ranks."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("extra", [[], ["--no-pipelined-exchange"]], ids=["pipelined", "lock-step"])
def test_two_ranks_one_card_boundary_exchange(extra):
    """default: the LightGlue of step t runs behind the extraction of step t + 1 (all-gather posted in between, two engine
    sequences); --no-pipelined-exchange: extract, all-gather, match inside one step.  Same matches either way."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, DANI_BENCH_BACKEND="gloo", DANI_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"] + extra
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak" and d["value"] > 0 and d["unit"] == "frames/s"
    assert "all-gather" in d["config"]["parallelism"]
    fm = d["config"]["first_frame_matches_per_rank"]
    # rank 1's first frame (global frame 8) is matched against rank 0's last frame (global frame 7): consecutive frames of
    # the trajectory, ~100 matches; rank 0's first frame meets the LAST rank's last frame of the previous step (a jump)
    assert len(fm) == 2 and fm[1] >= 40, fm
    assert min(d["config"]["matches_per_frame"][1:]) >= 40
