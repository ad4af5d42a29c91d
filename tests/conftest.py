import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_first(request):
    """GPU runs only: tests/test_bench_shape_gpu.py keeps its frames in HBM through torch, as bench.py does.  torch ships
    its own HIP runtime; when the engine's library has touched the GPU first, torch's later initialisation reports "No HIP
    GPUs are available" (seen when that test ran after tests/test_frontend_gpu.py).  Initialising torch's runtime once,
    before any engine exists, makes the order of the test files irrelevant -- the order bench.py uses."""
    expr = request.config.getoption("-m") or ""
    if "gpu" in expr and "not gpu" not in expr:
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
                torch.zeros(1, device="cuda")
        except Exception:
            pass
    yield


@pytest.fixture(scope="session")
def sp_weights():
    from dani_slam_amd import synth
    return synth.superpoint_weights(0)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
