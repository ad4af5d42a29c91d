"""CPU: the product's host logic (libdani_frontend.so, no GPU call) against the oracle, and the ABI surface."""
import ctypes
import os
import re

import numpy as np
import pytest

from dani_slam_amd import frontend
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "dani_frontend.h")).read()
    names = re.findall(r"DANI_API\s+[\w\s\*]+?\b(dani_\w+)\s*\(", hdr)
    assert len(names) >= 15
    lib = ctypes.CDLL(frontend.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/dani_frontend.h but not exported"
    assert b"gfx950" in frontend.load_library().dani_version()


def test_cell_grid_640x480():
    cells = frontend.host_cell_grid(640, 480)
    assert len(cells) == 204                      # 12 x 17, SURVEY.md section 8 notation
    w = cells[:, 1] - cells[:, 0]; h = cells[:, 3] - cells[:, 2]
    assert w.max() == 42 and h.max() == 44
    assert cells[0].tolist() == [16, 58, 16, 60, 0, 0]
    assert cells[:, 1].max() == 624 and cells[:, 3].max() == 464
    # the oracle walks the same cells: candidates of a flat prob map above threshold cover exactly these cells
    for (W, H) in [(640, 480), (320, 240), (752, 480), (1280, 720)]:
        c = frontend.host_cell_grid(W, H)
        assert len(c) > 0 and (c[:, 1] - c[:, 0]).max() <= 64 and (c[:, 3] - c[:, 2]).max() <= 64


def test_features_per_level_matches_oracle():
    for nf, sf, nl in [(1024, 1.2, 1), (1000, 1.2, 8), (2000, 1.2, 8), (1250, 1.1, 4)]:
        a, sa = frontend.host_features_per_level(nf, sf, nl)
        b, sb = oracle.features_per_level(nf, sf, nl)
        assert a.tolist() == b.tolist() and np.array_equal(sa, sb)
        assert a.sum() >= nf - nl


def _random_candidates(rng, n, w=608, h=448, clustered=False):
    if clustered:
        cx = rng.randint(0, w, 12); cy = rng.randint(0, h, 12)
        k = rng.randint(0, 12, n)
        x = np.clip(cx[k] + rng.randint(-25, 26, n), 0, w - 1)
        y = np.clip(cy[k] + rng.randint(-25, 26, n), 0, h - 1)
    else:
        x = rng.randint(0, w, n); y = rng.randint(0, h, n)
    r = rng.uniform(0.007, 1.0, n)
    if n > 10:   # duplicates (overlap strips emit a pixel twice, quirk Q5) and response ties
        x[5] = x[3]; y[5] = y[3]; r[7] = r[2]
    return np.stack([x, y, r], 1).astype(np.float32)


@pytest.mark.parametrize("n,N,clustered", [(0, 100, False), (1, 100, False), (2, 1, False), (7, 3, False),
                                           (500, 1024, False), (3000, 1024, False), (3000, 1024, True),
                                           (9000, 1024, False), (9000, 2000, True), (20000, 1000, False),
                                           (1500, 1500, True), (4000, 300, False)])
def test_quadtree_matches_oracle(n, N, clustered):
    rng = np.random.RandomState(n + N)
    for rep in range(3):
        xyr = _random_candidates(rng, n, clustered=clustered)
        ref = oracle.distribute_octtree(xyr, 16, 624, 16, 464, N)
        idx = frontend.host_distribute_octtree(xyr, 16, 624, 16, 464, N)
        got = xyr[idx] if len(idx) else np.zeros((0, 3), np.float32)
        assert got.shape == ref.shape
        assert np.array_equal(got, ref)            # same points, same std::list order
        if n:
            assert len(idx) <= max(N + 3, 1) or len(idx) <= n


def test_quadtree_other_aspect_ratios():
    rng = np.random.RandomState(5)
    for (minX, maxX, minY, maxY) in [(16, 736, 16, 464), (16, 1264, 16, 704), (16, 304, 16, 224)]:
        w, h = maxX - minX, maxY - minY
        xyr = _random_candidates(rng, 2500, w=w, h=h)
        ref = oracle.distribute_octtree(xyr, minX, maxX, minY, maxY, 800)
        idx = frontend.host_distribute_octtree(xyr, minX, maxX, minY, maxY, 800)
        assert np.array_equal(xyr[idx], ref)


def test_quadtree_refuses_what_the_reference_leaves_undefined():
    """nIni = round(width / height) = 0 for portrait regions: the reference indexes an empty vector
    (src/SPextractor.cpp:410-433).  The product returns DANI_ERR_UNSUPPORTED instead of walking it; the same for a key
    outside [0, maxX - minX)."""
    rng = np.random.RandomState(3)
    xyr = np.stack([rng.randint(0, 68, 40), rng.randint(0, 368, 40), rng.uniform(0.01, 1, 40)], 1).astype(np.float32)
    for box in [(16, 84, 16, 384), (16, 144, 16, 464)]:                       # 100x400 and 160x480 images
        with pytest.raises(frontend.DaniError) as ei:
            frontend.host_distribute_octtree(xyr, *box, 100)
        assert ei.value.code == frontend.DANI_ERR_UNSUPPORTED
    for bad in ([[608.0, 5.0, 0.5]], [[-1.0, 5.0, 0.5]], [[float("nan"), 5.0, 0.5]]):
        with pytest.raises(frontend.DaniError) as ei:
            frontend.host_distribute_octtree(np.array(bad, np.float32), 16, 624, 16, 464, 100)
        assert ei.value.code == frontend.DANI_ERR_UNSUPPORTED
    assert len(frontend.host_distribute_octtree(np.array([[607.0, 5.0, 0.5]], np.float32), 16, 624, 16, 464, 100)) == 1


def test_malformed_weight_file_is_an_error_code(tmp_path):
    """dani_load_weights_file: a name length or tensor count that would size a huge allocation is rejected before
    anything is allocated; nothing is thrown across the C ABI.  (Needs an engine handle, so only the NULL-engine
    argument check runs without a GPU.)"""
    lib = frontend.load_library()
    lib.dani_load_weights_file.restype = ctypes.c_int
    assert lib.dani_load_weights_file(None, b"/nonexistent") == frontend.DANI_ERR_INVALID


def test_process_wide_options_validate_their_arguments():
    """dani_lg_set_kernel_option / dani_set_host_wait touch no device: unknown names and out-of-range values are error codes,
    the documented values are accepted (and put back)."""
    lib = frontend.load_library()
    lib.dani_lg_set_kernel_option.restype = ctypes.c_int
    lib.dani_lg_set_kernel_option.argtypes = [ctypes.c_char_p, ctypes.c_int32]
    lib.dani_set_host_wait.restype = ctypes.c_int
    lib.dani_set_host_wait.argtypes = [ctypes.c_int32]
    assert lib.dani_lg_set_kernel_option(None, 0) == frontend.DANI_ERR_INVALID
    assert lib.dani_lg_set_kernel_option(b"no_such_option", 0) == frontend.DANI_ERR_INVALID
    for name, good, bad in ((b"lin_rs", (-1, 0, 1, 2, 3, 4), (-2, 5)), (b"attn", (0, 2, 6, 7, 8), (-1, 1, 3, 4, 5, 9)), (b"ffn_chain", (-1, 0, 1), (-2, 2))):
        for v in bad:
            assert lib.dani_lg_set_kernel_option(name, v) == frontend.DANI_ERR_INVALID, (name, v)
        for v in good:
            assert lib.dani_lg_set_kernel_option(name, v) == 0, (name, v)
    assert lib.dani_lg_set_kernel_option(b"lin_rs", -1) == 0 and lib.dani_lg_set_kernel_option(b"attn", 0) == 0 and lib.dani_lg_set_kernel_option(b"ffn_chain", -1) == 0   # the defaults
    assert lib.dani_set_host_wait(7) == frontend.DANI_ERR_INVALID
    assert lib.dani_set_host_wait(-1) == frontend.DANI_ERR_INVALID and lib.dani_set_host_wait(3) == frontend.DANI_ERR_INVALID
    assert lib.dani_set_host_wait(0) == 0 and lib.dani_get_host_wait() == 0 and lib.dani_set_host_wait(2) == 0 and lib.dani_get_host_wait() == 2
    assert lib.dani_set_host_wait(1) == 0 and lib.dani_get_host_wait() == 1   # block, hybrid, spin (the default)


def test_no_gpu_means_loud_failure_not_fallback():
    """On a machine without an MI355X the engine must refuse to exist (there is no CPU compute path)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    with pytest.raises(frontend.DaniError) as ei:
        frontend.Engine(device=0)
    assert ei.value.code == frontend.DANI_ERR_NO_DEVICE
    assert "no CPU fallback" in str(ei.value) or "HIP device" in str(ei.value)


def test_product_never_imports_the_oracle():
    import ast
    pkg = os.path.join(ROOT, "dani_slam_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            tree = ast.parse(open(os.path.join(pkg, fn)).read())
            for node in ast.walk(tree):
                names = []
                if isinstance(node, ast.Import):
                    names = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom):
                    names = [node.module or ""]
                assert not any(n.split(".")[0] == "oracle" for n in names), f"{fn} imports the oracle"
    for fn in os.listdir(os.path.join(pkg, "csrc")):
        if not os.path.isfile(os.path.join(pkg, "csrc", fn)):
            continue
        src = open(os.path.join(pkg, "csrc", fn)).read()
        assert "#include \"../../oracle" not in src and "liboracle" not in src


def test_shim_header_compiles_as_cxx17(tmp_path):
    """include/dani_shims.hpp (the reference-facing C++ classes) must at least compile here: against the mock OpenCV
    core types of tests/cpp/mock_opencv, since the real library is absent (run end to end in tests/test_shims_gpu.py)."""
    import shutil
    import subprocess
    cxx = shutil.which("g++")
    if not cxx:
        pytest.skip("no g++")
    subprocess.check_call([cxx, "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "tests/cpp/mock_opencv"),
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests/cpp/shim_smoke.cpp")])
