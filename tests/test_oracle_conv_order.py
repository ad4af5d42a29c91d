"""CPU: the canonical evaluation order of the general oracle convolution (oracle/sp_oracle.cpp orc_conv2d_ex, DESIGN.md
section 3), pinned against an independent scalar restatement: one fmaf chain per K segment in (channel pair, ky, kx, channel)
order, the bias opening segment 0, the segment sums added left to right.  The segment rule is a function of the layer's shape:
3x3 stride 1: 4 segments from 256 input channels, 2 from 128; 3x3 stride 2: 2 from 128; 1x1: 2 from 512; otherwise 1.  The HIP
kernels are held to the same oracle bit for bit on the GPU (tests/test_yolo_gpu.py)."""
import math

import numpy as np
import pytest

from oracle import oracle as O


def _f32(x):
    return np.float32(x)


def _fmaf(a, b, c):
    # exact for f32 operands: the product of two f32 fits a double, the sum is rounded once to f64 and once to f32; a double
    # rounding could differ from a true fma only in a tie that 48-bit products plus a 24-bit addend cannot produce here
    return _f32(np.float64(a) * np.float64(b) + np.float64(c))


def _kseg(cin, k, stride):
    if k == 3 and stride == 1:
        return 4 if cin >= 256 else (2 if cin >= 128 else 1)
    if k == 3:
        return 2 if cin >= 128 else 1
    if k == 1:
        return 2 if cin >= 512 else 1
    return 1


def _ref_point(x, w, b, co, y, xx, k, stride, pad):
    cin = x.shape[0]
    kseg = _kseg(cin, k, stride)
    ncp = (cin + 1) // 2
    if ncp % kseg:
        kseg = 1
    seg = ncp // kseg
    total = None
    for sg in range(kseg):
        acc = _f32(b[co]) if sg == 0 else _f32(0.0)
        for cp in range(sg * seg, (sg + 1) * seg):
            for ky in range(k):
                for kx in range(k):
                    for cc in range(2):
                        c = 2 * cp + cc
                        if c >= cin:
                            continue
                        iy, ix = y * stride + ky - pad, xx * stride + kx - pad
                        v = x[c, iy, ix] if (0 <= iy < x.shape[1] and 0 <= ix < x.shape[2]) else _f32(0.0)
                        acc = _fmaf(v, w[co, c, ky, kx], acc)
        total = acc if total is None else _f32(total + acc)
    return total


@pytest.mark.parametrize("cin,k,stride", [(64, 3, 1), (128, 3, 1), (256, 3, 1), (128, 3, 2), (256, 3, 2), (512, 1, 1), (256, 1, 1), (1024, 1, 1), (3, 6, 2)])
def test_general_conv_follows_the_segmented_canonical_order(cin, k, stride):
    rng = np.random.RandomState(cin * 10 + k + stride)
    h, wd, cout = 7, 9, 3
    x = rng.randn(cin, h, wd).astype(np.float32)
    w = (rng.randn(cout, cin, k, k) / math.sqrt(cin * k * k)).astype(np.float32)
    b = rng.randn(cout).astype(np.float32)
    pad = {1: 0, 3: 1, 6: 2}[k]
    out = O.conv2d_ex(x, w, b, stride, pad, 0)
    pts = [(0, 0, 0), (1, out.shape[1] - 1, out.shape[2] - 1), (2, out.shape[1] // 2, out.shape[2] // 2), (0, 1, out.shape[2] - 1)]
    for co, y, xx in pts:
        want = _ref_point(x, w, b, co, y, xx, k, stride, pad)
        got = out[co, y, xx]
        assert got.tobytes() == np.float32(want).tobytes(), (cin, k, stride, co, y, xx, float(got), float(want))
