"""oracle/lg_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU (numpy, float32) restatement of the reference's LightGlue matcher entry point
    LightGlue::matchDescriptors      /root/reference/src/LightGlue.cpp:56-94
    keypointsToTensor                src/LightGlue.cpp:18-26
    normalizeKeypoints               src/LightGlue.cpp:28-51
    filterMatches                    src/LightGlue.cpp:98-122
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

PARITY UNPINNED.  The network the reference runs is the TorchScript blob /sly_slam/lightglue_scripted_v2.pt
(src/LightGlue.cpp:9), which is not in the repository and whose upstream commit is not recorded; the
reference holds no test or golden vector at this boundary.  The third-party algorithm is
therefore restated from the published LightGlue model (Lindenberger et al., ICCV 2023, SuperPoint variant:
descriptor_dim 256, 9 layers, 4 heads x 64, learnable-Fourier rotary positional encoding of (x, y), per layer a
self block and a cross block each with a [512->512, LayerNorm, GELU, 512->256] FFN, MatchAssignment with
final_proj / matchability, sigmoid-log-double-softmax, mutual-nearest-neighbour filter at 0.1; no early exit, no
point pruning -- the exported "end-to-end" forward(kpts0, kpts1, desc0, desc1) -> (matches[K,2], scores[K])
convention the reference's call site uses).  Weight names follow the public checkpoint
(posenc.Wr.weight, transformers.{i}.self_attn.Wqkv.weight, ..., log_assignment.{i}.final_proj.weight).
What IS pinned is the arithmetic of this restatement: tests/golden/make_golden_nets.py writes the same architecture a
second time with ATen operators (scaled_dot_product_attention, layer_norm, gelu, log_softmax, logsigmoid) and
tests/test_oracle_golden_nets.py holds this module to the committed vectors (tests/golden/lg_g7.npz).
"""
from __future__ import annotations

import math

import numpy as np

N_LAYERS = 9
N_HEADS = 4
DIM = 256
HEAD_DIM = 64
FILTER_THRESHOLD = 0.1

f32 = np.float32


def normalize_keypoints(kpts, size):
    """src/LightGlue.cpp:28-51.  kpts [n,2] (x,y) f32; size (w,h) or None/empty -> 1 + max - min."""
    kpts = np.asarray(kpts, f32).reshape(-1, 2)
    if size is None or len(size) == 0:
        size = f32(1) + (kpts.max(0) - kpts.min(0))
    size = np.asarray(size, f32)
    shift = size / f32(2)
    scale = size.max() / f32(2)
    return ((kpts - shift[None]) / scale).astype(f32)


def _linear(x, w, b=None):
    y = x @ w.T
    if b is not None:
        y = y + b
    return y.astype(f32)


def _layernorm(x, g, b, eps=1e-5):
    mu = x.mean(-1, keepdims=True, dtype=f32)
    var = ((x - mu) ** 2).mean(-1, keepdims=True, dtype=f32)
    return ((x - mu) / np.sqrt(var + f32(eps)) * g + b).astype(f32)


def _gelu(x):
    from scipy.special import erf                   # exact (erf) GELU, nn.GELU() default
    return (f32(0.5) * x * (f32(1) + erf(x.astype(np.float64) / math.sqrt(2.0)).astype(f32))).astype(f32)


def _softmax(s, axis=-1):
    m = s.max(axis, keepdims=True)
    e = np.exp(s - m, dtype=f32)
    return (e / e.sum(axis, keepdims=True, dtype=f32)).astype(f32)


def _log_softmax(s, axis):
    m = s.max(axis, keepdims=True)
    z = s - m
    return (z - np.log(np.exp(z, dtype=f32).sum(axis, keepdims=True, dtype=f32))).astype(f32)


def _logsigmoid(x):
    # log(1/(1+exp(-x))) computed stably
    return (-(np.maximum(-x, 0) + np.log1p(np.exp(-np.abs(x), dtype=f32), dtype=f32))).astype(f32)


def posenc(kpts_n, wr):
    """LearnableFourierPositionalEncoding(M=2, dim=64): returns (cos, sin) each [n, 64], pairs repeated."""
    proj = _linear(kpts_n, wr)                      # [n, 32]
    c, s = np.cos(proj, dtype=f32), np.sin(proj, dtype=f32)
    return np.repeat(c, 2, axis=-1), np.repeat(s, 2, axis=-1)


def _rotate_half(x):
    x2 = x.reshape(*x.shape[:-1], -1, 2)
    return np.stack((-x2[..., 1], x2[..., 0]), -1).reshape(x.shape)


def _rotary(enc, t):                                # t [h, n, 64], enc (cos, sin) [n, 64]
    return (t * enc[0][None] + _rotate_half(t) * enc[1][None]).astype(f32)


def _ffn(w, prefix, x, msg):
    h = _linear(np.concatenate([x, msg], -1), w[prefix + ".ffn.0.weight"], w[prefix + ".ffn.0.bias"])
    h = _gelu(_layernorm(h, w[prefix + ".ffn.1.weight"], w[prefix + ".ffn.1.bias"]))
    return _linear(h, w[prefix + ".ffn.3.weight"], w[prefix + ".ffn.3.bias"])


def self_block(w, i, x, enc):
    p = f"transformers.{i}.self_attn"
    n = x.shape[0]
    qkv = _linear(x, w[p + ".Wqkv.weight"], w[p + ".Wqkv.bias"])            # [n, 768]
    qkv = qkv.reshape(n, N_HEADS, HEAD_DIM, 3).transpose(1, 0, 2, 3)         # unflatten(-1,(heads,-1,3)).transpose(1,2)
    q, k, v = qkv[..., 0], qkv[..., 1], qkv[..., 2]                          # [h, n, 64]
    q, k = _rotary(enc, q), _rotary(enc, k)
    s = (q @ k.transpose(0, 2, 1)) * f32(HEAD_DIM ** -0.5)
    ctx = _softmax(s, -1) @ v                                                # [h, n, 64]
    msg = _linear(ctx.transpose(1, 0, 2).reshape(n, DIM), w[p + ".out_proj.weight"], w[p + ".out_proj.bias"])
    return (x + _ffn(w, p, x, msg)).astype(f32)


def cross_block(w, i, x0, x1):
    p = f"transformers.{i}.cross_attn"
    def heads(t):
        return t.reshape(t.shape[0], N_HEADS, HEAD_DIM).transpose(1, 0, 2)
    qk0 = heads(_linear(x0, w[p + ".to_qk.weight"], w[p + ".to_qk.bias"]))
    qk1 = heads(_linear(x1, w[p + ".to_qk.weight"], w[p + ".to_qk.bias"]))
    v0 = heads(_linear(x0, w[p + ".to_v.weight"], w[p + ".to_v.bias"]))
    v1 = heads(_linear(x1, w[p + ".to_v.weight"], w[p + ".to_v.bias"]))
    sc = f32((HEAD_DIM ** -0.5) ** 0.5)
    qk0, qk1 = qk0 * sc, qk1 * sc
    sim = qk0 @ qk1.transpose(0, 2, 1)                                       # [h, n0, n1]
    m0 = _softmax(sim, -1) @ v1
    m1 = _softmax(sim.transpose(0, 2, 1), -1) @ v0
    def merge(t):
        return t.transpose(1, 0, 2).reshape(t.shape[1], DIM)
    m0 = _linear(merge(m0), w[p + ".to_out.weight"], w[p + ".to_out.bias"])
    m1 = _linear(merge(m1), w[p + ".to_out.weight"], w[p + ".to_out.bias"])
    return (x0 + _ffn(w, p, x0, m0)).astype(f32), (x1 + _ffn(w, p, x1, m1)).astype(f32)


def assignment(w, i, d0, d1):
    p = f"log_assignment.{i}"
    md0 = _linear(d0, w[p + ".final_proj.weight"], w[p + ".final_proj.bias"]) / f32(DIM ** 0.25)
    md1 = _linear(d1, w[p + ".final_proj.weight"], w[p + ".final_proj.bias"]) / f32(DIM ** 0.25)
    sim = (md0 @ md1.T).astype(f32)
    z0 = _linear(d0, w[p + ".matchability.weight"], w[p + ".matchability.bias"])   # [n0, 1]
    z1 = _linear(d1, w[p + ".matchability.weight"], w[p + ".matchability.bias"])
    cert = _logsigmoid(z0) + _logsigmoid(z1).T
    scores = _log_softmax(sim, 1) + _log_softmax(sim, 0) + cert
    return scores.astype(f32), sim


def filter_matches(scores, th):
    """filter_matches of the public model on the [n0, n1] block of the assignment matrix."""
    n0, n1 = scores.shape
    if n0 == 0 or n1 == 0:
        return np.zeros((0, 2), np.int64), np.zeros(0, f32)
    m0 = scores.argmax(1); m1 = scores.argmax(0)          # first maximum on ties (torch.max / np.argmax)
    max0 = scores.max(1)
    mutual0 = np.arange(n0) == m1[m0]
    ms0 = np.where(mutual0, np.exp(max0, dtype=f32), f32(0))
    valid0 = mutual0 & (ms0 > f32(th))
    idx0 = np.nonzero(valid0)[0]
    matches = np.stack([idx0, m0[idx0]], -1).astype(np.int64)
    return matches, ms0[idx0].astype(f32)


def forward(w, kpts0_n, kpts1_n, desc0, desc1, want_intermediates=False):
    """The blob's forward(kpts0, kpts1, desc0, desc1) with already-normalised keypoints."""
    d0 = np.asarray(desc0, f32); d1 = np.asarray(desc1, f32)
    e0 = posenc(kpts0_n, w["posenc.Wr.weight"]); e1 = posenc(kpts1_n, w["posenc.Wr.weight"])
    inter = []
    for i in range(N_LAYERS):
        d0 = self_block(w, i, d0, e0); d1 = self_block(w, i, d1, e1)
        if want_intermediates:
            inter.append(("self", i, d0.copy(), d1.copy()))
        d0, d1 = cross_block(w, i, d0, d1)
        if want_intermediates:
            inter.append(("cross", i, d0.copy(), d1.copy()))
    scores, sim = assignment(w, N_LAYERS - 1, d0, d1)
    matches, mscores = filter_matches(scores, FILTER_THRESHOLD)
    if want_intermediates:
        return matches, mscores, dict(layers=inter, scores=scores, sim=sim, d0=d0, d1=d1)
    return matches, mscores


def match_descriptors(w, kpts0, desc0, kpts1, desc1, size0, size1, threshold=0.05):
    """LightGlue::matchDescriptors + filterMatches (src/LightGlue.cpp:56-122).
    Returns (matches [K,2] int32 (queryIdx, trainIdx), scores [K] f32 = DMatch.distance)."""
    kpts0 = np.asarray(kpts0, f32).reshape(-1, 2); kpts1 = np.asarray(kpts1, f32).reshape(-1, 2)
    if len(kpts0) == 0 or len(kpts1) == 0:
        return np.zeros((0, 2), np.int32), np.zeros(0, f32)
    k0 = normalize_keypoints(kpts0, size0); k1 = normalize_keypoints(kpts1, size1)
    matches, scores = forward(w, k0, k1, desc0, desc1)
    keep = scores > f32(threshold)                              # filterMatches, :113
    return matches[keep].astype(np.int32), scores[keep]
