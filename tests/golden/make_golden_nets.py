#!/usr/bin/env python3
"""Golden vectors for the LightGlue and YOLOv5s oracles, generated with ATen CPU operators (torch 2.10).

Both networks are absent TorchScript blobs in the reference (src/LightGlue.cpp:9, src/YoloDetect.cpp:9), so the
ARCHITECTURE the oracles restate is the published one and stays "parity unpinned".  What this script pins is the
ARITHMETIC: the same architecture is written here a second time with torch ops (F.conv2d, F.silu, F.max_pool2d,
F.interpolate, F.scaled_dot_product_attention, F.layer_norm, F.gelu, log_softmax, logsigmoid), run on the seeded
synthetic weights of dani_slam_amd/synth.py, and the results are committed for tests/test_oracle_golden_nets.py.

Run:  python tests/golden/make_golden_nets.py        (writes tests/golden/lg_g7.npz, yolo_g8.npz)
Nothing under /root/reference is read or executed.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from dani_slam_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)


# ---------------------------------------------------------------- LightGlue (public model, torch ops)
def lg_forward(w, k0, k1, d0, d1):
    t = {k: torch.from_numpy(v) for k, v in w.items()}
    H, D = 4, 64

    def posenc(k):
        proj = k @ t["posenc.Wr.weight"].T
        c, s = torch.cos(proj), torch.sin(proj)
        return torch.stack([c, s], 0).unsqueeze(-3).repeat_interleave(2, dim=-1)     # [2, 1, n, 64]

    def rotate_half(x):
        x = x.unflatten(-1, (-1, 2))
        x1, x2 = x.unbind(dim=-1)
        return torch.stack((-x2, x1), dim=-1).flatten(start_dim=-2)

    def rot(freqs, x):
        return x * freqs[0] + rotate_half(x) * freqs[1]

    def ffn(p, x, m):
        h = F.linear(torch.cat([x, m], -1), t[p + ".ffn.0.weight"], t[p + ".ffn.0.bias"])
        h = F.gelu(F.layer_norm(h, (512,), t[p + ".ffn.1.weight"], t[p + ".ffn.1.bias"]))
        return F.linear(h, t[p + ".ffn.3.weight"], t[p + ".ffn.3.bias"])

    def self_block(i, x, enc):
        p = f"transformers.{i}.self_attn"
        qkv = F.linear(x, t[p + ".Wqkv.weight"], t[p + ".Wqkv.bias"])
        qkv = qkv.unflatten(-1, (H, -1, 3)).transpose(0, 1)                          # [h, n, 64, 3]
        q, k, v = qkv[..., 0], qkv[..., 1], qkv[..., 2]
        q, k = rot(enc, q), rot(enc, k)
        ctx = F.scaled_dot_product_attention(q[None], k[None], v[None])[0]
        m = F.linear(ctx.transpose(0, 1).flatten(start_dim=-2), t[p + ".out_proj.weight"], t[p + ".out_proj.bias"])
        return x + ffn(p, x, m)

    def cross_block(i, x0, x1):
        p = f"transformers.{i}.cross_attn"
        heads = lambda z: z.unflatten(-1, (H, -1)).transpose(0, 1)
        qk0, qk1 = heads(F.linear(x0, t[p + ".to_qk.weight"], t[p + ".to_qk.bias"])), heads(F.linear(x1, t[p + ".to_qk.weight"], t[p + ".to_qk.bias"]))
        v0, v1 = heads(F.linear(x0, t[p + ".to_v.weight"], t[p + ".to_v.bias"])), heads(F.linear(x1, t[p + ".to_v.weight"], t[p + ".to_v.bias"]))
        s = D ** -0.25
        sim = torch.einsum("hid,hjd->hij", qk0 * s, qk1 * s)
        m0 = torch.einsum("hij,hjd->hid", F.softmax(sim, -1), v1)
        m1 = torch.einsum("hji,hjd->hid", F.softmax(sim.transpose(-2, -1).contiguous(), -1).transpose(-2, -1), v0)
        merge = lambda z: z.transpose(0, 1).flatten(start_dim=-2)
        m0 = F.linear(merge(m0), t[p + ".to_out.weight"], t[p + ".to_out.bias"])
        m1 = F.linear(merge(m1), t[p + ".to_out.weight"], t[p + ".to_out.bias"])
        return x0 + ffn(p, x0, m0), x1 + ffn(p, x1, m1)

    x0, x1 = torch.from_numpy(d0), torch.from_numpy(d1)
    e0, e1 = posenc(torch.from_numpy(k0)), posenc(torch.from_numpy(k1))
    e0, e1 = e0[:, 0], e1[:, 0]
    for i in range(9):
        x0, x1 = self_block(i, x0, e0), self_block(i, x1, e1)
        x0, x1 = cross_block(i, x0, x1)
    p = "log_assignment.8"
    md0 = F.linear(x0, t[p + ".final_proj.weight"], t[p + ".final_proj.bias"]) / 256 ** 0.25
    md1 = F.linear(x1, t[p + ".final_proj.weight"], t[p + ".final_proj.bias"]) / 256 ** 0.25
    sim = md0 @ md1.T
    z0 = F.linear(x0, t[p + ".matchability.weight"], t[p + ".matchability.bias"])
    z1 = F.linear(x1, t[p + ".matchability.weight"], t[p + ".matchability.bias"])
    scores = F.log_softmax(sim, 1) + F.log_softmax(sim, 0) + F.logsigmoid(z0) + F.logsigmoid(z1).T
    return x0.numpy(), x1.numpy(), scores.numpy()


# ---------------------------------------------------------------- YOLOv5s v6 (public model, torch ops)
def yolo_forward(w, x):
    t = {k: torch.from_numpy(v) for k, v in w.items()}

    def conv(name, x, k, s, act=True):
        y = F.conv2d(x, t[name + ".weight"], t[name + ".bias"], stride=s, padding={1: 0, 3: 1, 6: 2}[k])
        return F.silu(y) if act else y

    def c3(name, x, n, shortcut):
        y = conv(f"{name}.cv1.conv", x, 1, 1)
        for j in range(n):
            z = conv(f"{name}.m.{j}.cv2.conv", conv(f"{name}.m.{j}.cv1.conv", y, 1, 1), 3, 1)
            y = y + z if shortcut else z
        return conv(f"{name}.cv3.conv", torch.cat([y, conv(f"{name}.cv2.conv", x, 1, 1)], 1), 1, 1)

    x0 = conv("model.0.conv", x, 6, 2); x1 = conv("model.1.conv", x0, 3, 2); x2 = c3("model.2", x1, 1, True)
    x3 = conv("model.3.conv", x2, 3, 2); x4 = c3("model.4", x3, 2, True)
    x5 = conv("model.5.conv", x4, 3, 2); x6 = c3("model.6", x5, 3, True)
    x7 = conv("model.7.conv", x6, 3, 2); x8 = c3("model.8", x7, 1, True)
    y = conv("model.9.cv1.conv", x8, 1, 1)
    y1 = F.max_pool2d(y, 5, 1, 2); y2 = F.max_pool2d(y1, 5, 1, 2); y3 = F.max_pool2d(y2, 5, 1, 2)
    x9 = conv("model.9.cv2.conv", torch.cat([y, y1, y2, y3], 1), 1, 1)
    x10 = conv("model.10.conv", x9, 1, 1)
    up = lambda z: F.interpolate(z, scale_factor=2, mode="nearest")
    x13 = c3("model.13", torch.cat([up(x10), x6], 1), 1, False)
    x14 = conv("model.14.conv", x13, 1, 1)
    x17 = c3("model.17", torch.cat([up(x14), x4], 1), 1, False)
    x20 = c3("model.20", torch.cat([conv("model.18.conv", x17, 3, 2), x14], 1), 1, False)
    x23 = c3("model.23", torch.cat([conv("model.21.conv", x20, 3, 2), x10], 1), 1, False)
    return [F.conv2d(f, t[f"model.24.m.{i}.weight"], t[f"model.24.m.{i}.bias"])[0].numpy() for i, f in enumerate((x17, x20, x23))]


def main():
    with torch.no_grad():
        w = synth.lightglue_weights(0)
        k0, d0, k1, d1, _ = synth.matching_pair(21, n0=96, n1=80)
        from oracle import lg_oracle
        k0n, k1n = lg_oracle.normalize_keypoints(k0, (640, 480)), lg_oracle.normalize_keypoints(k1, (640, 480))
        x0, x1, scores = lg_forward(w, k0n, k1n, d0, d1)
        np.savez_compressed(os.path.join(HERE, "lg_g7.npz"), k0n=k0n, k1n=k1n, d0=d0, d1=d1, x0=x0, x1=x1, scores=scores)

        yw = synth.yolo_weights(0)
        from oracle import yolo_oracle
        x = yolo_oracle.preprocess(synth.frame_bgr(1))
        raws = yolo_forward(yw, torch.from_numpy(x)[None])
        np.savez_compressed(os.path.join(HERE, "yolo_g8.npz"), raw0=raws[0][:, ::3, ::3], raw1=raws[1][:, ::2, ::2], raw2=raws[2])
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
