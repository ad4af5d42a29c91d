"""ATen-CPU restatement of the reference's LibTorch op sequence (TEST INFRASTRUCTURE and CPU BASELINE only).

The shipped reference runs its whole front-end on LibTorch-CPU (src/SPextractor.cpp:230 `detect(img, false)`,
src/LightGlue.cpp:10, src/YoloDetect.cpp:9).  Its sources cannot be compiled here (every file of the path includes
OpenCV), but the third-party library that does the arithmetic -- torch -- is importable, so this module replays the
SAME operator sequence with the SAME ATen CPU kernels:
  * sp_forward / sp_compute_descriptors   SuperPoint::forward, SPDetector::computeDescriptors
                                          (/root/reference/src/SuperPoint.cpp:56-99, :246-274)
  * lg_forward                            the public LightGlue architecture (the reference's network is the absent
                                          TorchScript blob src/LightGlue.cpp:9: architecture parity unpinned)
  * yolo_forward / yolo_decode            the public YOLOv5s v6 architecture (absent blob src/YoloDetect.cpp:9) and the
                                          Detect layer's decode that lives inside that blob
It is used by (1) tests/golden/make_golden_*.py to generate the committed golden vectors that pin oracle/ and the HIP
path, and (2) bench.py's `cpu_baseline` leg: frontend_frame() below is the reference-faithful per-frame CPU path --
ATen operators for the three networks, the literal restatement of the reference's host logic (oracle/sp_oracle.cpp,
oracle/yolo_oracle.py, oracle/lg_oracle.py) for the per-cell loop / NMS2 / quadtree / box logic, as BASELINE.md
section 4 defines the baseline.  Nothing under dani_slam_amd/ may import this module.
"""
from __future__ import annotations

import time

import numpy as np
import torch
import torch.nn.functional as F


# ---------------------------------------------------------------- SuperPoint (src/SuperPoint.cpp:56-99, :246-274)
def sp_forward(x, w, want_taps=True):
    """SuperPoint::forward.  x [1,1,H,W] f32 tensor; returns (prob [H,W], desc [256,Hc,Wc], taps) as numpy."""
    t = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(v)) for k, v in w.items() if k.startswith("conv")}
    taps = []

    def conv(name, x, relu, pad):
        y = F.conv2d(x, t[name + ".weight"], t[name + ".bias"], stride=1, padding=pad)
        if relu:
            y = torch.relu(y)
        if want_taps:
            taps.append(y[0].numpy().copy())
        return y

    x = conv("conv1a", x, True, 1); x = conv("conv1b", x, True, 1); x = F.max_pool2d(x, 2, 2)
    x = conv("conv2a", x, True, 1); x = conv("conv2b", x, True, 1); x = F.max_pool2d(x, 2, 2)
    x = conv("conv3a", x, True, 1); x = conv("conv3b", x, True, 1); x = F.max_pool2d(x, 2, 2)
    x = conv("conv4a", x, True, 1); x = conv("conv4b", x, True, 1)
    cPa = conv("convPa", x, True, 1); semi = conv("convPb", cPa, False, 0)
    cDa = conv("convDa", x, True, 1); desc = conv("convDb", cDa, False, 0)
    dn = torch.norm(desc, 2, 1)
    desc = desc.div(torch.unsqueeze(dn, 1))
    semi = torch.softmax(semi, 1)
    semi = semi[:, 0:64]
    semi = semi.permute(0, 2, 3, 1)
    Hc, Wc = semi.size(1), semi.size(2)
    semi = semi.contiguous().view(-1, Hc, Wc, 8, 8)
    semi = semi.permute(0, 1, 3, 2, 4)
    semi = semi.contiguous().view(-1, Hc * 8, Wc * 8)
    return semi[0].numpy(), desc[0].numpy(), taps


def sp_compute_descriptors(desc_map, kxy, PH, PW):
    """SPDetector::computeDescriptors, SuperPoint.cpp:246-274 (kxy = n x (x, y))."""
    mDesc = torch.from_numpy(np.ascontiguousarray(desc_map))[None]
    fk = torch.from_numpy(np.ascontiguousarray(kxy, np.float32))
    grid = torch.zeros(1, 1, fk.size(0), 2)
    grid[0, 0, :, 0] = 2.0 * fk[:, 0] / PW - 1
    grid[0, 0, :, 1] = 2.0 * fk[:, 1] / PH - 1
    d = F.grid_sample(mDesc, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
    d = d.squeeze(0).squeeze(1)
    dn = torch.norm(d, 2, 1)
    d = d.div(torch.unsqueeze(dn, 1))
    return d.transpose(0, 1).contiguous().numpy()


# ---------------------------------------------------------------- LightGlue (public model, torch ops)
def lg_forward(w, k0, k1, d0, d1):
    t = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(v)) for k, v in w.items()
         if k.startswith(("posenc", "transformers", "log_assignment"))}
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

    x0, x1 = torch.from_numpy(np.ascontiguousarray(d0, np.float32)), torch.from_numpy(np.ascontiguousarray(d1, np.float32))
    e0, e1 = posenc(torch.from_numpy(np.ascontiguousarray(k0, np.float32))), posenc(torch.from_numpy(np.ascontiguousarray(k1, np.float32)))
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
    """x [1,3,380,640] tensor -> the three raw Detect-layer maps [255, ny, nx] (numpy)."""
    t = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(v)) for k, v in w.items() if k.startswith("model.")}

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


_ANCHORS = torch.tensor([[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]], dtype=torch.float32).reshape(3, 3, 2)


def yolo_decode(raws):
    """The Detect layer's inference branch (inside the reference's TorchScript blob): raw maps -> preds [15120, 85]."""
    outs = []
    for i, r in enumerate(raws):
        r = torch.from_numpy(r)
        ny, nx = r.shape[1:]
        y = torch.sigmoid(r.reshape(3, 85, ny, nx).permute(0, 2, 3, 1))
        gy, gx = torch.meshgrid(torch.arange(ny, dtype=torch.float32), torch.arange(nx, dtype=torch.float32), indexing="ij")
        grid = torch.stack([gx, gy], -1)[None]
        xy = (y[..., 0:2] * 2 - 0.5 + grid) * float((8, 16, 32)[i])
        wh = (y[..., 2:4] * 2) ** 2 * _ANCHORS[i][:, None, None, :]
        outs.append(torch.cat([xy, wh, y[..., 4:]], -1).reshape(-1, 85))
    return torch.cat(outs, 0).numpy()


# ---------------------------------------------------------------- the reference-faithful per-frame CPU path (bench.py cpu_baseline)
def torch_weights(w):
    """numpy weight dict -> torch tensors, converted once (the reference loads its modules once, too)."""
    return {k: torch.from_numpy(np.ascontiguousarray(v, np.float32)) for k, v in w.items()}


def frontend_frame(tw, gray, bgr, prev, nfeatures=1024, ini_th=0.015, min_th=0.007, lg_threshold=0.05, stages=None):
    """One frame through the reference's front-end as shipped (LibTorch-CPU, batch 1):
    YoloDetection::Detect (src/YoloDetect.cpp:27-82) -> mvDynamicArea -> SPextractor::operator() (src/SPextractor.cpp:130-217:
    SuperPoint::forward, the 204-cell getKeyPoints / NMS2 loop with the box cull, DistributeOctTree, computeDescriptors) ->
    LightGlue::matchDescriptors against the previous frame (src/LightGlue.cpp:56-122).
    tw: torch_weights(...); prev: the dict this function returned for the previous frame (or None).
    stages: optional dict accumulating seconds per stage."""
    from . import lg_oracle, oracle, yolo_oracle
    H, W = gray.shape
    tick = time.perf_counter
    acc = stages if stages is not None else {}

    def add(k, t0):
        acc[k] = acc.get(k, 0.0) + (tick() - t0)

    with torch.no_grad():
        t0 = tick()
        x = torch.from_numpy(yolo_oracle.preprocess(bgr))[None]              # resize 640x380, BGR->RGB, /255 (:39-46)
        preds = yolo_decode(yolo_forward(tw, x))
        add("yolo_forward", t0); t0 = tick()
        dets = yolo_oracle.non_max_suppression(preds, 0.4, 0.5)               # :84-146
        dyn, _ = yolo_oracle.boxes_from_dets(dets, W, H)                      # :51-80, :148-208
        add("yolo_nms_boxes", t0); t0 = tick()
        xin = torch.from_numpy(gray)[None, None].to(torch.float32) / 255       # SPDetector::detect, src/SuperPoint.cpp:190-208
        prob, desc, _ = sp_forward(xin, tw, want_taps=False)
        add("sp_forward", t0); t0 = tick()
        cand = oracle.collect_candidates(np.ascontiguousarray(prob, np.float32), W, H, ini_th, min_th, dyn, 1.0)   # cell loop + NMS2 + cull
        add("sp_cell_loop_nms", t0); t0 = tick()
        sel = oracle.distribute_octtree(cand, 16, W - 16, 16, H - 16, nfeatures)
        kpts = (sel[:, :2] + np.float32(16)).astype(np.float32)
        add("sp_quadtree", t0); t0 = tick()
        d = sp_compute_descriptors(desc, kpts, (H // 8) * 8, (W // 8) * 8) if len(kpts) else np.zeros((0, 256), np.float32)
        add("sp_descriptors", t0); t0 = tick()
        cur = dict(kpts=kpts, desc=d, boxes=dyn)
        if prev is not None and len(prev["kpts"]) and len(kpts):
            k0 = lg_oracle.normalize_keypoints(prev["kpts"], (W, H)); k1 = lg_oracle.normalize_keypoints(kpts, (W, H))
            _, _, scores = lg_forward(tw, k0, k1, prev["desc"], d)
            m, s = lg_oracle.filter_matches(scores, lg_oracle.FILTER_THRESHOLD)
            keep = s > np.float32(lg_threshold)
            cur["matches"], cur["scores"] = m[keep], s[keep]
            add("lg_forward", t0)
    return cur
