#!/usr/bin/env python3
"""Convert the reference's three weight archives to the engine's .dsw container (needs PyTorch only on the converting machine).

usage: python tools/convert_weights.py --superpoint superpoint_new.pt [--lightglue lightglue_scripted_v2.pt]
                                       [--yolo yolov5s.torchscript.pt] out.dsw
       python tools/convert_weights.py <superpoint.pt> <out.dsw>                (the SuperPoint-only form of earlier rounds)

Inputs (none of them is shipped with the reference):
  --superpoint  what /root/reference/src/SPextractor.cpp:86 loads with torch::load: a torch::save(module) archive (a
                TorchScript zip that torch.jit.load reads), a scripted module, or a plain state_dict; the 24 parameters keep the
                reference's names conv1a.weight ... convDb.bias (src/SuperPoint.cpp:37-52).  A wrapper prefix ("model.",
                "module.", "superpoint." ...) is stripped.
  --lightglue   what src/LightGlue.cpp:9 jit::loads: a scripted wrapper around the public LightGlue module, or the public
                checkpoint (superpoint_lightglue.pth).  Names are mapped to the public module's:
                  <prefix>transformers.{i}.self_attn.* / cross_attn.*, log_assignment.{i}.*, posenc.Wr.weight
                the v0.1 checkpoint layout self_attn.{i}.* / cross_attn.{i}.* is renamed as the public loader does;
                token_confidence.* (early exit, unused: the engine runs all 9 layers) and num_batches_tracked are dropped;
                an input_proj that is not the identity is refused (the SuperPoint variant has none: 256 -> 256).
  --yolo        what src/YoloDetect.cpp:9 jit::loads: a YOLOv5s v6 export or checkpoint.  Convolutions that still carry
                a separate BatchNorm (model.N.conv.weight + model.N.bn.{weight,bias,running_mean,running_var}) are fused as
                ultralytics' fuse_conv_and_bn does (eps = 1e-3, the value YOLOv5 sets on every BatchNorm2d):
                    s = gamma / sqrt(running_var + eps);  W' = W * s[:, None, None, None];  b' = beta + (b - running_mean) * s
                (b = the conv's own bias, 0 when it has none), computed in float64 and rounded once.  model.24.anchors must
                equal the standard anchors the decode kernel holds (yolo_kernels.hip) or the conversion fails.
Every problem is an error naming the tensor; nothing is guessed.  After writing, the file is checked with the engine's own
host-side validator (dani_check_weights_file in libdani_frontend.so) when the library has been built.

Container: b"DSW1", u32 n, then n x { u32 name_len, name bytes, u64 count, count x f32 }.
"""
import argparse
import os
import re
import struct
import sys

import numpy as np


class ConversionError(ValueError):
    pass


def write_dsw(path, tensors: dict):
    with open(path, "wb") as f:
        f.write(b"DSW1")
        f.write(struct.pack("<I", len(tensors)))
        for name, arr in tensors.items():
            a = np.ascontiguousarray(arr, np.float32).reshape(-1)
            nb = name.encode()
            f.write(struct.pack("<I", len(nb))); f.write(nb)
            f.write(struct.pack("<Q", a.size)); f.write(a.tobytes())


def load_any(path):
    """name -> numpy array for a TorchScript archive (torch::save(module) writes one), a pickled module, a plain
    state_dict, or a checkpoint dict holding one under "model" / "state_dict" / "model_state_dict"."""
    import torch
    try:
        m = torch.jit.load(path, map_location="cpu")
        sd = m.state_dict()
    except Exception:
        sd = torch.load(path, map_location="cpu", weights_only=False)
        if isinstance(sd, dict):
            for key in ("model", "ema", "state_dict", "model_state_dict"):
                if key in sd and sd[key] is not None and not hasattr(sd[key], "shape"):
                    sd = sd[key]
                    break
        if hasattr(sd, "state_dict"):
            sd = sd.float().state_dict() if hasattr(sd, "float") else sd.state_dict()
    if not isinstance(sd, dict) or not sd:
        raise ConversionError(f"{path}: no tensors found")
    out = {}
    for k, v in sd.items():
        if hasattr(v, "detach"):
            out[k] = v.detach().cpu().to(torch.float32).numpy() if v.dtype.is_floating_point else v.detach().cpu().numpy()
    return out


# ---- expected tensors (mirrors dani_slam_amd/csrc/weights_spec.cpp; dani_check_weights_file has the last word) -----------
SP_SPEC = [("conv1a", 1, 64, 3), ("conv1b", 64, 64, 3), ("conv2a", 64, 64, 3), ("conv2b", 64, 64, 3), ("conv3a", 64, 128, 3),
           ("conv3b", 128, 128, 3), ("conv4a", 128, 128, 3), ("conv4b", 128, 128, 3), ("convPa", 128, 256, 3),
           ("convPb", 256, 65, 1), ("convDa", 128, 256, 3), ("convDb", 256, 256, 1)]


def lightglue_spec():
    s = {"posenc.Wr.weight": (32, 2)}
    def lin(n, o, i): s[n + ".weight"] = (o, i); s[n + ".bias"] = (o,)
    for i in range(9):
        ps, pc = f"transformers.{i}.self_attn", f"transformers.{i}.cross_attn"
        lin(ps + ".Wqkv", 768, 256); lin(ps + ".out_proj", 256, 256)
        lin(pc + ".to_qk", 256, 256); lin(pc + ".to_v", 256, 256); lin(pc + ".to_out", 256, 256)
        for p in (ps, pc):
            lin(p + ".ffn.0", 512, 512); s[p + ".ffn.1.weight"] = (512,); s[p + ".ffn.1.bias"] = (512,); lin(p + ".ffn.3", 256, 512)
    lin("log_assignment.8.final_proj", 256, 256)
    s["log_assignment.8.matchability.weight"] = (1, 256); s["log_assignment.8.matchability.bias"] = (1,)
    return s


def yolo_conv_list():
    L = []
    def conv(n, cin, cout, k): L.append((n, cin, cout, k))
    def c3(n, c1, c2, reps):
        c_ = c2 // 2
        conv(f"{n}.cv1", c1, c_, 1); conv(f"{n}.cv2", c1, c_, 1); conv(f"{n}.cv3", 2 * c_, c2, 1)
        for j in range(reps):
            conv(f"{n}.m.{j}.cv1", c_, c_, 1); conv(f"{n}.m.{j}.cv2", c_, c_, 3)
    conv("model.0", 3, 32, 6); conv("model.1", 32, 64, 3); c3("model.2", 64, 64, 1)
    conv("model.3", 64, 128, 3); c3("model.4", 128, 128, 2)
    conv("model.5", 128, 256, 3); c3("model.6", 256, 256, 3)
    conv("model.7", 256, 512, 3); c3("model.8", 512, 512, 1)
    conv("model.9.cv1", 512, 256, 1); conv("model.9.cv2", 1024, 512, 1)
    conv("model.10", 512, 256, 1); c3("model.13", 512, 256, 1)
    conv("model.14", 256, 128, 1); c3("model.17", 256, 128, 1)
    conv("model.18", 128, 128, 3); c3("model.20", 256, 256, 1)
    conv("model.21", 256, 256, 3); c3("model.23", 512, 512, 1)
    return L


YOLO_ANCHORS = np.array([[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]], np.float64).reshape(3, 3, 2)
YOLO_STRIDES = (8.0, 16.0, 32.0)
YOLO_BN_EPS = 1e-3


def _find_prefix(sd, anchor_key, what):
    """The wrapper prefix P such that P + anchor_key is a key of sd (shortest one); '' when the names are already bare."""
    c = sorted((k[: -len(anchor_key)] for k in sd if k.endswith(anchor_key)), key=len)
    if not c:
        raise ConversionError(f"{what}: tensor {anchor_key} not found under any prefix (keys start with: {sorted(sd)[:3]} ...)")
    return c[0]


def _want(sd, name, shape, what):
    if name not in sd:
        raise ConversionError(f"{what}: missing tensor {name} {list(shape)}")
    a = np.asarray(sd[name])
    if tuple(a.shape) != tuple(shape):
        raise ConversionError(f"{what}: tensor {name} has shape {list(a.shape)}, expected {list(shape)}")
    return np.ascontiguousarray(a, np.float32)


def map_superpoint(sd):
    """The reference's 24 parameters under their own names (src/SuperPoint.cpp:37-52), any wrapper prefix stripped."""
    p = _find_prefix(sd, "conv1a.weight", "SuperPoint")
    out = {}
    for n, cin, cout, k in SP_SPEC:
        out[n + ".weight"] = _want(sd, p + n + ".weight", (cout, cin, k, k), "SuperPoint")
        out[n + ".bias"] = _want(sd, p + n + ".bias", (cout,), "SuperPoint")
    return out


def map_lightglue(sd):
    """Public LightGlue (SuperPoint variant) names, from a wrapped / scripted module or either public checkpoint layout."""
    sd = dict(sd)
    if not any(k.endswith("transformers.0.self_attn.Wqkv.weight") for k in sd) and any(re.search(r"(^|\.)self_attn\.0\.Wqkv\.weight$", k) for k in sd):
        # v0.1 checkpoints: self_attn.{i}.X / cross_attn.{i}.X -> transformers.{i}.self_attn.X (the public loader's rename)
        ren = {}
        for k, v in sd.items():
            m = re.match(r"^(.*?)(self_attn|cross_attn)\.(\d+)\.(.*)$", k)
            ren[f"{m.group(1)}transformers.{m.group(3)}.{m.group(2)}.{m.group(4)}" if m else k] = v
        sd = ren
    p = _find_prefix(sd, "transformers.0.self_attn.Wqkv.weight", "LightGlue")
    iw, ib = p + "input_proj.weight", p + "input_proj.bias"
    if iw in sd:
        w = np.asarray(sd[iw], np.float64)
        b = np.asarray(sd.get(ib, np.zeros(w.shape[0])), np.float64)
        if w.shape != (256, 256) or not np.array_equal(w, np.eye(256)) or np.any(b != 0):
            raise ConversionError(f"LightGlue: {iw} {list(w.shape)} is not the identity: this is not the SuperPoint variant "
                                  "(descriptor dim 256 = model dim 256, no input projection); the engine has no such layer")
    out = {}
    for name, shape in lightglue_spec().items():
        out[name] = _want(sd, p + name, shape, "LightGlue")
    return out


def fuse_conv_bn(w, b, gamma, beta, mean, var, eps=YOLO_BN_EPS):
    """Conv2d followed by BatchNorm2d (eval mode) as one convolution; float64 arithmetic, one rounding to f32."""
    w = np.asarray(w, np.float64); s = np.asarray(gamma, np.float64) / np.sqrt(np.asarray(var, np.float64) + eps)
    b0 = np.zeros(w.shape[0]) if b is None else np.asarray(b, np.float64)
    return (w * s[:, None, None, None]).astype(np.float32), (np.asarray(beta, np.float64) + (b0 - np.asarray(mean, np.float64)) * s).astype(np.float32)


def map_yolov5s(sd):
    """YOLOv5s v6 under the fused public names model.N[...].conv.{weight,bias} and model.24.m.{i}.{weight,bias}."""
    p = _find_prefix(sd, "model.0.conv.weight", "YOLOv5s")
    out = {}
    for n, cin, cout, k in yolo_conv_list():
        w = _want(sd, f"{p}{n}.conv.weight", (cout, cin, k, k), "YOLOv5s")
        bn = f"{p}{n}.bn."
        bias_key = f"{p}{n}.conv.bias"
        if bn + "weight" in sd:      # un-fused export: fold the BatchNorm in
            g = _want(sd, bn + "weight", (cout,), "YOLOv5s"); be = _want(sd, bn + "bias", (cout,), "YOLOv5s")
            mu = _want(sd, bn + "running_mean", (cout,), "YOLOv5s"); var = _want(sd, bn + "running_var", (cout,), "YOLOv5s")
            if np.any(var < 0):
                raise ConversionError(f"YOLOv5s: {bn}running_var has negative entries")
            cb = _want(sd, bias_key, (cout,), "YOLOv5s") if bias_key in sd else None
            w, b = fuse_conv_bn(w, cb, g, be, mu, var)
        else:
            b = _want(sd, bias_key, (cout,), "YOLOv5s")
        out[f"{n}.conv.weight"], out[f"{n}.conv.bias"] = w, b
    for i, ch in enumerate((128, 256, 512)):
        out[f"model.24.m.{i}.weight"] = _want(sd, f"{p}model.24.m.{i}.weight", (255, ch, 1, 1), "YOLOv5s")
        out[f"model.24.m.{i}.bias"] = _want(sd, f"{p}model.24.m.{i}.bias", (255,), "YOLOv5s")
    ak = p + "model.24.anchors"
    if ak in sd:      # stored divided by the level's stride
        a = np.asarray(sd[ak], np.float64).reshape(3, 3, 2) * np.array(YOLO_STRIDES)[:, None, None]
        if not np.allclose(a, YOLO_ANCHORS, rtol=0, atol=1e-3):
            raise ConversionError(f"YOLOv5s: {ak} are not the standard anchors the decode kernel holds: {a.reshape(3, 6).tolist()}")
    return out


def check_dsw(path):
    """dani_check_weights_file of the built library (host-only); returns (rc, groups, message) or None when it is not built."""
    import ctypes as C
    here = os.path.dirname(os.path.abspath(__file__))
    lib_path = os.environ.get("DANI_FRONTEND_LIB", os.path.join(here, "..", "dani_slam_amd", "lib", "libdani_frontend.so"))
    if not os.path.exists(lib_path):
        return None
    lib = C.CDLL(lib_path)
    lib.dani_check_weights_file.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.c_char_p, C.c_int32]
    msg = C.create_string_buffer(512); groups = C.c_int32(0)
    rc = lib.dani_check_weights_file(path.encode(), C.byref(groups), msg, 512)
    return rc, groups.value, msg.value.decode()


def convert(superpoint, lightglue=None, yolo=None):
    t = map_superpoint(load_any(superpoint))
    if lightglue:
        t.update(map_lightglue(load_any(lightglue)))
    if yolo:
        t.update(map_yolov5s(load_any(yolo)))
    return t


def main(argv):
    if len(argv) == 2 and not argv[0].startswith("-"):      # legacy form: <superpoint.pt> <out.dsw>
        argv = ["--superpoint", argv[0], argv[1]]
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--superpoint", required=True); ap.add_argument("--lightglue"); ap.add_argument("--yolo")
    ap.add_argument("out")
    a = ap.parse_args(argv)
    try:
        t = convert(a.superpoint, a.lightglue, a.yolo)
    except ConversionError as ex:
        sys.exit(f"convert_weights: {ex}")
    write_dsw(a.out, t)
    chk = check_dsw(a.out)
    if chk is not None and chk[0] != 0:
        sys.exit(f"convert_weights: {a.out} fails the engine's check: {chk[2]}")
    print(f"wrote {len(t)} tensors to {a.out}" + ("" if chk is None else f" (engine check ok, groups {chk[1]:03b})"))


if __name__ == "__main__":
    main(sys.argv[1:])
