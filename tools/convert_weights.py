#!/usr/bin/env python3
"""Convert a torch archive / state_dict of the reference's SuperPoint module to the engine's .dsw container.

usage: python tools/convert_weights.py <in.pt> <out.dsw>
<in.pt> may be a torch::save(module) archive (what /root/reference/src/SPextractor.cpp:86 loads; readable with
torch.jit.load), a TorchScript module, or a plain state_dict.  Tensor names are kept (conv1a.weight ...).
Container: b"DSW1", u32 n, then n x { u32 name_len, name bytes, u64 count, count x f32 }.
"""
import struct
import sys

import numpy as np


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
    import torch
    try:
        m = torch.jit.load(path, map_location="cpu")
        sd = m.state_dict()
    except Exception:
        sd = torch.load(path, map_location="cpu")
        if hasattr(sd, "state_dict"):
            sd = sd.state_dict()
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


if __name__ == "__main__":
    if len(sys.argv) != 3:
        sys.exit(__doc__)
    t = load_any(sys.argv[1])
    write_dsw(sys.argv[2], t)
    print(f"wrote {len(t)} tensors to {sys.argv[2]}")
