"""CPU: real-archive ingestion.  The reference loads three files that are not in its repository:
  torch::load(model, "/sly_slam/superpoint_new.pt")            /root/reference/src/SPextractor.cpp:86-91  (exit on failure)
  torch::jit::load("/sly_slam/lightglue_scripted_v2.pt")       src/LightGlue.cpp:9
  torch::jit::load("yolov5s.torchscript.pt")                   src/YoloDetect.cpp:9
tools/convert_weights.py turns such archives into the engine's .dsw container; the engine's own host-side validator
(dani_check_weights_file, the check dani_finalize_weights runs) must accept the result or name the offending tensor.
Archives are built here with torch in the layouts a user would hold: a scripted module / torch::save-style archive with the
reference's parameter names (src/SuperPoint.cpp:37-52), a plain state_dict, wrapper prefixes, the public LightGlue checkpoint's
extra keys, and a YOLOv5s export whose BatchNorms have not been fused."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest
import torch

from dani_slam_amd import frontend, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("convert_weights", os.path.join(ROOT, "tools", "convert_weights.py"))
cw = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(cw)


def check_file(path):
    lib = frontend.load_library()
    lib.dani_check_weights_file.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.c_char_p, C.c_int32]
    msg = C.create_string_buffer(512); groups = C.c_int32(0)
    rc = lib.dani_check_weights_file(str(path).encode(), C.byref(groups), msg, 512)
    return rc, groups.value, msg.value.decode()


class RefSuperPoint(torch.nn.Module):
    """The reference's module: 12 Conv2d registered under these names (src/SuperPoint.cpp:16-52)."""

    def __init__(self, w):
        super().__init__()
        for name, cin, cout, k in synth.SP_LAYERS:
            conv = torch.nn.Conv2d(cin, cout, k, padding=k // 2)
            with torch.no_grad():
                conv.weight.copy_(torch.from_numpy(w[name + ".weight"])); conv.bias.copy_(torch.from_numpy(w[name + ".bias"]))
            setattr(self, name, conv)

    def forward(self, x):
        return self.conv1b(torch.relu(self.conv1a(x)))


class Wrapped(torch.nn.Module):
    def __init__(self, inner):
        super().__init__()
        self.model = inner

    def forward(self, x):
        return self.model(x)


def test_superpoint_archives_round_trip(tmp_path, sp_weights):
    m = RefSuperPoint(sp_weights).eval()
    paths = {}
    torch.jit.script(m).save(str(tmp_path / "scripted.pt")); paths["scripted"] = tmp_path / "scripted.pt"       # what torch::save(module) writes
    torch.save(m.state_dict(), str(tmp_path / "sd.pt")); paths["state_dict"] = tmp_path / "sd.pt"
    torch.save(m, str(tmp_path / "module.pt")); paths["pickled module"] = tmp_path / "module.pt"
    torch.jit.script(Wrapped(m)).save(str(tmp_path / "wrapped.pt")); paths["wrapper prefix"] = tmp_path / "wrapped.pt"
    torch.save({"epoch": 3, "model_state_dict": m.state_dict()}, str(tmp_path / "ckpt.pt")); paths["checkpoint dict"] = tmp_path / "ckpt.pt"
    for what, p in paths.items():
        t = cw.map_superpoint(cw.load_any(str(p)))
        assert list(t) == [f"{n}.{s}" for n, *_ in synth.SP_LAYERS for s in ("weight", "bias")], what
        for k, v in t.items():
            assert v.dtype == np.float32 and np.array_equal(v, sp_weights[k]), (what, k)      # OIHW, bit for bit
        out = tmp_path / f"{p.stem}.dsw"
        cw.write_dsw(str(out), t)
        rc, groups, msg = check_file(out)
        assert rc == 0 and groups == 1 and msg == "", (what, rc, groups, msg)
    # the command line, legacy two-argument form
    cw.main([str(paths["scripted"]), str(tmp_path / "cli.dsw")])
    assert check_file(tmp_path / "cli.dsw")[0] == 0


def test_superpoint_archives_that_must_fail(tmp_path, sp_weights):
    sd = {k: torch.from_numpy(v) for k, v in sp_weights.items()}
    missing = dict(sd); del missing["convDb.bias"]
    torch.save(missing, str(tmp_path / "missing.pt"))
    with pytest.raises(cw.ConversionError, match="convDb.bias"):
        cw.map_superpoint(cw.load_any(str(tmp_path / "missing.pt")))
    bad = dict(sd); bad["convPb.weight"] = torch.zeros(64, 256, 1, 1)          # a 64-channel detector head (no dustbin)
    torch.save(bad, str(tmp_path / "bad.pt"))
    with pytest.raises(cw.ConversionError, match=r"convPb.weight has shape \[64, 256, 1, 1\], expected \[65, 256, 1, 1\]"):
        cw.map_superpoint(cw.load_any(str(tmp_path / "bad.pt")))
    with pytest.raises(SystemExit):
        cw.main([str(tmp_path / "bad.pt"), str(tmp_path / "never.dsw")])
    assert not (tmp_path / "never.dsw").exists()
    # the engine's own check on containers written around the converter: same verdicts, tensor named
    w = {k: v for k, v in sp_weights.items() if k != "conv3a.weight"}
    cw.write_dsw(str(tmp_path / "m.dsw"), w)
    rc, _, msg = check_file(tmp_path / "m.dsw")
    assert rc == frontend.DANI_ERR_WEIGHTS and "missing tensor conv3a.weight [128,64,3,3]" in msg
    w = dict(sp_weights); w["conv1a.weight"] = np.zeros((64, 3, 3, 3), np.float32)   # an RGB first layer
    cw.write_dsw(str(tmp_path / "s.dsw"), w)
    rc, _, msg = check_file(tmp_path / "s.dsw")
    assert rc == frontend.DANI_ERR_WEIGHTS and "conv1a.weight has 1728 floats, expected 576 [64,1,3,3]" in msg
    (tmp_path / "junk.dsw").write_bytes(b"DSW1" + b"\xff" * 40)
    assert check_file(tmp_path / "junk.dsw")[0] == frontend.DANI_ERR_WEIGHTS
    full = tmp_path / "full.dsw"; cw.write_dsw(str(full), dict(sp_weights))
    (tmp_path / "trunc.dsw").write_bytes(full.read_bytes()[:-100])                    # last payload cut short
    assert check_file(tmp_path / "trunc.dsw")[0] == frontend.DANI_ERR_WEIGHTS
    assert check_file(tmp_path / "absent.dsw")[0] == frontend.DANI_ERR_WEIGHTS


def _public_lightglue_checkpoint(w, prefix=""):
    """The public checkpoint's key set for the SuperPoint variant: every layer's assignment head and the early-exit
    token_confidence heads next to what the engine uses."""
    sd = {prefix + k: torch.from_numpy(v) for k, v in w.items()}
    for i in range(8):
        sd[f"{prefix}token_confidence.{i}.token.0.weight"] = torch.randn(1, 256)
        sd[f"{prefix}token_confidence.{i}.token.0.bias"] = torch.randn(1)
    sd[prefix + "confidence_thresholds"] = torch.randn(9)
    return sd


def test_lightglue_checkpoint_layouts(tmp_path, sp_weights):
    w = synth.lightglue_weights(0)
    want = cw.lightglue_spec()
    for prefix in ("", "net.", "matcher.module."):                   # bare checkpoint / scripted wrapper prefixes
        sd = _public_lightglue_checkpoint(w, prefix)
        torch.save(sd, str(tmp_path / "lg.pt"))
        t = cw.map_lightglue(cw.load_any(str(tmp_path / "lg.pt")))
        assert set(t) == set(want) and not any("token_confidence" in k for k in t)
        assert all(np.array_equal(t[k], w[k]) for k in t), prefix
    # v0.1 layout: self_attn.{i}.X / cross_attn.{i}.X, renamed as the public loader does
    old = {}
    for k, v in _public_lightglue_checkpoint(w).items():
        parts = k.split(".")
        old[f"{parts[2]}.{parts[1]}." + ".".join(parts[3:]) if parts[0] == "transformers" else k] = v
    assert "self_attn.0.Wqkv.weight" in old and "cross_attn.8.to_out.bias" in old
    t = cw.map_lightglue(old)
    assert all(np.array_equal(t[k], w[k]) for k in want)
    # an identity input_proj is accepted (and dropped), any other is not the SuperPoint variant
    sd = _public_lightglue_checkpoint(w); sd["input_proj.weight"] = torch.eye(256); sd["input_proj.bias"] = torch.zeros(256)
    assert "input_proj.weight" not in cw.map_lightglue(sd)
    sd["input_proj.weight"] = torch.randn(256, 128)                   # e.g. the DISK / ALIKED variants (128-d descriptors)
    with pytest.raises(cw.ConversionError, match="input_proj.weight .* not the identity"):
        cw.map_lightglue(sd)
    # a missing key is named, by the converter and by the engine's check of a container written without it
    sd = _public_lightglue_checkpoint(w, "net."); del sd["net.transformers.4.cross_attn.to_v.bias"]
    with pytest.raises(cw.ConversionError, match=r"missing tensor net\.transformers\.4\.cross_attn\.to_v\.bias \[256\]"):
        cw.map_lightglue(sd)
    both = dict(sp_weights); both.update({k: v for k, v in w.items() if k != "transformers.4.cross_attn.to_v.bias"})
    cw.write_dsw(str(tmp_path / "lgmiss.dsw"), both)
    rc, _, msg = check_file(tmp_path / "lgmiss.dsw")
    assert rc == frontend.DANI_ERR_WEIGHTS and "missing tensor transformers.4.cross_attn.to_v.bias [256] (LightGlue" in msg
    # complete: SuperPoint + LightGlue present, the unused per-layer assignment heads reported as ignored
    both = dict(sp_weights); both.update(w)
    cw.write_dsw(str(tmp_path / "splg.dsw"), both)
    rc, groups, msg = check_file(tmp_path / "splg.dsw")
    assert rc == 0 and groups == 3 and "32 tensor(s) the engine does not use are ignored" in msg, (rc, groups, msg)


def _unfused_yolo(seed=0):
    """A YOLOv5s v6 state_dict as ultralytics saves it before fuse(): conv without bias + BatchNorm2d (eps 1e-3), plus the
    Detect layer's anchors buffer and the BatchNorm counters."""
    rng = np.random.RandomState(seed)
    sd = {}
    for n, cin, cout, k in cw.yolo_conv_list():
        sd[f"{n}.conv.weight"] = torch.from_numpy((rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32))
        sd[f"{n}.bn.weight"] = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32))
        sd[f"{n}.bn.bias"] = torch.from_numpy(rng.uniform(-0.2, 0.2, cout).astype(np.float32))
        sd[f"{n}.bn.running_mean"] = torch.from_numpy(rng.uniform(-0.3, 0.3, cout).astype(np.float32))
        sd[f"{n}.bn.running_var"] = torch.from_numpy(rng.uniform(0.2, 2.0, cout).astype(np.float32))
        sd[f"{n}.bn.num_batches_tracked"] = torch.tensor(1234)
    for i, ch in enumerate((128, 256, 512)):
        sd[f"model.24.m.{i}.weight"] = torch.from_numpy((rng.standard_normal((255, ch, 1, 1)) / np.sqrt(ch)).astype(np.float32))
        sd[f"model.24.m.{i}.bias"] = torch.from_numpy(rng.uniform(-0.2, 0.2, 255).astype(np.float32))
    sd["model.24.anchors"] = torch.from_numpy((cw.YOLO_ANCHORS / np.array(cw.YOLO_STRIDES)[:, None, None]).astype(np.float32))
    return sd


def test_yolov5s_batchnorm_is_fused_on_the_host(tmp_path, sp_weights):
    sd = _unfused_yolo()
    torch.save(sd, str(tmp_path / "y.pt"))
    t = cw.map_yolov5s(cw.load_any(str(tmp_path / "y.pt")))
    assert set(t) == set(synth.yolo_weights(0)) and all(t[k].shape == v.shape for k, v in synth.yolo_weights(0).items())
    # (a) the fused tensors equal torch's own fusion of the same conv + bn (eps 1e-3, ultralytics' fuse_conv_and_bn)
    # (b) the fused convolution computes what conv -> bn computes in eval mode
    for n, cin, cout, k in [c for c in cw.yolo_conv_list() if c[0] in ("model.0", "model.4.m.1.cv2", "model.9.cv2", "model.23.cv3")]:
        conv = torch.nn.Conv2d(cin, cout, k, padding=k // 2, bias=False)
        bn = torch.nn.BatchNorm2d(cout, eps=1e-3)
        conv.load_state_dict({"weight": sd[f"{n}.conv.weight"]})
        bn.load_state_dict({kk: sd[f"{n}.bn.{kk}"] for kk in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")})
        conv.eval(); bn.eval()
        fused = torch.nn.utils.fusion.fuse_conv_bn_eval(conv, bn)
        np.testing.assert_allclose(t[f"{n}.conv.weight"], fused.weight.detach().numpy(), rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(t[f"{n}.conv.bias"], fused.bias.detach().numpy(), rtol=2e-6, atol=1e-7)
        x = torch.randn(1, cin, 12, 20)
        with torch.no_grad():
            want = bn(conv(x))
            got = torch.nn.functional.conv2d(x, torch.from_numpy(t[f"{n}.conv.weight"]), torch.from_numpy(t[f"{n}.conv.bias"]), padding=k // 2)
        np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-4, atol=1e-5)
    # an already fused export (what the .torchscript.pt of the reference holds) passes through unchanged, under a prefix too
    pre = {"yolo." + k: torch.from_numpy(v) for k, v in t.items()}
    t2 = cw.map_yolov5s(pre)
    assert all(np.array_equal(t2[k], t[k]) for k in t)
    # other anchors than the decode kernel's: refused
    sd2 = dict(sd); sd2["model.24.anchors"] = sd["model.24.anchors"] * 1.5
    with pytest.raises(cw.ConversionError, match="not the standard anchors"):
        cw.map_yolov5s(sd2)
    sd2 = dict(sd); del sd2["model.13.m.0.cv1.bn.running_var"]
    with pytest.raises(cw.ConversionError, match=r"missing tensor model\.13\.m\.0\.cv1\.bn\.running_var \[128\]"):
        cw.map_yolov5s(sd2)
    # all three archives through the command line into one container: the engine's check sees every group
    torch.save({k: torch.from_numpy(v) for k, v in sp_weights.items()}, str(tmp_path / "sp.pt"))
    torch.save(_public_lightglue_checkpoint(synth.lightglue_weights(0), "net."), str(tmp_path / "lg.pt"))
    cw.main(["--superpoint", str(tmp_path / "sp.pt"), "--lightglue", str(tmp_path / "lg.pt"), "--yolo", str(tmp_path / "y.pt"), str(tmp_path / "all.dsw")])
    rc, groups, msg = check_file(tmp_path / "all.dsw")
    assert rc == 0 and groups == 7 and msg == "", (rc, groups, msg)
    # a half YOLO group is an error that names the first missing tensor
    some = dict(sp_weights); some.update({k: v for k, v in t.items() if not k.startswith("model.24.m.2")})
    cw.write_dsw(str(tmp_path / "half.dsw"), some)
    rc, _, msg = check_file(tmp_path / "half.dsw")
    assert rc == frontend.DANI_ERR_WEIGHTS and "missing tensor model.24.m.2.weight [255,512,1,1] (YOLOv5s" in msg


def test_damaged_containers_are_error_codes_not_crashes(tmp_path, sp_weights):
    """dani_check_weights_file on 300 damaged copies of a valid .dsw -- truncated at every kind of boundary, name lengths and
    element counts overwritten with huge / zero / negative-looking values, random byte flips in the directory part -- answers with
    a status and a message every time (the parser runs inside this process: a wild read would take pytest down with it), and
    accepts only containers whose directory is still intact (src/SPextractor.cpp:86-91: the reference exits on a bad archive)."""
    good = tmp_path / "good.dsw"
    cw.write_dsw(str(good), sp_weights)
    blob = bytearray(good.read_bytes())
    assert check_file(good)[0] == 0
    rng = np.random.default_rng(7)
    first_name_len = 8                                       # "DSW1", u32 n, then the first record's u32 name length
    cases = []
    for cut in [0, 3, 4, 7, 8, 11, 12, 20, 21, 29, len(blob) // 2, len(blob) - 1]:
        cases.append(bytes(blob[:cut]))
    for v in (0, 255, 256, 0x7fffffff, 0xffffffff):
        b = bytearray(blob); b[first_name_len:first_name_len + 4] = int(v).to_bytes(4, "little"); cases.append(bytes(b))
    nl = int.from_bytes(blob[first_name_len:first_name_len + 4], "little")
    cnt_at = first_name_len + 4 + nl
    for v in (0, 1, (1 << 28) - 1, 1 << 28, (1 << 63) - 1, (1 << 64) - 1):
        b = bytearray(blob); b[cnt_at:cnt_at + 8] = int(v).to_bytes(8, "little"); cases.append(bytes(b))
    for v in (0, 1, 11, 13, 0xffffffff):                     # record count of the container
        b = bytearray(blob); b[4:8] = int(v).to_bytes(4, "little"); cases.append(bytes(b))
    n_structured = len(cases)
    while len(cases) < 300:
        b = bytearray(blob)
        for _ in range(int(rng.integers(1, 6))):
            b[int(rng.integers(0, min(len(b), 4096)))] = int(rng.integers(0, 256))
        if rng.random() < 0.3:
            b = b[:int(rng.integers(8, len(b)))]
        cases.append(bytes(b))
    rejected_random = 0
    for i, c in enumerate(cases):
        p = tmp_path / "bad.dsw"
        p.write_bytes(c)
        rc, groups, msg = check_file(p)
        assert rc in (0, frontend.DANI_ERR_WEIGHTS), (i, rc, msg)
        assert rc == 0 or msg, i
        if i < n_structured:
            assert rc == frontend.DANI_ERR_WEIGHTS, (i, msg)          # every truncation and every damaged length / count is refused
        else:
            rejected_random += rc != 0                                  # (a flip inside a float payload leaves a valid container)
    assert rejected_random > 20
