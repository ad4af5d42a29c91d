"""CPU: static audit of the EMITTED code of the register-streaming kernels (lin_rs.h, attn_rs.h, ffn_chain.h).

Their operand loads are inline asm whose completion hipcc does not track; the kernels order every use behind a hand-counted
`s_waitcnt vmcnt(N)`.  A compiler upgrade or an edit that moves a use in front of its wait would give wrong operands on some
waves of some launches, which a GPU test may not catch.  tools/debug/check_asm_ring.py proves, on the gfx950 assembly hipcc
emits here (cross-compiled, no GPU needed), that every instruction naming an asm-loaded register follows a wait that covers
the load -- and the second test shows that the audit does flag a kernel whose waits are too weak."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
pytestmark = pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not available")


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    out = tmp_path_factory.mktemp("asm") / "lg_kernels.s"
    src = os.path.join(ROOT, "dani_slam_amd", "csrc", "lg_kernels.hip")
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-S", "--cuda-device-only",
                           "-o", str(out), src], cwd=os.path.dirname(src), stderr=subprocess.DEVNULL)
    return str(out)


def _audit(path, *names):
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", "debug", "check_asm_ring.py"), path, *names],
                          capture_output=True, text=True)


def test_every_use_of_an_asm_loaded_register_is_behind_a_covering_wait(asm):
    r = _audit(asm, "linear_rs_kernel", "attn_rs_kernel", "ffn_chain_kernel")
    assert r.returncode == 0, r.stdout[-2000:]
    checked = [int(x) for x in re.findall(r"(\d+) scalar-base loads checked", r.stdout)]
    assert len(checked) >= 6 and min(checked) >= 30, r.stdout      # three linear tiles, two attention ring depths, the FFN chain
    assert re.search(r"ffn_chain_kernel\w*: (\d+) scalar-base loads checked", r.stdout), r.stdout


def test_the_audit_flags_a_weakened_wait_in_the_ffn_chain(asm, tmp_path):
    s = open(asm).read()
    m = re.search(r"\n(_ZN4dani16ffn_chain_kernel\w*):[^\n]*\n(.*?)\.end_amdhsa_kernel", s, re.S)
    body = m.group(2)
    assert "s_waitcnt vmcnt(15)" in body and "s_waitcnt vmcnt(17)" in body   # chain_pass16's ring of 16; stage A's 8 + 8 + B ring
    bad = tmp_path / "bad_chain.s"
    bad.write_text(s.replace(body, body.replace("s_waitcnt vmcnt(15)", "s_waitcnt vmcnt(40)").replace("s_waitcnt vmcnt(17)", "s_waitcnt vmcnt(40)")))
    r = _audit(str(bad), "ffn_chain_kernel")
    assert r.returncode == 1 and "first use before a covering wait" in r.stdout


def test_the_audit_flags_a_weakened_wait(asm, tmp_path):
    s = open(asm).read()
    m = re.search(r"\n(_ZN4dani16linear_rs_kernelILi1ELi4E\w*):[^\n]*\n(.*?)\.end_amdhsa_kernel", s, re.S)
    body = m.group(2)
    assert "s_waitcnt vmcnt(15)" in body                            # (UQ - 1) * (NCB + 4) for NCB = 1, UQ = 4
    bad = tmp_path / "bad.s"
    bad.write_text(s.replace(body, body.replace("s_waitcnt vmcnt(15)", "s_waitcnt vmcnt(40)")))
    r = _audit(str(bad), "linear_rs_kernelILi1ELi4")
    assert r.returncode == 1 and "first use before a covering wait" in r.stdout
