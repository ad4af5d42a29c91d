#!/usr/bin/env python3
"""Static audit of the hand-counted operand rings of the register-streaming kernels (lin_rs.h, attn_rs.h).

hipcc does not know that the inline-asm loads write their destination registers asynchronously: the kernels order every use
behind an `s_waitcnt vmcnt(N)` by naming the registers as in/out operands of the wait statement.  This script checks the
EMITTED code: for every `global_load_dwordx{2,4}` whose address uses an SGPR base (the asm loads), the first later instruction
that names one of its destination registers must be preceded by a wait whose count N is <= the number of vector-memory
operations issued between the load and that wait (vmcnt retires in issue order, so the load has then landed).
Loop bodies are scanned twice in a row so that loads consumed in the next iteration are followed across the back edge.

usage: hipcc ... -S --cuda-device-only -o lg.s lg_kernels.hip ; python tools/debug/check_asm_ring.py lg.s linear_rs_kernel attn_rs_kernel
"""
import re
import sys


def regs_of(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def operands(line):
    body = line.split(None, 1)
    if len(body) < 2:
        return []
    return [t.strip() for t in re.split(r",\s*", body[1].split(";")[0]) if t.strip()]


def check(name, lines):
    # basic blocks; a block ending in a backward branch to its own label is a loop body: duplicate it
    # (a block also ends BEHIND every branch: the straight-line code that follows a loop is not part of its body)
    labels = {l.split(":")[0]: i for i, l in enumerate(lines) if re.match(r"\.LBB\d+_\d+:", l)}
    seq = []
    cuts = set(labels.values()) | {i + 1 for i, l in enumerate(lines) if re.match(r"\s*s_c?branch", l)}
    starts = sorted(c for c in cuts if 0 < c < len(lines)) + [len(lines)]
    for a, b in zip([0] + starts[:-1], starts):
        blk = lines[a:b]
        lab = lines[a].split(":")[0] if a in labels.values() else None
        loop = lab and re.search(r"s_cbranch\w*\s+" + re.escape(lab) + r"\b", blk[-1])
        seq += blk + (blk if loop else [])
    bad = 0
    nloads = 0
    is_vmem = lambda t: t.startswith(("global_", "buffer_", "scratch_", "flat_"))
    for i, l in enumerate(seq):
        t = l.strip()
        m = re.match(r"global_load_dword(?:x[24])?\s+(v\[\d+:\d+\]|v\d+),\s*v\d+,\s*s\[\d+:\d+\]", t)
        if not m:
            continue
        nloads += 1
        dst = regs_of(m.group(1))
        issued = 0          # vector-memory operations issued after this load
        landed = False
        for j in range(i + 1, len(seq)):
            u = seq[j].strip()
            if not u or u.startswith((";", ".")):
                continue
            w = re.match(r"s_waitcnt\s+.*vmcnt\((\d+)\)", u)
            if w:
                if int(w.group(1)) <= issued:
                    landed = True
                continue
            ops = operands(u)
            touched = set().union(*[regs_of(o) for o in ops]) if ops else set()
            if touched & dst:
                if is_vmem(u.split()[0]) and regs_of(ops[0]) & dst and not (set().union(*[regs_of(o) for o in ops[1:]]) & dst):
                    if not landed:   # the slot is being refilled before its previous contents were ever waited for: unused load
                        pass
                    break
                if not landed:
                    bad += 1
                    print(f"  {name}: {t}  ->  first use before a covering wait: {u}")
                break
            if is_vmem(u.split()[0]):
                issued += 1
    print(f"{name}: {nloads} scalar-base loads checked, {bad} uses not covered by a wait")
    return bad


def main():
    s = open(sys.argv[1]).read()
    total = 0
    for want in sys.argv[2:]:
        for m in re.finditer(r"\n(_ZN4dani\w*?" + want + r"\w*):[^\n]*\n(.*?)\.end_amdhsa_kernel", s, re.S):
            lines = [l for l in m.group(2).split("\n") if l.strip() and not l.strip().startswith(";")]
            total += check(m.group(1), lines)
    sys.exit(1 if total else 0)


if __name__ == "__main__":
    main()
