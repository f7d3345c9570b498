#!/usr/bin/env python3
"""Instruction mix of the hottest loop of each kernel in a gfx950 assembly file (hipcc -S --cuda-device-only).

    python tools/loop_histogram.py file.s [kernel-substring]

The scans are issue-bound (two waves per SIMD, a wave64 VALU instruction occupies its SIMD for 4 cycles, a
transcendental for 16): what to look at is the VALU count per step, not the MFMA count.  For every kernel the
largest backward-branch loop is taken as "the" loop; the output is the number of instructions per class and an
issue-cycle estimate (valu*4 + trans*16 [the quarter-rate ops] + mfma*8 (16x16x32 bf16: 8 passes of 4 cycles = 32, but
issued back to back on the matrix pipe beside the VALU) ...) -- a rough guide, not a simulator.
"""
import re
import sys
from collections import Counter

TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


def kernels(path):
    name, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+|[A-Za-z_]\w*):\s*(;.*)?$", line)
        if m and not line.startswith(".L"):
            if name and body:
                yield name, body
            name, body = m.group(1), []
            continue
        if name is not None:
            if line.startswith(".Lfunc_end"):
                yield name, body
                name, body = None, []
            else:
                body.append(line.rstrip())
    if name and body:
        yield name, body


def hottest_loop(body):
    labels = {}
    for k, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = k
    best = None
    for k, l in enumerate(body):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < k:
            span = (labels[m.group(1)], k)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    return best


def main():
    path = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, body in kernels(path):
        if sub not in name:
            continue
        span = hottest_loop(body)
        if not span:
            continue
        c = Counter()
        ops = Counter()
        for l in body[span[0]:span[1] + 1]:
            m = re.match(r"\s+([a-z]\w+)", l)
            if not m:
                continue
            c[classify(m.group(1))] += 1
            ops[m.group(1)] += 1
        est = c["valu"] * 4 + c["valu_pk"] * 4 + c["trans"] * 16
        print("%s\n  loop %d lines: %s  ~VALU issue cycles %d" % (name, span[1] - span[0], dict(c), est))
        print("  top ops: " + ", ".join("%s %d" % kv for kv in ops.most_common(14)))


if __name__ == "__main__":
    main()
