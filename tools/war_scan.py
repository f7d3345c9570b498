#!/usr/bin/env python3
"""Static check for the MFMA operand hazard: a load (LDS / global / scratch) whose
destination is a register that an MFMA issued fewer than N instructions earlier reads as A or B, with no read
of that MFMA's (or a later MFMA's) result in between -- such a read proves the matrix pipe has drained past it.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -o /tmp/ks.s kws_amd/csrc/kernels_split.hip
    python tools/war_scan.py /tmp/ks.s [N=24]
Prints one line per kernel with the number of such pairs (0 everywhere is the goal) and the closest few.
"""
import re
import sys

path = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 24
LOADS = tuple(sys.argv[3].split(",")) if len(sys.argv) > 3 else ("ds_read", "global_load", "buffer_load", "scratch_load", "flat_load")


def regs(tok, kinds="v"):
    """register numbers named by one operand token; accumulation registers are offset by 1000"""
    tok = tok.strip()
    m = re.match(r"([va])\[(\d+):(\d+)\]", tok)
    if m and m.group(1) in kinds:
        return {(1000 if m.group(1) == "a" else 0) + r for r in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"([va])(\d+)$", tok)
    if m and m.group(1) in kinds:
        return {(1000 if m.group(1) == "a" else 0) + int(m.group(2))}
    return set()


def srcs(ins):
    parts = ins.split(None, 1)
    if len(parts) < 2:
        return set()
    ops = parts[1].split(",")
    out = set()
    for o in (ops if ins.startswith(("global_store", "ds_write", "buffer_store", "scratch_store", "v_cmp")) else ops[1:]):
        out |= regs(o.split()[0] if o.split() else "", "va")
    return out


kernels, cur, name = {}, None, None
for line in open(path):
    if re.match(r"^_Z\w+:", line):
        name = line.split(":")[0]
        cur = kernels.setdefault(name, [])
        continue
    s = line.strip()
    if cur is None or not line.startswith("\t") or not s or s[0] in ".;":
        continue
    cur.append(s.split(";")[0].strip())
    if s.startswith("s_endpgm"):
        cur = None

bad_total = 0
for name, ins in kernels.items():
    hits = []
    for i, l in enumerate(ins):
        if not l.startswith("v_mfma"):
            continue
        ops = [o.strip() for o in l.split(None, 1)[1].split(",")]
        ab = regs(ops[1]) | regs(ops[2])
        done = regs(ops[0], "va")                # results of this and of later MFMAs: reading one means this one retired
        for j in range(i + 1, min(i + 1 + N, len(ins))):
            m = ins[j]
            if m.startswith(("s_branch", "s_endpgm", "s_setpc")):
                break                            # (a conditional branch falls through: keep scanning that path)
            if m.startswith("v_mfma"):
                done |= regs(m.split(None, 1)[1].split(",")[0], "va")
                continue
            if srcs(m) & done:
                break                            # a completion read: the matrix pipe has drained past MFMA i
            if not m.startswith(LOADS):
                continue
            if regs(m.split(None, 1)[1].split(",")[0].strip()) & ab:
                hits.append((j - i, l, m))
                break
    n_mfma = sum(1 for l in ins if l.startswith("v_mfma"))
    if n_mfma:
        short = re.sub(r"^_ZN8fastgrnn12_GLOBAL__N_1\d+", "", name)[:48]
        print("%-50s mfma %4d  load-behind-mfma pairs %3d%s" % (
            short, n_mfma, len(hits), ("   closest +%d" % min(h[0] for h in hits)) if hits else ""))
        bad_total += len(hits)
print("total pairs:", bad_total)
sys.exit(1 if bad_total else 0)
