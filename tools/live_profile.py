#!/usr/bin/env python3
"""Rough VGPR liveness profile of a kernel's main loop from the compiler's assembly (linear approximation: the loop body
in program order, live-in = live-out around the back edge).  Prints the live count at markers (MFMA runs, barriers,
buffer/global loads, spills) so that one can see WHICH phase of a scan step is at the register ceiling.

    python tools/live_profile.py /tmp/x.s <mangled-name-substring>
"""
import re
import sys

REG = re.compile(r"\bv(?:\[(\d+):(\d+)\]|(\d+)\b)")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def main():
    txt = open(sys.argv[1]).read()
    i = txt.index(sys.argv[2]); i = txt.index(":\n", i); j = txt.index(".Lfunc_end", i)
    body = txt[i:j].split("\n")
    # loop = from the first "Loop Header" label to the last branch back to it
    hdr = next(k for k, l in enumerate(body) if "Loop Header" in l)
    label = body[hdr].split(":")[0].strip()
    tag = "Header=" + label.lstrip(".L")
    end = max(k for k, l in enumerate(body) if tag in l)
    while end + 1 < len(body) and not body[end + 1].strip().startswith((".LBB", ";")):   # to the end of that block
        end += 1
    ins = []
    for k in range(hdr, end + 1):
        t = body[k].split(";")[0].strip()
        if not t or t.startswith("."):
            continue
        op, _, rest = t.partition(" ")
        ops = [o.strip() for o in rest.split(",")] if rest else []
        stores = op.startswith(("global_store", "buffer_store", "ds_write", "scratch_store", "v_cmp", "s_", "ds_add"))
        if stores or not ops:
            d, u = set(), set().union(*[regs(o) for o in ops]) if ops else set()
        else:
            d = regs(ops[0]); u = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
            if op.startswith("v_mfma") or op.startswith(("v_fmac", "v_mac", "v_pk_fmac")):
                u |= regs(ops[0]) if op.startswith(("v_fmac", "v_mac", "v_pk_fmac")) else set()
        ins.append((k, op, d, u, t))
    live = set()
    for _ in range(2):                              # two passes: around the back edge
        prof = []
        for k, op, d, u, t in reversed(ins):
            live = (live - d) | u
            prof.append((k, len(live), op, t))
    prof.reverse()
    peak = max(p[1] for p in prof)
    print("loop %s: %d instructions, peak live VGPRs (linear approx) %d" % (label, len(prof), peak))
    last = None
    for k, n, op, t in prof:
        mark = op.startswith(("v_mfma", "s_barrier", "buffer_load", "global_load", "scratch_", "global_store", "ds_read_b64_tr", "ds_write_b128"))
        if mark:
            key = op if op.startswith("v_mfma") else None
            if key and key == last:
                continue
            last = key
            print("%5d live %3d  %s" % (k, n, t[:80]))
        elif n >= peak - 2:
            print("%5d live %3d  * %s" % (k, n, t[:80]))
            last = None


if __name__ == "__main__":
    main()
