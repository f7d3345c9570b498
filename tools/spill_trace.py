#!/usr/bin/env python3
"""Condensed trace of one kernel's assembly: spills / reloads, global and LDS-transposed memory instructions, barriers,
branches and MFMA runs, with line numbers -- to see WHERE in a scan step the register allocator ran out.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -Iinclude kws_amd/csrc/X.hip -o /tmp/x.s
    python tools/spill_trace.py /tmp/x.s <mangled-name-substring> [--loop]
"""
import sys


def main():
    txt = open(sys.argv[1]).read()
    key = sys.argv[2]
    only_loop = "--loop" in sys.argv
    i = txt.index(key)
    i = txt.index(":\n", i)
    j = txt.index(".Lfunc_end", i)
    body = txt[i:j].split("\n")
    out = []
    in_loop = False
    for k, ln in enumerate(body):
        t = ln.strip()
        if not t or t.startswith(";"):
            continue
        if t.startswith("."):
            if t.startswith(".LBB"):
                if "Loop Header" in t:
                    in_loop = True
                out.append((k, t, in_loop))
            continue
        op = t.split()[0]
        if op.startswith("v_mfma"):
            if out and out[-1][1].startswith("MFMA x"):
                out[-1] = (out[-1][0], "MFMA x%d" % (int(out[-1][1].split("x")[1]) + 1), in_loop)
            else:
                out.append((k, "MFMA x1", in_loop))
        elif op.startswith(("scratch_", "global_", "s_cbranch", "ds_read_b64_tr", "s_barrier", "buffer_")):
            out.append((k, t[:100], in_loop))
            if op.startswith("s_cbranch") and in_loop and "Loop" not in t:
                pass
    for k, t, il in out:
        if only_loop and not il:
            continue
        print(k, t)


if __name__ == "__main__":
    main()
