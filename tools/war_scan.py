#!/usr/bin/env python3
"""Static check for the MFMA operand hazard: a load (LDS / global / scratch) whose
destination is a register that an earlier MFMA reads as A or B, with no read of that MFMA's (or a younger MFMA's)
result in between -- MFMAs retire in order, so such a read proves the matrix pipe has drained past it.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -o /tmp/ks.s kws_amd/csrc/kernels_split.hip
    python tools/war_scan.py /tmp/ks.s [max_steps=6000]

Every MFMA is followed along the control-flow graph of the compiler's assembly -- both arms of every conditional
branch, through unconditional branches and around loops -- until each path reaches a completion read or the end
of the program; there is no fixed look-ahead window.  A path that is still open after ``max_steps`` instructions
is reported as UNBOUNDED and counts as a failure.  Prints one line per kernel with the number of violating
(MFMA, load) pairs (0 everywhere is the goal) and the closest few; ``--list`` prints only the scanned kernel names.
"""
import re
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
path = args[0]
MAX_STEPS = int(args[1]) if len(args) > 1 else 6000
LIST_ONLY = "--list" in sys.argv
NO_PRUNE = "--no-prune" in sys.argv      # follow both arms of every branch (reports paths the scalar flags rule out)
ONLY = next((a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")), None)      # kernel-name substring
TRACE = next((a.split("=", 1)[1] for a in sys.argv if a.startswith("--trace=")), None)   # kernel-name substring
LOADS = ("ds_read", "ds_load", "global_load", "buffer_load", "scratch_load", "flat_load")
STORES = ("global_store", "ds_write", "ds_store", "buffer_store", "scratch_store", "flat_store", "v_cmp", "v_cmpx",
          "global_atomic", "ds_add", "ds_max")

REG = re.compile(r"\b([va])(?:\[(\d+):(\d+)\]|(\d+)\b)")


def regs(tok, kinds="va"):
    """register numbers named in one operand string; accumulation registers are offset by 1000"""
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) not in kinds:
            continue
        base = 1000 if m.group(1) == "a" else 0
        if m.group(2) is not None:
            out.update(base + r for r in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add(base + int(m.group(4)))
    return out


class Ins:
    __slots__ = ("text", "kind", "src", "dst", "ab", "target", "sdst", "sconst", "vccop", "cond")


SREG = re.compile(r"\b(?:s\[(\d+):(\d+)\]|s(\d+)\b|(vcc)\b)")


def sregs(tok):
    """scalar registers named in one operand string (vcc = 'vcc')"""
    out = set()
    for m in SREG.finditer(tok):
        if m.group(4):
            out.add("vcc")
        elif m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def parse_kernel(lines):
    """lines: (label or None, instruction text).  Returns the instruction list with register sets and branch targets."""
    labels, ins = {}, []
    for lab, text in lines:
        if lab is not None:
            labels[lab] = len(ins)
            continue
        x = Ins()
        x.text = text
        parts = text.split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        x.target = None
        x.ab = set()
        # scalar side, for pruning infeasible paths: the structurizer guards the arms of a wave-uniform if / else-if
        # chain with flags ("s_mov_b64 s[a:b], -1 ... s_andn2_b64 vcc, exec, s[a:b]; s_cbranch_vccnz L"); following
        # such a branch against a flag set on the same path would report paths that cannot execute
        x.sdst = sregs(ops[0]) if ops and not op.startswith(("s_cbranch", "s_branch", "s_cmp", "s_waitcnt", "s_nop", "s_barrier")) else set()
        x.sconst = None            # (first sgpr of the pair, value) for s_mov_b64 s[a:b], 0 / -1
        x.vccop = None             # ("and" | "andn2", first sgpr) for s_and(n2)_b64 vcc, exec, s[a:b]
        x.cond = None              # "vccz" | "vccnz"
        if op == "s_mov_b64" and len(ops) == 2 and ops[1] in ("0", "-1"):
            m2 = re.match(r"s\[(\d+):(\d+)\]$", ops[0])
            if m2:
                x.sconst = (int(m2.group(1)), int(ops[1]))
        if op in ("s_and_b64", "s_andn2_b64") and len(ops) == 3 and ops[0] == "vcc" and ops[1] == "exec":
            m2 = re.match(r"s\[(\d+):(\d+)\]$", ops[2])
            if m2:
                x.vccop = ("andn2" if op == "s_andn2_b64" else "and", int(m2.group(1)))
        if op in ("s_cbranch_vccz", "s_cbranch_vccnz"):
            x.cond = op[len("s_cbranch_"):]
        if op.startswith("v_mfma") or op.startswith("v_smfmac"):
            x.kind = "mfma"
            x.dst = regs(ops[0])
            # every register the matrix pipe fetches: A, B and the C input (a C that is the result of an older MFMA
            # is forwarded inside the pipe, but one written by the VALU -- the chain's z*g -- is fetched like A and B)
            x.ab = regs(ops[1]) | regs(ops[2]) | (regs(ops[3]) if len(ops) > 3 else set())
            x.src = set(x.ab)
        elif op.startswith(LOADS):
            x.kind = "load"
            x.dst = regs(ops[0]) if ops else set()
            x.src = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
        elif op.startswith(STORES):
            x.kind = "other"
            x.dst = set()
            x.src = set().union(*[regs(o) for o in ops]) if ops else set()
        elif op in ("s_endpgm", "s_setpc_b64", "s_trap"):
            x.kind = "end"
            x.dst = x.src = set()
        elif op == "s_branch":
            x.kind = "jump"
            x.target = ops[0]
            x.dst = x.src = set()
        elif op.startswith("s_cbranch"):
            x.kind = "fork"
            x.target = ops[0]
            x.dst = x.src = set()
        else:
            x.kind = "other"
            x.dst = regs(ops[0]) if ops else set()
            x.src = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
        ins.append(x)
    for x in ins:
        if x.target is not None:
            x.target = labels.get(x.target)         # None: a target outside the kernel ends the path
    return ins, set(labels.values())


def scratch_in_loops(lines):
    """Spill traffic inside a loop: (label, instruction) pairs for every scratch_* instruction that sits between a
    label and a later branch back to it.  The scans are written to run without spills (a spill reload is a load the
    author never placed, and the one build of the 8-wave backward that spilled inside its loop was the one that gave
    run-to-run differences on the GPU: DESIGN.md 4.0)."""
    pos, out = {}, []
    texts = []
    for lab, text in lines:
        if lab is not None:
            pos[lab] = len(texts)
        else:
            texts.append(text)
    for j, text in enumerate(texts):
        m = re.match(r"s_c?branch\w*\s+(\.L\w+)", text)
        if m and m.group(1) in pos and pos[m.group(1)] <= j:
            for k in range(pos[m.group(1)], j):
                if texts[k].startswith("scratch_"):
                    out.append((m.group(1), texts[k]))
    return sorted(set(out))


def track_scalars(x, known):
    """apply instruction x to the dict of known scalar flags (see parse_kernel)"""
    if not x.sdst:
        return
    vcc_val = None
    if x.vccop is not None and x.vccop[1] in known:
        flag = known[x.vccop[1]]
        vcc_val = (0 if flag == -1 else 1) if x.vccop[0] == "andn2" else (1 if flag == -1 else 0)
    for r in x.sdst:
        known.pop(r, None)
        if isinstance(r, int):
            known.pop(r - 1, None)                  # a write to the upper half of a tracked pair
    if x.sconst is not None:
        known[x.sconst[0]] = x.sconst[1]
    if vcc_val is not None:
        known["vcc"] = vcc_val


TRACE_ON = [False]


def scan_kernel(ins, block_starts):
    """-> (number of MFMAs, hits [(distance, mfma text, load text)], unbounded count)"""
    hits, unbounded = [], 0
    n = len(ins)
    for i, m in enumerate(ins):
        if m.kind != "mfma":
            continue
        ab = m.ab
        # depth-first over paths; a state is (next instruction, registers whose read proves retirement, known scalar
        # flags: {first sgpr of a pair: 0 / -1}, "vcc": 0 (zero) / 1 (non-zero: exec is non-zero in a running wave))
        # flags already established in the MFMA's own basic block (straight-line code from the nearest label or
        # branch above it)
        k0 = i
        while k0 > 0 and k0 not in block_starts and ins[k0 - 1].kind not in ("fork", "jump", "end"):
            k0 -= 1
        known0 = {}
        for k1 in range(k0, i + 1):
            track_scalars(ins[k1], known0)
        stack = [(i + 1, frozenset(m.dst), frozenset(known0.items()), 1)]
        trail = []
        seen = set()
        steps = 0
        found = None
        while stack and found is None:
            j, done, known_f, dist = stack.pop()
            known = dict(known_f)
            while True:
                if j >= n:
                    break
                key = (j, done, frozenset(known.items()))
                if key in seen:
                    break
                seen.add(key)
                steps += 1
                if steps > MAX_STEPS:
                    unbounded += 1
                    stack = []
                    break
                x = ins[j]
                k = x.kind
                trail.append((j, x.text))
                track_scalars(x, known)
                if k == "mfma":
                    # a younger MFMA that CONSUMES a result (as A/B/C) does not prove retirement by itself (the pipe
                    # is in order: it simply queues), but reading its own result later does
                    done = done | x.dst
                    j += 1; dist += 1
                    continue
                if k == "end":
                    break
                if k == "jump":
                    if x.target is None:
                        break
                    j = x.target; dist += 1
                    continue
                if k == "fork":
                    take, fall = True, True
                    if x.cond is not None and "vcc" in known and not NO_PRUNE:
                        nz = known["vcc"] == 1
                        take = nz if x.cond == "vccnz" else not nz
                        fall = not take
                    if take and x.target is not None:
                        if fall:
                            stack.append((x.target, done, frozenset(known.items()), dist + 1))
                        else:
                            j = x.target; dist += 1
                            continue
                    if not fall:
                        break
                    j += 1; dist += 1
                    continue
                if x.src & done:
                    break                            # completion read: the matrix pipe has drained past MFMA i
                if x.dst & done:
                    # the register no longer holds an MFMA result once something else has written it: a later read of
                    # it proves nothing about the matrix pipe (a result consumed only as the C operand of a younger
                    # MFMA and then reused as a load destination would otherwise pass for a completion read)
                    done = done - x.dst
                if k == "load" and (x.dst & ab):
                    found = (dist, m.text, x.text)
                    if TRACE_ON[0]:
                        TRACE_ON[0] = False
                        print("---- trace from [%d] %s" % (i, m.text))
                        for jj, tt in trail:
                            if tt.startswith(("s_cbranch", "s_branch", "v_accvgpr_read", "s_barrier", "ds_write", "global_store")) or jj == j:
                                print("   [%d] %s" % (jj, tt))
                    break
                j += 1; dist += 1
        if found is not None:
            hits.append(found)
    return sum(1 for x in ins if x.kind == "mfma"), hits, unbounded


kernels, cur, name = {}, None, None
declared = set()                                 # every kernel the object declares (.amdhsa_kernel directives)
for line in open(path):
    m = re.match(r"^\s*\.amdhsa_kernel\s+(\S+)", line)
    if m:
        declared.add(m.group(1))
    m = re.match(r"^(_Z\w+):", line)
    if m:
        name = m.group(1)
        cur = kernels.setdefault(name, [])
        continue
    if cur is None:
        continue
    m = re.match(r"^(\.L\w+):", line)
    if m:
        if m.group(1).startswith(".Lfunc_end"):
            cur = None                           # end of this kernel's text
        else:
            cur.append((m.group(1), None))
        continue
    s = line.strip()
    if not line.startswith("\t") or not s or s[0] in ".;":
        continue
    text = s.split(";")[0].strip()
    cur.append((None, text))

if LIST_ONLY:
    for name in kernels:
        print(name)
    sys.exit(0)

bad_total = 0
for name, lines in kernels.items():
    if ONLY and ONLY not in name:
        continue
    ins, block_starts = parse_kernel(lines)
    TRACE_ON[0] = bool(TRACE) and TRACE in name
    n_mfma, hits, unbounded = scan_kernel(ins, block_starts)
    spills = scratch_in_loops(lines) if n_mfma else []
    if True:
        short = re.sub(r"^_ZN8fastgrnn12_GLOBAL__N_1\d+", "", name)[:48]
        print("%-50s mfma %4d  load-behind-mfma pairs %3d%s%s" % (
            short, n_mfma, len(hits), ("   closest +%d" % min(h[0] for h in hits)) if hits else "",
            ("   UNBOUNDED paths %d" % unbounded) if unbounded else ""))
        for h in sorted(hits)[:(40 if ONLY else 3)]:
            print("      +%d  %s   <-   %s" % h)
        if spills:
            print("      SPILLS IN LOOP %d, e.g. %s: %s" % (len(spills), spills[0][0], spills[0][1]))
        bad_total += len(hits) + unbounded + (1 if spills else 0)
missing = sorted(declared - set(kernels))
print("kernels declared %d, scanned %d%s" % (len(declared), len(kernels), (", NOT SCANNED: " + " ".join(missing)) if missing else ""))
if missing:
    bad_total += len(missing)
print("total pairs:", bad_total)
sys.exit(1 if bad_total else 0)
