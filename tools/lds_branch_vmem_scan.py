#!/usr/bin/env python3
"""Sites of the pattern  LDS instruction ... conditional branch ... vector-memory instruction  with no
`s_waitcnt lgkmcnt(0)` (or barrier) between the LDS instruction and the branch, inside loops of gfx950 assembly.

    python tools/lds_branch_vmem_scan.py file.s [window=12]

Background: the ragged wide-layer backward had exactly this in its scan loop -- three ds_write2st64,
`s_and_saveexec` + `s_cbranch_execz`, address arithmetic, `global_store` -- and returned wrong results in some
processes.  Any ONE of these removed the failures in interleaved A/B runs (0 of 10 processes against 6 of 10): idle
cycles in front of the block; idle cycles behind the store; `s_waitcnt lgkmcnt(0)` in front of the block; the address
arithmetic hoisted out of the block; the block moved in front of the LDS writes.  gfx10 has a documented hazard of this
shape (LLVM's LdsBranchVmemWARHazard); nothing is documented for gfx9.  The shipped scans therefore keep conditional
branches away from memory instructions in their loops; this scanner lists what is left.  Exit status 0 always
(informational); prints one line per kernel with a finding.
"""
import re
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
path = args[0]
W = int(args[1]) if len(args) > 1 else 12
VMEM = ("global_", "buffer_", "flat_", "scratch_")


def kernels(path):
    name, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        if line.startswith(".Lfunc_end"):
            yield name, body
            name = None
            continue
        t = line.split(";")[0].strip()
        if t and (not t.startswith(".") or re.match(r"^\.L\w+:", t)):
            body.append(t)


REG = re.compile(r"\bv(?:\[(\d+):(\d+)\]|(\d+)\b)")


def vregs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def scc(succ):
    """component id per node for nodes on a cycle (Tarjan, iterative); -1 for the others"""
    n = len(succ)
    index = [None] * n
    low = [0] * n
    on = [False] * n
    comp = [-1] * n
    stack, counter, ncomp = [], 0, 0
    for root in range(n):
        if index[root] is not None:
            continue
        work = [(root, 0)]
        while work:
            v, pi = work.pop()
            if pi == 0:
                index[v] = low[v] = counter
                counter += 1
                stack.append(v)
                on[v] = True
            recurse = False
            for qi in range(pi, len(succ[v])):
                w = succ[v][qi]
                if w >= n:
                    continue
                if index[w] is None:
                    work.append((v, qi + 1))
                    work.append((w, 0))
                    recurse = True
                    break
                if on[w]:
                    low[v] = min(low[v], index[w])
            if recurse:
                continue
            if low[v] == index[v]:
                members = []
                while True:
                    w = stack.pop()
                    on[w] = False
                    members.append(w)
                    if w == v:
                        break
                if len(members) > 1 or v in succ[v]:
                    for w in members:
                        comp[w] = ncomp
                    ncomp += 1
            if work:
                u = work[-1][0]
                low[u] = min(low[u], low[v])
    return comp


NARROW = "--narrow" in sys.argv      # only: LDS WRITE whose data registers a VALU instruction behind the branch overwrites
WRITES = "--writes" in sys.argv      # any LDS WRITE in front of the branch (whatever happens to its data registers behind it)
# --strict: --writes with NO window in front of the branch: the look-back runs to the nearest `s_waitcnt lgkmcnt(0)`,
# barrier or label (a label = a join: another path may arrive there with its own pending writes, so the walk
# continues above it only if everything in between is fall-through code).  This is the form tests/test_operand_rule_cpu.py
# requires to be empty in every kernel of the library since round 3: an LDS write may never be pending when a conditional
# branch -- exec-masked or wave-uniform -- with a vector-memory instruction behind it is taken.
STRICT = "--strict" in sys.argv
if STRICT:
    WRITES = True
total = 0
for name, body in kernels(path):
    labels = {}
    ins = []
    for t in body:
        m = re.match(r"^(\.L\w+):", t)
        if m:
            labels[m.group(1)] = len(ins)
        else:
            ins.append(t)
    # loop ranges: a branch back to an earlier label
    loops = []
    for j, t in enumerate(ins):
        m = re.match(r"s_c?branch\w*\s+(\.L\w+)", t)
        if m and m.group(1) in labels and labels[m.group(1)] <= j:
            loops.append((labels[m.group(1)], j))
    hits = []
    if STRICT:
        # control-flow graph of the kernel: successors / predecessors by instruction index
        n = len(ins)
        succ = [[] for _ in range(n)]
        for j, t in enumerate(ins):
            m = re.match(r"s_(c?)branch\w*\s+(\.L\w+)", t)
            if t.startswith("s_endpgm"):
                continue
            if m and m.group(2) in labels:
                succ[j].append(labels[m.group(2)])
                if m.group(1) and j + 1 < n:
                    succ[j].append(j + 1)
            elif j + 1 < n:
                succ[j].append(j + 1)
        pred = [[] for _ in range(n)]
        for j in range(n):
            for k in succ[j]:
                if k < n:
                    pred[k].append(j)
        # loops = strongly connected components of the graph (a backward jump alone is not one: the compiler parks
        # blocks behind the kernel's end and jumps back from them)
        comp = scc(succ)
        for j, t in enumerate(ins):
            if not t.startswith("s_cbranch"):
                continue
            if comp[j] < 0:
                continue
            # a vector-memory instruction within W instructions behind the branch, along either arm
            vm, seen, front = None, set(), [(k, 1) for k in succ[j]]
            while front and vm is None:
                k, d = front.pop()
                if k in seen or k >= n or d > W:
                    continue
                seen.add(k)
                if ins[k].startswith(VMEM):
                    vm = ins[k]
                    break
                front.extend((q, d + 1) for q in succ[k])
            if vm is None:
                continue
            # an LDS write on some path to the branch with no `s_waitcnt lgkmcnt(0)` / barrier behind it (paths are
            # followed backwards inside the loop; its entry from outside ends a path)
            lds, seen, front = None, set(), list(pred[j])
            while front and lds is None:
                k = front.pop()
                if k in seen or comp[k] != comp[j]:
                    continue
                seen.add(k)
                u = ins[k]
                if u.startswith("s_barrier") or (u.startswith("s_waitcnt") and "lgkmcnt(0)" in u):
                    continue
                if u.startswith(("ds_write", "ds_store", "ds_add", "ds_max", "ds_min")):
                    lds = u
                    break
                front.extend(pred[k])
            if lds:
                hits.append((lds.split()[0], t.split()[0], vm.split()[0]))
    for j, t in enumerate(ins):
        if STRICT:
            break
        if not t.startswith("s_cbranch"):
            continue
        if not any(a <= j <= b for a, b in loops):
            continue
        loop_lo = max(a for a, b in loops if a <= j <= b)        # innermost enclosing loop's first instruction
        lds = None
        data = set()
        for k in range(j - 1, (loop_lo - 1) if STRICT else max(-1, j - 1 - W), -1):
            u = ins[k]
            if u.startswith("s_barrier") or (u.startswith("s_waitcnt") and "lgkmcnt(0)" in u):
                break
            if u.startswith("ds_") and (not (NARROW or WRITES) or u.startswith(("ds_write", "ds_store", "ds_add", "ds_max", "ds_min"))):
                if lds is None:
                    lds = u
                ops = u.split(None, 1)[1].split(",") if " " in u else []
                for o in ops[1:]:
                    data |= vregs(o)
                if not (NARROW or WRITES):
                    break
        if lds is None:
            continue
        if NARROW:
            clobber = None
            for k in range(j + 1, min(len(ins), j + 1 + W)):
                u = ins[k]
                if u.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_waitcnt")):
                    break
                if u.startswith("v_") and not u.startswith("v_cmp"):
                    ops = u.split(None, 1)[1].split(",") if " " in u else []
                    if ops and (vregs(ops[0]) & data):
                        clobber = u
                        break
            if clobber is None:
                continue
        vm = None
        for k in range(j + 1, min(len(ins), j + 1 + W)):
            u = ins[k]
            if u.startswith(VMEM):
                vm = u
                break
            if u.startswith(("s_cbranch", "s_branch", "s_endpgm")):
                break
        if vm:
            hits.append((lds.split()[0], t.split()[0], vm.split()[0]))
    if hits:
        total += len(hits)
        short = re.sub(r"^_ZN8fastgrnn12_GLOBAL__N_1\d+", "", name)[:60]
        print("%-62s %2d  e.g. %s -> %s -> %s" % (short, len(hits), *hits[0]))
print("sites in loops:", total)
