"""Bitwise repeatability of a SEQUENCE of different shapes run back to back (as a test suite does): every pass runs
all shapes in order; outputs are compared with the first pass.  Usage: python tools/repro_seq.py [passes]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kws_amd import fastgrnn_cuda, _lib
dev = torch.device("cuda:0")
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 30
e = torch.empty(0)
SHAPES = [(99, 64, 256, 128, 1), (99, 50, 256, 128, 1), (23, 37, 256, 128, 0), (12, 16, 128, 128, 1), (7, 33, 64, 128, 1),
          (1, 5, 256, 128, 1), (2, 1, 256, 128, 0), (6, 130, 128, 128, 0), (99, 32, 32, 256, 1), (24, 37, 32, 256, 1),
          (9, 16, 32, 256, 0), (7, 33, 64, 128, 1), (5, 37, 32, 128, 1), (7, 33, 64, 128, 0), (8, 17, 64, 128, 1),
          (8, 33, 128, 128, 1), (7, 50, 256, 128, 0), (7, 32, 64, 128, 1), (7, 33, 32, 128, 1), (7, 33, 32, 128, 0),
          # low-rank (6th entry = rank) and H=256 ragged batches
          (9, 33, 32, 256, 1, 16), (7, 50, 32, 256, 1, 8), (8, 17, 32, 256, 1, 16), (9, 33, 32, 256, 1), (8, 50, 32, 256, 0),
          (7, 37, 32, 256, 1, 32)]                     # rank 32: factors multiplied out, dense H=256 kernels
if os.environ.get("REPRO_BIG"):
    # full-size ragged batches (255 full workgroups + one with 10 utterances): what the last batch of an epoch looks like
    SHAPES += [(99, 4090, 32, 128, 1), (99, 4090, 32, 128, 0), (99, 4090, 256, 128, 1), (99, 4090, 32, 256, 1),
               (99, 4090, 32, 256, 1, 16), (99, 4096, 32, 128, 1), (99, 4096, 32, 256, 1, 16)]


def make(T, B, F, H, seed, r=0):
    torch.manual_seed(seed)
    if r:
        fac = dict(w=e, u=e, w1=0.1 * torch.randn(r, F, device=dev), w2=0.1 * torch.randn(H, r, device=dev),
                   u1=0.1 * torch.randn(r, H, device=dev), u2=0.1 * torch.randn(H, r, device=dev))
    else:
        fac = dict(w=0.1 * torch.randn(H, F, device=dev), u=0.1 * torch.randn(H, H, device=dev), w1=e, w2=e, u1=e, u2=e)
    return dict(fac,
                bz=torch.randn(1, H, device=dev), bh=torch.randn(1, H, device=dev), zeta=torch.ones(1, 1, device=dev),
                nu=-4 * torch.ones(1, 1, device=dev), x=torch.randn(T, B, F, device=dev),
                h0=0.3 * torch.randn(B, H, device=dev), G=torch.randn(T, B, H, device=dev))


def run(d, fl):
    outs = fastgrnn_cuda.forward_unroll(d["x"], d["w"], d["u"], d["bz"], d["bh"], d["zeta"], d["nu"], d["h0"], 0,
                                        d["w1"], d["w2"], d["u1"], d["u2"], flags=fl)
    g = fastgrnn_cuda.backward_unroll(d["G"], d["x"], outs[0], d["zeta"], d["nu"], d["w"], d["u"], outs[1], outs[-1], d["h0"],
                                      d["w1"], d["w2"], d["u1"], d["u2"], 0, flags=fl, bias_gate=d["bz"] if fl & 4 else None,
                                      bias_update=d["bh"] if fl & 4 else None)
    return [o for o in list(outs) + list(g) if o.numel()]


BF16 = bool(os.environ.get("REPRO_BF16"))              # bf16 sequences (one-saved-tensor contract; dense shapes)
if BF16:
    SHAPES = [(25, 40, 256, 128, 1), (25, 40, 32, 128, 1), (25, 40, 32, 256, 1), (25, 40, 64, 256, 1), (99, 64, 256, 128, 1),
              (23, 37, 256, 128, 1), (12, 16, 128, 128, 1), (7, 33, 64, 128, 1), (31, 48, 256, 128, 1), (31, 37, 64, 256, 1),
              (9, 33, 32, 256, 1), (31, 64, 32, 128, 1), (8, 17, 64, 128, 1), (7, 50, 256, 128, 1)]
if os.environ.get("REPRO_SHORT"):
    SHAPES = SHAPES[:15]
POISON = [int(v, 0) for v in os.environ.get("REPRO_POISON", "").split(",") if v]
data = [make(sh[0], sh[1], sh[2], sh[3], 10 + k, sh[5] if len(sh) > 5 else 0) for k, sh in enumerate(SHAPES)]
if BF16:
    for d_ in data:
        d_["x"], d_["G"] = d_["x"].to(torch.bfloat16), d_["G"].to(torch.bfloat16)
first = {}
nbad = 0
per_shape = {}
for ps in range(passes):
    for k, sh in enumerate(SHAPES):
        T, B, F, H, p = sh[:5]
        # fresh tensors every pass, like a test does (allocator reuse)
        d = {n: (v.clone() if v.numel() else v) for n, v in data[k].items()}
        if POISON:
            st = _lib.load().fastgrnn_hip_debug_poison_cu_state(POISON[ps % len(POISON)], None)
            assert st == 0, st
        outs = run(d, 4 if p else 0)
        torch.cuda.synchronize()
        if k not in first:
            first[k] = [o.clone() for o in outs]
            continue
        diff = [int((a != b).sum()) for a, b in zip(outs, first[k])]
        if any(diff):
            nbad += 1
            per_shape[k] = per_shape.get(k, 0) + 1
            if nbad <= int(os.environ.get("REPRO_SHOW", "8")):
                print("pass %d shape %s: differing element counts %s" % (ps, SHAPES[k], diff), flush=True)
                names = ["hs", "aux0", "aux1"][:2 if p else 3] + ["d_x", "d_bz", "d_bh", "d_zeta", "d_nu", "d_h0", "d_w", "d_u"]
                for n, a, b in zip(names, outs, first[k]):
                    if n in ("d_h0", "d_x"):
                        a2 = a.reshape(-1, a.shape[-1]); b2 = b.reshape(-1, b.shape[-1])
                        rows = ((a2 != b2).sum(1) > 0).nonzero().flatten().tolist()
                        print("   %s rows differing: %s  max|diff| %.3e  max|ref| %.3e" % (
                            n, rows[:24], float((a2 - b2).abs().max()), float(b2.abs().max())), flush=True)
for k, c in sorted(per_shape.items()):
    print("shape #%d %s: %d of %d runs differ from the first" % (k, SHAPES[k], c, passes - 1))
print("%d differing runs in %d passes" % (nbad, passes))
sys.exit(1 if nbad else 0)
