#!/usr/bin/env python3
"""Kernel micro-bench: times the forward / backward C-ABI calls separately with HIP events.
    python tools/kbench.py [--B 4096] [--iters 20] [--what fwd,fwd_nogates,bwd] [--H 128 --F 32]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from kws_amd import fastgrnn_cuda  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=4096)
    ap.add_argument("--T", type=int, default=99)
    ap.add_argument("--H", type=int, default=128)
    ap.add_argument("--F", type=int, default=32)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--what", default="fwd,fwd_nogates,bwd")
    ap.add_argument("--flags", type=int, default=0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    T, B, F, H = a.T, a.B, a.F, a.H
    torch.manual_seed(0)
    e = torch.empty(0)
    if a.rank:
        w, u = e, e
        w1 = 0.1 * torch.randn(a.rank, F, device=dev); w2 = 0.1 * torch.randn(H, a.rank, device=dev)
        u1 = 0.1 * torch.randn(a.rank, H, device=dev); u2 = 0.1 * torch.randn(H, a.rank, device=dev)
    else:
        w = 0.1 * torch.randn(H, F, device=dev); u = 0.1 * torch.randn(H, H, device=dev)
        w1 = w2 = u1 = u2 = e
    bz = torch.ones(1, H, device=dev); bh = torch.ones(1, H, device=dev)
    zeta = torch.ones(1, 1, device=dev); nu = -4 * torch.ones(1, 1, device=dev)
    x = torch.randn(T, B, F, device=dev); G = torch.randn(T, B, H, device=dev)
    h0 = torch.zeros(B, H, device=dev)

    def fwd(gates=True):
        return fastgrnn_cuda.forward_unroll(x, w, u, bz, bh, zeta, nu, h0, 0, w1, w2, u1, u2,
                                            want_gates=gates, flags=a.flags)

    outs = fwd()
    preact = bool(a.flags & 4)
    hs, zs = outs[0], outs[1]
    cs = outs[2] if len(outs) > 2 else outs[1]

    def bwd():
        return fastgrnn_cuda.backward_unroll(G, x, hs, zeta, nu, w, u, zs, cs, h0, w1, w2, u1, u2, 0, flags=a.flags,
                                             bias_gate=bz if preact else None, bias_update=bh if preact else None)

    fns = {"fwd": lambda: fwd(True), "fwd_nogates": lambda: fwd(False), "bwd": bwd}
    for name in a.what.split(","):
        fn = fns[name]
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(a.iters):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        med = ts[len(ts) // 2]
        print("%-12s B=%d H=%d F=%d r=%d: median %.1f us  min %.1f us  -> %.3g utt/s  (%.2f us/step)"
              % (name, B, H, F, a.rank, med * 1e3, ts[0] * 1e3, B / (med * 1e-3), med * 1e3 / T), flush=True)


if __name__ == "__main__":
    main()
