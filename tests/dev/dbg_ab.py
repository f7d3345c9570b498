"""Which forward variant is off?  Each against the float64 oracle on the A/B test's inputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from kws_amd import fastgrnn_cuda
from oracle import fastgrnn_oracle as O
DEV = torch.device("cuda:0")
T, F, H = 40, 32, 128
e = torch.empty(0)
for B in (48, 37, 64):
    p = O.make_params(F, H, seed=12, randomize_scalars=True)
    g = torch.Generator().manual_seed(14)
    x = torch.randn(T, B, F, generator=g); G = torch.randn(T, B, H, generator=g); h0 = 0.3 * torch.randn(B, H, generator=g)
    p64 = {k: (v.astype(np.float64) if isinstance(v, np.ndarray) else v) for k, v in p.items()}
    ref = O.unroll_forward(x.numpy().astype(np.float64), p64, h0.numpy().astype(np.float64))[0]
    P = {k: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in p.items() if isinstance(v, np.ndarray)}
    assert "w" in P and "u" in P, list(P)
    for fl in (0, 4, 8, 12, 64, 72):
        for rep in range(2):
            hs = fastgrnn_cuda.forward_unroll(x.to(DEV), P["w"], P["u"],
                                              P["bias_gate"], P["bias_update"], P["zeta"], P["nu"], h0.to(DEV), 0,
                                              e, e, e, e, flags=fl)[0]
            d = np.abs(hs.cpu().numpy().astype(np.float64) - ref)
            per_t = d.reshape(T, -1).max(1)
            print("B=%d flags=%3d rep%d  max %.3e  first bad t %s  per-t %s" % (
                B, fl, rep, d.max(), next((t for t in range(T) if per_t[t] > 1e-4), None),
                " ".join("%.1e" % v for v in per_t[::6])), flush=True)
