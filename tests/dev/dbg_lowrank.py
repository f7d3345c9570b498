"""Low-rank forward: every (saved-tensor contract, wave shape) against the float64 oracle, twice."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from kws_amd import fastgrnn_cuda
from oracle import fastgrnn_oracle as O
DEV = torch.device("cuda:0")
F, H, r = 32, 256, 16
for T, B in ((15, 32), (15, 21), (40, 64), (99, 4096)):
    p = O.make_params(F, H, r, r, seed=51, randomize_scalars=True)
    g = torch.Generator().manual_seed(52)
    x = torch.randn(T, B, F, generator=g); h0 = 0.3 * torch.randn(B, H, generator=g)
    nb = min(B, 64)
    p64 = {k: (v.astype(np.float64) if isinstance(v, np.ndarray) else v) for k, v in p.items()}
    ref = O.unroll_forward(x[:, :nb].numpy().astype(np.float64), p64, h0[:nb].numpy().astype(np.float64))[0]
    P = {k: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in p.items() if isinstance(v, np.ndarray)}
    e = torch.empty(0)
    for fl in (0, 8, 4, 12):
        for want in ((True, False) if fl in (0, 8) else (True,)):
            for rep in range(2):
                hs = fastgrnn_cuda.forward_unroll(x.to(DEV), e, e, P["bias_gate"], P["bias_update"], P["zeta"], P["nu"],
                                                  h0.to(DEV), 0, P["w1"], P["w2"], P["u1"], P["u2"], want_gates=want, flags=fl)[0]
                d = np.abs(hs[:, :nb].cpu().numpy().astype(np.float64) - ref)
                per_t = d.reshape(T, -1).max(1)
                print("T=%d B=%d flags=%2d gates=%d rep%d  max %.2e  first t>1e-4: %s   %s" % (
                    T, B, fl, want, rep, d.max(), next((t for t in range(T) if per_t[t] > 1e-4), None),
                    " ".join("%.0e" % v for v in per_t[::max(1, T // 6)])), flush=True)
