"""Is the split-precision product biased?  pre = W.x with U = 0 against fp64: mean SIGNED relative error."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from kws_amd import fastgrnn_cuda
dev = torch.device("cuda:0")
T, B, F, H = 8, 256, 32, 128
e = torch.empty(0)
rng = np.random.default_rng(0)
for tag, wgen, xgen in (("all positive", lambda: np.abs(rng.standard_normal((H, F))), lambda: np.abs(rng.standard_normal((T, B, F)))),
                        ("mixed signs", lambda: rng.standard_normal((H, F)), lambda: rng.standard_normal((T, B, F)))):
    w = wgen().astype(np.float32); x = xgen().astype(np.float32)
    ref = np.einsum("tbf,hf->tbh", x.astype(np.float64), w.astype(np.float64))
    P = dict(w=torch.from_numpy(w).to(dev), u=torch.zeros(H, H, device=dev), bz=torch.zeros(1, H, device=dev),
             bh=torch.zeros(1, H, device=dev), zeta=torch.ones(1, 1, device=dev), nu=torch.ones(1, 1, device=dev))
    h0 = torch.zeros(B, H, device=dev)
    for name, fl in (("split (w8)", 4), ("split 4-wave", 4 | 8), ("fp32 matmul (torch)", None)):
        if fl is None:
            pre = (torch.from_numpy(x).to(dev).reshape(-1, F) @ P["w"].t()).reshape(T, B, H).cpu().numpy().astype(np.float64)
        else:
            outs = fastgrnn_cuda.forward_unroll(torch.from_numpy(x).to(dev), P["w"], P["u"], P["bz"], P["bh"], P["zeta"], P["nu"], h0, 0,
                                                e, e, e, e, flags=fl)
            pre = outs[1].cpu().numpy().astype(np.float64)
        rel = (pre - ref) / np.maximum(np.abs(ref), 1e-3)
        print("%-13s %-20s mean signed rel err %+.3e   mean |rel err| %.3e   (fp32 ulp 6e-8)" % (tag, name, rel.mean(), np.abs(rel).mean()))
