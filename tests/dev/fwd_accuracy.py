"""Forward accuracy vs the fp64 oracle and kernel time for the forward variants:
flags 4 = default (fp16 two-plane state product), 4|64 = three bf16 planes, 4|8 = 4-wave kernel; also the
generic fp32 scan (flag 1) as the plain-fp32 yardstick.  Weight scales 0.1 (reference init), 0.01, 1.0."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from kws_amd import fastgrnn_cuda
from oracle import fastgrnn_oracle as O
dev = torch.device("cuda:0")
T, B, F, H = 99, 64, 32, 128
e = torch.empty(0)
for scale in (0.1, 0.01, 1.0, 3.0):
    p = O.make_params(F, H, dtype=np.float32, seed=3, randomize_scalars=True)
    p["u"] = (p["u"] * (scale / 0.1)).astype(np.float32)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((T, B, F)).astype(np.float32)
    h0 = (0.5 * rng.standard_normal((B, H))).astype(np.float32)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    hs_o, _, _ = O.unroll_forward(x.astype(np.float64), p64, h0.astype(np.float64))
    hprev = np.concatenate([h0.astype(np.float64)[None], hs_o[:-1]], 0)
    pre_o = x.astype(np.float64) @ p64["w"].T + hprev @ p64["u"].T
    P = {k: torch.from_numpy(v).to(dev) for k, v in p.items()}
    xt, ht = torch.from_numpy(x).to(dev), torch.from_numpy(h0).to(dev)
    line = "|U| scale %-4g:" % scale
    for name, fl in (("fp16x2", 4), ("bf16x3", 4 | 64), ("4-wave", 4 | 8), ("generic fp32", 1)):
        outs = fastgrnn_cuda.forward_unroll(xt, P["w"], P["u"], P["bias_gate"], P["bias_update"], P["zeta"], P["nu"], ht, 0,
                                            e, e, e, e, flags=fl, want_gates=(fl == 1))
        hs = outs[0].cpu().numpy().astype(np.float64)
        err_h = np.abs(hs - hs_o).max()
        line += "  %s: hs %.2e" % (name, err_h)
        if fl != 1:
            line += " pre %.2e" % np.abs(outs[1].cpu().numpy() - pre_o).max()
    print(line, flush=True)
# time at B=4096
B = 4096
x = torch.randn(T, B, F, device=dev); h0 = torch.zeros(B, H, device=dev)
w = 0.1 * torch.randn(H, F, device=dev); u = 0.1 * torch.randn(H, H, device=dev)
bz = torch.ones(1, H, device=dev); bh = torch.ones(1, H, device=dev)
zeta = torch.ones(1, 1, device=dev); nu = -4 * torch.ones(1, 1, device=dev)
for rep in range(2):
    for name, fl in (("fp16x2", 4), ("bf16x3", 4 | 64), ("4-wave", 4 | 8)):
        fn = lambda: fastgrnn_cuda.forward_unroll(x, w, u, bz, bh, zeta, nu, h0, 0, e, e, e, e, flags=fl)
        for _ in range(3): fn()
        ts = []
        for _ in range(30):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        print("B=4096 %s: median %.1f us (incl. launch path)" % (name, ts[len(ts) // 2]), flush=True)
