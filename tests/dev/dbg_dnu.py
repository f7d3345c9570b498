import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from kws_amd import fastgrnn_cuda
from oracle import fastgrnn_oracle as O
dev = torch.device("cuda:0")
F, H = 32, 128
e = torch.empty(0)
for (T, B, seed) in ((82, 95, 7), (99, 64, 8)):
    rng = np.random.default_rng(seed)
    p = O.make_params(F, H, dtype=np.float32, seed=seed, randomize_scalars=True)
    x = rng.standard_normal((T, B, F)).astype(np.float32)
    h0 = (0.5 * rng.standard_normal((B, H))).astype(np.float32)
    G = rng.standard_normal((T, B, H)).astype(np.float32)
    P = {k: torch.from_numpy(v).to(dev) for k, v in p.items()}
    xt, Gt, ht = torch.from_numpy(x).to(dev), torch.from_numpy(G).to(dev), torch.from_numpy(h0).to(dev)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    hs_o, zs_o, cs_o = O.unroll_forward(x.astype(np.float64), p64, h0.astype(np.float64))
    g_o = O.unroll_backward(G.astype(np.float64), x.astype(np.float64), hs_o, zs_o, cs_o, p64, h0.astype(np.float64))
    # the raw sums before sigma': what an exact evaluation gives, and their conditioning
    print("T=%d B=%d zeta=%.3f nu=%.3f  ref d_zeta=%.6g d_nu=%.6g" % (T, B, p["zeta"].item(), p["nu"].item(), g_o["d_zeta"].item(), g_o["d_nu"].item()))
    for name, fl in (("w8", 4), ("4-wave", 4 | 32), ("f32-mfma", 2)):
        pre = fl & 4
        outs = fastgrnn_cuda.forward_unroll(xt, P["w"], P["u"], P["bias_gate"], P["bias_update"], P["zeta"], P["nu"], ht, 0, e, e, e, e, flags=fl)
        gr = fastgrnn_cuda.backward_unroll(Gt, xt, outs[0], P["zeta"], P["nu"], P["w"], P["u"], outs[1], outs[-1], ht, e, e, e, e, 0, flags=fl,
                                           bias_gate=P["bias_gate"] if pre else None, bias_update=P["bias_update"] if pre else None)
        print("   %-9s d_zeta %.6g (rel %.2e)  d_nu %.6g (rel %.2e)" % (name, gr[3].item(), abs(gr[3].item() - g_o["d_zeta"].item()) / max(1, abs(g_o["d_zeta"].item())),
                                                                     gr[4].item(), abs(gr[4].item() - g_o["d_nu"].item()) / max(1, abs(g_o["d_nu"].item()))))
