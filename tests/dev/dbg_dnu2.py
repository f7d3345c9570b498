import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from kws_amd import fastgrnn_cuda
from oracle import fastgrnn_oracle as O
dev = torch.device("cuda:0")
F, H = 32, 128
e = torch.empty(0)
T, B, seed = 99, 64, 8
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
fw = lambda fl: fastgrnn_cuda.forward_unroll(xt, P["w"], P["u"], P["bias_gate"], P["bias_update"], P["zeta"], P["nu"], ht, 0, e, e, e, e, flags=fl)
bw = lambda o, fl: fastgrnn_cuda.backward_unroll(Gt, xt, o[0], P["zeta"], P["nu"], P["w"], P["u"], o[1], o[2], ht, e, e, e, e, 0, flags=fl)
exact = [torch.from_numpy(a.astype(np.float32)).to(dev) for a in (hs_o, zs_o, cs_o)]
for fname, o in (("split fwd", fw(0)), ("f32-mfma fwd", fw(2)), ("oracle fwd (fp64 rounded)", exact)):
    for bname, fl in (("split bwd (4-wave, z/c given)", 0), ("f32-mfma bwd", 2), ("generic bwd", 1)):
        gr = bw(o, fl)
        rel = lambda k, i: abs(gr[i].item() - g_o[k].item()) / max(1, abs(g_o[k].item()))
        print("%-28s + %-30s d_zeta rel %.2e  d_nu rel %.2e   d_u rel %.2e" % (fname, bname, rel("d_zeta", 3), rel("d_nu", 4),
              float((gr[7].cpu().numpy() - g_o["d_u"]).__abs__().max() / np.abs(g_o["d_u"]).max())))
print("mean SIGNED error (kernel - oracle) and mean |error| of two backward outputs, oracle forward:")
for bname, fl in (("split bwd", 0), ("f32-mfma bwd", 2), ("generic bwd", 1)):
    gr = bw(exact, fl)
    for k, i in (("d_h0", 5), ("d_x", 0), ("d_bias_update", 2), ("d_bias_gate", 1)):
        d = gr[i].cpu().numpy().astype(np.float64).reshape(g_o[k].shape) - g_o[k]
        print("   %-13s %-14s mean signed %+.3e   mean abs %.3e   (mean |ref| %.3e, mean ref %+.3e)" % (bname, k, d.mean(), np.abs(d).mean(), np.abs(g_o[k]).mean(), g_o[k].mean()))
