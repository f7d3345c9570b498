import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kws_amd import fastgrnn_cuda
from oracle import fastgrnn_oracle as O
F, H, r = 32, 256, 16
dev = torch.device("cuda:0")
T, B = 6, 32
rng = np.random.default_rng(1)
p = O.make_params(F, H, r, r, dtype=np.float32, seed=17, randomize_scalars=True)
x = rng.standard_normal((T, B, F)).astype(np.float32)
h0 = 0.5 * rng.standard_normal((B, H)).astype(np.float32)
P = {k: torch.from_numpy(v).to(dev) for k, v in p.items()}
e = torch.empty(0)
xt, ht = torch.from_numpy(x).to(dev), torch.from_numpy(h0).to(dev)
res = {}
for it in range(3):
    outs = fastgrnn_cuda.forward_unroll(xt, e, e, P["bias_gate"], P["bias_update"], P["zeta"], P["nu"], ht, 0,
                                        P["w1"], P["w2"], P["u1"], P["u2"], flags=4)
    torch.cuda.synchronize()
    res["hs%d" % it] = outs[0].cpu().numpy(); res["pre%d" % it] = outs[1].cpu().numpy(); res["m%d" % it] = outs[2].cpu().numpy()
p64 = {k: v.astype(np.float64) for k, v in p.items()}
hs_o, zs_o, cs_o = O.unroll_forward(x.astype(np.float64), p64, h0.astype(np.float64))
np.savez("gpurun_out/dbg_lr.npz", x=x, h0=h0, hs_o=hs_o, zs_o=zs_o, cs_o=cs_o, **{k: v for k, v in p.items()}, **res)
