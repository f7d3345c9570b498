"""Randomised parity sweep (developer check, not collected by pytest): dense H=128/F=32 one-saved-tensor
contract over random (T, B, layout, sequence dtype) against the fp64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from kws_amd import fastgrnn_cuda
from oracle import fastgrnn_oracle as O
dev = torch.device("cuda:0")
F, H = 32, 128
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
worst = {}
e = torch.empty(0)
for it in range(n):
    T = int(rng.integers(1, 130)); B = int(rng.integers(1, 150))
    layout = int(rng.integers(0, 3))          # 0 time-major, 1 batch-major, 2 x as [B,F,T] (hs time-major)
    p = O.make_params(F, H, dtype=np.float32, seed=int(rng.integers(1 << 30)), randomize_scalars=True)
    x = rng.standard_normal((T, B, F)).astype(np.float32)
    h0 = (0.5 * rng.standard_normal((B, H))).astype(np.float32)
    G = rng.standard_normal((T, B, H)).astype(np.float32)
    P = {k: torch.from_numpy(v).to(dev) for k, v in p.items()}
    xt, Gt, ht = torch.from_numpy(x).to(dev), torch.from_numpy(G).to(dev), torch.from_numpy(h0).to(dev)
    flags = 4
    if layout == 1:
        flags |= 16; xi, Gi = xt.transpose(0, 1).contiguous(), Gt.transpose(0, 1).contiguous()
        un = lambda t: t.transpose(0, 1); unx = un
    elif layout == 2:
        flags |= 128; xi, Gi = xt.permute(1, 2, 0).contiguous(), Gt
        un = lambda t: t; unx = lambda t: t.permute(2, 0, 1)
    else:
        xi, Gi = xt, Gt; un = lambda t: t; unx = un
    hs, pre = fastgrnn_cuda.forward_unroll(xi, P["w"], P["u"], P["bias_gate"], P["bias_update"], P["zeta"], P["nu"], ht, 0,
                                           e, e, e, e, flags=flags)
    outs = fastgrnn_cuda.backward_unroll(Gi, xi, hs, P["zeta"], P["nu"], P["w"], P["u"], pre, pre, ht, e, e, e, e, 0,
                                         flags=flags, bias_gate=P["bias_gate"], bias_update=P["bias_update"])
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    hs_o, zs_o, cs_o = O.unroll_forward(x.astype(np.float64), p64, h0.astype(np.float64))
    g_o = O.unroll_backward(G.astype(np.float64), x.astype(np.float64), hs_o, zs_o, cs_o, p64, h0.astype(np.float64))
    errs = {"hs": np.abs(un(hs).cpu().numpy() - hs_o).max()}
    names = ["d_x", "d_bias_gate", "d_bias_update", "d_zeta", "d_nu", "d_h0", "d_w", "d_u"]
    for nme, o in zip(names, outs[:8]):
        o = unx(o) if nme == "d_x" else o
        ref = g_o[nme]
        errs[nme] = np.abs(o.cpu().numpy().reshape(ref.shape) - ref).max() / max(1.0, np.abs(ref).max())
    bad = {k: v for k, v in errs.items() if v > (1e-4 if k in ("d_zeta", "d_nu") else 2e-5)}
    for k, v in errs.items():
        worst[k] = max(worst.get(k, 0.0), float(v))
    if bad:
        print("FAIL T=%d B=%d layout=%d:" % (T, B, layout), bad, flush=True)
print("worst errors over %d random cases:" % n, {k: "%.2e" % v for k, v in worst.items()})
