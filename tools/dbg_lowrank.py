"""Debug helper: low-rank forward (H=256, r=16) per-timestep error vs the fp64 oracle for several shapes/flags."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kws_amd import fastgrnn_cuda
from oracle import fastgrnn_oracle as O

F, H, r = 32, 256, 16
dev = torch.device("cuda:0")
for (T, B, flags, rs, hz) in [(6, 32, 0, False, True), (6, 32, 0, True, True), (6, 32, 0, True, False), (6, 37, 0, True, False),
                              (6, 32, 4, True, False), (7, 32, 4, True, False), (6, 37, 4, True, False)]:
    rng = np.random.default_rng(1)
    p = O.make_params(F, H, r, r, dtype=np.float32, seed=17, randomize_scalars=rs)
    x = rng.standard_normal((T, B, F)).astype(np.float32)
    h0 = (0.0 if hz else 0.5) * rng.standard_normal((B, H)).astype(np.float32)
    P = {k: torch.from_numpy(v).to(dev) for k, v in p.items()}
    e = torch.empty(0)
    outs = fastgrnn_cuda.forward_unroll(torch.from_numpy(x).to(dev), e, e, P["bias_gate"], P["bias_update"], P["zeta"], P["nu"],
                                        torch.from_numpy(h0).to(dev), 0, P["w1"], P["w2"], P["u1"], P["u2"], flags=flags)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    hs_o, zs_o, cs_o = O.unroll_forward(x.astype(np.float64), p64, h0.astype(np.float64))
    err = np.abs(outs[0].cpu().numpy() - hs_o).reshape(T, -1).max(1)
    print(T, B, flags, "rs", rs, "h0zero", hz, "zeta/nu", p["zeta"].ravel(), p["nu"].ravel(), "err/t", np.array2string(err, precision=2))
    if flags == 4:
        h64 = h0.astype(np.float64); x64 = x.astype(np.float64)
        hprev = np.concatenate([h64[None], hs_o[:-1]], 0)
        m_o = np.concatenate([hprev @ p64["u1"].T, x64 @ p64["w1"].T], -1)
        pre_o = m_o[..., r:] @ p64["w2"].T + m_o[..., :r] @ p64["u2"].T
        m = outs[2].cpu().numpy(); pre = outs[1].cpu().numpy()
        print("   m_h err/t", np.array2string(np.abs(m - m_o)[..., :r].reshape(T, -1).max(1), precision=2))
        print("   m_x err/t", np.array2string(np.abs(m - m_o)[..., r:].reshape(T, -1).max(1), precision=2))
        print("   pre err/t", np.array2string(np.abs(pre - pre_o).reshape(T, -1).max(1), precision=2))
        eh = np.abs(outs[0].cpu().numpy() - hs_o)[0]
        print("   t=0 err per wave-quarter of units", [float(eh[:, q * 64:(q + 1) * 64].max()) for q in range(4)],
              "per utterance block", [float(eh[k:k + 16].max()) for k in range(0, B, 16)])
        # recompute h_0 from the kernel's own pre
        z = 1 / (1 + np.exp(-(pre[0] + p64["bias_gate"]))); c = np.tanh(pre[0] + p64["bias_update"])
        sz = 1 / (1 + np.exp(-p64["zeta"])); sn = 1 / (1 + np.exp(-p64["nu"]))
        h_re = (sz * (1 - z) + sn) * c + z * h64
        print("   h_0 recomputed from the kernel's pre vs kernel h_0:", float(np.abs(h_re - outs[0].cpu().numpy()[0]).max()))
