"""A/B: forward kernel variants (flag bit 8 = older 4-wave shape) -- equality + time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kws_amd import fastgrnn_cuda
dev = torch.device("cuda:0")
T, F, H = 99, 32, 128
torch.manual_seed(0)
e = torch.empty(0)
w = 0.1 * torch.randn(H, F, device=dev); u = 0.1 * torch.randn(H, H, device=dev)
bz = torch.randn(1, H, device=dev); bh = torch.randn(1, H, device=dev)
zeta = torch.ones(1, 1, device=dev); nu = -4 * torch.ones(1, 1, device=dev)
for B in (50, 4096):
    x = torch.randn(T, B, F, device=dev); h0 = 0.3 * torch.randn(B, H, device=dev)
    for base in (0, 4):
        for want in (True, False):
            if base == 4 and not want:
                continue
            ref = fastgrnn_cuda.forward_unroll(x, w, u, bz, bh, zeta, nu, h0, 0, e, e, e, e, want_gates=want, flags=base)
            new = fastgrnn_cuda.forward_unroll(x, w, u, bz, bh, zeta, nu, h0, 0, e, e, e, e, want_gates=want, flags=base | 8)
            errs = [float((a - b).abs().max()) for a, b in zip(ref, new)]
            line = "B=%d flags=%d gates=%s maxdiff %s" % (B, base, want, errs)
            if B == 4096:
                for fl in (base, base | 8):
                    fn = lambda: fastgrnn_cuda.forward_unroll(x, w, u, bz, bh, zeta, nu, h0, 0, e, e, e, e, want_gates=want, flags=fl)
                    for _ in range(3): fn()
                    ts = []
                    for _ in range(20):
                        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
                    ts.sort(); line += "  | flags %d: %.1f us" % (fl, ts[len(ts) // 2])
            print(line, flush=True)
