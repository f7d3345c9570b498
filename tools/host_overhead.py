#!/usr/bin/env python3
"""Host cost of one training step of the headline configuration (module forward + autograd backward through the
ctypes shim): time to ENQUEUE n steps (no synchronise) against the time until the GPU has finished them.  When the
two are close the step is host-bound on this box.    python3 tools/host_overhead.py [n] [B]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from kws_amd import FastGRNNCUDA  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda:0")
T, F, H = 99, 32, 128
m = FastGRNNCUDA(F, H, device=dev)
x = torch.randn(T, B, F, device=dev)
G = torch.randn(T, B, H, device=dev)
params = list(m.parameters())


def step():
    for p in params:
        p.grad = None
    m(x).backward(G)


for _ in range(300):
    step()
torch.cuda.synchronize()
from kws_amd import fastgrnn_cuda as _fc  # noqa: E402
_fast = _fc._get_raw_stream
_slow = lambda idx: torch.cuda.current_stream(idx).cuda_stream      # what the shim did before (A/B in one process)
for rep in range(6):
    _fc._get_raw_stream = _slow if rep % 2 else _fast
    print("stream handle via %s:" % ("torch.cuda.current_stream()" if rep % 2 else "raw query"), end=" ")
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("B=%d: enqueue %.1f us/step, until done %.1f us/step" % (B, 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n))
_fc._get_raw_stream = _fast
# the same loop at one workgroup's worth of utterances: the floor of a step (99 serial frames per scan, twice, however
# small the batch) or the host cost, whichever is larger
xs, Gs = x[:, :16].contiguous(), G[:, :16].contiguous()
x, G = xs, Gs
for _ in range(50):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
print("B=16: %.1f us/step (serial-chain / host floor)" % (1e6 * (time.perf_counter() - t0) / n))
# the host's own cost: the same calls on a two-frame sequence (the GPU side is a few microseconds, so the loop time is
# what the host spends per step: module forward + autograd + the ctypes shim + three launches).  A/B in ONE process
# (the same loop differs by 2x between processes on one box): with and without the shim's validated-signature path.
x, G = xs[:2].contiguous(), Gs[:2].contiguous()
for _ in range(200):
    step()
torch.cuda.synchronize()
for rep in range(6):
    _fc._use_seen = (rep % 2 == 0)
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("T=2 B=16, %s: host %.1f us/step" % ("validated-signature path" if _fc._use_seen else "full checks every call",
                                                1e6 * (t1 - t0) / n))
_fc._use_seen = True
