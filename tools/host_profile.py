#!/usr/bin/env python3
"""cProfile of the host side of the headline training step (tools/host_overhead.py's loop)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from kws_amd import FastGRNNCUDA  # noqa: E402

dev = torch.device("cuda:0")
T, B, F, H = 99, 4096, 32, 128
m = FastGRNNCUDA(F, H, device=dev)
x = torch.randn(T, B, F, device=dev)
G = torch.randn(T, B, H, device=dev)
params = list(m.parameters())


def step():
    for p in params:
        p.grad = None
    m(x).backward(G)


for _ in range(100):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
