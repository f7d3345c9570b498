#!/usr/bin/env python3
"""The headline training step (FastGRNNCUDA forward + autograd backward through the C ABI) captured ONCE in a HIP graph
(torch.cuda.CUDAGraph) and replayed: the operator makes no host-side decisions that depend on data, allocates through
torch's allocator only and launches on the current stream, so the whole step is capturable.  Checks that a replay
produces the eager step's gradients bit for bit, then times eager steps against replays.
    python3 tools/graph_step.py [n] [B]"""
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
torch.manual_seed(0)
m = FastGRNNCUDA(F, H, device=dev)
x = torch.randn(T, B, F, device=dev)
G = torch.randn(T, B, H, device=dev)
params = list(m.parameters())


def step():
    for p in params:
        p.grad = None
    m(x).backward(G)


for _ in range(20):
    step()
torch.cuda.synchronize()
eager = [p.grad.clone() for p in params]

# warm up on a side stream (workspace cache of that stream, plan cache), then capture one step
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
for p in params:
    p.grad = None
with torch.cuda.graph(graph):
    m(x).backward(G)
torch.cuda.synchronize()
for p in params:
    p.grad.zero_()
graph.replay()
torch.cuda.synchronize()
same = all(torch.equal(a, p.grad) for a, p in zip(eager, params))
print("replay == eager step, bit for bit:", same)
assert same

for name, fn in (("eager", step), ("graph replay", graph.replay), ("eager", step), ("graph replay", graph.replay)):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-12s B=%d: host %.1f us/step, until done %.1f us/step" % (name, B, 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n))
