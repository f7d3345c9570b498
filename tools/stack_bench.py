#!/usr/bin/env python3
"""The reference's default model (two dense layers 32 -> 256 -> 128 + head) as a training step, for profiling:
    rocprofv3 --kernel-trace --stats -d <dir> -- python3 tools/stack_bench.py [steps] [B] [which]
`which`: stack (default) | stack_bft (fed the loader's [B,F,T] batch as the trainer's permuted view: layer 1 takes
it through FASTGRNN_FLAG_X_BFT) | stack_copy (same view, copied with .contiguous() first as the reference does) | lowrank (BASELINE config 4) | l1 (H=256/F=32 layer alone) | l2 (H=128/F=256 alone)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from kws_amd import FastGRNNCUDA, RNNClassifierModel  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
which = sys.argv[3] if len(sys.argv) > 3 else "stack"
dev = torch.device("cuda:0")
T, F, C = 99, 32, 12
torch.manual_seed(0)
g = torch.Generator().manual_seed(1)
if which.startswith("stack"):
    m = RNNClassifierModel("FastGRNNCUDA", F, 2, [256, 128], [None, None], [None, None], [1.0, 1.0], [1.0, 1.0],
                           "sigmoid", "tanh", num_classes=C, device=dev)
    x = torch.randn(T, B, F, generator=g).to(dev)
    y = torch.randint(0, C, (B,), generator=g).to(dev)
    audio = x.permute(1, 2, 0).contiguous() if which != "stack" else None      # [B,F,T] (trainClassifier.py:204)

    def step():
        for p in m.parameters():
            p.grad = None
        m.init_hidden()
        xin = x if which == "stack" else (audio.permute(2, 0, 1) if which == "stack_bft" else audio.permute(2, 0, 1).contiguous())
        m.loss(xin, y).backward()
else:
    Fi, H, r = {"lowrank": (32, 256, 16), "l1": (32, 256, None), "l2": (256, 128, None)}[which]
    m = FastGRNNCUDA(Fi, H, wRank=r, uRank=r, device=dev)
    x = torch.randn(T, B, Fi, generator=g).to(dev).requires_grad_(which == "l2")
    G = torch.randn(T, B, H, generator=g).to(dev)

    def step():
        for p in m.parameters():
            p.grad = None
        x.grad = None
        m(x).backward(G)
for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
print("%s B=%d: %.3f ms/step" % (which, B, 1e3 * (time.perf_counter() - t0) / steps))
