"""what the last layer costs when only h_T is consumed (model.py:227).
Module path, B=4096 T=99 F=32 H=128: hs[-1].backward(g) against forward(..., last_state=True).backward(g), and
the inference forward with / without the hidden-state sequence."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from kws_amd.rnn import FastGRNNCUDA

dev = torch.device("cuda:0")
T, B, F, H = 99, 4096, 32, 128
torch.manual_seed(0)
m = FastGRNNCUDA(F, H, device=dev)
x = torch.randn(T, B, F, device=dev)
gl = torch.randn(B, H, device=dev)


def timed(fn, n=40):
    for _ in range(5):
        fn()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def step_indexed():
    m.zero_grad(set_to_none=True)
    m(x)[-1].backward(gl)


def step_last():
    m.zero_grad(set_to_none=True)
    m(x, last_state=True).backward(gl)


def infer_full():
    with torch.no_grad():
        return m(x)[-1]


def infer_last():
    with torch.no_grad():
        return m(x, last_state=True)


for rep in range(2):
    a, b = timed(step_indexed), timed(step_last)
    c, d = timed(infer_full), timed(infer_last)
    print("train step  hs[-1].backward: %.1f us   last_state=True: %.1f us   (%.2fx)" % (a, b, a / b))
    print("inference   full sequence:   %.1f us   last_state=True: %.1f us   (%.2fx)" % (c, d, c / d), flush=True)
