"""what the last layer costs when only h_T is consumed (model.py:227).
Module path, B=4096 T=99 F=32 H=128: hs[-1].backward(g) against forward(..., last_state=True).backward(g), and
the inference forward with / without the hidden-state sequence."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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

# ---- the whole classifier tail (model.py:226-230 + trainClassifier.py:236): torch modules vs last_state + fused head
from kws_amd.head import KeywordHead
head = KeywordHead(H, 12, device=dev)
y = torch.randint(0, 12, (B,), device=dev)
nll = torch.nn.NLLLoss()


def tail_reference():
    m.zero_grad(set_to_none=True); head.zero_grad(set_to_none=True)
    nll(head(m(x)[-1]), y).backward()


def tail_fused():
    m.zero_grad(set_to_none=True); head.zero_grad(set_to_none=True)
    head.loss(m(x, last_state=True), y).backward()


def head_only_torch():
    head.zero_grad(set_to_none=True)
    hl = hlast.detach().requires_grad_(True)
    nll(head(hl), y).backward()


def head_only_fused():
    head.zero_grad(set_to_none=True)
    hl = hlast.detach().requires_grad_(True)
    head.loss(hl, y).backward()


with torch.no_grad():
    hlast = m(x, last_state=True)
for rep in range(2):
    a, b = timed(tail_reference), timed(tail_fused)
    c, d = timed(head_only_torch), timed(head_only_fused)
    print("layer + head + loss, fwd+bwd   torch tail: %.1f us   last_state + fused head: %.1f us   (%.2fx)" % (a, b, a / b))
    print("head + loss alone, fwd+bwd     torch: %.1f us   fused: %.1f us" % (c, d), flush=True)
