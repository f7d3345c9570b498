"""CPU baseline ("port") for the FastGRNN cell -- TEST/BENCH INFRASTRUCTURE ONLY.

A torch-CPU restatement of the reference's CPU path with the SAME op sequence:
``FastGRNNCell.forward`` (/root/reference rnn.py:273-297: two ``matmul`` (four if
low-rank), ``pre+bias`` twice, sigmoid/tanh, the zeta/nu update) driven by the
per-timestep loop of ``BaseRNN.forward`` (rnn.py:588-591,657-668) with torch
autograd for the backward.  The reference's Python cannot travel to the GPU box,
so ``bench.py``'s ``cpu_baseline`` leg times THIS on the host cores
(``kind: "port"``).  Parameters use the CPU cell's layout (W:[F,H], U:[H,H], ...).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg
may import this file; the product never does.
"""
from __future__ import annotations

import torch


def _nl(a, name):
    # rnn.py:40-67 (relu restated as torch.relu(a); the reference's relu raises)
    if name == "tanh":
        return torch.tanh(a)
    if name == "sigmoid":
        return torch.sigmoid(a)
    if name == "relu":
        return torch.relu(a)
    if name == "quantTanh":
        return torch.max(torch.min(a, torch.ones_like(a)), -1.0 * torch.ones_like(a))
    if name == "quantSigm":
        a = (a + 1.0) / 2.0
        return torch.max(torch.min(a, torch.ones_like(a)), torch.zeros_like(a))
    if name == "quantSigm4":
        a = (a + 2.0) / 4.0
        return torch.max(torch.min(a, torch.ones_like(a)), torch.zeros_like(a))
    raise ValueError(name)


class FastGRNNCellPort(torch.nn.Module):
    """Same parameters, init and forward as rnn.py:192-297 (CPU layout)."""

    def __init__(self, input_size, hidden_size, gate_nonlinearity="sigmoid",
                 update_nonlinearity="tanh", wRank=None, uRank=None,
                 zetaInit=1.0, nuInit=-4.0, dtype=torch.float32):
        super().__init__()
        self._gate, self._update = gate_nonlinearity, update_nonlinearity
        self._wRank, self._uRank = wRank, uRank
        P = torch.nn.Parameter
        if wRank is None:
            self.W = P(0.1 * torch.randn([input_size, hidden_size], dtype=dtype))
        else:
            self.W1 = P(0.1 * torch.randn([input_size, wRank], dtype=dtype))
            self.W2 = P(0.1 * torch.randn([wRank, hidden_size], dtype=dtype))
        if uRank is None:
            self.U = P(0.1 * torch.randn([hidden_size, hidden_size], dtype=dtype))
        else:
            self.U1 = P(0.1 * torch.randn([hidden_size, uRank], dtype=dtype))
            self.U2 = P(0.1 * torch.randn([uRank, hidden_size], dtype=dtype))
        self.bias_gate = P(torch.ones([1, hidden_size], dtype=dtype))
        self.bias_update = P(torch.ones([1, hidden_size], dtype=dtype))
        self.zeta = P(zetaInit * torch.ones([1, 1], dtype=dtype))
        self.nu = P(nuInit * torch.ones([1, 1], dtype=dtype))
        self.hidden_size = hidden_size

    def forward(self, input, state):
        if self._wRank is None:
            wComp = torch.matmul(input, self.W)
        else:
            wComp = torch.matmul(torch.matmul(input, self.W1), self.W2)
        if self._uRank is None:
            uComp = torch.matmul(state, self.U)
        else:
            uComp = torch.matmul(torch.matmul(state, self.U1), self.U2)
        pre_comp = wComp + uComp
        z = _nl(pre_comp + self.bias_gate, self._gate)
        c = _nl(pre_comp + self.bias_update, self._update)
        return z * state + (torch.sigmoid(self.zeta) * (1.0 - z) + torch.sigmoid(self.nu)) * c


def unroll(cell, x, h0=None):
    """BaseRNN.forward semantics (rnn.py:588-591,657-668): x:[T,B,F] -> hs:[T,B,H]."""
    T, B, _ = x.shape
    h = torch.zeros(B, cell.hidden_size, dtype=x.dtype) if h0 is None else h0
    hs = []
    for t in range(T):
        h = cell(x[t], h)
        hs.append(h)
    return torch.stack(hs, 0)


def time_fwd_bwd(B, T=99, F=32, H=128, wRank=None, uRank=None, threads=None,
                 budget_s=20.0, min_iters=3, seed=0):
    """Times (zero_grad, unroll forward, (hs*G).sum().backward()) on the host.
    Returns dict(utt_per_s, fwd_utt_per_s, iters, threads)."""
    import time
    if threads:
        torch.set_num_threads(threads)
    torch.manual_seed(seed)
    cell = FastGRNNCellPort(F, H, wRank=wRank, uRank=uRank)
    x = torch.randn(T, B, F)
    G = torch.randn(T, B, H)
    # one warm-up
    (unroll(cell, x) * G).sum().backward()
    ts, tf = [], []
    t_start = time.perf_counter()
    while len(ts) < min_iters or (time.perf_counter() - t_start) < budget_s:
        cell.zero_grad(set_to_none=True)
        t0 = time.perf_counter()
        hs = unroll(cell, x)
        t1 = time.perf_counter()
        (hs * G).sum().backward()
        t2 = time.perf_counter()
        ts.append(t2 - t0)
        tf.append(t1 - t0)
        if len(ts) >= 200:
            break
    ts.sort(); tf.sort()
    med = ts[len(ts) // 2]
    medf = tf[len(tf) // 2]
    return {"utt_per_s": B / med, "fwd_utt_per_s": B / medf, "iters": len(ts),
            "threads": torch.get_num_threads()}
