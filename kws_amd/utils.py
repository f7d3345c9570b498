"""Device-side mirrors of the reference's sparsification helpers (/root/reference utils.py:53-114), the
bookkeeping around the recurrent cell that SURVEY.md section 8f lists as N4.

The reference moves every weight matrix GPU -> CPU -> numpy -> GPU on each call
(``hardThreshold``: ``A.data.cpu().detach().numpy()``, utils.py:57-62); here everything stays on the
tensor's device and is done in place with a handful of torch ops -- no HIP kernel is needed for 20 K
elements.  Same results: the threshold is numpy's ``percentile(|A|, (1-s)*100, interpolation='higher')``,
i.e. the element of rank ``ceil((1-s)*(n-1))`` of the sorted magnitudes, and entries strictly below it
are zeroed (ties with the threshold survive, as in the reference).
"""
from __future__ import annotations

import math

import torch


def hardThreshold(A: torch.Tensor, s: float) -> torch.Tensor:
    """utils.py:53-64: keep (about) a fraction ``s`` of the entries of ``A`` with the largest magnitude and
    zero the rest, IN PLACE on ``A.data`` (the reference returns a new CPU tensor that the CPU cell assigns
    to ``.data``, rnn.py:441-443); returns ``A``."""
    data = A.data
    n = data.numel()
    if n == 0:
        return A
    mag = data.abs().reshape(-1)
    pos = (1.0 - float(s)) * (n - 1)                      # numpy: virtual index of the q-th percentile
    k = min(max(int(math.ceil(pos - 1e-12)), 0), n - 1)   # interpolation='higher'
    th = torch.kthvalue(mag.float(), k + 1).values        # k-th smallest, 1-based
    data.mul_((data.abs() >= th).to(data.dtype))
    return A


def supportBasedThreshold(dst: torch.Tensor, src: torch.Tensor) -> torch.Tensor:
    """utils.py:66-81: zero the entries of ``dst.data`` where ``src`` is zero (in place); returns ``dst``."""
    dst.data.mul_((src.to(dst.device) != 0).to(dst.dtype).reshape(dst.shape))
    return dst


def estimateNNZ(A, s, bytesPerVar=4):
    """utils.py:84-100 (pure arithmetic)."""
    params = 1
    for d in A.shape:
        params *= int(d)
    if s < 0.5:
        nnz = math.ceil(params * s)
        return nnz, nnz * 2 * bytesPerVar, True
    return params, params * bytesPerVar, False


def countNNZ(A: torch.Tensor, isSparse) -> int:
    """utils.py:103-114, on the tensor's own device (one scalar crosses to the host)."""
    if isSparse:
        return int(torch.count_nonzero(A.detach()).item())
    n = 1
    for d in A.shape:
        n *= int(d)
    return n
