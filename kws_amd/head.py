"""Classifier head on the last hidden state behind the C ABI's ``fastgrnn_hip_head_xent``.

The reference chains three torch modules after the last FastGRNN layer -- ``hidden2keyword = nn.Linear(H, C)``
on ``hs[T-1]`` (model.py:86-88, 226-227), ``F.log_softmax(dim=1)`` (model.py:229-230) and ``nn.NLLLoss()``
(trainClassifier.py:154,236) -- i.e. about eight small launches per training step for forward and backward.
``keyword_loss`` does all of it, gradients included, in two launches; ``KeywordHead`` owns the ``Linear``
parameters under the reference's names so a ``RNNClassifierModel`` state dict loads unchanged
(``hidden2keyword.weight`` / ``.bias``).
"""
import ctypes as C

import torch
from torch import nn
from torch.autograd import Function

from . import _lib
from .fastgrnn_cuda import _check_input, _ptr, _stream, _workspace


def head_xent(h_last, weight, bias, labels, want_log_probs=False):
    """One fused pass: returns (loss[1], log_probs[B,C] or None, d_h_last[B,H], d_weight[C,H], d_bias[C]) for
    loss = NLLLoss(mean)(log_softmax(h_last @ weight.T + bias), labels)."""
    lib = _lib.load()
    for t, n in ((h_last, "h_last"), (weight, "weight"), (bias, "bias"), (labels, "labels")):
        _check_input(t, n)
    if h_last.dim() != 2 or weight.dim() != 2 or weight.shape[1] != h_last.shape[1] or bias.numel() != weight.shape[0]:
        raise RuntimeError("head_xent: h_last [B,H], weight [C,H], bias [C]")
    if labels.dtype != torch.int64 or labels.numel() != h_last.shape[0]:
        raise RuntimeError("head_xent: labels must be int64 [B]")
    if h_last.dtype != torch.float32 or weight.dtype != torch.float32 or bias.dtype != torch.float32:
        raise RuntimeError("head_xent: float32 operands")
    B, H = h_last.shape
    Cn = weight.shape[0]
    dev = h_last.device
    with torch.cuda.device(dev):
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        logp = torch.empty((B, Cn), dtype=torch.float32, device=dev) if want_log_probs else None
        d_h = torch.empty_like(h_last)
        d_w = torch.empty_like(weight)
        d_b = torch.empty(Cn, dtype=torch.float32, device=dev)
        nbytes = lib.fastgrnn_hip_head_workspace_bytes(B, H, Cn)
        ws, wsp = _workspace(nbytes, dev)
        st = lib.fastgrnn_hip_head_xent(B, H, Cn, _ptr(h_last), _ptr(weight), _ptr(bias), _ptr(labels), _ptr(loss),
                                        _ptr(logp), _ptr(d_h), _ptr(d_w), _ptr(d_b), wsp, nbytes, _stream(dev))
        _lib.check(st, "fastgrnn head_xent")
        del ws
    return loss, logp, d_h, d_w, d_b


class KeywordLossFunction(Function):
    """loss = NLLLoss()(log_softmax(Linear(h_last)), labels); the gradients are computed with the forward (the
    loss is always differentiated in training) and scaled by the incoming grad in backward."""

    @staticmethod
    def forward(ctx, h_last, weight, bias, labels):
        loss, _, d_h, d_w, d_b = head_xent(h_last.contiguous(), weight.contiguous(), bias.contiguous(),
                                           labels.contiguous())
        ctx.save_for_backward(d_h, d_w, d_b)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_loss):
        d_h, d_w, d_b = ctx.saved_tensors
        return d_h * grad_loss, d_w * grad_loss, d_b * grad_loss, None


def keyword_loss(h_last, weight, bias, labels):
    return KeywordLossFunction.apply(h_last, weight, bias, labels)


class KeywordHead(nn.Module):
    """``hidden2keyword`` (model.py:86-88) + log_softmax + NLLLoss.  ``forward(h_last)`` gives the reference's
    ``keyword_scores`` (log-probabilities, model.py:226-230, plain torch); ``loss(h_last, labels)`` the fused
    training loss (trainClassifier.py:236)."""

    def __init__(self, hidden_size, num_classes, device=None):
        super().__init__()
        self.hidden2keyword = nn.Linear(hidden_size, num_classes, device=device)

    def forward(self, h_last):
        return torch.log_softmax(self.hidden2keyword(h_last), dim=1)

    def loss(self, h_last, labels):
        return keyword_loss(h_last, self.hidden2keyword.weight, self.hidden2keyword.bias, labels)
