"""The stacked FastGRNN classifier around the recurrent cell, mirroring the reference's
``RNNClassifierModel`` for ``rnn_name == "FastGRNNCUDA"`` (/root/reference model.py:22-233) as far as the hot
path goes: 1-3 ``FastGRNNCUDA`` layers, layer l's ``[T,B,H_l]`` output handed to layer l+1 (model.py:196-203),
``hidden2keyword = nn.Linear`` on the LAST state of the top layer (model.py:226-227), ``log_softmax``
(model.py:229-230), hidden-state carry between batches (``init_hidden`` / ``hidden_states``, model.py:150-156,
200-201) and the IHT hooks (model.py:91-107), all on the GPU.

Same constructor arguments, attribute and parameter names as the reference (``rnn_list.{l}.W|U|...``,
``hidden2keyword.weight|bias``: a reference state dict loads), with these differences:

* the top layer is asked for its last state only (``FastGRNNCUDA.forward(..., last_state=True)``): its backward
  then takes the ``[B,H]`` gradient directly instead of a dense ``[T,B,H]`` tensor that is zero except for one row,
  and under ``torch.no_grad()`` its hidden-state sequence is never written;
* ``loss(input, labels)`` computes ``nn.NLLLoss()(forward(input), labels)`` (trainClassifier.py:154,236) with the
  fused head kernel -- Linear, log_softmax, NLL and all their gradients in two launches;
* ``sparsify`` / ``sparsifyWithSupport`` work in place on the device (the reference moves every layer to the CPU
  and back, model.py:91-107);
* the shadow ``rnn_list_`` / ``tracking`` ONNX-export path (model.py:72-84,187-195) is not built (export is
  disabled in the reference, trainClassifier.py:42-52), nor are the rolling hidden-state bags
  (model.py:135-148: data-loader bookkeeping, no arithmetic).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .head import keyword_loss
from .rnn import FastGRNNCUDA


class RNNClassifierModel(nn.Module):
    """1-, 2- or 3-layer FastGRNN classifier (model.py:22-233), ``rnn_name`` fixed to ``"FastGRNNCUDA"``."""

    def __init__(self, rnn_name, input_dim, num_layers, hidden_units_list, wRank_list, uRank_list,
                 wSparsity_list, uSparsity_list, gate_nonlinearity, update_nonlinearity, num_classes=None,
                 linear=True, batch_first=False, apply_softmax=True, device=None):
        if rnn_name != "FastGRNNCUDA":
            raise ValueError("kws_amd builds the FastGRNNCUDA model family only (got %r)" % (rnn_name,))
        if linear and not num_classes:
            raise Exception("num_classes need to be specified if linear is True")      # model.py:54-56
        super().__init__()
        self.rnn_name = rnn_name
        self.input_dim = input_dim
        self.hidden_units_list = list(hidden_units_list)
        self.num_layers = num_layers
        self.num_classes = num_classes
        self.wRank_list, self.uRank_list = list(wRank_list), list(uRank_list)
        self.wSparsity_list, self.uSparsity_list = list(wSparsity_list), list(uSparsity_list)
        self.gate_nonlinearity = gate_nonlinearity
        self.update_nonlinearity = update_nonlinearity
        self.linear = linear
        self.batch_first = batch_first
        self.apply_softmax = apply_softmax
        self.rnn_list = nn.ModuleList([                                                 # model.py:61-70
            FastGRNNCUDA(self.input_dim if l == 0 else self.hidden_units_list[l - 1], self.hidden_units_list[l],
                         gate_nonlinearity=gate_nonlinearity, update_nonlinearity=update_nonlinearity,
                         wRank=self.wRank_list[l], uRank=self.uRank_list[l],
                         wSparsity=self.wSparsity_list[l], uSparsity=self.uSparsity_list[l],
                         batch_first=batch_first, device=device)
            for l in range(num_layers)])
        if self.linear:                                                                 # model.py:85-88
            self.hidden2keyword = nn.Linear(self.hidden_units_list[num_layers - 1], num_classes,
                                            device=self.rnn_list[0].device)
        self.init_hidden()

    # ---- bookkeeping (model.py:91-156) -----------------------------------------------------------------
    def sparsify(self):
        for rnn in self.rnn_list:
            rnn.sparsify()

    def sparsifyWithSupport(self):
        for rnn in self.rnn_list:
            rnn.sparsifyWithSupport()

    def get_model_size(self):
        total_size = 4 * self.hidden_units_list[self.num_layers - 1] * self.num_classes
        for rnn in self.rnn_list:
            total_size += rnn.get_model_size()
        return total_size

    def name(self):
        return f"{self.num_layers} layer {self.rnn_name}"

    def init_hidden(self):
        """Clear the carried hidden states (model.py:150-156)."""
        self.hidden_states = [None] * self.num_layers

    # ---- the hot path ------------------------------------------------------------------------------------
    def _last_state(self, input):
        """Layers chained as model.py:196-203 does; returns the top layer's final state [B, H_top]."""
        rnn_in = input
        top = self.num_layers - 1
        for l, rnn in enumerate(self.rnn_list):
            out = rnn(rnn_in, hiddenState=self.hidden_states[l], last_state=(l == top))
            # (bf16 sequences: the state a layer carries over is fp32, like the one it starts from)
            if l == top:
                self.hidden_states[l] = out.detach().float()
            else:
                self.hidden_states[l] = (out.detach()[:, -1, :] if self.batch_first else out.detach()[-1, :, :]).float()
            rnn_in = out
        return rnn_in.float()                              # (the head is fp32; a no-op for fp32 sequences)

    def forward(self, input):
        """[T,B,F] (or [B,T,F] with ``batch_first``) -> keyword scores [B,C] (model.py:185-231)."""
        model_output = self._last_state(input)
        if self.linear:
            model_output = self.hidden2keyword(model_output)
        if self.apply_softmax:
            model_output = F.log_softmax(model_output, dim=1)
        return model_output

    def loss(self, input, labels):
        """``nn.NLLLoss()(self(input), labels)`` (trainClassifier.py:233-236) with the fused head."""
        if not (self.linear and self.apply_softmax):
            raise RuntimeError("loss() is the Linear + log_softmax + NLLLoss tail (linear=True, apply_softmax=True)")
        h_last = self._last_state(input)
        return keyword_loss(h_last, self.hidden2keyword.weight, self.hidden2keyword.bias, labels)
