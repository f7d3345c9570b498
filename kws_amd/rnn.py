"""Host-side mirror of the reference's GPU FastGRNN classes (/root/reference rnn.py):

* ``FastGRNNCUDACell``        rnn.py:454-549   single-step cell
* ``FastGRNNCUDA``            rnn.py:738-889   unrolled module (the class the build sits behind)
* ``FastGRNNFunction``        rnn.py:891-905   autograd glue, single step
* ``FastGRNNUnrollFunction``  rnn.py:907-972   autograd glue, unrolled

Same constructor keywords, same parameter names and ``[out,in]`` shapes (so reference
state dicts ``rnn_list.{l}.W|U|W1|W2|U1|U2|bias_gate|bias_update|zeta|nu`` load), same
gate-code table, zero ``h0`` default, ``batch_first`` handling and error behaviour, with
these deliberate differences:

* the GPU gate is ``torch.cuda.is_available()`` (ROCm reports HIP devices there), not
  ``utils.findCUDA()`` which looks for nvcc (utils.py:12-36);
* ``device=`` may be passed explicitly (e.g. a specific ``cuda:N`` for one-process-per-GPU
  data parallel, or ``"cpu"`` to inspect parameter layout without a GPU -- calling
  ``forward`` on CPU tensors still raises like CHECK_CUDA, fastgrnn_cuda.cpp:69);
* non-Parameter placeholders for absent operands are registered buffers so ``.to()``
  keeps them beside the parameters;
* the tanh-gate backward uses d_tanh (reference defect .cu:519-521 not reproduced).
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib, fastgrnn_cuda, utils

NON_LINEARITY = {"sigmoid": 0, "relu": 1, "tanh": 2}   # rnn.py:478,751


def _resolve_device(device):
    if device is not None:
        return torch.device(device)
    if not torch.cuda.is_available():
        # same message and exception type as rnn.py:476-477,749-750
        raise Exception('FastGRNNCUDA is supported only on GPU devices.')
    return torch.device("cuda", torch.cuda.current_device())


class FastGRNNFunction(Function):
    """rnn.py:891-905."""

    @staticmethod
    def forward(ctx, input, bias_gate, bias_update, zeta, nu, old_h, w, u, w1, w2, u1, u2, gate_non_linearity):
        outputs = fastgrnn_cuda.forward(input.contiguous(), w, u, bias_gate, bias_update, zeta, nu,
                                        old_h.contiguous(), gate_non_linearity, w1, w2, u1, u2)
        new_h = outputs[0]
        variables = [input, old_h, zeta, nu, w, u] + outputs[1:] + [w1, w2, u1, u2]
        ctx.save_for_backward(*variables)
        ctx.non_linearity = gate_non_linearity
        return new_h

    @staticmethod
    def backward(ctx, grad_h):
        input, old_h, zeta, nu, w, u, z, h_prime, w1, w2, u1, u2 = ctx.saved_tensors
        outputs = fastgrnn_cuda.backward(grad_h.contiguous(), input.contiguous(), old_h.contiguous(), zeta, nu,
                                         w, u, z, h_prime, w1, w2, u1, u2, ctx.non_linearity)
        return _as_autograd_grads(outputs, ctx.needs_input_grad)


_unroll_decisions = {}
_zero_states = {}        # (B, H, dtype, device) -> the default h0
_inference_ok = {}       # call signature -> "the hs-only forward is on kernel path 2"


class FastGRNNUnrollFunction(Function):
    """rnn.py:907-972.  Same inputs, same gradients; what is SAVED between the two passes is
    an internal matter: on the split-precision kernel path the forward keeps one auxiliary
    tensor (the pre-activation W.x+U.h) instead of the reference's two (z_s, h_prime_s) and the
    backward recomputes the gates from it -- one [T,B,H] HBM write and read less per step."""

    @staticmethod
    def forward(ctx, input, bias_gate, bias_update, zeta, nu, old_h, w, u, w1, w2, u1, u2, gate_non_linearity,
                batch_major=False, last_state=False):
        """``batch_major`` (not in the reference signature): ``input`` / the result are [B,T,.] and the
        kernels index them in place (FLAG_BATCH_MAJOR) instead of working on transposed copies.
        ``last_state``: return h_T [B,H] alone -- what the classifier head reads
        (model.py:227) -- so that the backward receives a [B,H] gradient instead of the dense, all-but-one-
        step-zero [T,B,H] tensor autograd builds for ``hs[-1]`` (FLAG_GRAD_LAST: not written, not read)."""
        old_h = old_h.contiguous()
        # Which contract / layout flags this call runs under is a pure function of its shapes, dtype and strides:
        # decided once per signature (three descriptor look-ups), then one dictionary hit per step.
        key = (input.shape, input.stride(), input.dtype, input.is_cuda, old_h.shape, w1.shape, u1.shape,
               gate_non_linearity, batch_major, last_state)
        dec = _unroll_decisions.get(key)
        if dec is None:
            if batch_major:
                B, T, F = input.shape
            else:
                T, B, F = input.shape
            H = old_h.shape[1]
            rw = w1.shape[0] if w1.numel() else 0
            ru = u1.shape[0] if u1.numel() else 0
            # The trainer hands over permute(2,0,1) of the loader's [B,F,T] batch (trainClassifier.py:204,299), a
            # [T,B,F] VIEW; the reference copies it here with .contiguous().  Where the kernels can read [B,F,T]
            # in place (FLAG_X_BFT) the view's base is passed instead and d_input comes back as the same view.
            x_bft = (not batch_major and input.is_cuda and not input.is_contiguous()
                     and input.permute(1, 2, 0).is_contiguous()
                     and input.dtype in (torch.float32, torch.bfloat16)
                     and fastgrnn_cuda.kernel_path(T, B, F, H, rw, ru, gate_non_linearity, 2, input.dtype, 1,
                                                   _lib.FLAG_SAVE_PREACT | _lib.FLAG_X_BFT) == 2)
            preact = (input.dtype in (torch.float32, torch.bfloat16) and input.is_cuda and
                      fastgrnn_cuda.kernel_path(T, B, F, H, rw, ru, gate_non_linearity, 2, input.dtype, 1,
                                                _lib.FLAG_SAVE_PREACT) == 2)
            flags = _lib.FLAG_SAVE_PREACT if preact else 0
            if batch_major:
                flags |= _lib.FLAG_BATCH_MAJOR
            if x_bft:
                flags |= _lib.FLAG_X_BFT
            grad_last = bool(last_state and input.is_cuda and fastgrnn_cuda.kernel_path(
                T, B, F, H, rw, ru, gate_non_linearity, 2, input.dtype, 1, flags | _lib.FLAG_GRAD_LAST) == 2)
            dec = _unroll_decisions[key] = (x_bft, preact, flags, grad_last)
        x_bft, preact, flags, grad_last = dec
        input = input.permute(1, 2, 0) if x_bft else input.contiguous()
        ctx.flags = flags
        outputs = fastgrnn_cuda.forward_unroll(input, w, u, bias_gate, bias_update, zeta, nu, old_h,
                                               gate_non_linearity, w1, w2, u1, u2, flags=flags)
        hidden_states = outputs[0]
        if preact:
            # outputs[2] (factorised operands only): the rank-space vector [U1.h | W1.x] per step
            aux2 = outputs[2] if len(outputs) > 2 else outputs[1]
            variables = [input, hidden_states, zeta, nu, w, u, outputs[1], aux2, bias_gate, bias_update,
                         old_h, w1, w2, u1, u2]
        else:
            variables = [input, hidden_states, zeta, nu, w, u] + outputs[1:] + [old_h, w1, w2, u1, u2]
        ctx.save_for_backward(*variables)
        ctx.gate_non_linearity = gate_non_linearity
        ctx.preact = preact
        ctx.last_state = bool(last_state)
        if last_state:
            ctx.grad_last = grad_last
            return (hidden_states[:, -1] if batch_major else hidden_states[-1]).clone()
        return hidden_states

    @staticmethod
    def backward(ctx, grad_h):
        flags = ctx.flags
        if ctx.last_state:
            if ctx.grad_last:
                flags |= _lib.FLAG_GRAD_LAST            # the kernel takes the [B,H] gradient as it is
            else:                                        # other kernel paths: the dense form autograd would build
                hs_saved = ctx.saved_tensors[1]
                dense = torch.zeros_like(hs_saved)
                (dense[:, -1] if flags & _lib.FLAG_BATCH_MAJOR else dense[-1]).copy_(grad_h)
                grad_h = dense
        if ctx.preact:
            (input, hidden_states, zeta, nu, w, u, pre_s, aux2, bias_gate, bias_update, old_h,
             w1, w2, u1, u2) = ctx.saved_tensors
            outputs = fastgrnn_cuda.backward_unroll(grad_h.contiguous(), input, hidden_states, zeta, nu, w, u,
                                                    pre_s, aux2, old_h, w1, w2, u1, u2, ctx.gate_non_linearity,
                                                    flags=flags, bias_gate=bias_gate,
                                                    bias_update=bias_update, need_dx=ctx.needs_input_grad[0])
        else:
            (input, hidden_states, zeta, nu, w, u, z_s, h_prime_s, old_h, w1, w2, u1, u2) = ctx.saved_tensors
            outputs = fastgrnn_cuda.backward_unroll(grad_h.contiguous(), input, hidden_states, zeta, nu, w, u,
                                                    z_s, h_prime_s, old_h, w1, w2, u1, u2,
                                                    ctx.gate_non_linearity, flags=flags,
                                                    need_dx=ctx.needs_input_grad[0])
        if ctx.flags & _lib.FLAG_X_BFT:                 # d_input was produced as [B,F,T]: hand back the [T,B,F] view
            outputs = [outputs[0].permute(2, 0, 1) if outputs[0].numel() else outputs[0]] + list(outputs[1:])
        return _as_autograd_grads(outputs, ctx.needs_input_grad) + (None, None)


def _as_autograd_grads(outputs, needs):
    """Map the operator's 12-tuple (.cu:556) onto the Function's 13 inputs
    (input, bias_gate, bias_update, zeta, nu, old_h, w, u, w1, w2, u1, u2, nl):
    same order + trailing None (rnn.py:905,972); empty(0) placeholders become None."""
    grads = []
    for g, need in zip(outputs, needs[:12]):
        grads.append(g if (need and g.numel() != 0) else None)
    return tuple(grads + [None])


def _make_params(mod, input_size, hidden_size, wRank, uRank, zetaInit, nuInit, device):
    """Parameter set of rnn.py:491-517 / 782-805: [out,in] layout, 0.1*randn, biases one."""
    if wRank is None:
        mod.W = nn.Parameter(0.1 * torch.randn([hidden_size, input_size], device=device))
        mod.register_buffer("W1", torch.empty(0), persistent=False)
        mod.register_buffer("W2", torch.empty(0), persistent=False)
    else:
        mod.register_buffer("W", torch.empty(0), persistent=False)
        mod.W1 = nn.Parameter(0.1 * torch.randn([wRank, input_size], device=device))
        mod.W2 = nn.Parameter(0.1 * torch.randn([hidden_size, wRank], device=device))
    if uRank is None:
        mod.U = nn.Parameter(0.1 * torch.randn([hidden_size, hidden_size], device=device))
        mod.register_buffer("U1", torch.empty(0), persistent=False)
        mod.register_buffer("U2", torch.empty(0), persistent=False)
    else:
        mod.register_buffer("U", torch.empty(0), persistent=False)
        mod.U1 = nn.Parameter(0.1 * torch.randn([uRank, hidden_size], device=device))
        mod.U2 = nn.Parameter(0.1 * torch.randn([hidden_size, uRank], device=device))
    mod.bias_gate = nn.Parameter(torch.ones([1, hidden_size], device=device))
    mod.bias_update = nn.Parameter(torch.ones([1, hidden_size], device=device))
    mod.zeta = nn.Parameter(zetaInit * torch.ones([1, 1], device=device))
    mod.nu = nn.Parameter(nuInit * torch.ones([1, 1], device=device))


def _get_vars(mod):
    # rnn.py:536-549, 828-841
    Vars = []
    if mod._num_W_matrices == 1:
        Vars.append(mod.W)
    else:
        Vars.extend([mod.W1, mod.W2])
    if mod._num_U_matrices == 1:
        Vars.append(mod.U)
    else:
        Vars.extend([mod.U1, mod.U2])
    Vars.extend([mod.bias_gate, mod.bias_update, mod.zeta, mod.nu])
    return Vars


# ---- sparsification bookkeeping (rnn.py:843-889), all on the parameters' device ----------
def _get_model_size(mod):
    """rnn.py:843-865: 4 bytes x (non-zeros of the W/U matrices when their sparsity flag is set, else their
    sizes) + biases + zeta, nu.  The reference counts on CPU copies; ``isSparse`` is the truthiness of the
    sparsity value exactly as there (rnn.py:855-860)."""
    mats = mod.getVars()
    endW = mod._num_W_matrices
    endU = endW + mod._num_U_matrices
    total = 2
    for i in range(0, endW):
        total += utils.countNNZ(mats[i], mod._wSparsity)
    for i in range(endW, endU):
        total += utils.countNNZ(mats[i], mod._uSparsity)
    for i in range(endU, len(mats)):
        total += utils.countNNZ(mats[i], False)
    return total * 4


def _copy_previous_UW(mod):
    mats = mod.getVars()
    n = mod._num_W_matrices + mod._num_U_matrices
    mod.oldmats = [mats[i].detach().clone() for i in range(n)]


def _sparsify(mod):
    """Hard-threshold W/U (or their factors) IN PLACE and remember the support.  This is what the CPU cell does
    (``mats[i].data = utils.hardThreshold(...)``, rnn.py:433-443); the reference's CUDA class rebinds list
    entries instead (rnn.py:875-883), which leaves its parameters untouched -- a defect not reproduced."""
    mats = mod.getVars()
    endW = mod._num_W_matrices
    endU = endW + mod._num_U_matrices
    for i in range(0, endW):
        utils.hardThreshold(mats[i], mod._wSparsity)
    for i in range(endW, endU):
        utils.hardThreshold(mats[i], mod._uSparsity)
    _copy_previous_UW(mod)


def _sparsify_with_support(mod):
    mats = mod.getVars()
    endU = mod._num_W_matrices + mod._num_U_matrices
    for i in range(0, endU):
        utils.supportBasedThreshold(mats[i], mod.oldmats[i])


class FastGRNNCUDACell(nn.Module):
    """Single-step GPU cell (rnn.py:454-549).

    z_t = non_linearity(W x_t + U h_{t-1} + B_g);  h_t^ = tanh(W x_t + U h_{t-1} + B_h)
    h_t = z_t*h_{t-1} + (sigmoid(zeta)(1-z_t) + sigmoid(nu))*h_t^
    """

    def __init__(self, input_size, hidden_size, gate_nonlinearity="sigmoid",
                 update_nonlinearity="tanh", wRank=None, uRank=None, zetaInit=1.0, nuInit=-4.0,
                 wSparsity=1.0, uSparsity=1.0, name="FastGRNNCUDACell", device=None):
        super().__init__()
        self.device = _resolve_device(device)
        if update_nonlinearity != "tanh":
            raise ValueError("FastGRNNCUDA fixes update_nonlinearity to tanh (rnn.py:741-742)")
        self._input_size = input_size
        self._hidden_size = hidden_size
        self._gate_nonlinearity = gate_nonlinearity
        self._update_nonlinearity = update_nonlinearity
        self._zetaInit = zetaInit
        self._nuInit = nuInit
        self._name = name
        self._wRank, self._uRank = wRank, uRank
        self._wSparsity, self._uSparsity = wSparsity, uSparsity
        self._num_W_matrices = 1 if wRank is None else 2
        self._num_U_matrices = 1 if uRank is None else 2
        self._num_biases = 2
        self._num_weight_matrices = [self._num_W_matrices, self._num_U_matrices, self._num_biases]
        _make_params(self, input_size, hidden_size, wRank, uRank, zetaInit, nuInit, self.device)
        self._gate_non_linearity = NON_LINEARITY[gate_nonlinearity]   # KeyError on others, as rnn.py:512
        self.oldmats = []

    @property
    def name(self):
        return self._name

    @property
    def cellType(self):
        return "FastGRNNCUDACell"

    @property
    def state_size(self):
        return self._hidden_size

    @property
    def input_size(self):
        return self._input_size

    @property
    def output_size(self):
        return self._hidden_size

    def forward(self, input, state):
        return FastGRNNFunction.apply(input, self.bias_gate, self.bias_update, self.zeta, self.nu, state,
                                      self.W, self.U, self.W1, self.W2, self.U1, self.U2,
                                      self._gate_non_linearity)

    def getVars(self):
        return _get_vars(self)

    def get_model_size(self):
        return _get_model_size(self)

    def copy_previous_UW(self):
        _copy_previous_UW(self)

    def sparsify(self):
        _sparsify(self)

    def sparsifyWithSupport(self):
        _sparsify_with_support(self)


class FastGRNNCUDA(nn.Module):
    """Unrolled GPU FastGRNN (rnn.py:738-889).  ``update_nonlinearity`` is fixed to tanh,
    only ``gate_nonlinearity`` is configurable (rnn.py:741-742)."""

    def __init__(self, input_size, hidden_size, gate_nonlinearity="sigmoid",
                 update_nonlinearity="tanh", wRank=None, uRank=None,
                 wSparsity=1.0, uSparsity=1.0, zetaInit=1.0, nuInit=-4.0,
                 batch_first=False, name="FastGRNNCUDA", device=None):
        super().__init__()
        self.device = _resolve_device(device)
        if update_nonlinearity != "tanh":
            raise ValueError("FastGRNNCUDA fixes update_nonlinearity to tanh (rnn.py:741-742)")
        self._input_size = input_size
        self._hidden_size = hidden_size
        self._zetaInit = zetaInit
        self._nuInit = nuInit
        self._name = name
        self._wRank, self._uRank = wRank, uRank
        self._wSparsity, self._uSparsity = wSparsity, uSparsity
        self._num_W_matrices = 1 if wRank is None else 2
        self._num_U_matrices = 1 if uRank is None else 2
        self._num_biases = 2
        self._num_weight_matrices = [self._num_W_matrices, self._num_U_matrices, self._num_biases]
        self.oldmats = []
        self.batch_first = batch_first
        _make_params(self, input_size, hidden_size, wRank, uRank, zetaInit, nuInit, self.device)
        self._gate_non_linearity = NON_LINEARITY[gate_nonlinearity]

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self.device = self.bias_gate.device
        return out

    def forward(self, input, hiddenState=None, cell_state=None, last_state=False):
        """input: [timesteps, batch, features] (or [batch, timesteps, features] when
        ``batch_first``); hiddenState: [batch, state_size], zeros if not provided
        (rnn.py:807-826).  Returns every hidden state, same leading layout as the input -- or, with
        ``last_state=True`` (not in the reference signature; the last layer under the classifier head,
        model.py:227), the final state [batch, state_size] alone: equal to ``forward(...)[-1]`` with the same
        gradients, without the dense mostly-zero grad_hs on the way back; under ``torch.no_grad()`` the
        hidden-state sequence is not written at all (FLAG_HS_LAST)."""
        if not input.is_cuda:
            input = input.to(self.device)
        # batch_first: the reference transposes to [T,B,F] and back (rnn.py:812-813,823-825); where the
        # kernels can index [B,T,.] in place (FLAG_BATCH_MAJOR) no copy is made at all
        in_place = False
        if self.batch_first is True:
            Bn, Tn, Fn = input.shape
            rw = self.W1.shape[0] if self.W1.numel() else 0
            ru = self.U1.shape[0] if self.U1.numel() else 0
            in_place = (input.dtype in (torch.float32, torch.bfloat16) and fastgrnn_cuda.kernel_path(
                Tn, Bn, Fn, self._hidden_size, rw, ru, self._gate_non_linearity, 2, input.dtype, 1,
                _lib.FLAG_SAVE_PREACT | _lib.FLAG_BATCH_MAJOR) == 2)
            if not in_place:
                input = input.transpose(0, 1).contiguous()
        nbatch = input.shape[0] if in_place else input.shape[1]
        if hiddenState is None:
            # bf16 sequences (BASELINE config "bf16 with fp32 master grads") keep an fp32 state and parameters
            hdt = torch.float32 if input.dtype == torch.bfloat16 else input.dtype
            # (the default state is read-only to every kernel and never handed out: one tensor per shape instead of a
            # fill launch per call)
            zkey = (nbatch, self._hidden_size, hdt, input.device)
            hiddenState = None if torch.is_inference_mode_enabled() else _zero_states.get(zkey)
            if hiddenState is None:
                hiddenState = torch.zeros([nbatch, self._hidden_size], dtype=hdt, device=input.device)
                if not torch.is_inference_mode_enabled():   # (an inference tensor could not be saved for a backward later)
                    if len(_zero_states) >= 16:
                        _zero_states.clear()
                    _zero_states[zkey] = hiddenState
        if not hiddenState.is_cuda:
            hiddenState = hiddenState.to(self.device)
        if last_state and not torch.is_grad_enabled():
            out = self._last_state_inference(input, hiddenState, in_place)
            if out is not None:
                return out
        if not last_state and not torch.is_grad_enabled():
            out = self._inference_forward(input, hiddenState, in_place)
            if out is not None:
                return out.transpose(0, 1) if (self.batch_first is True and not in_place) else out
        result = FastGRNNUnrollFunction.apply(input, self.bias_gate, self.bias_update, self.zeta, self.nu,
                                              hiddenState, self.W, self.U, self.W1, self.W2, self.U1, self.U2,
                                              self._gate_non_linearity, in_place, bool(last_state))
        if self.batch_first is True and not in_place and not last_state:
            return result.transpose(0, 1)
        return result

    def _inference_forward(self, input, hiddenState, batch_major):
        """hs alone when nothing will be differentiated (torch.no_grad()): no tensor is saved for a backward, so the
        scan writes one [T,B,H] tensor instead of two.  None where the kernels have no such variant (the caller then
        goes through the autograd function as before)."""
        if not (input.is_cuda and input.is_contiguous() and input.dtype in (torch.float32, torch.bfloat16)):
            return None
        key = (input.shape, input.dtype, batch_major, self.W1.shape, self.U1.shape, self._gate_non_linearity)
        ok = _inference_ok.get(key)
        flags = _lib.FLAG_BATCH_MAJOR if batch_major else 0
        if ok is None:
            if batch_major:
                Bn, Tn, Fn = input.shape
            else:
                Tn, Bn, Fn = input.shape
            rw = self.W1.shape[0] if self.W1.numel() else 0
            ru = self.U1.shape[0] if self.U1.numel() else 0
            ok = _inference_ok[key] = fastgrnn_cuda.kernel_path(
                Tn, Bn, Fn, self._hidden_size, rw, ru, self._gate_non_linearity, 2, input.dtype, 0, flags) == 2
        if not ok:
            return None
        return fastgrnn_cuda.forward_unroll(input, self.W, self.U, self.bias_gate, self.bias_update, self.zeta, self.nu,
                                            hiddenState.contiguous(), self._gate_non_linearity, self.W1, self.W2,
                                            self.U1, self.U2, want_gates=False, flags=flags)[0]

    def _last_state_inference(self, input, hiddenState, batch_major):
        """h_T without writing hs[T,B,H] (FLAG_HS_LAST); None where the kernels cannot (other shapes / paths)."""
        if batch_major:
            Bn, Tn, Fn = input.shape
        else:
            Tn, Bn, Fn = input.shape
        rw = self.W1.shape[0] if self.W1.numel() else 0
        ru = self.U1.shape[0] if self.U1.numel() else 0
        flags = _lib.FLAG_HS_LAST | (_lib.FLAG_BATCH_MAJOR if batch_major else 0)
        if input.dtype not in (torch.float32, torch.bfloat16) or fastgrnn_cuda.kernel_path(
                Tn, Bn, Fn, self._hidden_size, rw, ru, self._gate_non_linearity, 2, input.dtype, 0, flags) != 2:
            return None
        return fastgrnn_cuda.forward_unroll(input.contiguous(), self.W, self.U, self.bias_gate, self.bias_update,
                                            self.zeta, self.nu, hiddenState.contiguous(), self._gate_non_linearity,
                                            self.W1, self.W2, self.U1, self.U2, want_gates=False, flags=flags)[0]

    def getVars(self):
        return _get_vars(self)

    def get_model_size(self):
        """rnn.py:843-865."""
        return _get_model_size(self)

    def copy_previous_UW(self):
        """rnn.py:867-873."""
        _copy_previous_UW(self)

    def sparsify(self):
        """rnn.py:875-883 (with the in-place effect the CPU cell has, rnn.py:433-443)."""
        _sparsify(self)

    def sparsifyWithSupport(self):
        """rnn.py:885-889."""
        _sparsify_with_support(self)
