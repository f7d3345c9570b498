"""Drop-in for the reference's native operator module ``fastgrnn_cuda``
(/root/reference cuda/fastgrnn_cuda.cpp:235-240): the same four functions with the
same positional argument orders and return lists, backed by the C ABI of
``libfastgrnn_hip.so`` (include/fastgrnn_hip.h).

    forward(input, w, u, bias_gate, bias_update, zeta, nu, old_h, z_non_linearity,
            w1, w2, u1, u2)                                  -> [new_h, z, h_prime]
    backward(grad_h, input, old_h, zeta, nu, w, u, z, h_prime, w1, w2, u1, u2,
             z_non_linearity)                                -> 12 tensors
    forward_unroll(input, w, u, bias_gate, bias_update, zeta, nu, initial_h,
                   z_non_linearity, w1, w2, u1, u2)          -> [hs, z_s, h_prime_s]
    backward_unroll(grad_h, input, hidden_states, zeta, nu, w, u, z, h_prime,
                    initial_h, w1, w2, u1, u2, z_non_linearity) -> 12 tensors

(fastgrnn_cuda.cpp:73-232).  The 12-tuple order is
``d_input, d_bias_z, d_bias_h_prime, d_zeta, d_nu, d_old_h, d_w, d_u, d_w1, d_w2,
d_u1, d_u2`` (.cu:317,556); operands that do not apply are ``torch.empty(0)`` on both
sides of the call (rnn.py:783-798, .cu:221-224).

This shim only checks arguments the way ``CHECK_INPUT`` does
(fastgrnn_cuda.cpp:69-71 -> RuntimeError), allocates outputs + workspace with torch,
and launches on torch's CURRENT stream of the input's device.  All arithmetic is in
the HIP kernels; there is no eager fallback.
"""
from __future__ import annotations

import ctypes as C
import functools

import torch

from . import _lib

_DTYPES = {torch.float32: _lib.F32, torch.float64: _lib.F64, torch.bfloat16: _lib.BF16_IO}


def _param_dtype(io_dtype):
    """bf16 sequences (x, hs, grad_hs, d_x) go with fp32 parameters, h0 and gradients."""
    return torch.float32 if io_dtype == torch.bfloat16 else io_dtype

# Optional per-launch timing (bench.py): when set to a list, every C-ABI call appends
# (tag, start_event, end_event) recorded on the stream the kernels are launched on.
_timing = None


class _Timed:
    def __init__(self, tag, device):
        self.tag, self.device = tag, device

    def __enter__(self):
        if _timing is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record(torch.cuda.current_stream(self.device))

    def __exit__(self, *exc):
        if _timing is not None:
            self.e1.record(torch.cuda.current_stream(self.device))
            _timing.append((self.tag, self.e0, self.e1))
        return False


def _check_input(t, name):
    # CHECK_CUDA / CHECK_CONTIGUOUS, fastgrnn_cuda.cpp:69-71
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA tensor" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)


def _present(t):
    # reference: low-rank is detected by w1.size(0) != 0 (.cu:138-139)
    return t is not None and t.numel() != 0


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if _present(t) else C.c_void_p(None)


def _expect(t, shape, name):
    if tuple(t.shape) != tuple(shape):
        raise RuntimeError("%s has shape %s, expected %s" % (name, tuple(t.shape), tuple(shape)))


def _describe(T, B, F, H, w, u, w1, w2, u1, u2, bias_gate, bias_update, zeta, nu, dtype, gate_nl,
              update_nl, flags):
    """Validate parameters against (F,H) and build the C descriptor + params struct."""
    if dtype not in _DTYPES:
        raise RuntimeError("fastgrnn: unsupported dtype %s (float32/float64, or bfloat16 sequences)" % dtype)
    io_dtype, dtype = dtype, _param_dtype(dtype)
    w_lr, u_lr = _present(w1), _present(u1)
    if w_lr:
        _check_input(w1, "w1"); _check_input(w2, "w2")
        rw = w1.shape[0]
        _expect(w1, (rw, F), "w1"); _expect(w2, (H, rw), "w2")
    else:
        _check_input(w, "w")
        rw = 0
        _expect(w, (H, F), "w")
    if u_lr:
        _check_input(u1, "u1"); _check_input(u2, "u2")
        ru = u1.shape[0]
        _expect(u1, (ru, H), "u1"); _expect(u2, (H, ru), "u2")
    else:
        _check_input(u, "u")
        ru = 0
        _expect(u, (H, H), "u")
    for t, n in ((zeta, "zeta"), (nu, "nu")):
        _check_input(t, n)
        if t.numel() != 1:
            raise RuntimeError("%s must hold one element" % n)
    tensors = [t for t in (w, u, w1, w2, u1, u2, bias_gate, bias_update, zeta, nu) if _present(t)]
    for t in tensors:
        if t.dtype != dtype:
            raise RuntimeError("fastgrnn: all operands must share dtype %s (got %s)" % (dtype, t.dtype))
    desc = _plan(T, B, F, H, rw, ru, int(gate_nl), int(update_nl), _DTYPES[io_dtype], int(flags))
    params = _lib.Params(_ptr(None if w_lr else w), _ptr(None if u_lr else u),
                         _ptr(w1 if w_lr else None), _ptr(w2 if w_lr else None),
                         _ptr(u1 if u_lr else None), _ptr(u2 if u_lr else None),
                         _ptr(bias_gate), _ptr(bias_update), _ptr(zeta), _ptr(nu))
    return desc, params, w_lr, u_lr          # desc: the cached plan tuple (struct, paths, workspace sizes)


# Scratch workspace, kept per (device, stream) and grown on demand: its contents never outlive a call and every use
# is stream-ordered, so consecutive calls on one stream can share it (a fresh torch.empty per call was a few
# microseconds of host time per step; the 8-GPU step adds the collective's latency on top of whatever the host spends)
_ws_cache = {}


def _workspace(nbytes, device):
    if nbytes == 0:
        return None, C.c_void_p(None)
    key = (device.index, _raw_stream(device))
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws, C.c_void_p(ws.data_ptr())


@functools.lru_cache(maxsize=1024)
def _plan(T, B, F, H, rw, ru, gate_nl, update_nl, dtype_code, flags):
    """Everything about a descriptor that does not depend on the tensors: the C struct itself, the kernel family
    each direction dispatches to and the workspace sizes (pure functions of the descriptor in the C ABI).  One
    dictionary lookup per call instead of four ctypes round trips."""
    lib = _lib.load()
    desc = _lib.Desc(T, B, F, H, rw, ru, int(gate_nl), int(update_nl), dtype_code, int(flags))
    return (desc, lib.fastgrnn_hip_kernel_path(C.byref(desc), 0), lib.fastgrnn_hip_kernel_path(C.byref(desc), 1),
            int(lib.fastgrnn_hip_forward_workspace_bytes(C.byref(desc))),
            int(lib.fastgrnn_hip_backward_workspace_bytes(C.byref(desc))))


_cliff_warned = set()


def _warn_fallback(plan, direction):
    """Perf cliffs are not silent: the first call of a shape that lands on the generic scan (kernel path 0) at a size
    where that matters says so once (include/fastgrnn_hip.h lists what runs on the matrix pipe)."""
    d = plan[0]
    if plan[1 + direction] != 0 or d.T * d.B < 4096 or (d.flags & _lib.FLAG_FORCE_GENERIC):
        return
    key = (d.T, d.B, d.F, d.H, d.w_rank, d.u_rank, d.gate_nl, d.update_nl, d.dtype, d.flags, direction)
    if key in _cliff_warned:
        return
    _cliff_warned.add(key)
    import warnings
    warnings.warn("fastgrnn: %s of shape T=%d B=%d F=%d H=%d ranks=(%d,%d) dtype=%d flags=0x%x runs on the generic "
                  "scan (kernel path 0), typically 20-30x slower than the matrix-pipe kernels; see "
                  "include/fastgrnn_hip.h (fastgrnn_hip_kernel_path) for the shapes those cover"
                  % ("backward" if direction else "forward", d.T, d.B, d.F, d.H, d.w_rank, d.u_rank, d.dtype, d.flags),
                  RuntimeWarning, stacklevel=4)


def _raw_stream(device):
    """The current stream's handle as an integer (the raw-stream query is one C call; building a torch.cuda.Stream
    object for it cost ~5 us, twice per operator call)."""
    idx = device.index
    return _get_raw_stream(torch.cuda.current_device() if idx is None else idx)


_get_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or \
    (lambda idx: torch.cuda.current_stream(idx).cuda_stream)


def _stream(device):
    return C.c_void_p(_raw_stream(device))


def kernel_path(T, B, F, H, w_rank=0, u_rank=0, gate_nl=0, update_nl=2, dtype=torch.float32,
                direction=0, flags=0):
    """0 = generic scan, 1 = fp32-MFMA scan, 2 = split-precision scan (pure function of the descriptor; cached)."""
    return _plan(T, B, F, H, w_rank, u_rank, int(gate_nl), int(update_nl), _DTYPES[dtype], int(flags))[1 + int(direction)]


# Signatures (shapes, dtypes, flags of one call) that have passed the full argument checks once.  A training loop makes
# the same call thousands of times: after the first one only what can change from call to call without changing the
# signature is re-checked (device, contiguity), and the checks that are pure functions of the signature are skipped --
# half of the host time of a step (tools/host_overhead.py: the host's enqueue time is within reach of the GPU's).
_seen = {}
_use_seen = True                 # (tools/host_overhead.py switches it off for its A/B)


def _all_dense_cuda(*ts):
    for t in ts:
        if t.numel() and not (t.is_cuda and t.is_contiguous()):
            return False
    return True


def _forward_impl(input, w, u, bias_gate, bias_update, zeta, nu, h0, gate_nl, w1, w2, u1, u2,
                  unrolled, update_nl, want_gates, flags):
    lib = _lib.load()
    sig = ("f", unrolled, input.shape, input.dtype, h0.shape, h0.dtype, w.shape, u.shape, w1.shape, w2.shape,
           u1.shape, u2.shape, bias_gate.shape, bias_update.shape, zeta.shape, nu.shape,
           (w if w.numel() else w1).dtype, (u if u.numel() else u1).dtype, bias_gate.dtype, bias_update.dtype,
           zeta.dtype, nu.dtype, gate_nl, update_nl, flags, want_gates, input.device.index)
    ent = _seen.get(sig) if _use_seen else None
    if ent is not None and _all_dense_cuda(input, h0, w, u, w1, w2, u1, u2, bias_gate, bias_update, zeta, nu):
        plan, w_lr, u_lr, oshape, hshape, rank_space, nbytes, T, B, H = ent
        desc = plan[0]
        params = _lib.Params(_ptr(None if w_lr else w), _ptr(None if u_lr else u),
                             _ptr(w1 if w_lr else None), _ptr(w2 if w_lr else None),
                             _ptr(u1 if u_lr else None), _ptr(u2 if u_lr else None),
                             _ptr(bias_gate), _ptr(bias_update), _ptr(zeta), _ptr(nu))
        dev = input.device
        preact = bool(flags & _lib.FLAG_SAVE_PREACT)
        pdt = h0.dtype
        with torch.cuda.device(dev):
            hs = torch.empty(hshape, dtype=input.dtype, device=dev)
            zs = torch.empty(oshape, dtype=pdt, device=dev) if (want_gates or preact) else None
            cs = torch.empty(oshape, dtype=pdt, device=dev) if (want_gates and not preact) else None
            if rank_space:
                cs = torch.empty((T * B, 32), dtype=pdt, device=dev)
            ws, wsp = _workspace(nbytes, dev)
            fn = lib.fastgrnn_hip_forward_unroll if unrolled else lib.fastgrnn_hip_forward
            with _Timed("forward", dev):
                st = fn(C.byref(desc), C.byref(params), _ptr(input), _ptr(h0), _ptr(hs), _ptr(zs), _ptr(cs),
                        wsp, nbytes, _stream(dev))
            _lib.check(st, "fastgrnn forward_unroll" if unrolled else "fastgrnn forward")
            del ws
        if preact:
            return [hs, zs] if cs is None else [hs, zs, cs]
        return [hs, zs, cs] if want_gates else [hs]
    _check_input(input, "input")
    _check_input(bias_gate, "bias_gate"); _check_input(bias_update, "bias_update")
    _check_input(h0, "initial_h" if unrolled else "old_h")
    batch_major = bool(flags & _lib.FLAG_BATCH_MAJOR)
    if unrolled:
        if input.dim() != 3:
            raise RuntimeError("input must be [timesteps, batch, features]")
        if flags & _lib.FLAG_X_BFT:          # the trainer's [B,F,T] (trainClassifier.py:204)
            B, F, T = input.shape
        elif batch_major:
            B, T, F = input.shape
        else:
            T, B, F = input.shape
    else:
        if input.dim() != 2:
            raise RuntimeError("input must be [batch, features]")
        T = 1
        B, F = input.shape
    H = h0.shape[-1]
    _expect(h0, (B, H), "initial_h" if unrolled else "old_h")
    if bias_gate.numel() != H or bias_update.numel() != H:
        raise RuntimeError("bias_gate/bias_update must hold H=%d elements" % H)
    pdt = _param_dtype(input.dtype)
    if h0.dtype != pdt:
        raise RuntimeError("input and hidden state dtypes differ" if pdt == input.dtype
                           else "bfloat16 sequences take a float32 hidden state")
    plan, params, _, _ = _describe(T, B, F, H, w, u, w1, w2, u1, u2, bias_gate, bias_update, zeta, nu,
                                   input.dtype, gate_nl, update_nl, flags)
    desc = plan[0]
    _warn_fallback(plan, 0)
    dev = input.device
    oshape = ((B, T, H) if batch_major else (T, B, H)) if unrolled else (B, H)
    preact = bool(flags & _lib.FLAG_SAVE_PREACT)
    hs_last = bool(unrolled and (flags & _lib.FLAG_HS_LAST))
    if hs_last and (want_gates or preact):
        raise RuntimeError("FLAG_HS_LAST is an inference mode: nothing can be saved for a backward "
                           "(want_gates=False and no FLAG_SAVE_PREACT)")
    with torch.cuda.device(dev):
        hs = torch.empty((B, H) if hs_last else oshape, dtype=input.dtype, device=dev)
        zs = torch.empty(oshape, dtype=pdt, device=dev) if (want_gates or preact) else None
        cs = torch.empty(oshape, dtype=pdt, device=dev) if (want_gates and not preact) else None
        if preact and desc.H == 256 and desc.F == 32 and 0 < desc.w_rank <= 16 and 0 < desc.u_rank <= 16:
            # factorised operands: the forward also saves the rank-space vector [U1.h | W1.x] per step, always
            # time-major and zero-extended to 16 + 16 columns (an opaque tensor for the backward)
            cs = torch.empty((T * B, 32), dtype=pdt, device=dev)
        # (dense H=128 layers with a wide input keep the frame product in the workspace only when no auxiliary
        # output is requested: include/fastgrnn_hip.h, forward workspace)
        wide = desc.H == 128 and desc.F > 32 and not desc.w_rank and not desc.u_rank
        nbytes = 0 if (plan[1] == 2 and zs is not None and wide) else plan[3]
        ws, wsp = _workspace(nbytes, dev)
        fn = lib.fastgrnn_hip_forward_unroll if unrolled else lib.fastgrnn_hip_forward
        _seen[sig] = (plan, _present(w1), _present(u1), oshape, (B, H) if hs_last else oshape,
                      bool(preact and cs is not None), nbytes, T, B, H)
        with _Timed("forward", dev):
            st = fn(C.byref(desc), C.byref(params), _ptr(input), _ptr(h0), _ptr(hs), _ptr(zs), _ptr(cs),
                    wsp, nbytes, _stream(dev))
        _lib.check(st, "fastgrnn forward_unroll" if unrolled else "fastgrnn forward")
        del ws                   # (cached per stream: reuse by the next call is stream-ordered behind these launches)
    if preact:                   # zs holds the pre-activation W.x + U.h
        if zs is None:
            zs = torch.empty(0, dtype=pdt, device=dev)
        return [hs, zs] if cs is None else [hs, zs, cs]
    return [hs, zs, cs] if want_gates else [hs]


def _launch_backward(lib, plan, ent, unrolled, preact, grad_h, input, hs_or_old_h, z, h_prime, rank_space, h0,
                     w, u, w1, w2, u1, u2, b0, b1, zeta, nu, need_dx):
    """Allocate the 12 outputs (the parameter gradients as views of ONE flat buffer) and launch."""
    w_lr, u_lr, shapes, sizes, dx_is_gemm, B, H = ent
    desc = plan[0]
    dev = input.device
    dt = input.dtype
    pdt = h0.dtype
    params = _lib.Params(_ptr(None if w_lr else w), _ptr(None if u_lr else u),
                         _ptr(w1 if w_lr else None), _ptr(w2 if w_lr else None),
                         _ptr(u1 if u_lr else None), _ptr(u2 if u_lr else None),
                         _ptr(b0), _ptr(b1), _ptr(zeta), _ptr(nu))
    with torch.cuda.device(dev):
        none = _NONE
        # the input's gradient is a GEMM of its own on these shapes (fastgrnn_hip.h, fastgrnn_grads.d_x): skipped
        # when autograd does not ask for it (a model's first layer)
        d_input = none if (dx_is_gemm and not need_dx) else torch.empty(input.shape, dtype=dt, device=dev)
        d_old_h = torch.empty((B, H), dtype=pdt, device=dev)
        # The parameter gradients are views of ONE flat buffer, laid out in the order the modules register
        # their parameters (W | W1,W2 ; U | U1,U2 ; bias_gate ; bias_update ; zeta ; nu).  autograd adopts
        # them as the .grad tensors, so a data-parallel step can all-reduce that buffer in place
        # (kws_amd.dp.GradBucket) instead of packing and unpacking six tensors.
        flat = torch.empty(sum(sizes), dtype=pdt, device=dev)
        views = [v.view(sh) for v, sh in zip(flat.split(sizes), shapes)]
        nw = 2 if w_lr else 1
        nu_ = 2 if u_lr else 1
        d_w = none if w_lr else views[0]
        d_w1, d_w2 = (views[0], views[1]) if w_lr else (none, none)
        d_u = none if u_lr else views[nw]
        d_u1, d_u2 = (views[nw], views[nw + 1]) if u_lr else (none, none)
        d_bz, d_bh, d_zeta, d_nu = views[nw + nu_:nw + nu_ + 4]
        del flat, views
        grads = _lib.Grads(_ptr(d_input), _ptr(d_bz), _ptr(d_bh), _ptr(d_zeta), _ptr(d_nu), _ptr(d_old_h),
                           _ptr(d_w), _ptr(d_u), _ptr(d_w1), _ptr(d_w2), _ptr(d_u1), _ptr(d_u2))
        nbytes = plan[4]
        ws, wsp = _workspace(nbytes, dev)
        with _Timed("backward", dev):
            if unrolled:
                st = lib.fastgrnn_hip_backward_unroll(C.byref(desc), C.byref(params), _ptr(grad_h), _ptr(input),
                                                      _ptr(hs_or_old_h), _ptr(z),
                                                      _ptr(rank_space if preact else h_prime),
                                                      _ptr(h0), C.byref(grads), wsp, nbytes, _stream(dev))
            else:
                st = lib.fastgrnn_hip_backward(C.byref(desc), C.byref(params), _ptr(grad_h), _ptr(input),
                                               _ptr(h0), _ptr(z), _ptr(h_prime), C.byref(grads), wsp, nbytes,
                                               _stream(dev))
        _lib.check(st, "fastgrnn backward_unroll" if unrolled else "fastgrnn backward")
        del ws                   # (cached per stream: reuse by the next call is stream-ordered behind these launches)
    return [d_input, d_bz, d_bh, d_zeta, d_nu, d_old_h, d_w, d_u, d_w1, d_w2, d_u1, d_u2]


_NONE = torch.empty(0)           # the reference's placeholder for operands / gradients that do not apply


def _backward_impl(grad_h, input, hs_or_old_h, zeta, nu, w, u, z, h_prime, h0, w1, w2, u1, u2, gate_nl,
                   unrolled, update_nl, flags, bias_gate=None, bias_update=None, need_dx=True):
    lib = _lib.load()
    preact = bool(flags & _lib.FLAG_SAVE_PREACT)
    sig = ("b", unrolled, grad_h.shape, grad_h.dtype, input.shape, input.dtype, hs_or_old_h.shape, hs_or_old_h.dtype,
           z.shape, z.dtype, h_prime.shape, h_prime.dtype, h0.shape, h0.dtype, w.shape, u.shape, w1.shape, w2.shape,
           u1.shape, u2.shape, (w if w.numel() else w1).dtype, (u if u.numel() else u1).dtype, zeta.shape, nu.shape,
           zeta.dtype, nu.dtype, None if bias_gate is None else (bias_gate.shape, bias_gate.dtype),
           None if bias_update is None else (bias_update.shape, bias_update.dtype),
           gate_nl, update_nl, flags, input.device.index)
    ent = _seen.get(sig) if _use_seen else None
    if ent is not None and _all_dense_cuda(grad_h, input, hs_or_old_h, z, h_prime, h0, w, u, w1, w2, u1, u2, zeta, nu) \
            and (not preact or _all_dense_cuda(bias_gate, bias_update)):
        plan, rs, tail = ent
        return _launch_backward(lib, plan, tail, unrolled, preact, grad_h, input, hs_or_old_h, z,
                                z if preact else h_prime, h_prime if rs else None, h0, w, u, w1, w2, u1, u2,
                                bias_gate if preact else zeta, bias_update if preact else zeta, zeta, nu, need_dx)
    if preact:
        if bias_gate is None or bias_update is None:
            raise RuntimeError("FLAG_SAVE_PREACT backward needs bias_gate and bias_update")
        # (the register-resident low-rank scans, both ranks 1..16, save a rank-space vector; other factorised cells
        # run on the dense kernels and save the pre-activation alone)
        rank_space = h_prime if (_present(w1) and _present(u1) and w1.shape[0] <= 16 and u1.shape[0] <= 16
                                 and tuple(w1.shape[1:]) == (32,) and u1.shape[1] == 256) else None
        if rank_space is not None:
            _check_input(rank_space, "rank_space")
            _expect(rank_space, (hs_or_old_h.numel() // hs_or_old_h.shape[-1], 32), "rank_space")
        h_prime = z              # keeps the shape checks below uniform
    for t, n in ((grad_h, "grad_h"), (input, "input"), (hs_or_old_h, "hidden_states" if unrolled else "old_h"),
                 (z, "z"), (h_prime, "h_prime"), (h0, "initial_h")):
        if not (preact and t.device.type == "meta"):
            _check_input(t, n)
    if unrolled:
        if flags & _lib.FLAG_X_BFT:
            B, F, T = input.shape
        elif flags & _lib.FLAG_BATCH_MAJOR:
            B, T, F = input.shape
        else:
            T, B, F = input.shape
        H = grad_h.shape[-1]
        lead = (B, T) if flags & _lib.FLAG_BATCH_MAJOR else (T, B)
        # FLAG_GRAD_LAST: the gradient of the last state only (the classifier head's view, model.py:227)
        _expect(grad_h, (B, H) if flags & _lib.FLAG_GRAD_LAST else lead + (H,), "grad_h")
        _expect(hs_or_old_h, lead + (H,), "hidden_states")
        _expect(z, lead + (H,), "z"); _expect(h_prime, lead + (H,), "h_prime"); _expect(h0, (B, H), "initial_h")
    else:
        T = 1
        B, F = input.shape
        H = grad_h.shape[-1]
        _expect(grad_h, (B, H), "grad_h"); _expect(h0, (B, H), "old_h")
        _expect(z, (B, H), "z"); _expect(h_prime, (B, H), "h_prime")
    dt = input.dtype
    pdt = _param_dtype(dt)
    for t, want in ((grad_h, dt), (hs_or_old_h, dt if unrolled else pdt), (z, pdt), (h_prime, pdt), (h0, pdt)):
        if t.dtype != want:
            raise RuntimeError("fastgrnn backward: operand dtypes differ")
    # biases are not needed by the backward when z, h_prime are given: pass zeta as a dummy
    plan, params, w_lr, u_lr = _describe(T, B, F, H, w, u, w1, w2, u1, u2,
                                         bias_gate if preact else zeta, bias_update if preact else zeta,
                                         zeta, nu, dt, gate_nl, update_nl, flags)
    desc = plan[0]
    _warn_fallback(plan, 1)
    shapes = ([tuple(w1.shape), tuple(w2.shape)] if w_lr else [(H, F)]) + \
             ([tuple(u1.shape), tuple(u2.shape)] if u_lr else [(H, H)]) + [(1, H), (1, H), (1, 1), (1, 1)]
    sizes = [a * b for a, b in shapes]
    dx_is_gemm = (plan[2] == 2 and not w_lr and not u_lr and (desc.H == 256 or (desc.H == 128 and desc.F > 32)))
    ent = (w_lr, u_lr, shapes, sizes, dx_is_gemm, B, H)
    _seen[sig] = (plan, preact and rank_space is not None, ent)
    return _launch_backward(lib, plan, ent, unrolled, preact, grad_h, input, hs_or_old_h, z, h_prime,
                            rank_space if preact else None, h0, w, u, w1, w2, u1, u2,
                            bias_gate if preact else zeta, bias_update if preact else zeta, zeta, nu, need_dx)


def frame_gemm(x, w):
    """P[rows, H] = X[rows, F] . W^T (W:[H,F]): the batched frame product a wide-input layer's forward runs in front of
    its scan (include/fastgrnn_hip.h, fastgrnn_hip_frame_gemm), as a call of its own -- for measuring it."""
    _check_input(x, "x"); _check_input(w, "w")
    rows, F = x.shape
    H = w.shape[0]
    _expect(w, (H, F), "w")
    p = torch.empty((rows, H), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        with _Timed("frame_gemm", x.device):
            st = _lib.load().fastgrnn_hip_frame_gemm(rows, H, F, _ptr(x), _ptr(w), _ptr(p), _DTYPES[x.dtype], _stream(x.device))
    _lib.check(st, "fastgrnn frame_gemm")
    return p


# ---- the four reference entry points (fastgrnn_cuda.cpp:235-240) -------------------------

def forward(input, w, u, bias_gate, bias_update, zeta, nu, old_h, z_non_linearity, w1, w2, u1, u2,
            *, update_non_linearity=2, flags=0):
    """fastgrnn_cuda.cpp:73-107 -> [new_h, z, h_prime]."""
    return _forward_impl(input, w, u, bias_gate, bias_update, zeta, nu, old_h, z_non_linearity,
                         w1, w2, u1, u2, False, update_non_linearity, True, flags)


def backward(grad_h, input, old_h, zeta, nu, w, u, z, h_prime, w1, w2, u1, u2, z_non_linearity,
             *, update_non_linearity=2, flags=0):
    """fastgrnn_cuda.cpp:109-145 -> 12 tensors (.cu:317)."""
    return _backward_impl(grad_h, input, old_h, zeta, nu, w, u, z, h_prime, old_h, w1, w2, u1, u2,
                          z_non_linearity, False, update_non_linearity, flags)


def forward_unroll(input, w, u, bias_gate, bias_update, zeta, nu, initial_h, z_non_linearity,
                   w1, w2, u1, u2, *, update_non_linearity=2, want_gates=True, flags=0):
    """fastgrnn_cuda.cpp:147-180 -> [hidden_states, z_s, h_prime_s] (each [T,B,H]).
    ``want_gates=False`` (extension) returns ``[hidden_states]`` only and skips the two
    extra [T,B,H] stores -- forward-only / inference use."""
    return _forward_impl(input, w, u, bias_gate, bias_update, zeta, nu, initial_h, z_non_linearity,
                         w1, w2, u1, u2, True, update_non_linearity, want_gates, flags)


def backward_unroll(grad_h, input, hidden_states, zeta, nu, w, u, z, h_prime, initial_h, w1, w2, u1, u2,
                    z_non_linearity, *, update_non_linearity=2, flags=0, bias_gate=None, bias_update=None,
                    need_dx=True):
    """fastgrnn_cuda.cpp:182-232 -> 12 tensors (.cu:556).  With ``flags & FLAG_SAVE_PREACT``
    (extension, kernel path 2) ``z`` is the pre-activation tensor returned by
    ``forward_unroll(..., flags=FLAG_SAVE_PREACT)``, ``h_prime`` is ignored and the two bias
    tensors must be given."""
    return _backward_impl(grad_h, input, hidden_states, zeta, nu, w, u, z, h_prime, initial_h,
                          w1, w2, u1, u2, z_non_linearity, True, update_non_linearity, flags,
                          bias_gate=bias_gate, bias_update=bias_update, need_dx=need_dx)
