"""GPU tests on input distributions the other tests do not visit.  Every parity test draws 0.1 * randn weights and
N(0,1) data; one bug of round 2 (a per-block scale applied by the wrong wave) was invisible on exactly that
distribution.  Here each kernel family sees: large, tiny and sparsified weights (whole zero blocks), saturated gates, extreme zeta / nu, large frames,
gradients of 1e-9 and 1e+5 (the H=256 backward rescales d_pre per slice), all-zero frames and all-zero gradients.
Checked against the fp64 oracle RELATIVE to each output's own largest element (no absolute floor: a gradient of size
1e-9 must be right to 2e-5 of 1e-9)."""
import zlib

import numpy as np
import pytest
import torch

from oracle import fastgrnn_oracle as O

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from kws_amd import _lib, fastgrnn_cuda
DEV = "cuda:0"

FAMILIES = {"dense128": (32, 128, None), "wide256": (256, 128, None), "h256": (32, 256, None), "lowrank16": (32, 256, 16),
            "rank32": (32, 256, 32), "lowrank_h128": (32, 128, 16)}
SCENARIOS = ["big_weights", "tiny_weights", "saturated", "zeta_nu_extreme", "big_frames", "tiny_grad", "huge_grad",
             "zero_frames", "zero_grad", "two_steps", "sparse_weights"]


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("scenario", SCENARIOS)
@pytest.mark.parametrize("family", list(FAMILIES))
def test_unusual_inputs_against_the_oracle(family, scenario):
    F, H, r = FAMILIES[family]
    T, B = (2 if scenario == "two_steps" else 12), 21
    rng = np.random.default_rng(zlib.crc32(("%s/%s" % (family, scenario)).encode()))     # (hash() of a str differs per process)
    p = O.make_params(F, H, r, r, dtype=np.float32, seed=53, randomize_scalars=True)
    mats = [k for k in ("w", "u", "w1", "w2", "u1", "u2") if k in p]
    x = rng.standard_normal((T, B, F)).astype(np.float32)
    h0 = (0.5 * rng.standard_normal((B, H))).astype(np.float32)
    G = rng.standard_normal((T, B, H)).astype(np.float32)
    if scenario == "big_weights":
        for k in mats:
            p[k] = (p[k] * (3.0 if r else 8.0)).astype(np.float32)
    elif scenario == "tiny_weights":
        for k in mats:
            p[k] = (p[k] * 1e-3).astype(np.float32)
    elif scenario == "sparse_weights":
        # what the trainer's IHT phase leaves (utils.py:53-63): most entries zero, and whole blocks of rows / columns
        for k in mats:
            m = p[k].copy()
            m[rng.random(m.shape) < 0.9] = 0.0
            if m.shape[0] >= 64:
                m[32:64, :] = 0.0
            if m.shape[1] >= 64:
                m[:, 0:32] = 0.0
            p[k] = m.astype(np.float32)
    elif scenario == "saturated":
        p["bias_gate"] = (p["bias_gate"] + 12.0 * np.sign(rng.standard_normal((1, H)))).astype(np.float32)
        p["bias_update"] = (p["bias_update"] * 6.0).astype(np.float32)
    elif scenario == "zeta_nu_extreme":
        p["zeta"] = np.asarray([[9.0]], np.float32); p["nu"] = np.asarray([[-11.0]], np.float32)
    elif scenario == "big_frames":
        x = (x * 50.0).astype(np.float32)
    elif scenario == "tiny_grad":
        G = (G * 1e-9).astype(np.float32)
    elif scenario == "huge_grad":
        G = (G * 1e5).astype(np.float32)
    elif scenario == "zero_frames":
        x[:] = 0.0
    elif scenario == "zero_grad":
        G[:] = 0.0
    e = torch.empty(0)
    P = {k: e for k in ("w", "u", "w1", "w2", "u1", "u2")}
    P.update({k: _t(v) for k, v in p.items()})
    fl = _lib.FLAG_SAVE_PREACT
    assert fastgrnn_cuda.kernel_path(T, B, F, H, r or 0, r or 0, 0, direction=1, flags=fl) == 2
    outs = fastgrnn_cuda.forward_unroll(_t(x), P["w"], P["u"], P["bias_gate"], P["bias_update"], P["zeta"], P["nu"], _t(h0), 0,
                                        P["w1"], P["w2"], P["u1"], P["u2"], flags=fl)
    gr = fastgrnn_cuda.backward_unroll(_t(G), _t(x), outs[0], P["zeta"], P["nu"], P["w"], P["u"], outs[1], outs[-1], _t(h0),
                                       P["w1"], P["w2"], P["u1"], P["u2"], 0, flags=fl,
                                       bias_gate=P["bias_gate"], bias_update=P["bias_update"])
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    hs_o, zs_o, cs_o = O.unroll_forward(x.astype(np.float64), p64, h0.astype(np.float64))
    hs = outs[0].cpu().numpy()
    assert np.isfinite(hs).all()
    # weights 8x the reference's scale make the recurrence expansive: ANY fp32 evaluation drifts from fp64 there, so
    # that scenario is judged beside the oracle itself run in fp32 (numpy), as the relu-gate test does
    ill = scenario == "big_weights"
    hs_32, zs_32, cs_32 = O.unroll_forward(x, p, h0) if ill else (None, None, None)
    rel = lambda a: float((np.abs(a - hs_o) / np.maximum(1.0, np.abs(hs_o))).max())
    assert rel(hs) <= (max(1e-5, 3.0 * rel(hs_32)) if ill else 1e-5), (rel(hs), rel(hs_32) if ill else None)
    g_o = O.unroll_backward(G.astype(np.float64), x.astype(np.float64), hs_o, zs_o, cs_o, p64, h0.astype(np.float64),
                            diagnostics=True)
    g_32 = O.unroll_backward(G, x, hs_32, zs_32, cs_32, p, h0) if ill else None
    names = ["d_x", "d_bias_gate", "d_bias_update", "d_zeta", "d_nu", "d_h0", "d_w", "d_u", "d_w1", "d_w2", "d_u1", "d_u2"]
    g = {n: v.cpu().numpy() for n, v in zip(names, gr) if v.numel()}
    gscale = max(float(np.abs(G).max()), 0.0)
    for k, v in g_o.items():
        if k.startswith("_"):
            continue
        got = g[k].reshape(v.shape)
        assert np.isfinite(got).all(), k
        ref_max = float(np.abs(v).max())
        err = float(np.abs(got - v).max())
        # relative to the output's own size; gradients that cancel to (near) nothing are judged against what one
        # fp32 rounding of their terms amounts to: the terms' magnitude sum for zeta / nu, the gradient scale otherwise
        lim = (6e-5 if ill else 3e-5) * ref_max
        if k in ("d_zeta", "d_nu"):
            lim = max(lim, 2e-7 * g_o["_abs_" + k[2:]])
        lim = max(lim, 1e-6 * gscale)
        if ill:
            # (three times what the same formulas evaluated in fp32 are off by.  The dense H=256 backward gets six: its
            # chain carries d_pre exactly since round 3, but U^T as two fp16 planes, 22 bits -- a FIXED relative
            # perturbation of the weights of 2^-23 that an expansive recurrence amplifies step after step the same way,
            # where independent roundings average out: d_nu 4.5e-4 here against 8e-5, DESIGN.md 4.1c.  It was twelve.)
            fac = 6.0 if H == 256 and r != 16 else 3.0
            lim = max(lim, fac * float(np.abs(g_32[k].reshape(v.shape) - v).max()))
        if scenario == "zero_grad":
            assert err == 0.0, (k, err)
        else:
            assert err <= lim, (family, scenario, k, err, lim, ref_max)
