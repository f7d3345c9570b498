"""GPU tests: the low-rank scans (H=256, F=32, ranks <= 16; kernels_lowrank.hip) offer what the dense H=128 scans do
-- batch-major sequences, the data loader's [B,F,T] input, last-state-only gradients / outputs, bf16 sequences with
fp32 master gradients (SURVEY 8(f) N1/N2 and BASELINE's bf16 config, for the factorised cell of rnn.py:783-798).
The layout flags change WHERE a row lives, not the arithmetic: outputs equal the time-major run bit for bit.
"""
import numpy as np
import pytest
import torch

from oracle import fastgrnn_oracle as O

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from kws_amd import FastGRNNCUDA, fastgrnn_cuda
DEV = "cuda:0"
SAVE_PREACT, BATCH_MAJOR, X_BFT, GRAD_LAST, HS_LAST = 4, 16, 128, 256, 512
F, H = 32, 256


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _params(rw, ru, seed=4):
    p = O.make_params(F, H, rw, ru, dtype=np.float32, seed=seed, randomize_scalars=True)
    e = torch.empty(0)
    P = {k: e for k in ("w", "u", "w1", "w2", "u1", "u2")}
    P.update({k: _t(v) for k, v in p.items()})
    return p, P


def _run(P, x, G, h0, flags, gate=0):
    outs = fastgrnn_cuda.forward_unroll(x, P["w"], P["u"], P["bias_gate"], P["bias_update"], P["zeta"], P["nu"], h0, gate,
                                        P["w1"], P["w2"], P["u1"], P["u2"], flags=flags)
    gr = fastgrnn_cuda.backward_unroll(G, x, outs[0], P["zeta"], P["nu"], P["w"], P["u"], outs[1], outs[2], h0,
                                       P["w1"], P["w2"], P["u1"], P["u2"], gate, flags=flags,
                                       bias_gate=P["bias_gate"], bias_update=P["bias_update"])
    return list(outs), list(gr)


@pytest.mark.parametrize("B,rw,ru,bf16", [(37, 16, 16, False), (64, 16, 16, False), (48, 8, 12, False), (1, 16, 16, False),
                                          (37, 16, 16, True), (64, 16, 8, True)])
def test_batch_major_and_bft_layouts_equal_time_major(B, rw, ru, bf16):
    T = 23
    _, P = _params(rw, ru)
    dt = torch.bfloat16 if bf16 else torch.float32
    g = torch.Generator().manual_seed(9 + B)
    x = torch.randn(T, B, F, generator=g).to(dt).to(DEV)
    G = torch.randn(T, B, H, generator=g).to(dt).to(DEV)
    h0 = (0.3 * torch.randn(B, H, generator=g)).to(DEV)
    for fl in (BATCH_MAJOR, X_BFT, BATCH_MAJOR | X_BFT):
        for direction in (0, 1):
            assert fastgrnn_cuda.kernel_path(T, B, F, H, rw, ru, 0, dtype=dt, direction=direction, flags=SAVE_PREACT | fl) == 2
    o_t, g_t = _run(P, x, G, h0, SAVE_PREACT)
    tb = lambda a: a.transpose(0, 1).contiguous()
    for fl in (BATCH_MAJOR, X_BFT, BATCH_MAJOR | X_BFT):
        bm, bft = bool(fl & BATCH_MAJOR), bool(fl & X_BFT)
        xi = x.permute(1, 2, 0).contiguous() if bft else (tb(x) if bm else x)
        o, gr = _run(P, xi, tb(G) if bm else G, h0, SAVE_PREACT | fl)
        for a, b in zip(o_t[:2], o[:2]):                                   # hs, pre-activation
            assert torch.equal(a, b.transpose(0, 1) if bm else b), fl
        assert torch.equal(o_t[2], o[2])                                   # the rank-space vector is always time-major
        dx = gr[0].permute(2, 0, 1) if bft else (gr[0].transpose(0, 1) if bm else gr[0])
        assert gr[0].shape == xi.shape and torch.equal(g_t[0], dx), fl
        for k in (1, 2, 3, 4, 5, 8, 9, 10, 11):
            assert torch.equal(g_t[k], gr[k]), (fl, k)


@pytest.mark.parametrize("B,bf16,batch_major", [(64, False, False), (37, False, False), (48, True, False), (37, False, True)])
def test_grad_last_and_hs_last(B, bf16, batch_major):
    """GRAD_LAST: the [B,H] gradient of the last state equals the dense, zero-padded [T,B,H] gradient bit for bit.
    HS_LAST: the [B,H] output equals the last row of the full forward (a separately compiled kernel variant:
    fp32 rounding, one bf16 ulp for bf16 sequences)."""
    T, r = 19, 16
    _, P = _params(r, r, seed=8)
    dt = torch.bfloat16 if bf16 else torch.float32
    g = torch.Generator().manual_seed(31 + B)
    x = torch.randn(T, B, F, generator=g).to(dt).to(DEV)
    gl = torch.randn(B, H, generator=g).to(dt).to(DEV)
    h0 = (0.3 * torch.randn(B, H, generator=g)).to(DEV)
    base = SAVE_PREACT | (BATCH_MAJOR if batch_major else 0)
    if batch_major:
        x = x.transpose(0, 1).contiguous()
    G = torch.zeros((B, T, H) if batch_major else (T, B, H), dtype=dt, device=DEV)
    (G[:, -1] if batch_major else G[-1]).copy_(gl)
    assert fastgrnn_cuda.kernel_path(T, B, F, H, r, r, 0, dtype=dt, direction=1, flags=base | GRAD_LAST) == 2
    o, g_dense = _run(P, x, G, h0, base)
    gr = fastgrnn_cuda.backward_unroll(gl, x, o[0], P["zeta"], P["nu"], P["w"], P["u"], o[1], o[2], h0,
                                       P["w1"], P["w2"], P["u1"], P["u2"], 0, flags=base | GRAD_LAST,
                                       bias_gate=P["bias_gate"], bias_update=P["bias_update"])
    for k in (0, 1, 2, 3, 4, 5, 8, 9, 10, 11):
        assert torch.equal(g_dense[k], gr[k]), k
    fl = HS_LAST | (BATCH_MAJOR if batch_major else 0)
    assert fastgrnn_cuda.kernel_path(T, B, F, H, r, r, 0, dtype=dt, direction=0, flags=fl) == 2
    hl = fastgrnn_cuda.forward_unroll(x, P["w"], P["u"], P["bias_gate"], P["bias_update"], P["zeta"], P["nu"], h0, 0,
                                      P["w1"], P["w2"], P["u1"], P["u2"], want_gates=False, flags=fl)[0]
    last = o[0][:, -1] if batch_major else o[0][-1]
    assert hl.shape == (B, H) and hl.dtype == dt
    assert float((hl.float() - last.float()).abs().max()) <= (2.0 ** -7 if bf16 else 2e-6)
    with pytest.raises(RuntimeError):                  # HS_LAST saves nothing for a backward
        fastgrnn_cuda.forward_unroll(x, P["w"], P["u"], P["bias_gate"], P["bias_update"], P["zeta"], P["nu"], h0, 0,
                                     P["w1"], P["w2"], P["u1"], P["u2"], flags=fl | SAVE_PREACT)


@pytest.mark.parametrize("B,rw,ru", [(64, 16, 16), (37, 16, 16), (48, 8, 8)])
def test_bf16_sequences_fp32_master_grads_lowrank(B, rw, ru):
    """x, hs, grad_hs, d_x bf16 in HBM; state, factors, saved tensors and every parameter gradient fp32.  Against the
    fp64 oracle on the SAME rounded tensors (as the dense test in test_hip_parity.py)."""
    T = 31
    rng = np.random.default_rng(77 + B)
    p, P = _params(rw, ru, seed=23)
    bf = lambda a: torch.from_numpy(a).to(torch.bfloat16)
    x_bf = bf(rng.standard_normal((T, B, F)).astype(np.float32))
    G_bf = bf(rng.standard_normal((T, B, H)).astype(np.float32))
    h0 = (0.5 * rng.standard_normal((B, H))).astype(np.float32)
    assert fastgrnn_cuda.kernel_path(T, B, F, H, rw, ru, 0, dtype=torch.bfloat16, direction=1, flags=SAVE_PREACT) == 2
    o, gr = _run(P, x_bf.to(DEV), G_bf.to(DEV), _t(h0), SAVE_PREACT)
    hs, pre = o[0], o[1]
    assert hs.dtype == torch.bfloat16 and pre.dtype == torch.float32 and gr[0].dtype == torch.bfloat16
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    x64 = x_bf.to(torch.float64).numpy(); G64 = G_bf.to(torch.float64).numpy(); h64 = h0.astype(np.float64)
    hs_o, zs_o, cs_o = O.unroll_forward(x64, p64, h64)
    hs_k = hs.to(torch.float64).cpu().numpy()
    assert (np.abs(hs_k - hs_o) / np.maximum(1.0, np.abs(hs_o))).max() <= 2.0 ** -8 + 1e-5
    # backward: the kernel sees the ROUNDED hs as h_prev; oracle on the same tensors, gates from the saved pre-activation
    pre_k = pre.cpu().numpy().astype(np.float64)
    z_k = 1.0 / (1.0 + np.exp(-(pre_k + p64["bias_gate"]))); c_k = np.tanh(pre_k + p64["bias_update"])
    g_o = O.unroll_backward(G64, x64, hs_k, z_k, c_k, p64, h64, diagnostics=True)
    # d_u2 = sum_t d_pre_t^T (U1 h_{t-1}) is contracted with the rank-space vector the FORWARD saved, i.e. with the fp32
    # state before it was rounded for storage -- the exact value of the function that ran -- while the oracle above
    # sees the rounded hs everywhere.  Its bound is the distance between the oracle on rounded and on exact states.
    g_x = O.unroll_backward(G64, x64, hs_o, z_k, c_k, p64, h64)
    names = ["d_x", "d_bias_gate", "d_bias_update", "d_zeta", "d_nu", "d_h0", "d_w", "d_u", "d_w1", "d_w2", "d_u1", "d_u2"]
    g = {n: v for n, v in zip(names, gr) if v.numel()}
    dx = g.pop("d_x").to(torch.float64).cpu().numpy()
    ref = g_o.pop("d_x")
    assert (np.abs(dx - ref) / np.maximum(1.0, np.abs(ref))).max() <= 2.0 ** -8 + 2e-5
    for k, v in g_o.items():
        if k.startswith("_"):
            continue
        err = float(np.abs(g[k].cpu().numpy().reshape(v.shape) - v).max())
        lim = 2e-5 * max(1.0, float(np.abs(v).max()))
        if k in ("d_zeta", "d_nu"):
            lim = max(lim, 2e-7 * g_o["_abs_" + k[2:]])
        if k == "d_u2":
            lim += 1.5 * float(np.abs(g_x[k] - v).max())
        assert err <= lim, (k, err, lim)


@pytest.mark.parametrize("kind", ["batch_first", "bft_view", "bf16", "rank8"])
def test_module_lowrank_uses_the_layout_flags(kind):
    """FastGRNNCUDA with factorised weights: batch_first input, the trainer's permuted [B,F,T] view and bf16 frames all
    reach kernel path 2 in place; the results equal a plain time-major fp32/bf16 run of the same module."""
    T, B = 17, 40
    r = 8 if kind == "rank8" else 16
    torch.manual_seed(3)
    m = FastGRNNCUDA(F, H, wRank=r, uRank=r, batch_first=(kind == "batch_first"), device=DEV)
    ref = FastGRNNCUDA(F, H, wRank=r, uRank=r, device=DEV)
    ref.load_state_dict(m.state_dict())
    dt = torch.bfloat16 if kind == "bf16" else torch.float32
    x = torch.randn(T, B, F, device=DEV).to(dt)
    x_ref = x.clone().requires_grad_(True)
    out_ref = ref(x_ref)
    out_ref.float().square().mean().backward()
    if kind == "batch_first":
        xi = x.transpose(0, 1).contiguous().requires_grad_(True)
        out = m(xi).transpose(0, 1)
    elif kind == "bft_view":
        base = x.permute(1, 2, 0).contiguous().requires_grad_(True)       # the loader's [B,F,T]
        xi = base
        out = m(base.permute(2, 0, 1))
    else:
        xi = x.clone().requires_grad_(True)
        out = m(xi)
    out.float().square().mean().backward()
    assert torch.equal(out, out_ref)
    gx = xi.grad.transpose(0, 1) if kind == "batch_first" else (xi.grad.permute(2, 0, 1) if kind == "bft_view" else xi.grad)
    assert torch.equal(gx, x_ref.grad)
    for (n, a), (_, b) in zip(m.named_parameters(), ref.named_parameters()):
        assert torch.equal(a.grad, b.grad), n


@pytest.mark.parametrize("bm", [False, True], ids=["tm", "bm"])
@pytest.mark.parametrize("rw,ru,B,preact", [(32, 32, 37, True), (32, 32, 64, False), (16, None, 48, True), (None, 16, 33, True),
                                            (64, 8, 16, True), (17, 16, 21, False)])
def test_other_factorised_cells_run_on_the_dense_kernels(rw, ru, B, preact, bm):
    """Ranks above 16 and cells with only W or only U factorised (rnn.py:783-798): the factors are multiplied out per
    call, the dense H=256 scans run, the dense gradients are projected onto the factors (what the reference's CUDA
    operator does for every low-rank cell, .cu:353-362,546-555).  Against the fp64 oracle's FACTORISED evaluation."""
    T = 11
    rng = np.random.default_rng(5 + B)
    p = O.make_params(F, H, rw, ru, dtype=np.float32, seed=29, randomize_scalars=True)
    e = torch.empty(0)
    P = {k: e for k in ("w", "u", "w1", "w2", "u1", "u2")}
    P.update({k: _t(v) for k, v in p.items()})
    x = rng.standard_normal((T, B, F)).astype(np.float32)
    h0 = (0.5 * rng.standard_normal((B, H))).astype(np.float32)
    G = rng.standard_normal((T, B, H)).astype(np.float32)
    fl = (SAVE_PREACT if preact else 0) | (BATCH_MAJOR if bm else 0)     # bm: [B,T,.] sequences indexed in place
    for direction in (0, 1):
        assert fastgrnn_cuda.kernel_path(T, B, F, H, rw or 0, ru or 0, 0, direction=direction, flags=fl) == 2
    lay = (lambda a: np.ascontiguousarray(a.transpose(1, 0, 2))) if bm else (lambda a: a)
    outs = fastgrnn_cuda.forward_unroll(_t(lay(x)), P["w"], P["u"], P["bias_gate"], P["bias_update"], P["zeta"], P["nu"], _t(h0), 0,
                                        P["w1"], P["w2"], P["u1"], P["u2"], flags=fl)
    assert len(outs) == (2 if preact else 3)
    gr = fastgrnn_cuda.backward_unroll(_t(lay(G)), _t(lay(x)), outs[0], P["zeta"], P["nu"], P["w"], P["u"], outs[1], outs[-1], _t(h0),
                                       P["w1"], P["w2"], P["u1"], P["u2"], 0, flags=fl,
                                       bias_gate=P["bias_gate"], bias_update=P["bias_update"])
    if bm:
        outs = [o.transpose(0, 1) for o in outs]
        gr = [gr[0].transpose(0, 1)] + list(gr[1:])
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    hs_o, zs_o, cs_o = O.unroll_forward(x.astype(np.float64), p64, h0.astype(np.float64))
    assert (np.abs(outs[0].cpu().numpy() - hs_o) / np.maximum(1.0, np.abs(hs_o))).max() <= 1e-5
    g_o = O.unroll_backward(G.astype(np.float64), x.astype(np.float64), hs_o, zs_o, cs_o, p64, h0.astype(np.float64),
                            diagnostics=True)
    names = ["d_x", "d_bias_gate", "d_bias_update", "d_zeta", "d_nu", "d_h0", "d_w", "d_u", "d_w1", "d_w2", "d_u1", "d_u2"]
    g = {n: v.cpu().numpy() for n, v in zip(names, gr) if v.numel()}
    assert set(g) == {k for k in g_o if not k.startswith("_")}
    for k, v in g_o.items():
        if k.startswith("_"):
            continue
        err = float(np.abs(g[k].reshape(v.shape) - v).max())
        lim = 2e-5 * max(1.0, float(np.abs(v).max()))
        if k in ("d_zeta", "d_nu"):
            lim = max(lim, 2e-7 * g_o["_abs_" + k[2:]])
        assert err <= lim, (k, err, lim)


def test_module_with_rank_32_trains_on_the_matrix_pipe():
    torch.manual_seed(5)
    m = FastGRNNCUDA(F, H, wRank=32, uRank=32, device=DEV)
    x = torch.randn(13, 40, F, device=DEV, requires_grad=True)
    assert fastgrnn_cuda.kernel_path(13, 40, F, H, 32, 32, 0, direction=1, flags=SAVE_PREACT) == 2
    m(x).square().mean().backward()
    assert all(q.grad is not None and torch.isfinite(q.grad).all() for q in m.parameters()) and x.grad is not None


@pytest.mark.parametrize("Fi,Hi,rw,ru,B,flags", [(32, 128, 16, 16, 37, SAVE_PREACT), (32, 128, 8, None, 64, 0),
                                                 (256, 128, 16, 16, 21, SAVE_PREACT), (32, 128, 4, 24, 48, SAVE_PREACT | BATCH_MAJOR)])
def test_factorised_cells_on_the_dense_h128_kernels(Fi, Hi, rw, ru, B, flags):
    """Factorised cells with H = 128 (any ranks; the register-resident low-rank scans are H = 256 only): multiplied out
    per call onto the dense H = 128 kernels, with everything those offer (here: batch-major)."""
    T = 10
    rng = np.random.default_rng(7 + B)
    p = O.make_params(Fi, Hi, rw, ru, dtype=np.float32, seed=31, randomize_scalars=True)
    e = torch.empty(0)
    P = {k: e for k in ("w", "u", "w1", "w2", "u1", "u2")}
    P.update({k: _t(v) for k, v in p.items()})
    x = rng.standard_normal((T, B, Fi)).astype(np.float32)
    h0 = (0.5 * rng.standard_normal((B, Hi))).astype(np.float32)
    G = rng.standard_normal((T, B, Hi)).astype(np.float32)
    bm = bool(flags & BATCH_MAJOR)
    lay = (lambda a: np.ascontiguousarray(a.transpose(1, 0, 2))) if bm else (lambda a: a)
    unlay = (lambda t: t.transpose(0, 1)) if bm else (lambda t: t)
    for direction in (0, 1):
        assert fastgrnn_cuda.kernel_path(T, B, Fi, Hi, rw or 0, ru or 0, 0, direction=direction, flags=flags) == 2
    outs = fastgrnn_cuda.forward_unroll(_t(lay(x)), P["w"], P["u"], P["bias_gate"], P["bias_update"], P["zeta"], P["nu"], _t(h0), 0,
                                        P["w1"], P["w2"], P["u1"], P["u2"], flags=flags)
    gr = fastgrnn_cuda.backward_unroll(_t(lay(G)), _t(lay(x)), outs[0], P["zeta"], P["nu"], P["w"], P["u"], outs[1], outs[-1], _t(h0),
                                       P["w1"], P["w2"], P["u1"], P["u2"], 0, flags=flags,
                                       bias_gate=P["bias_gate"], bias_update=P["bias_update"])
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    hs_o, zs_o, cs_o = O.unroll_forward(x.astype(np.float64), p64, h0.astype(np.float64))
    assert (np.abs(unlay(outs[0]).cpu().numpy() - hs_o) / np.maximum(1.0, np.abs(hs_o))).max() <= 1e-5
    g_o = O.unroll_backward(G.astype(np.float64), x.astype(np.float64), hs_o, zs_o, cs_o, p64, h0.astype(np.float64),
                            diagnostics=True)
    names = ["d_x", "d_bias_gate", "d_bias_update", "d_zeta", "d_nu", "d_h0", "d_w", "d_u", "d_w1", "d_w2", "d_u1", "d_u2"]
    g = {n: v for n, v in zip(names, gr) if v.numel()}
    g["d_x"] = unlay(g["d_x"])
    for k, v in g_o.items():
        if k.startswith("_"):
            continue
        err = float(np.abs(g[k].cpu().numpy().reshape(v.shape) - v).max())
        lim = 2e-5 * max(1.0, float(np.abs(v).max()))
        if k in ("d_zeta", "d_nu"):
            lim = max(lim, 2e-7 * g_o["_abs_" + k[2:]])
        assert err <= lim, (k, err, lim)


@pytest.mark.parametrize("B,rw,ru,bf16", [(4096, 16, 16, False), (4096 + 5, 16, 16, False), (2048, 7, 13, False), (4096, 16, 16, True)])
def test_factor_gradients_contracted_inside_the_scan(B, rw, ru, bf16):
    """The factor gradients (d_w1, d_w2, d_u1, d_u2: .cu:546-555) are sums over the T*B rows that the backward scan
    contracts itself, per workgroup of 16 utterances, into slabs that one fixed-order reduction adds up (round 3; they
    were three GEMMs over a d_pre[T,B,H] round trip).  At the BASELINE size, where the oracle comparison lives in
    tests/test_hip_fullsize.py, the size-independent properties: the same bits from one launch to the next, the
    data-parallel identity (the gradient of the batch is the sum of its shards' gradients) and linearity in grad_hs."""
    T = 99
    _, P = _params(rw, ru, seed=21)
    dt = torch.bfloat16 if bf16 else torch.float32
    g = torch.Generator().manual_seed(3 + B)
    x = torch.randn(T, B, F, generator=g).to(dt).to(DEV)
    G = torch.randn(T, B, H, generator=g).to(dt).to(DEV)
    h0 = (0.3 * torch.randn(B, H, generator=g)).to(DEV)
    outs, g1 = _run(P, x, G, h0, SAVE_PREACT)
    _, g2 = _run(P, x, G, h0, SAVE_PREACT)
    for k, (a, b) in enumerate(zip(g1, g2)):
        assert torch.equal(a, b), k                                   # no atomics, fixed reduction order
    # shards: utterances [0, cut) and [cut, B), cut not a multiple of 16
    cut = B // 2 + 3
    parts = []
    for lo, hi in ((0, cut), (cut, B)):
        _, gp = _run(P, x[:, lo:hi].contiguous(), G[:, lo:hi].contiguous(), h0[lo:hi].contiguous(), SAVE_PREACT)
        parts.append(gp)
    tol = 2.0 ** -6 if bf16 else 2e-5
    for k in (1, 2, 3, 4, 8, 9, 10, 11):                              # d_bias_z, d_bias_h, d_zeta, d_nu, d_w1, d_w2, d_u1, d_u2
        whole, summed = g1[k].float(), parts[0][k].float() + parts[1][k].float()
        assert float((whole - summed).abs().max()) <= tol * max(1.0, float(whole.abs().max())), k
    for k in (0, 5):                                                  # d_input, d_old_h: per utterance, bit for bit
        cat = torch.cat([parts[0][k], parts[1][k]], dim=1 if k == 0 else 0)
        assert torch.equal(g1[k], cat), k
    if not bf16:
        gr = fastgrnn_cuda.backward_unroll(-0.5 * G, x, outs[0], P["zeta"], P["nu"], P["w"], P["u"], outs[1], outs[2], h0,
                                           P["w1"], P["w2"], P["u1"], P["u2"], 0, flags=SAVE_PREACT,
                                           bias_gate=P["bias_gate"], bias_update=P["bias_update"])
        for k in (8, 9, 10, 11):
            assert float((gr[k] + 0.5 * g1[k]).abs().max()) <= 2e-6 * max(1.0, float(g1[k].abs().max())), k


def test_lowrank_backward_wants_both_saved_tensors():
    """z_s = NULL (round 2's recomputing variant) is an error again: include/fastgrnn_hip.h at fastgrnn_hip_kernel_path."""
    _, P = _params(16, 16, seed=2)
    x = torch.randn(5, 16, F, device=DEV); G = torch.randn(5, 16, H, device=DEV); h0 = torch.zeros(16, H, device=DEV)
    outs, _ = _run(P, x, G, h0, SAVE_PREACT)
    with pytest.raises(RuntimeError):
        fastgrnn_cuda.backward_unroll(G, x, outs[0], P["zeta"], P["nu"], P["w"], P["u"], torch.empty(0, device=DEV), outs[2], h0,
                                      P["w1"], P["w2"], P["u1"], P["u2"], 0, flags=SAVE_PREACT,
                                      bias_gate=P["bias_gate"], bias_update=P["bias_update"])
