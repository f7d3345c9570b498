"""GPU tests for the round-2 robustness items:
  * the fp16 two-plane forward state product is only used where h provably stays inside fp16's range
    (gate with z in [0,1] AND a bounded h0, checked on the device) -- relu gate past 1e5, user h0 of 1e5;
  * bf16 stores keep a NaN a NaN (hardware converter instead of integer rounding);
  * the fused head follows nn.NLLLoss (ignore_index = -100; other out-of-range labels are loud);
  * SURVEY 8(f) N4 on the device: hardThreshold / supportBasedThreshold / get_model_size and the
    sparsify -> train step -> sparsifyWithSupport cycle through the HIP operator (trainClassifier.py:251-260).
Reference semantics: rnn.py:290-295 (cell), utils.py:53-81,103-114, rnn.py:843-889.
"""
import numpy as np
import pytest
import torch

from oracle import fastgrnn_oracle as O

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from kws_amd import FastGRNNCUDA, _lib, fastgrnn_cuda, utils as U
    from kws_amd.head import head_xent
DEV = "cuda:0"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _fwd(x, h0, p, gate, flags=0):
    e = torch.empty(0)
    return fastgrnn_cuda.forward_unroll(_t(x), _t(p["w"]), _t(p["u"]), _t(p["bias_gate"]), _t(p["bias_update"]),
                                        _t(p["zeta"]), _t(p["nu"]), _t(h0), gate, e, e, e, e, want_gates=False,
                                        flags=flags)[0].cpu().numpy()


def test_relu_gate_state_beyond_fp16_range_matches_oracle():
    """A relu gate does not bound h (z = relu(.) can exceed 1): here z ~ 2, so |h| doubles every frame and passes
    1e5 within the sequence.  The fp16 two-plane product would overflow to inf; the dispatcher must give this gate
    the three-bf16-plane product; accuracy is judged against what an fp32 evaluation of the same formula reaches."""
    T, B, F, H = 18, 48, 32, 128
    rng = np.random.default_rng(7)
    p = O.make_params(F, H, dtype=np.float32, seed=3, randomize_scalars=True)
    p["u"] = (1e-8 * rng.standard_normal((H, H))).astype(np.float32)      # keeps U.h << 1 while |h| reaches ~1e6
    p["w"] = (0.02 * rng.standard_normal((H, F))).astype(np.float32)
    p["bias_gate"] = (2.0 + 0.05 * rng.standard_normal((1, H))).astype(np.float32)
    x = rng.standard_normal((T, B, F)).astype(np.float32)
    h0 = rng.standard_normal((B, H)).astype(np.float32)
    assert fastgrnn_cuda.kernel_path(T, B, F, H, gate_nl=1, direction=0) == 2
    hs = _fwd(x, h0, p, 1)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    hs_o, _, _ = O.unroll_forward(x.astype(np.float64), p64, h0.astype(np.float64), gate="relu")
    assert np.abs(hs_o).max() > 1e5 and np.isfinite(hs).all()
    # Error relative to the largest state of the same utterance and frame (single elements pass through zero while
    # their neighbours are at 1e6: what a consumer of h_t -- a dot product -- sees is the error against the row's
    # scale), and judged beside the oracle itself run in fp32 (numpy), since a gate above 1 is not contractive.
    hs_32, _, _ = O.unroll_forward(x, p, h0, gate="relu")
    rel = lambda a: float((np.abs(a - hs_o) / np.maximum(1.0, np.abs(hs_o).max(axis=2, keepdims=True))).max())
    assert rel(hs) <= max(1e-5, 3.0 * rel(hs_32)), (rel(hs), rel(hs_32))


@pytest.mark.parametrize("scale,B", [(1e5, 48), (1e5, 37), (2e4, 48), (6e4, 16)])
def test_sigmoid_gate_with_a_large_user_h0_matches_oracle(scale, B):
    """h0 ~ 1e5 with the (default) sigmoid gate: outside fp16's range, so the workgroup's device-side check must
    route the state product to the three-bf16-plane form; 2e4 stays on the fp16 path (|h| <= |h0| + T + 1 < 3e4)
    and must be just as accurate.  U is small so that U.h stays O(1) (with the reference's 0.1-scale U a state of
    1e5 saturates every gate and cancellation makes ANY fp32 evaluation differ from fp64)."""
    T, F, H = 40, 32, 128
    rng = np.random.default_rng(11)
    p = O.make_params(F, H, dtype=np.float32, seed=5, randomize_scalars=True)
    p["u"] = ((1.0 / scale) * 0.1 * rng.standard_normal((H, H))).astype(np.float32)
    x = rng.standard_normal((T, B, F)).astype(np.float32)
    h0 = (scale * rng.uniform(0.5, 1.0, (B, H)) * rng.choice([-1.0, 1.0], (B, H))).astype(np.float32)
    hs = _fwd(x, h0, p, 0)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    hs_o, _, _ = O.unroll_forward(x.astype(np.float64), p64, h0.astype(np.float64))
    assert np.isfinite(hs).all()
    assert (np.abs(hs - hs_o) / np.maximum(1.0, np.abs(hs_o))).max() <= 1e-5
    # a mixed batch: one workgroup tile with a large h0, the others zero -- the choice is per workgroup
    if B == 48:
        h0m = h0.copy(); h0m[16:] = 0.0
        hs_m = _fwd(x, h0m, p, 0)
        hs_om, _, _ = O.unroll_forward(x.astype(np.float64), p64, h0m.astype(np.float64))
        assert (np.abs(hs_m - hs_om) / np.maximum(1.0, np.abs(hs_om))).max() <= 1e-5


def test_nan_in_h0_stays_nan_in_bf16_outputs():
    """A diverged run must stay visible: NaN state -> NaN in the bf16 hs (integer rounding used to turn some NaN
    payloads into 0 or inf).  Only the poisoned utterance is affected."""
    T, B, F, H = 5, 16, 32, 128
    p = O.make_params(F, H, dtype=np.float32, seed=1)
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.standard_normal((T, B, F)).astype(np.float32)).to(torch.bfloat16).to(DEV)
    h0 = np.zeros((B, H), np.float32)
    h0[3, :] = np.frombuffer(np.uint32(0x7F800001).tobytes(), np.float32)[0]     # a NaN the integer form turned into inf
    e = torch.empty(0)
    hs = fastgrnn_cuda.forward_unroll(x, _t(p["w"]), _t(p["u"]), _t(p["bias_gate"]), _t(p["bias_update"]), _t(p["zeta"]),
                                      _t(p["nu"]), _t(h0), 0, e, e, e, e, want_gates=False)[0].float().cpu().numpy()
    assert np.isnan(hs[:, 3, :]).all()
    assert np.isfinite(np.delete(hs, 3, axis=1)).all()


def test_head_follows_nllloss_ignore_index_and_flags_bad_labels():
    B, H, C = 70, 128, 12
    g = torch.Generator().manual_seed(3)
    h = torch.randn(B, H, generator=g)
    W = 0.2 * torch.randn(C, H, generator=g)
    b = 0.1 * torch.randn(C, generator=g)
    y = torch.randint(0, C, (B,), generator=g)
    y[[2, 17, 69]] = -100
    # torch float64 on the CPU: the modules the reference chains (model.py:226-230, trainClassifier.py:154,236)
    h64 = h.double().requires_grad_(True); W64 = W.double().requires_grad_(True); b64 = b.double().requires_grad_(True)
    loss_ref = torch.nn.NLLLoss()(torch.log_softmax(h64 @ W64.T + b64, dim=1), y)
    loss_ref.backward()
    loss, logp, d_h, d_w, d_b = head_xent(h.to(DEV), W.to(DEV), b.to(DEV), y.to(DEV), want_log_probs=True)
    assert abs(float(loss) - float(loss_ref)) <= 1e-5
    assert float((d_h.cpu().double() - h64.grad).abs().max()) <= 1e-6
    assert float((d_w.cpu().double() - W64.grad).abs().max()) <= 1e-5
    assert float((d_b.cpu().double() - b64.grad).abs().max()) <= 1e-5
    assert float(d_h[[2, 17, 69]].abs().max()) == 0.0
    # an out-of-range label that is not ignore_index: torch raises; the fused head makes the loss NaN
    y2 = y.clone(); y2[5] = C + 3
    loss2 = head_xent(h.to(DEV), W.to(DEV), b.to(DEV), y2.to(DEV))[0]
    assert torch.isnan(loss2).all()


# ---- SURVEY 8(f) N4 on the device ------------------------------------------------------------------------------
def _np_hard_threshold(a, s):
    """utils.py:57-63 restated with numpy."""
    a = a.copy().ravel()
    if len(a):
        th = np.percentile(np.abs(a), (1 - s) * 100.0, method="higher")
        a[np.abs(a) < th] = 0.0
    return a


def test_hard_threshold_support_and_count_on_device_match_numpy():
    rng = np.random.default_rng(0)
    for shape in ((256, 256), (128, 256), (128, 32), (16, 256), (7, 3), (1, 1)):
        for s in (1.0, 0.9, 0.8, 0.5, 0.31, 0.013, 0.0):
            a = rng.standard_normal(shape).astype(np.float32)
            if a.size > 4:
                a.ravel()[:3] = a.ravel()[3]              # ties with each other
            t = _t(a.copy())
            out = U.hardThreshold(t, s)
            assert out is t and t.is_cuda
            ref = _np_hard_threshold(a, s)
            np.testing.assert_array_equal(t.cpu().numpy().ravel(), ref, err_msg="%s s=%g" % (shape, s))
            assert U.countNNZ(t, True) == int(np.count_nonzero(ref)) and U.countNNZ(t, False) == a.size
            fresh = _t(a.copy())
            U.supportBasedThreshold(fresh, t)
            np.testing.assert_array_equal(fresh.cpu().numpy().ravel(), np.where(ref != 0, a.ravel(), 0.0))


@pytest.mark.parametrize("lowrank", [False, True])
def test_sparsify_train_step_sparsify_with_support_through_the_operator(lowrank):
    """The trainer's IHT phases (trainClassifier.py:251-260): sparsify() once, then after every optimizer step
    sparsifyWithSupport().  All of it stays on the GPU here (the reference moves the module to the CPU and back,
    model.py:91-107); the training step in between runs the HIP operator."""
    T, B, F = 20, 64, 32
    H, r = (256, 16) if lowrank else (128, None)
    torch.manual_seed(0)
    m = FastGRNNCUDA(F, H, wRank=r, uRank=r, wSparsity=0.8, uSparsity=0.3, device=DEV)
    mats = m.getVars()[:m._num_W_matrices + m._num_U_matrices]
    before = [w.detach().cpu().numpy().copy() for w in mats]
    dense_size = m.get_model_size()
    m.sparsify()
    spars = [0.8] * m._num_W_matrices + [0.3] * m._num_U_matrices
    for w, a, s in zip(mats, before, spars):
        assert w.is_cuda
        np.testing.assert_array_equal(w.detach().cpu().numpy().ravel(), _np_hard_threshold(a, s))
    nnz = [int((w != 0).sum()) for w in mats]
    assert m.get_model_size() == dense_size - 4 * sum(w.numel() - n for w, n in zip(mats, nnz))
    opt = torch.optim.SGD(m.parameters(), lr=0.05)
    x = torch.randn(T, B, F, device=DEV)
    for _ in range(2):
        opt.zero_grad()
        m(x).square().mean().backward()
        opt.step()                                          # fills the zeros again ...
        assert any(int((w != 0).sum()) > n for w, n in zip(mats, nnz))
        m.sparsifyWithSupport()                             # ... and the remembered support removes them
        for w, old in zip(mats, m.oldmats):
            assert torch.equal(w == 0, (old == 0) | (w == 0)) and int(((old == 0) & (w != 0)).sum()) == 0


@pytest.mark.parametrize("H,F,r", [(128, 32, None), (256, 32, None), (256, 32, 16), (128, 256, None)])
def test_weights_whose_blocks_sit_in_different_binades(H, F, r):
    """Several kernels scale a wave's block of a weight matrix by its own power of two (fp16 two-plane operands).  With
    0.1 * randn matrices every block's maximum falls into the same binade and a mixed-up factor goes unnoticed (the
    H=256 backward shipped one in round 2); here row AND column blocks of U (and W) differ by up to 2^5."""
    T, B = 9, 37
    rng = np.random.default_rng(13)
    p = O.make_params(F, H, r, r, dtype=np.float32, seed=47, randomize_scalars=True)
    nb = H // 16
    rowf = np.repeat(np.array([(0.07, 1.9, 0.45, 1.0, 0.12, 2.6, 0.8, 0.3)[k % 8] for k in range(nb)], np.float32), 16)
    colf = np.repeat(np.array([(1.3, 0.09, 0.6, 2.1, 0.25, 1.0, 0.5, 0.15)[k % 8] for k in range(nb)], np.float32), 16)
    if r:
        p["u2"] = (p["u2"] * rowf[:, None]).astype(np.float32); p["u1"] = (p["u1"] * colf[None, :]).astype(np.float32)
        p["w2"] = (p["w2"] * rowf[::-1, None]).astype(np.float32)
    else:
        p["u"] = (0.6 * p["u"] * rowf[:, None] * colf[None, :]).astype(np.float32)
        p["w"] = (p["w"] * rowf[::-1, None]).astype(np.float32)
    x = rng.standard_normal((T, B, F)).astype(np.float32)
    h0 = (0.5 * rng.standard_normal((B, H))).astype(np.float32)
    G = rng.standard_normal((T, B, H)).astype(np.float32)
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
    assert (np.abs(outs[0].cpu().numpy() - hs_o) / np.maximum(1.0, np.abs(hs_o))).max() <= 1e-5
    g_o = O.unroll_backward(G.astype(np.float64), x.astype(np.float64), hs_o, zs_o, cs_o, p64, h0.astype(np.float64),
                            diagnostics=True)
    names = ["d_x", "d_bias_gate", "d_bias_update", "d_zeta", "d_nu", "d_h0", "d_w", "d_u", "d_w1", "d_w2", "d_u1", "d_u2"]
    g = {n: v.cpu().numpy() for n, v in zip(names, gr) if v.numel()}
    for k, v in g_o.items():
        if k.startswith("_"):
            continue
        err = float(np.abs(g[k].reshape(v.shape) - v).max())
        lim = 2e-5 * max(1.0, float(np.abs(v).max()))
        if k in ("d_zeta", "d_nu"):
            lim = max(lim, 2e-7 * g_o["_abs_" + k[2:]])
        assert err <= lim, (k, err, lim)


def test_fallback_to_the_generic_scan_warns_once():
    """A shape without matrix-pipe kernels (H = 100) at a size where it matters: one RuntimeWarning per shape and
    direction, none for shapes that are covered or too small to matter."""
    import warnings
    e = torch.empty(0)
    def fwd(T, B, F, H):
        p = O.make_params(F, H, dtype=np.float32, seed=1)
        x = torch.zeros(T, B, F, device=DEV); h0 = torch.zeros(B, H, device=DEV)
        return fastgrnn_cuda.forward_unroll(x, _t(p["w"]), _t(p["u"]), _t(p["bias_gate"]), _t(p["bias_update"]), _t(p["zeta"]),
                                            _t(p["nu"]), h0, 0, e, e, e, e, want_gates=False)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        fwd(64, 128, 20, 100); fwd(64, 128, 20, 100)            # path 0, T*B = 8192
        fwd(64, 128, 32, 128)                                   # path 2
        fwd(4, 8, 20, 100)                                      # path 0 but tiny
    msgs = [str(m.message) for m in w if issubclass(m.category, RuntimeWarning) and "generic scan" in str(m.message)]
    assert len(msgs) == 1 and "H=100" in msgs[0]
