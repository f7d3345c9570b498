"""Pin the oracle (oracle/fastgrnn_oracle.py) against the reference's own CPU cell.

The fixtures in tests/golden/*.npz were produced by importing /root/reference's
rnn.FastGRNNCell (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import fastgrnn_oracle as O


def _tol(dtype):
    # fp64: the oracle re-associates nothing in the forward; backward is hand-derived
    # algebra vs autograd -> rounding-level agreement.  fp32: summation order differs
    # (numpy vs ATen) over K = F + H terms (160 for the north-star cell, up to 384 for the stack's layers: g10,
    # g11) and 99 steps -- two fp32 evaluations of the same formula, each ~2e-6 from the fp64 value.
    return (1e-12, 1e-10) if dtype == "f64" else (6e-6, 2e-4)


def test_oracle_forward_matches_reference(golden):
    fw_tol, _ = _tol(golden["dtype"])
    hs, zs, cs = O.unroll_forward(golden["x"], golden["params"], golden["h0"],
                                  gate=golden["gate"], update=golden["update"])
    assert hs.dtype == golden["hs"].dtype
    err = np.abs(hs - golden["hs"]).max()
    assert err <= fw_tol, (golden["name"], err)


def test_oracle_backward_matches_reference(golden):
    _, bw_tol = _tol(golden["dtype"])
    p = golden["params"]
    hs, zs, cs = O.unroll_forward(golden["x"], p, golden["h0"], gate=golden["gate"], update=golden["update"])
    g = O.unroll_backward(golden["G"], golden["x"], hs, zs, cs, p, golden["h0"],
                          gate=golden["gate"], update=golden["update"])
    ref = dict(golden["dparams"])
    ref["d_x"] = golden["dx"]
    ref["d_h0"] = golden["dh0"]
    for k, v in ref.items():
        scale = max(1.0, np.abs(v).max())
        err = np.abs(g[k].reshape(v.shape) - v).max() / scale
        assert err <= bw_tol, (golden["name"], k, err)


def test_cpu_layout_cell_matches_boundary_layout():
    """rnn.py:273-297 restated in CPU layout == boundary-layout unroll step."""
    rng = np.random.default_rng(0)
    F, H, B = 6, 10, 4
    p = O.make_params(F, H, dtype=np.float64, seed=3, randomize_scalars=True)
    x = rng.standard_normal((1, B, F))
    h0 = rng.standard_normal((B, H))
    hs, zs, cs = O.unroll_forward(x, p, h0)
    new_h, z, c = O.cell_forward_cpu_layout(x[0], h0, W=p["w"].T, U=p["u"].T,
                                            bias_gate=p["bias_gate"], bias_update=p["bias_update"],
                                            zeta=p["zeta"], nu=p["nu"])
    np.testing.assert_allclose(hs[0], new_h, rtol=0, atol=1e-14)


@pytest.mark.parametrize("gate", ["sigmoid", "tanh", "relu"])
@pytest.mark.parametrize("lowrank", [False, True])
def test_oracle_backward_finite_difference(gate, lowrank):
    """fp64 central differences of L = sum(hs*G) against unroll_backward (covers relu,
    which the reference cannot run on CPU: parity unpinned there)."""
    rng = np.random.default_rng(5)
    T, B, F, H = 4, 3, 5, 6
    p = O.make_params(F, H, w_rank=2 if lowrank else None, u_rank=3 if lowrank else None,
                      dtype=np.float64, seed=7, randomize_scalars=True)
    x = rng.standard_normal((T, B, F))
    h0 = 0.3 * rng.standard_normal((B, H))
    G = rng.standard_normal((T, B, H))

    def loss(pp, xx, hh):
        return float((O.unroll_forward(xx, pp, hh, gate=gate)[0] * G).sum())

    hs, zs, cs = O.unroll_forward(x, p, h0, gate=gate)
    g = O.unroll_backward(G, x, hs, zs, cs, p, h0, gate=gate)
    eps = 1e-6
    names = [k for k in p]
    for k in names:
        flat = p[k].reshape(-1)
        idxs = rng.choice(flat.size, size=min(5, flat.size), replace=False)
        for i in idxs:
            old = flat[i]
            flat[i] = old + eps; lp = loss(p, x, h0)
            flat[i] = old - eps; lm = loss(p, x, h0)
            flat[i] = old
            fd = (lp - lm) / (2 * eps)
            an = g["d_" + k].reshape(-1)[i]
            assert abs(fd - an) <= 1e-6 * max(1.0, abs(fd)), (k, i, fd, an)
    for (arr, key) in ((x, "d_x"), (h0, "d_h0")):
        flat = arr.reshape(-1)
        for i in rng.choice(flat.size, size=5, replace=False):
            old = flat[i]
            flat[i] = old + eps; lp = loss(p, x, h0)
            flat[i] = old - eps; lm = loss(p, x, h0)
            flat[i] = old
            fd = (lp - lm) / (2 * eps)
            an = g[key].reshape(-1)[i]
            assert abs(fd - an) <= 1e-6 * max(1.0, abs(fd)), (key, i, fd, an)


@pytest.mark.parametrize("B,H,C", [(7, 16, 3), (64, 128, 12), (33, 40, 35)])
def test_head_oracle_matches_torch_modules(B, H, C):
    """The numpy head (Linear + log_softmax + NLLLoss and gradients) against the torch modules the reference chains
    (model.py:86-88, 226-230; trainClassifier.py:154,236), float64, autograd gradients."""
    import torch
    rng = np.random.default_rng(B + H + C)
    h = rng.standard_normal((B, H)); W = 0.3 * rng.standard_normal((C, H)); b = 0.1 * rng.standard_normal(C)
    y = rng.integers(0, C, B)
    lin = torch.nn.Linear(H, C).double()
    with torch.no_grad():
        lin.weight.copy_(torch.from_numpy(W)); lin.bias.copy_(torch.from_numpy(b))
    ht = torch.from_numpy(h).requires_grad_(True)
    scores = torch.nn.functional.log_softmax(lin(ht), dim=1)
    loss = torch.nn.NLLLoss()(scores, torch.from_numpy(y))
    loss.backward()
    got = O.head_loss_and_grads(h, W, b, y)
    for a, ref in zip(got, (loss.detach().numpy(), scores.detach().numpy(), ht.grad.numpy(), lin.weight.grad.numpy(),
                            lin.bias.grad.numpy())):
        assert np.allclose(a, ref, rtol=1e-12, atol=1e-14)


def test_oracle_stack_matches_the_reference_cells_chained(stack_golden):
    """The two-layer default model (trainingConfig.py:12-15) end to end: loss, keyword scores, last state, d_x and
    every layer's gradients against the fixture produced by chaining the reference's own cells (fp64)."""
    g = stack_golden
    layers = [p for p, _ in g["layers"]]
    loss, scores, h_last, d_x, grads, d_w, d_b = O.stack_forward_backward(g["x"], layers, g["fc_w"], g["fc_b"],
                                                                         g["labels"])
    assert abs(float(loss) - float(g["loss"])) <= 1e-12
    assert np.abs(scores - g["scores"]).max() <= 1e-12 and np.abs(h_last - g["h_last"]).max() <= 1e-12
    assert np.abs(d_x - g["dx"]).max() <= 1e-12
    assert np.abs(d_w - g["dfc_w"]).max() <= 1e-12 and np.abs(d_b - g["dfc_b"]).max() <= 1e-12
    for (p, dp), got in zip(g["layers"], grads):
        for k, v in dp.items():
            assert np.abs(got[k].reshape(v.shape) - v).max() <= 1e-11 * max(1.0, np.abs(v).max()), k
