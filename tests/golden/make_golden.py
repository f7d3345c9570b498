#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own CPU cell.

Run in the build container only (needs /root/reference, read-only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports ``rnn.FastGRNNCell`` from /root/reference, drives it in the
per-timestep loop of ``BaseRNN.forward`` (rnn.py:588-591,657-668 -- BaseRNN itself
raises TypeError with FastGRNNCell, SURVEY.md section 0.2, so the loop is written
here), takes gradients of ``L = sum(hs * G)`` with torch autograd, and stores
inputs + expected outputs as small ``.npz`` fixtures next to this script.  Only
DATA is written; no reference source travels.

Parameters are stored in the CPU cell's layout (W:[F,H], U:[H,H], W1:[F,r],
W2:[r,H], U1:[H,r], U2:[r,H]; rnn.py:247-256); loaders transpose to the
operator boundary's [out,in] layout.
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
import rnn  # noqa: E402  (the reference)

HERE = os.path.dirname(os.path.abspath(__file__))


def run_case(name, T, B, F, H, dtype, seed, wRank=None, uRank=None, gate="sigmoid",
             update="tanh", nonzero_h0=False, randomize_scalars=False):
    torch.manual_seed(seed)
    cell = rnn.FastGRNNCell(F, H, gate_nonlinearity=gate, update_nonlinearity=update,
                            wRank=wRank, uRank=uRank)
    if randomize_scalars:
        with torch.no_grad():
            cell.bias_gate.add_(0.5 * torch.randn_like(cell.bias_gate))
            cell.bias_update.add_(0.5 * torch.randn_like(cell.bias_update))
            cell.zeta.fill_(0.7)
            cell.nu.fill_(-2.5)
    tdt = torch.float64 if dtype == "f64" else torch.float32
    cell = cell.to(tdt)
    if wRank is not None:
        # FastGRNNCell.forward reads self.W.device unconditionally (rnn.py:274);
        # with wRank set there is no self.W.  Give the instance a
        # plain non-Parameter attribute so the reference arithmetic at :277-297 runs.
        object.__setattr__(cell, "W", cell.W1.detach())
    x = torch.randn(T, B, F, dtype=tdt, requires_grad=True)
    h0 = (0.5 * torch.randn(B, H, dtype=tdt)) if nonzero_h0 else torch.zeros(B, H, dtype=tdt)
    h0.requires_grad_(True)
    G = torch.randn(T, B, H, dtype=tdt)
    h = h0
    hs = []
    for t in range(T):                    # BaseRNN semantics, rnn.py:657-660
        h = cell(x[t], h)
        hs.append(h)
    hs = torch.stack(hs, 0)
    (hs * G).sum().backward()
    out = {"x": x.detach().numpy(), "h0": h0.detach().numpy(), "G": G.numpy(),
           "hs": hs.detach().numpy(), "dx": x.grad.numpy(), "dh0": h0.grad.numpy()}
    for pname, par in cell.named_parameters():
        out[pname] = par.detach().numpy()
        out["d" + pname] = par.grad.numpy()
    out["meta_gate"] = np.array(gate)
    out["meta_update"] = np.array(update)
    out["meta_dtype"] = np.array(dtype)
    out["meta_torch"] = np.array(torch.__version__)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s T=%d B=%d F=%d H=%d %s -> %.1f KB" % (name, T, B, F, H, dtype, os.path.getsize(path) / 1024))


def run_stack_case(name, T, B, F, hidden, C, dtype, seed):
    """The reference's default model (trainingConfig.py:12-15: two dense layers 32 -> 256 -> 128) as
    RNNClassifierModel.forward chains it (model.py:196-203: layer l's [T,B,H] output is layer l+1's input;
    :226-230: Linear on the LAST state, then log_softmax) with the trainer's loss (trainClassifier.py:154,236:
    nn.NLLLoss()).  The cells are the reference's; the loop over layers and time is written here because its
    own FastGRNN wrapper raises with FastGRNNCell.  Stores inputs, every layer's parameters
    (CPU layout) and gradients, the loss, the keyword scores and d_x."""
    torch.manual_seed(seed)
    tdt = torch.float64 if dtype == "f64" else torch.float32
    sizes = [F] + list(hidden)
    cells = [rnn.FastGRNNCell(sizes[l], sizes[l + 1]).to(tdt) for l in range(len(hidden))]
    with torch.no_grad():
        for c in cells:
            c.bias_gate.add_(0.5 * torch.randn_like(c.bias_gate))
            c.bias_update.add_(0.5 * torch.randn_like(c.bias_update))
            c.zeta.fill_(0.7)
            c.nu.fill_(-2.5)
    head = torch.nn.Linear(hidden[-1], C).to(tdt)
    x = torch.randn(T, B, F, dtype=tdt, requires_grad=True)
    labels = torch.randint(0, C, (B,))
    seq = x
    for c in cells:                                   # model.py:196-203
        h = torch.zeros(B, c.state_size, dtype=tdt)
        outs = []
        for t in range(T):                            # rnn.py:657-660
            h = c(seq[t], h)
            outs.append(h)
        seq = torch.stack(outs, 0)
    scores = torch.log_softmax(head(seq[-1]), dim=1)  # model.py:226-230
    loss = torch.nn.NLLLoss()(scores, labels)         # trainClassifier.py:154,236
    loss.backward()
    out = {"x": x.detach().numpy(), "labels": labels.numpy(), "scores": scores.detach().numpy(),
           "loss": loss.detach().numpy(), "dx": x.grad.numpy(), "h_last": seq[-1].detach().numpy(),
           "fc_w": head.weight.detach().numpy(), "fc_b": head.bias.detach().numpy(),
           "dfc_w": head.weight.grad.numpy(), "dfc_b": head.bias.grad.numpy()}
    for l, c in enumerate(cells):
        for pname, par in c.named_parameters():
            out["l%d_%s" % (l, pname)] = par.detach().numpy()
            out["l%d_d%s" % (l, pname)] = par.grad.numpy()
    out["meta_dtype"] = np.array(dtype)
    out["meta_hidden"] = np.array(hidden)
    out["meta_torch"] = np.array(torch.__version__)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s T=%d B=%d F=%d hidden=%s %s -> %.1f KB" % (name, T, B, F, hidden, dtype, os.path.getsize(path) / 1024))


def main():
    only = sys.argv[1:]                               # optional: names of the cases to (re)generate
    if only:
        global run_case, run_stack_case
        _rc, _rs = run_case, run_stack_case
        run_case = lambda name, *a, **k: _rc(name, *a, **k) if name in only else None
        run_stack_case = lambda name, *a, **k: _rs(name, *a, **k) if name in only else None
    # G1 tiny, non-zero h0, randomised biases/zeta/nu
    run_case("g1_tiny_f64", 5, 3, 4, 8, "f64", 1, nonzero_h0=True, randomize_scalars=True)
    # G2 north-star shape (reference init) in fp32 and (randomised scalars) fp64
    run_case("g2_northstar_f32", 99, 4, 32, 128, "f32", 2)
    run_case("g2_northstar_f64", 99, 4, 32, 128, "f64", 3, randomize_scalars=True, nonzero_h0=True)
    # G3 low-rank, config (4) shape
    run_case("g3_lowrank_f32", 99, 2, 32, 256, "f32", 4, wRank=16, uRank=16)
    run_case("g3_lowrank_f64", 99, 2, 32, 256, "f64", 5, wRank=16, uRank=16, randomize_scalars=True)
    # G4 tanh gate
    run_case("g4_tanhgate_f64", 5, 3, 4, 8, "f64", 6, gate="tanh", nonzero_h0=True, randomize_scalars=True)
    # G5 mixed rank (dense W, low-rank U) and (low-rank W, dense U), ragged sizes
    run_case("g5_mixed_urank_f64", 7, 3, 5, 12, "f64", 7, uRank=4, nonzero_h0=True, randomize_scalars=True)
    run_case("g5_mixed_wrank_f64", 7, 3, 5, 12, "f64", 8, wRank=3, randomize_scalars=True)
    # G6 odd sizes fp32 (nothing a multiple of anything), B not a multiple of the batch tile
    run_case("g6_odd_f32", 11, 19, 7, 20, "f32", 9, nonzero_h0=True, randomize_scalars=True)
    # G7 quantised nonlinearities
    run_case("g7_quant_f64", 6, 4, 5, 8, "f64", 10, gate="quantSigm", update="quantTanh",
             nonzero_h0=True, randomize_scalars=True)
    # G8 T=1 (single-step operator), B=1
    run_case("g8_single_f64", 1, 1, 32, 128, "f64", 11, nonzero_h0=True, randomize_scalars=True)
    # G9 second north-star-width case with B=17 (ragged against the 16-utterance MFMA tile)
    run_case("g9_ragged17_f32", 23, 17, 32, 128, "f32", 12, nonzero_h0=True, randomize_scalars=True)
    # G10 / G11: the two layers of the reference's default stack (trainingConfig.py:12-15), each on its own
    run_case("g10_stack_l1_f32", 99, 4, 32, 256, "f32", 13, randomize_scalars=True)
    run_case("g11_stack_l2_f32", 99, 4, 256, 128, "f32", 14, randomize_scalars=True)
    # G12: the whole default model -- both layers chained, last-state Linear, log_softmax, NLLLoss
    run_stack_case("g12_stack2_f64", 99, 4, 32, (256, 128), 12, "f64", 15)
    # G13: the first layer of the model the reference ships and trains by default -- feature_type='delta'
    # (trainingConfig.py:36) = 32 MFCCs + 32 deltas = 64 features (mfccProcessor.py:6,27-28) into 256 hidden units
    run_case("g13_delta64_l1_f32", 99, 4, 64, 256, "f32", 16, randomize_scalars=True)


if __name__ == "__main__":
    main()
