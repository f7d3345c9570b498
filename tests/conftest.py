import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Load a fixture written by tests/golden/make_golden.py and convert the CPU-cell
    parameter layout to the operator boundary's [out,in] layout."""
    d = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    g = {"x": d["x"], "h0": d["h0"], "G": d["G"], "hs": d["hs"], "dx": d["dx"], "dh0": d["dh0"],
         "gate": str(d["meta_gate"]), "update": str(d["meta_update"]), "dtype": str(d["meta_dtype"])}
    p, dp = {}, {}
    for cpu, bnd in (("W", "w"), ("U", "u"), ("W1", "w1"), ("W2", "w2"), ("U1", "u1"), ("U2", "u2")):
        if cpu in d:
            p[bnd] = np.ascontiguousarray(d[cpu].T)
            dp["d_" + bnd] = np.ascontiguousarray(d["d" + cpu].T)
    for k in ("bias_gate", "bias_update", "zeta", "nu"):
        p[k] = d[k]
        dp["d_" + k] = d["d" + k]
    g["params"] = p
    g["dparams"] = dp
    return g


# single-layer fixtures (run_case) and whole-model fixtures (run_stack_case: layers chained + head + loss)
ALL_GOLDEN = sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and "_stack2_" not in f)
STACK_GOLDEN = sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and "_stack2_" in f)


def load_stack_golden(name):
    """Whole-model fixture: per-layer parameters / gradients converted to the boundary's [out,in] layout."""
    d = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    hidden = [int(h) for h in d["meta_hidden"]]
    g = {k: d[k] for k in ("x", "labels", "scores", "loss", "dx", "h_last", "fc_w", "fc_b", "dfc_w", "dfc_b")}
    g["dtype"] = str(d["meta_dtype"])
    g["hidden"] = hidden
    layers = []
    for l in range(len(hidden)):
        p, dp = {}, {}
        for cpu, bnd in (("W", "w"), ("U", "u")):
            p[bnd] = np.ascontiguousarray(d["l%d_%s" % (l, cpu)].T)
            dp["d_" + bnd] = np.ascontiguousarray(d["l%d_d%s" % (l, cpu)].T)
        for k in ("bias_gate", "bias_update", "zeta", "nu"):
            p[k] = d["l%d_%s" % (l, k)]
            dp["d_" + k] = d["l%d_d%s" % (l, k)]
        layers.append((p, dp))
    g["layers"] = layers
    g["name"] = name
    return g


@pytest.fixture(params=STACK_GOLDEN)
def stack_golden(request):
    return load_stack_golden(request.param)


@pytest.fixture(params=ALL_GOLDEN)
def golden(request):
    g = load_golden(request.param)
    g["name"] = request.param
    return g
