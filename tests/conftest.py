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


ALL_GOLDEN = sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz"))


@pytest.fixture(params=ALL_GOLDEN)
def golden(request):
    g = load_golden(request.param)
    g["name"] = request.param
    return g
