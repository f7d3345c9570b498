"""CPU-side checks of the C-ABI library: it loads, exports every symbol the header
declares, and its host-only functions (status strings, path selection, workspace sizes,
argument validation) behave.  No kernel is launched here."""
import ctypes as C
import os
import re

import pytest

from kws_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "fastgrnn_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(fastgrnn_hip_\w+)\s*\(", hdr)))
    assert len(declared) >= 9
    for name in declared:
        assert hasattr(lib, name), name
    assert set(declared) == set(_lib.EXPORTS)


def test_abi_version_and_status_strings(lib):
    assert lib.fastgrnn_hip_abi_version() == 1
    assert _lib.status_string(0) == "ok"
    for code in range(1, 8):
        assert _lib.status_string(code) not in ("ok", "unknown status")
    assert _lib.status_string(99) == "unknown status"


def _desc(**kw):
    base = dict(T=99, B=4096, F=32, H=128, w_rank=0, u_rank=0, gate_nl=0, update_nl=2, dtype=0, flags=0)
    base.update(kw)
    return _lib.Desc(**base)


def test_kernel_path_is_pure_and_respects_force_generic(lib):
    d = _desc()
    a = lib.fastgrnn_hip_kernel_path(C.byref(d), 0)
    assert a in (0, 1, 2) and a == lib.fastgrnn_hip_kernel_path(C.byref(d), 0)
    assert lib.fastgrnn_hip_kernel_path(C.byref(_desc(flags=_lib.FLAG_FORCE_F32_MFMA)), 0) == 1
    d2 = _desc(flags=_lib.FLAG_FORCE_GENERIC)
    assert lib.fastgrnn_hip_kernel_path(C.byref(d2), 0) == 0
    assert lib.fastgrnn_hip_kernel_path(C.byref(d2), 1) == 0
    assert lib.fastgrnn_hip_kernel_path(C.byref(_desc(dtype=1)), 0) == 0      # fp64 -> generic
    assert lib.fastgrnn_hip_kernel_path(C.byref(_desc(H=20, F=7)), 0) == 0     # odd shape -> generic
    assert lib.fastgrnn_hip_kernel_path(C.byref(_desc(T=0)), 0) == -1


def test_workspace_queries(lib):
    d = _desc(flags=_lib.FLAG_FORCE_GENERIC)
    fw = lib.fastgrnn_hip_forward_workspace_bytes(C.byref(d))
    bw = lib.fastgrnn_hip_backward_workspace_bytes(C.byref(d))
    assert fw >= (32 * 128 + 128 * 128) * 4 and fw % 256 == 0
    assert bw >= 99 * 4096 * 128 * 4 and bw % 256 == 0
    assert lib.fastgrnn_hip_forward_workspace_bytes(C.byref(_desc(B=0))) == 0


def test_argument_validation_without_launch(lib):
    """Bad arguments are rejected before anything touches the GPU."""
    null = C.c_void_p(None)
    d = _desc(B=4, T=3)
    p = _lib.Params()
    # NULL params / tensors
    st = lib.fastgrnn_hip_forward_unroll(C.byref(d), C.byref(p), null, null, null, null, null, null, 0, null)
    assert st == 1
    st = lib.fastgrnn_hip_forward_unroll(C.byref(_desc(H=0)), C.byref(p), null, null, null, null, null, null, 0, null)
    assert st == 2
    st = lib.fastgrnn_hip_forward_unroll(C.byref(_desc(gate_nl=9)), C.byref(p), null, null, null, null, null, null, 0, null)
    assert st == 3
    st = lib.fastgrnn_hip_forward_unroll(C.byref(_desc(dtype=5)), C.byref(p), null, null, null, null, null, null, 0, null)
    assert st == 4
    # single-step entry points insist on T == 1
    st = lib.fastgrnn_hip_forward(C.byref(d), C.byref(p), null, null, null, null, null, null, 0, null)
    assert st == 2
    # workspace too small: fake non-null pointers, valid desc -> must fail on the workspace check
    one = C.c_void_p(256)
    pf = _lib.Params(*([one] * 10))
    dg = _desc(B=4, T=3, flags=_lib.FLAG_FORCE_GENERIC)
    st = lib.fastgrnn_hip_forward_unroll(C.byref(dg), C.byref(pf), one, one, one, null, null, null, 0, null)
    assert st == 5
    g = _lib.Grads()
    st = lib.fastgrnn_hip_backward_unroll(C.byref(dg), C.byref(pf), one, one, one, one, one, one, C.byref(g), null, 0, null)
    assert st == 1
    # d_x alone missing: an error on the shapes whose scan produces it, accepted (-> the workspace check, 5) where it
    # is a GEMM of its own behind the scan (dense H=256, dense H=128 with F > 32; kernel path 2)
    gx = _lib.Grads(*([null] + [one] * 11))
    d128 = _desc(B=4, T=3, flags=_lib.FLAG_SAVE_PREACT)
    assert lib.fastgrnn_hip_backward_unroll(C.byref(d128), C.byref(pf), one, one, one, one, one, one, C.byref(gx), null, 0, null) == 1
    for kw in (dict(H=256), dict(F=256)):
        dd = _desc(B=4, T=3, flags=_lib.FLAG_SAVE_PREACT, **kw)
        assert lib.fastgrnn_hip_backward_unroll(C.byref(dd), C.byref(pf), one, one, one, one, one, one, C.byref(gx), null, 0, null) == 5


def test_last_state_flags_are_refused_where_no_kernel_implements_them(lib):
    """FLAG_GRAD_LAST / FLAG_HS_LAST exist on the split-precision kernels only: every other path --
    and HS_LAST together with tensors saved for a backward -- answers FASTGRNN_ERR_UNSUPPORTED (7) before any
    launch; path selection reports the same."""
    null, one = C.c_void_p(None), C.c_void_p(256)
    pf = _lib.Params(*([one] * 10))
    UNSUP = 7
    assert _lib.status_string(UNSUP) != "unknown status"
    for extra in (_lib.FLAG_FORCE_GENERIC, _lib.FLAG_FORCE_F32_MFMA):
        d = _desc(B=4, T=3, flags=_lib.FLAG_HS_LAST | extra)
        assert lib.fastgrnn_hip_forward_unroll(C.byref(d), C.byref(pf), one, one, one, null, null, null, 0, null) == UNSUP
        d = _desc(B=4, T=3, flags=_lib.FLAG_GRAD_LAST | extra)
        g = _lib.Grads(*([one] * 12))
        assert lib.fastgrnn_hip_backward_unroll(C.byref(d), C.byref(pf), one, one, one, one, one, one, C.byref(g),
                                                null, 0, null) == UNSUP
    # other shapes (ranks above 16, half-factorised cells, odd sizes) have no last-state kernels either
    assert lib.fastgrnn_hip_kernel_path(C.byref(_desc(flags=_lib.FLAG_GRAD_LAST)), 1) == 2
    assert lib.fastgrnn_hip_kernel_path(C.byref(_desc(flags=_lib.FLAG_HS_LAST)), 0) == 2
    lr = dict(H=256, w_rank=16, u_rank=16)
    assert lib.fastgrnn_hip_kernel_path(C.byref(_desc(flags=_lib.FLAG_GRAD_LAST | _lib.FLAG_SAVE_PREACT, **lr)), 1) == 2
    assert lib.fastgrnn_hip_kernel_path(C.byref(_desc(flags=_lib.FLAG_GRAD_LAST, **lr)), 1) != 2      # reference contract
    assert lib.fastgrnn_hip_kernel_path(C.byref(_desc(flags=_lib.FLAG_HS_LAST, **lr)), 0) == 2
    # the other factorised H=256 cells run on the dense H=256 kernels (factors multiplied out per call): path 2, with
    # that path's own limits (batch-major since round 3; no bf16, no [B,F,T] frames beside batch-major sequences in the
    # backward); ranks above H have no fast path
    for w_rank, u_rank in ((32, 32), (16, 0), (0, 16), (17, 16)):
        d = _desc(H=256, w_rank=w_rank, u_rank=u_rank, flags=_lib.FLAG_GRAD_LAST | _lib.FLAG_SAVE_PREACT)
        assert lib.fastgrnn_hip_kernel_path(C.byref(d), 1) == 2
        d = _desc(H=256, w_rank=w_rank, u_rank=u_rank, flags=_lib.FLAG_BATCH_MAJOR | _lib.FLAG_SAVE_PREACT)
        assert lib.fastgrnn_hip_kernel_path(C.byref(d), 1) == 2
        d = _desc(H=256, w_rank=w_rank, u_rank=u_rank, flags=_lib.FLAG_BATCH_MAJOR | _lib.FLAG_X_BFT | _lib.FLAG_SAVE_PREACT)
        assert lib.fastgrnn_hip_kernel_path(C.byref(d), 1) != 2
    assert lib.fastgrnn_hip_kernel_path(C.byref(_desc(H=256, w_rank=300, u_rank=8, flags=_lib.FLAG_SAVE_PREACT)), 1) != 2
    assert lib.fastgrnn_hip_kernel_path(C.byref(_desc(H=64, flags=_lib.FLAG_HS_LAST)), 0) != 2
    # HS_LAST with a tensor to save
    d = _desc(B=4, T=3, flags=_lib.FLAG_HS_LAST)
    assert lib.fastgrnn_hip_forward_unroll(C.byref(d), C.byref(pf), one, one, one, one, null, null, 0, null) == UNSUP
    assert lib.fastgrnn_hip_kernel_path(C.byref(_desc(flags=_lib.FLAG_HS_LAST | _lib.FLAG_SAVE_PREACT)), 0) != 2


def test_module_parameter_layout_matches_reference_classes():
    """rnn.py:782-805: [out,in] shapes and state-dict key names."""
    from kws_amd import FastGRNNCUDA, FastGRNNCUDACell
    m = FastGRNNCUDA(32, 128, device="cpu")
    sd = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert sd == {"W": (128, 32), "U": (128, 128), "bias_gate": (1, 128), "bias_update": (1, 128),
                  "zeta": (1, 1), "nu": (1, 1)}
    assert float(m.zeta) == 1.0 and float(m.nu) == -4.0 and float(m.bias_gate.min()) == 1.0
    m = FastGRNNCUDA(32, 256, wRank=16, uRank=16, device="cpu")
    sd = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert sd["W1"] == (16, 32) and sd["W2"] == (256, 16) and sd["U1"] == (16, 256) and sd["U2"] == (256, 16)
    assert "W" not in sd and "U" not in sd
    assert m.W.numel() == 0 and m.U.numel() == 0
    assert [tuple(v.shape) for v in m.getVars()] == [(16, 32), (256, 16), (16, 256), (256, 16), (1, 256), (1, 256), (1, 1), (1, 1)]
    c = FastGRNNCUDACell(32, 128, gate_nonlinearity="tanh", device="cpu")
    assert c._gate_non_linearity == 2 and c.cellType == "FastGRNNCUDACell"
    with pytest.raises(KeyError):
        FastGRNNCUDA(32, 128, gate_nonlinearity="quantSigm", device="cpu")


def test_no_gpu_means_loud_failure():
    import torch
    from kws_amd import FastGRNNCUDA
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(Exception, match="supported only on GPU"):
        FastGRNNCUDA(32, 128)
    m = FastGRNNCUDA(32, 128, device="cpu")
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        m(torch.randn(3, 2, 32))


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure; nothing under kws_amd/ may reference it."""
    pkg = os.path.join(ROOT, "kws_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("test oracle", ""), os.path.join(dirpath, f)


# ---- sparsification helpers: device-side mirrors of the reference's utils.py:53-114 ---------
def _np_hard_threshold(a, s):
    """The reference's arithmetic restated with numpy (utils.py:57-63)."""
    import numpy as np
    a = a.copy().ravel()
    if len(a):
        th = np.percentile(np.abs(a), (1 - s) * 100.0, method="higher")
        a[np.abs(a) < th] = 0.0
    return a


def test_hard_threshold_matches_numpy_percentile_higher():
    import numpy as np
    import torch
    from kws_amd import utils as U
    rng = np.random.default_rng(0)
    for shape in ((128, 128), (128, 32), (7, 3), (1, 1), (0,)):
        for s in (1.0, 0.5, 0.31, 0.1, 0.013, 0.0):
            a = rng.standard_normal(shape).astype(np.float32)
            if a.size > 4:
                a.ravel()[:3] = a.ravel()[3]              # ties
            t = torch.from_numpy(a.copy())
            out = U.hardThreshold(t, s)
            assert out is t
            np.testing.assert_array_equal(t.numpy().ravel(), _np_hard_threshold(a, s), err_msg="%s s=%g" % (shape, s))


def test_support_threshold_model_size_and_module_methods():
    import torch
    from kws_amd import utils as U
    from kws_amd.rnn import FastGRNNCUDA
    src = torch.tensor([[0.0, 1.0], [2.0, 0.0]])
    dst = torch.tensor([[5.0, 6.0], [7.0, 8.0]])
    assert torch.equal(U.supportBasedThreshold(dst, src), torch.tensor([[0.0, 6.0], [7.0, 0.0]]))
    assert U.estimateNNZ(torch.zeros(10, 10), 0.3) == (30, 240, True) and U.estimateNNZ(torch.zeros(10, 10), 0.5) == (100, 400, False)
    assert U.countNNZ(src, True) == 2 and U.countNNZ(src, False) == 4
    m = FastGRNNCUDA(32, 128, wSparsity=0.25, uSparsity=0.1, device="cpu")
    # zeta and nu are counted twice, as in the reference ("totalnnz = 2" plus their getVars entries, rnn.py:851,861)
    dense_bytes = 4 * (128 * 32 + 128 * 128 + 2 * 128 + 2 + 2)
    assert m.get_model_size() == dense_bytes                 # nothing is zero yet: non-zero counts = sizes
    m.sparsify()
    nzW, nzU = int((m.W != 0).sum()), int((m.U != 0).sum())
    assert abs(nzW - 0.25 * 128 * 32) <= 2 and abs(nzU - 0.1 * 128 * 128) <= 2
    assert m.get_model_size() == 4 * (nzW + nzU + 2 * 128 + 2 + 2)
    with torch.no_grad():
        m.W.add_(1.0); m.U.add_(1.0)                         # a training step fills the zeros again ...
    m.sparsifyWithSupport()                                  # ... and the remembered support removes them
    assert int((m.W != 0).sum()) <= nzW and int((m.U != 0).sum()) <= nzU
    assert torch.equal(m.W == 0, m.oldmats[0] == 0)
