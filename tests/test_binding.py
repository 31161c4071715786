"""The reference-side binding of INTEGRATION.md section B, compiled (integration/TCGNN_binding.cpp -> integration/TCGNN*.so by
integration/setup.py, part of build()): a pybind11 torch extension named `TCGNN` with the reference's seven names
(TCGNN_conv/TCGNN.cpp:260-272) whose bodies call the C ABI.  It is a SECOND backend next to the ctypes module: the parity
tests below run the three kernels and the layer library through it."""
import glob
import importlib.util
import os

import numpy as np
import pytest
import torch

import graphs
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def ext():
    found = glob.glob(os.path.join(ROOT, "integration", "TCGNN*.so"))
    assert found, "integration/TCGNN*.so is not built: python integration/setup.py build_ext --inplace (build() does it)"
    spec = importlib.util.spec_from_file_location("TCGNN", found[0])   # PyInit_TCGNN: the reference's module name
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_module_exports_the_reference_names(ext):
    for name in ("preprocess", "preprocess_gpu", "forward", "forward_ef", "forward_AGNN", "backward", "backward_ef"):
        assert callable(getattr(ext, name)), name
    # TCGNN.cpp:54-56 through the binding: the first check that fires names the offending argument
    x = torch.zeros(4, 8); i = torch.zeros(4, dtype=torch.int32)
    with pytest.raises(RuntimeError, match="input must be a CUDA tensor"):
        ext.forward(x, i, i, i, i, i)
    with pytest.raises(RuntimeError, match="input must be a CUDA tensor"):
        ext.forward_AGNN(x, i, i, x, i, i, i)


@pytest.mark.parametrize("name", ["uniform_n1000", "uniform_n32", "empty_middle_window_n48", "powerlaw_n1000"])
def test_preprocess_through_the_binding_equals_the_reference_fixture(ext, name, capfd):
    f = np.load(os.path.join(GOLD, "sgt_%s.npz" % name))
    rp, col, guard = f["rowptr"], f["col"], int(f["guard"])
    n = len(rp) - 1
    nw = (n + 15) // 16
    bp = torch.zeros(nw, dtype=torch.int32); e2c = torch.zeros(len(col), dtype=torch.int32); e2r = torch.zeros(len(col), dtype=torch.int32)
    assert ext.preprocess(torch.from_numpy(col), torch.from_numpy(rp), n, 16, 8, bp, e2c, e2r) is None
    assert np.array_equal(bp.numpy(), f["bp_with_guard"][:nw]) and np.array_equal(e2c.numpy(), f["e2c"]) and np.array_equal(e2r.numpy(), f["e2r"])
    out = capfd.readouterr().out
    assert "TC_Blocks:\t%d\nExp_Edges:\t%d\n" % (int(f["tc_blocks"]), int(f["tc_blocks"]) * 128) in out


CASES = [c for c in graphs.edge_case_graphs() if c[0] in ("uniform_n17", "uniform_n1000", "empty_middle_window_n48", "powerlaw_n1000", "no_edges_n20")]
CASES.append(("dense_n3000_deg150", *graphs.uniform_graph(3000, 150, seed=2)))


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("D", [16, 64, 41, 160])
def test_three_kernels_through_the_binding_match_the_oracle(ext, case, D):
    _, rp, col = case
    dev = torch.device("cuda:0")
    n, nnz = len(rp) - 1, len(col)
    trp, tcol = torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev)
    bp = torch.zeros((n + 15) // 16, dtype=torch.int32, device=dev); e2c = torch.zeros(nnz, dtype=torch.int32, device=dev); e2r = torch.zeros(nnz, dtype=torch.int32, device=dev)
    ext.preprocess_gpu(tcol, trp, n, 16, 8, bp, e2c, e2r)
    hbp, he2c, he2r, _ = graphs.host_sgt(rp, col)
    assert np.array_equal(bp.cpu().numpy(), hbp) and np.array_equal(e2c.cpu().numpy(), he2c) and np.array_equal(e2r.cpu().numpy(), he2r)
    rng = np.random.default_rng(D + n)
    X = rng.standard_normal((n, D)).astype(np.float32); att = rng.standard_normal(nnz).astype(np.float32)
    tX, tatt = torch.from_numpy(X).to(dev), torch.from_numpy(att).to(dev)

    def close(got, ref, ref64, scale):
        got = got.cpu().numpy()
        assert got.shape == ref.shape
        if got.size:
            assert (np.abs(got - ref) / np.maximum(1.0, np.abs(ref))).max() <= 1e-3            # the north-star bar
            assert (np.abs(got - ref) / (scale + 1.0)).max() <= 4e-6                             # accumulation-order noise only
            assert (np.abs(got - ref64) / (scale + 1.0)).max() <= 2.0 ** -9
    for fn in (ext.forward, ext.backward):
        Y = fn(tX, trp, tcol, bp, e2c, e2r)
        assert isinstance(Y, list) and len(Y) == 1 and Y[0].dtype == torch.float32 and Y[0].device == tX.device
        close(Y[0], O.spmm(X, rp, col, hbp, he2c, he2r, round_mode=O.ROUND_TF32), *O.spmm_f64(X, rp, col))
    Yv = ext.forward_AGNN(tX, trp, tcol, tatt.view(1, -1), bp, e2c, e2r)[0]
    close(Yv, O.spmm_val(X, rp, col, att, hbp, he2c, he2r, round_mode=O.ROUND_TF32), *O.spmm_f64(X, rp, col, att))
    for fn in (ext.forward_ef, ext.backward_ef):
        ef = fn(tX, trp, tcol, bp, e2c, e2r)[0]
        assert ef.shape == (nnz,)
        close(ef, O.sddmm(X, rp, col, hbp, he2c, he2r, round_mode=O.ROUND_TF32), *O.sddmm_f64(X, rp, col))
    # the other backend (ctypes module over the same library) gives the same bits
    import TCGNN as T
    assert torch.equal(T.forward(tX, trp, tcol, bp, e2c, e2r)[0], ext.forward(tX, trp, tcol, bp, e2c, e2r)[0])
    assert torch.equal(T.forward_ef(tX, trp, tcol, bp, e2c, e2r)[0], ext.forward_ef(tX, trp, tcol, bp, e2c, e2r)[0])
    ext.clear_plan_cache()


@pytest.mark.gpu
def test_layer_library_on_the_binding_reproduces_the_reference_fixture(ext):
    """tcgnn_layers with the compiled extension installed as its backend (exactly what the reference's gnn_conv.py would
    import): forward values and gradients captured from the reference's own gnn_conv.py (tests/golden/layers_n200.npz)."""
    import tcgnn_layers as L
    dev = torch.device("cuda:0")
    f = np.load(os.path.join(GOLD, "layers_n200.npz"))
    t = lambda k: torch.from_numpy(f[k]).to(dev)
    meta = (t("rowptr"), t("col"), t("bp"), t("e2c"), t("e2r"))
    dY = t("dY")
    old = L._backend
    L.set_backend(ext)
    try:
        def close(a, key, tol=2e-3):
            b = f[key]
            assert np.allclose(a.detach().cpu().numpy(), b, rtol=tol, atol=tol * max(1.0, float(np.abs(b).max()))), key
        x, w = t("X").clone().requires_grad_(True), t("W").clone().requires_grad_(True)
        y = L.TCGNNFunction.apply(x, w, *meta); y.backward(dY)
        close(y, "gcn_Y"); close(x.grad, "gcn_dX"); close(w.grad, "gcn_dW")
        x, w = t("X").clone().requires_grad_(True), t("W").clone().requires_grad_(True)
        y = L.TCGNNFunction_GIN.apply(x, w, *meta); y.backward(dY)
        close(y, "gin_Y"); close(x.grad, "gin_dX"); close(w.grad, "gin_dW")
        x, w, a = t("X").clone().requires_grad_(True), t("W").clone().requires_grad_(True), t("attention_w").clone().requires_grad_(True)
        y = L.TCGNNFunction_AGNN.apply(x, w, a, *meta); y.backward(dY)   # the extension has no fused pair: the separate calls
        close(y, "agnn_Y"); close(x.grad, "agnn_dX"); close(w.grad, "agnn_dW"); close(a.grad, "agnn_dattention_w", tol=5e-3)
    finally:
        L.set_backend(old)
        ext.clear_plan_cache()
