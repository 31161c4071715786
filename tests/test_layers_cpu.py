"""The layer library and the dataset loader against fixtures captured from the reference's own
gnn_conv.py / dataset.py (tests/golden/make_golden.py).  The operators come from the oracle here
(CPU); tests/test_gpu_parity.py repeats the layer check with the HIP kernels."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def oracle_backend():
    m = types.SimpleNamespace()
    npy = lambda *ts: [t.detach().cpu().numpy() for t in ts]
    m.forward = lambda X, rp, col, bp, e2c, e2r: [torch.from_numpy(O.spmm(*npy(X, rp, col, bp, e2c, e2r), round_mode=O.ROUND_TF32))]
    m.forward_ef = lambda X, rp, col, bp, e2c, e2r: [torch.from_numpy(O.sddmm(*npy(X, rp, col, bp, e2c, e2r), round_mode=O.ROUND_TF32))]

    def forward_AGNN(X, rp, col, att, bp, e2c, e2r):
        x, rp_, col_, att_, bp_, e2c_, e2r_ = npy(X, rp, col, att, bp, e2c, e2r)
        return [torch.from_numpy(O.spmm_val(x, rp_, col_, att_[0], bp_, e2c_, e2r_, round_mode=O.ROUND_TF32))]
    m.forward_AGNN = forward_AGNN
    return m


@pytest.fixture
def layers():
    import tcgnn_layers as L
    old = L._backend
    L.set_backend(oracle_backend())
    yield L
    L.set_backend(old)


def _close(a, b, tol=2e-5):
    b = np.asarray(b)
    return np.allclose(a.detach().numpy(), b, rtol=tol, atol=tol * max(1.0, float(np.abs(b).max())))


def test_functions_reproduce_reference_forward_and_backward(layers):
    f = np.load(os.path.join(GOLD, "layers_n200.npz"))
    t = lambda k: torch.from_numpy(f[k])
    meta = (t("rowptr"), t("col"), t("bp"), t("e2c"), t("e2r"))
    dY = t("dY")

    x = t("Xs").clone().requires_grad_(True)
    y = layers.TCGNNFunction_SAG.apply(x, *meta); y.backward(dY)
    assert _close(y, f["sag_Y"]) and _close(x.grad, f["sag_dX"])

    x, w = t("X").clone().requires_grad_(True), t("W").clone().requires_grad_(True)
    y = layers.TCGNNFunction.apply(x, w, *meta); y.backward(dY)
    assert _close(y, f["gcn_Y"]) and _close(x.grad, f["gcn_dX"]) and _close(w.grad, f["gcn_dW"])

    x, w = t("X").clone().requires_grad_(True), t("W").clone().requires_grad_(True)
    y = layers.TCGNNFunction_GIN.apply(x, w, *meta); y.backward(dY)
    assert _close(y, f["gin_Y"]) and _close(x.grad, f["gin_dX"]) and _close(w.grad, f["gin_dW"])

    x, w, a = t("X").clone().requires_grad_(True), t("W").clone().requires_grad_(True), t("attention_w").clone().requires_grad_(True)
    y = layers.TCGNNFunction_AGNN.apply(x, w, a, *meta); y.backward(dY)
    assert _close(y, f["agnn_Y"]) and _close(x.grad, f["agnn_dX"]) and _close(w.grad, f["agnn_dW"])
    assert _close(a.grad, f["agnn_dattention_w"], tol=1e-4)


def test_module_facts_match_reference(layers):
    f = np.load(os.path.join(GOLD, "layers_facts.npz"))
    assert layers.n_heads == int(f["n_heads"])
    torch.manual_seed(0)
    conv = layers.AGNNConv(8, 4)
    assert float(conv.weights.detach().abs().max()) <= float(f["agnn_weight_bound"]) + 1e-6
    assert list(conv.attention_w.shape) == list(f["agnn_attention_shape"])
    assert list(layers.GCNConv(8, 4).weights.shape) == list(f["gcn_weight_shape"])
    assert list(layers.GINConv(8, 4).weights.shape) == [8, 4]


def test_fuse_relu_without_a_fused_backend_still_masks_the_gradient(layers):
    """A direct TCGNNFunction.apply(..., True) caller on a backend that lacks forward_fused (the oracle backend here) gets the
    plain ReLU AND its backward mask: same values and gradients as relu(TCGNNFunction.apply(...)) through autograd."""
    f = np.load(os.path.join(GOLD, "layers_n200.npz"))
    t = lambda k: torch.from_numpy(f[k])
    meta = (t("rowptr"), t("col"), t("bp"), t("e2c"), t("e2r"))
    dY = t("dY")
    assert not hasattr(layers.backend(), "forward_fused")
    x1, w1 = t("X").clone().requires_grad_(True), t("W").clone().requires_grad_(True)
    y1 = layers.TCGNNFunction.apply(x1, w1, *meta, True); y1.backward(dY)
    x2, w2 = t("X").clone().requires_grad_(True), t("W").clone().requires_grad_(True)
    y2 = torch.relu(layers.TCGNNFunction.apply(x2, w2, *meta)); y2.backward(dY)
    assert (y1 < 0).sum() == 0 and (y1 == 0).any()
    assert torch.equal(y1, y2) and torch.equal(x1.grad, x2.grad) and torch.equal(w1.grad, w2.grad)


def test_dataset_loader_matches_reference_fixture(tmp_path):
    import tcgnn_graph as G
    f = np.load(os.path.join(GOLD, "dataset_toy.npz"))
    path = os.path.join(tmp_path, "toy.npz")
    np.savez(path, src_li=f["src_li"], dst_li=f["dst_li"], num_nodes=f["num_nodes"])
    ds = G.TCGNN_dataset(path, 12, 5, load_from_txt=False)
    assert np.array_equal(ds.row_pointers.numpy(), f["row_pointers"])
    assert np.array_equal(ds.column_index.numpy(), f["column_index"])
    assert ds.num_edges == int(f["num_edges"]) and ds.num_edges > ds.column_index.numel()   # raw count > nnz
    assert ds.row_pointers.dtype == torch.int32 and ds.column_index.dtype == torch.int32
    assert list(ds.x.shape) == list(f["x_shape"]) and np.array_equal(ds.y.numpy(), f["y"])
    assert ds.num_features == int(f["num_features"]) and ds.num_classes == int(f["num_classes"])
    txt = os.path.join(tmp_path, "toy.txt")
    np.savetxt(txt, np.stack([f["src_li"], f["dst_li"]], 1), fmt="%d")
    ds2 = G.TCGNN_dataset(txt, 12, 5, load_from_txt=True)
    assert np.array_equal(ds2.column_index.numpy()[: 10], f["column_index"][: 10])
    with pytest.raises(ValueError):
        G.TCGNN_dataset(txt, 12, 5, load_from_txt=False)


def test_synthetic_generator_is_seeded_symmetric_and_canonical():
    import tcgnn_graph as G
    rp, col = G.synthetic_csr(2000, 40000, seed=3)
    rp2, col2 = G.synthetic_csr(2000, 40000, seed=3)
    assert torch.equal(rp, rp2) and torch.equal(col, col2)
    assert abs(col.numel() - 40000) <= 40
    import scipy.sparse as sp
    a = sp.csr_matrix((np.ones(col.numel()), col.numpy(), rp.numpy()), shape=(2000, 2000))
    assert (a != a.T).nnz == 0 and a.diagonal().sum() == 0 and a.has_canonical_format
    rp3, col3, dim, classes = G.synthetic_shape("reddit", scale=0.01)
    assert (dim, classes) == (602, 41) and rp3.numel() - 1 == 2329


def test_input_gradient_is_skipped_when_not_needed_and_weights_are_unchanged(layers):
    """First-layer features carry no gradient: the Functions must not compute dX, and dW must not change."""
    f = np.load(os.path.join(GOLD, "layers_n200.npz"))
    t = lambda k: torch.from_numpy(f[k])
    meta = (t("rowptr"), t("col"), t("bp"), t("e2c"), t("e2r"))
    calls = []
    inner = layers.backend()
    spy = types.SimpleNamespace(forward=lambda *a: (calls.append("spmm"), inner.forward(*a))[1],
                                forward_ef=lambda *a: (calls.append("sddmm"), inner.forward_ef(*a))[1],
                                forward_AGNN=lambda *a: (calls.append("spmm_val"), inner.forward_AGNN(*a))[1])
    layers.set_backend(spy)
    for fn, key, n_bwd in ((layers.TCGNNFunction, "gcn_dW", 1), (layers.TCGNNFunction_GIN, "gin_dW", 0)):
        x, w = t("X").clone(), t("W").clone().requires_grad_(True)
        y = fn.apply(x, w, *meta)
        calls.clear()
        y.backward(t("dY"))
        assert x.grad is None and _close(w.grad, f[key]) and calls.count("spmm") == n_bwd
    x, w, a = t("X").clone(), t("W").clone().requires_grad_(True), t("attention_w").clone().requires_grad_(True)
    layers.TCGNNFunction_AGNN.apply(x, w, a, *meta).backward(t("dY"))
    assert x.grad is None and _close(w.grad, f["agnn_dW"]) and _close(a.grad, f["agnn_dattention_w"], tol=1e-4)


def test_harness_loss_is_nll_loss():
    import tcgnn_harness as H
    torch.manual_seed(0)
    z = torch.randn(37, 5, requires_grad=True)
    y = torch.randint(0, 5, (37,))
    a = H.node_nll_loss(torch.log_softmax(z, 1), y)
    ga, = torch.autograd.grad(a, z)
    b = torch.nn.functional.nll_loss(torch.log_softmax(z, 1), y)
    gb, = torch.autograd.grad(b, z)
    assert torch.allclose(a, b, rtol=1e-6, atol=1e-7) and torch.allclose(ga, gb, rtol=1e-6, atol=1e-8)


def test_split_weight_gradient_product_equals_plain_mm():
    import tcgnn_layers as L
    torch.manual_seed(1)
    for n in (100, L._SPLIT_ROWS + 37):
        A, B = torch.randn(n, 19), torch.randn(n, 7)
        ref = torch.mm(A.double().t(), B.double())
        out = L.tall_tn_mm(A, B)
        assert out.shape == (19, 7) and torch.allclose(out.double(), ref, rtol=1e-5, atol=1e-4 * float(ref.abs().max()))
    A = torch.randn(L._SPLIT_ROWS + 5, 24)[:, ::2]          # non-contiguous operand
    assert torch.allclose(L.tall_tn_mm(A, A), torch.mm(A.t(), A), rtol=1e-4, atol=1e-2)
