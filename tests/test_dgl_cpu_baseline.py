"""The CPU baseline bench.py times next to the GPU numbers (BASELINE.json configs[0]: "Cora GCN 2-layer hidden=16 via
dgl_baseline/gcn.py on CPU"): oracle/dgl_gcn_cpu.py restates DGL's GraphConv stack; DGL itself is third-party, absent and
unpinned by the reference, so this is checked against a dense-matrix evaluation of the same published formula."""
import numpy as np
import torch

import graphs
from oracle import dgl_gcn_cpu as B


def _dense(rp, col):
    n = len(rp) - 1
    A = np.zeros((n, n), np.float32)
    for r in range(n):
        A[r, col[rp[r]:rp[r + 1]]] = 1.0
    return torch.from_numpy(A)


def test_graphconv_stack_equals_dense_formula_forward_and_backward():
    # deliberately NOT symmetric: in-degree / out-degree normalisation and the transposed backward both matter
    rp, col = graphs.uniform_graph(300, 6, seed=4, symmetric=False)
    n, in_dim, hidden, classes = 300, 24, 16, 5
    rng = np.random.default_rng(0)
    X = torch.from_numpy(rng.standard_normal((n, in_dim)).astype(np.float32))
    y = torch.from_numpy(rng.integers(0, classes, size=n))
    torch.manual_seed(1)
    model = B.GCN(in_dim, hidden, classes, n_layers=2)
    assert model.layers[0].mult_first and model.layers[1].mult_first          # 24 > 16 > 5: weight first, as DGL does
    g = B.CpuGraph(rp, col, threads=2)
    loss = torch.nn.functional.cross_entropy(model(g, X), y)
    loss.backward()

    A = _dense(rp, col)
    din = A.sum(1).clamp(min=1).pow(-0.5).view(-1, 1)
    dout = A.sum(0).clamp(min=1).pow(-0.5).view(-1, 1)
    W = [l.weight.detach().clone().requires_grad_(True) for l in model.layers]
    b = [l.bias.detach().clone().requires_grad_(True) for l in model.layers]
    h = torch.relu((A @ ((X * dout) @ W[0])) * din + b[0])
    logits = (A @ ((h * dout) @ W[1])) * din + b[1]
    ref = torch.nn.functional.cross_entropy(logits, y)
    ref.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) < 1e-5
    for k in range(2):
        assert torch.allclose(model.layers[k].weight.grad, W[k].grad, rtol=1e-4, atol=1e-6)
        assert torch.allclose(model.layers[k].bias.grad, b[k].grad, rtol=1e-4, atol=1e-6)


def test_aggregate_first_when_the_layer_widens():
    rp, col = graphs.uniform_graph(64, 4, seed=2)
    g = B.CpuGraph(rp, col, threads=1)
    torch.manual_seed(0)
    layer = B.GraphConv(8, 12)                       # in <= out: aggregate, then multiply
    assert not layer.mult_first
    X = torch.randn(64, 8)
    A = _dense(rp, col)
    din = A.sum(1).clamp(min=1).pow(-0.5).view(-1, 1); dout = A.sum(0).clamp(min=1).pow(-0.5).view(-1, 1)
    assert torch.allclose(layer(g, X), ((A @ (X * dout)) @ layer.weight) * din + layer.bias, atol=1e-5)


def test_cora_shape_training_runs_and_learns():
    """configs[0] itself, on a Cora-shaped synthetic graph (N = 2 708, nnz = 10 556, 1 433 features, 7 classes, hidden 16)."""
    rp, col = graphs.uniform_graph(2708, 10556 / 2708, seed=0)
    rng = np.random.default_rng(0)
    X = rng.standard_normal((2708, 1433)).astype(np.float32)
    y = np.ones(2708, dtype=np.int64)                # dataset.py:122: all-ones labels
    short = B.time_training(rp, col, X, y, 16, 7, epochs=1, threads=2)
    long = B.time_training(rp, col, X, y, 16, 7, epochs=12, threads=2)
    assert np.isfinite(long["final_loss"]) and long["final_loss"] < short["final_loss"]
    assert long["train_ms"] > 0
