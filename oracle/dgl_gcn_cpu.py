"""CPU restatement of the reference's DGL GCN baseline - TEST / BASELINE INFRASTRUCTURE, never the product path
(only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import anything under oracle/).

What it follows: /root/reference/dgl_baseline/gcn.py:14-37 (GraphConv stack: in -> hidden with the activation, hidden ->
classes without) and dgl_baseline/train.py:57-88 (CrossEntropyLoss, Adam(lr=1e-2, weight_decay=5e-4), 3 forward-only dry runs,
then timed epochs).  The arithmetic of GraphConv lives in the third-party package DGL (dgl.nn.pytorch.GraphConv), which is not
in /root/reference and not installed here; version unpinned by the reference (docker/dockerfile:23 `conda install -c dglteam
dgl-cuda11.6`, README.md:49 says dgl-cuda11.0).  Its published algorithm, restated:
    GraphConv(in, out, norm='both', weight=True, bias=True, activation=a, allow_zero_in_degree=True)(g, h) =
        a( D_in^-1/2 * A * D_out^-1/2 * h * W + b ),  degrees clamped to >= 1,
        h*W applied BEFORE the aggregation when in_feats > out_feats, after it otherwise;
    weights Xavier-uniform, bias zeros.
A[r, c] = 1 for CSR entry (r, c): row r aggregates from its columns (in-degree = row length, out-degree = column count).
No reference test pins DGL's results: PARITY UNPINNED - this file is only ever timed, and checked against a dense-matrix
evaluation of the same formula (tests/test_dgl_cpu_baseline.py).
"""
import time

import numpy as np
import torch

from . import oracle as O


class _Aggregate(torch.autograd.Function):
    """Y = A X on the host cores (oracle_csr_spmm, OpenMP); backward dX = A^T dY through the transposed CSR."""

    @staticmethod
    def forward(ctx, X, graph):
        ctx.graph = graph
        return torch.from_numpy(O.csr_spmm(np.ascontiguousarray(X.detach().numpy(), dtype=np.float32), graph.rp, graph.col, threads=graph.threads))

    @staticmethod
    def backward(ctx, dY):
        g = ctx.graph
        return torch.from_numpy(O.csr_spmm(np.ascontiguousarray(dY.numpy(), dtype=np.float32), g.rp_t, g.col_t, threads=g.threads)), None


class CpuGraph:
    def __init__(self, rowptr, col, threads=0, symmetric=False):
        """symmetric=True: the caller vouches that A = A^T (bench.py's graphs are built that way): the transposed CSR is A itself."""
        self.rp = np.ascontiguousarray(rowptr, dtype=np.int32)
        self.col = np.ascontiguousarray(col, dtype=np.int32)
        n = len(self.rp) - 1
        self.n, self.threads = n, threads
        if symmetric:
            self.rp_t, self.col_t = self.rp, self.col
        else:
            import scipy.sparse as sp
            at = sp.csr_matrix((np.ones(len(self.col), np.float32), self.col, self.rp), shape=(n, n)).T.tocsr()
            self.rp_t, self.col_t = at.indptr.astype(np.int32), at.indices.astype(np.int32)
        in_deg = np.maximum(np.diff(self.rp), 1).astype(np.float32)
        out_deg = np.maximum(np.diff(self.rp_t), 1).astype(np.float32)
        self.norm_in = torch.from_numpy(in_deg ** -0.5).view(-1, 1)
        self.norm_out = torch.from_numpy(out_deg ** -0.5).view(-1, 1)


class GraphConv(torch.nn.Module):
    def __init__(self, in_feats, out_feats, activation=None):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.empty(in_feats, out_feats))
        self.bias = torch.nn.Parameter(torch.zeros(out_feats))
        torch.nn.init.xavier_uniform_(self.weight)
        self.activation, self.mult_first = activation, in_feats > out_feats

    def forward(self, graph, h):
        h = h * graph.norm_out
        if self.mult_first:
            h = _Aggregate.apply(torch.mm(h, self.weight), graph)
        else:
            h = torch.mm(_Aggregate.apply(h, graph), self.weight)
        h = h * graph.norm_in + self.bias
        return self.activation(h) if self.activation is not None else h


class GCN(torch.nn.Module):
    def __init__(self, in_feats, n_hidden, n_classes, n_layers=2):
        super().__init__()
        dims = [in_feats] + [n_hidden] * (n_layers - 1)
        self.layers = torch.nn.ModuleList(GraphConv(a, b, activation=torch.relu) for a, b in zip(dims[:-1], dims[1:]))
        self.layers.append(GraphConv(dims[-1], n_classes))

    def forward(self, graph, features):
        h = features
        for layer in self.layers:
            h = layer(graph, h)
        return h


def time_training(rowptr, col, features, labels, n_hidden, n_classes, epochs, n_layers=2, dry_runs=3, seed=0, threads=0, symmetric=False):
    """dgl_baseline/train.py:57-88 on CPU tensors.  Returns ms per epoch and the final loss."""
    torch.manual_seed(seed)
    torch.set_num_threads(threads or O.num_threads())
    graph = CpuGraph(rowptr, col, threads=threads, symmetric=symmetric)
    x = torch.as_tensor(features, dtype=torch.float32)
    y = torch.as_tensor(labels, dtype=torch.long)
    model = GCN(x.shape[1], n_hidden, n_classes, n_layers)
    loss_fn = torch.nn.CrossEntropyLoss()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2, weight_decay=5e-4)
    model.train()
    with torch.no_grad():
        for _ in range(dry_runs):
            model(graph, x)
    t0 = time.perf_counter()
    loss = torch.tensor(float("nan"))
    for _ in range(epochs):
        logits = model(graph, x)
        loss = loss_fn(logits, y)
        opt.zero_grad()
        loss.backward()
        opt.step()
    return {"train_ms": (time.perf_counter() - t0) * 1e3 / max(epochs, 1), "final_loss": float(loss.detach()), "threads": threads or O.num_threads()}
