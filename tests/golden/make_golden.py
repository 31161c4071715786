#!/usr/bin/env python3
"""Generates the golden fixtures in tests/golden/ from the REFERENCE ITSELF.  Runs only where
/root/reference exists (the build container); the fixtures it writes are plain data (.npz) and are
committed, this script is committed, nothing of the reference's text is.

  sgt_*.npz       inputs + outputs of the reference's own `preprocess` (the unmodified
                  TCGNN_conv/TCGNN.cpp compiled by oracle/build_ref.sh), on graphs that hit its
                  quirks: N % 16 == 0 (one-past-the-end write, observed through guard slots),
                  an edgeless window, N = 1, a 1000-node graph.  `tc_blocks` is parsed from the
                  "TC_Blocks:" line the reference prints (TCGNN.cpp:225).
  layers_*.npz    forward/backward values of the reference's autograd Functions (gnn_conv.py:26-158)
                  imported here with `TCGNN` bound to a CPU shim: preprocess = the compiled
                  reference, forward/forward_ef/forward_AGNN = the oracle restatement in TF32 mode.
                  Pins the COMPOSITION (which operator, operand order, A vs A^T, which gradients),
                  not kernel arithmetic.
  dataset_*.npz   the reference's dataset.py (dataset.py:69-104) run on a synthetic edge list with
                  duplicates and self loops: CSR arrays + the raw-vs-nnz num_edges discrepancy.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("TCGNN_REFERENCE_DIR", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import graphs  # noqa: E402
from oracle import oracle as O  # noqa: E402


def capture_stdout_fd(fn):
    """Run fn() while C-level stdout (fd 1) goes to a temp file; return (result, text)."""
    sys.stdout.flush()
    saved = os.dup(1)
    with tempfile.TemporaryFile(mode="w+b") as tmp:
        os.dup2(tmp.fileno(), 1)
        try:
            res = fn()
        finally:
            import ctypes
            ctypes.CDLL(None).fflush(None)
            os.dup2(saved, 1)
            os.close(saved)
        tmp.seek(0)
        return res, tmp.read().decode()


def ref_preprocess(ref, rowptr, col, guard):
    n = len(rowptr) - 1
    nw = (n + 15) // 16
    bp = torch.full((nw + guard,), -7, dtype=torch.int32)
    e2c = torch.zeros(len(col), dtype=torch.int32)
    e2r = torch.zeros(len(col), dtype=torch.int32)
    _, text = capture_stdout_fd(lambda: ref.preprocess(torch.from_numpy(col), torch.from_numpy(rowptr), n, 16, 8, bp, e2c, e2r))
    tc = int(text.split("TC_Blocks:")[1].split()[0])
    return bp.numpy(), e2c.numpy(), e2r.numpy(), tc


def make_sgt(ref):
    for name, rp, col in graphs.edge_case_graphs():
        bp, e2c, e2r, tc = ref_preprocess(ref, rp, col, guard=3)
        np.savez_compressed(os.path.join(HERE, "sgt_%s.npz" % name), rowptr=rp, col=col, bp_with_guard=bp, guard=3,
                            e2c=e2c, e2r=e2r, tc_blocks=tc)
        print("sgt_%s: N=%d nnz=%d TC_Blocks=%d bp(tail)=%s" % (name, len(rp) - 1, len(col), tc, bp[-4:]))


class Shim(types.ModuleType):
    """CPU stand-in for the extension module, used ONLY to drive the reference's Python callers."""

    def __init__(self, ref):
        super().__init__("TCGNN")
        self.preprocess = ref.preprocess

    @staticmethod
    def _np(*ts):
        return [t.detach().cpu().numpy() for t in ts]

    def forward(self, X, rp, col, bp, e2c, e2r):
        x, rp, col, bp, e2c, e2r = self._np(X, rp, col, bp, e2c, e2r)
        return [torch.from_numpy(O.spmm(x, rp, col, bp, e2c, e2r, round_mode=O.ROUND_TF32))]

    def forward_ef(self, X, rp, col, bp, e2c, e2r):
        x, rp, col, bp, e2c, e2r = self._np(X, rp, col, bp, e2c, e2r)
        return [torch.from_numpy(O.sddmm(x, rp, col, bp, e2c, e2r, round_mode=O.ROUND_TF32))]

    def forward_AGNN(self, X, rp, col, att, bp, e2c, e2r):
        x, rp, col, att, bp, e2c, e2r = self._np(X, rp, col, att, bp, e2c, e2r)
        return [torch.from_numpy(O.spmm_val(x, rp, col, att[0], bp, e2c, e2r, round_mode=O.ROUND_TF32))]


def make_layers(ref):
    sys.modules["TCGNN"] = Shim(ref)
    sys.path.insert(0, REF)
    import gnn_conv as G  # the reference's layer library, imported in place, never copied
    rp, col = graphs.uniform_graph(200, 6, seed=21, symmetric=True)
    n, nnz = len(rp) - 1, len(col)
    nw = (n + 15) // 16
    bp = torch.zeros(nw, dtype=torch.int32); e2c = torch.zeros(nnz, dtype=torch.int32); e2r = torch.zeros(nnz, dtype=torch.int32)
    capture_stdout_fd(lambda: ref.preprocess(torch.from_numpy(col), torch.from_numpy(rp), n, 16, 8, bp, e2c, e2r))
    trp, tcol = torch.from_numpy(rp), torch.from_numpy(col)
    g = torch.Generator().manual_seed(1234)
    din, dout = 48, 32
    X = torch.randn(n, din, generator=g)
    W = torch.randn(din, dout, generator=g) * 0.2
    Xs = torch.randn(n, dout, generator=g)
    dY = torch.randn(n, dout, generator=g)
    aw = torch.randn(1, 1, generator=g)
    out = dict(rowptr=rp, col=col, bp=bp.numpy(), e2c=e2c.numpy(), e2r=e2r.numpy(), X=X.numpy(), W=W.numpy(), Xs=Xs.numpy(),
               dY=dY.numpy(), attention_w=aw.numpy())
    meta = (trp, tcol, bp, e2c, e2r)

    x = Xs.clone().requires_grad_(True)
    y = G.TCGNNFunction_SAG.apply(x, *meta); y.backward(dY)
    out.update(sag_Y=y.detach().numpy(), sag_dX=x.grad.numpy())

    x, w = X.clone().requires_grad_(True), W.clone().requires_grad_(True)
    y = G.TCGNNFunction.apply(x, w, *meta); y.backward(dY)
    out.update(gcn_Y=y.detach().numpy(), gcn_dX=x.grad.numpy(), gcn_dW=w.grad.numpy())

    x, w = X.clone().requires_grad_(True), W.clone().requires_grad_(True)
    y = G.TCGNNFunction_GIN.apply(x, w, *meta); y.backward(dY)
    out.update(gin_Y=y.detach().numpy(), gin_dX=x.grad.numpy(), gin_dW=w.grad.numpy())

    x, w, a = X.clone().requires_grad_(True), W.clone().requires_grad_(True), aw.clone().requires_grad_(True)
    y = G.TCGNNFunction_AGNN.apply(x, w, a, *meta); y.backward(dY)
    out.update(agnn_Y=y.detach().numpy(), agnn_dX=x.grad.numpy(), agnn_dW=w.grad.numpy(), agnn_dattention_w=a.grad.numpy())

    np.savez_compressed(os.path.join(HERE, "layers_n200.npz"), **out)
    print("layers_n200: N=%d nnz=%d keys=%d" % (n, nnz, len(out)))
    # module-level facts a mirror must keep (gnn_conv.py:10, :195, :215, :231-237)
    torch.manual_seed(0)
    conv = G.AGNNConv(8, 4)
    np.savez_compressed(os.path.join(HERE, "layers_facts.npz"), n_heads=G.n_heads,
                        agnn_weight_absmax=float(conv.weights.abs().max()), agnn_weight_bound=1.0 / np.sqrt(4),
                        agnn_attention_shape=np.array(conv.attention_w.shape),
                        gcn_weight_shape=np.array(G.GCNConv(8, 4).weights.shape))


def make_dataset():
    sys.path.insert(0, REF)
    saved = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self  # dataset.py calls .cuda() unconditionally
    try:
        import dataset as D
        rng = np.random.default_rng(5)
        n, m = 60, 400
        src = rng.integers(0, n, size=m); dst = rng.integers(0, n, size=m)
        src[:20] = src[20:40]; dst[:20] = dst[20:40]  # duplicates
        src[40:45] = dst[40:45]                       # self loops
        path = os.path.join(tempfile.mkdtemp(), "toy.npz")
        np.savez(path, src_li=src, dst_li=dst, num_nodes=n)
        ds = D.TCGNN_dataset(path, 12, 5, load_from_txt=False)
        np.savez_compressed(os.path.join(HERE, "dataset_toy.npz"), src_li=src, dst_li=dst, num_nodes=n,
                            row_pointers=ds.row_pointers.numpy(), column_index=ds.column_index.numpy(),
                            num_edges=ds.num_edges, x_shape=np.array(ds.x.shape), y=ds.y.numpy(),
                            num_features=ds.num_features, num_classes=ds.num_classes)
        print("dataset_toy: raw num_edges=%d nnz=%d" % (ds.num_edges, ds.column_index.numel()))
    finally:
        torch.Tensor.cuda = saved


def make_tile_stats():
    """Runs find_dense() of 3_cnt_TC_blk_SpMM.py (16x8) and 3_cnt_TC_blk_SDDMM.py (16x16) on seeded graphs.
    The scripts execute their dataset loop at import, so only the function definition is compiled (from the
    reference checkout, at generation time) into a namespace that carries the script's tile constants."""
    import ast, contextlib, io
    out = {}
    graphs = {}
    rng = np.random.default_rng(9)
    for name, n, m in (("a", 50, 300), ("b", 200, 4000), ("c", 333, 900)):
        src = rng.integers(0, n, size=m); dst = rng.integers(0, n, size=m)
        if name == "c":   # clustered ids: sliding tiles cover several neighbours
            src = (src // 7) * 7 + rng.integers(0, 3, size=m); src = np.minimum(src, n - 1)
        key = np.unique(dst.astype(np.int64) * n + src)       # the scripts exit on duplicate edges
        graphs[name] = (key % n, key // n, n)
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.chdir(tmp)
    try:
        for script, tag in (("3_cnt_TC_blk_SpMM.py", "16x8"), ("3_cnt_TC_blk_SDDMM.py", "16x16")):
            tree = ast.parse(open(os.path.join(REF, script)).read())
            ns = {"np": np, "defaultdict": __import__("collections").defaultdict, "sys": sys}
            for node in tree.body:
                if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", "").startswith("dense_tile_"):
                    exec(compile(ast.Module([node], []), script, "exec"), ns)
                if isinstance(node, ast.FunctionDef) and node.name == "find_dense":
                    exec(compile(ast.Module([node], []), script, "exec"), ns)
            for name, (src, dst, n) in graphs.items():
                path = os.path.join(tmp, name)
                np.savez(path + ".npz", src_li=src, dst_li=dst, num_nodes=n)
                buf = io.StringIO()
                with contextlib.redirect_stdout(buf):
                    ns["find_dense"](path, name)
                _, origin, reduced, pct = buf.getvalue().strip().split(",")
                out["%s_%s" % (name, tag)] = np.array([int(origin), int(reduced)])
                out["%s_%s_pct" % (name, tag)] = pct
                out["%s_tile" % tag] = np.array([ns["dense_tile_H"], ns["dense_tile_W"]])
    finally:
        os.chdir(cwd)
    for name, (src, dst, n) in graphs.items():
        out[name + "_src"] = src.astype(np.int32); out[name + "_dst"] = dst.astype(np.int32); out[name + "_n"] = n
    np.savez_compressed(os.path.join(HERE, "tile_stats.npz"), **out)
    print("tile_stats:", {k: v.tolist() for k, v in out.items() if k.endswith("x8") or k.endswith("x16")})


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference checkout not found at %s - fixtures can only be generated in the build container" % REF)
    O.build()
    os.system(os.path.join(ROOT, "oracle", "build_ref.sh") + " > /dev/null")
    ref = O.load_ref()
    make_sgt(ref)
    make_layers(ref)
    make_dataset()
    make_tile_stats()
