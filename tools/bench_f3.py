#!/usr/bin/env python3
"""f3 timing: GCN layer 2 of the headline model (64 -> 41 on the Reddit shape) as A (H W) [tall product + SpMM at 48 columns] against
(A H) W in one launch [tcgnn_spmm_gemm]; and whole GCN epochs with conv2 in either order."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import torch
import TCGNN, tcgnn_graph as G, tcgnn_layers as L, tcgnn_harness as H
dev = torch.device("cuda:0")
shape = sys.argv[1] if len(sys.argv) > 1 else "reddit"
n, nnz, in_dim, classes = G.SHAPES[shape]
rp, col = G.synthetic_csr(n, nnz, seed=0, device=dev)
E = col.numel()
bp = torch.zeros((n + 15) // 16, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r)
meta = (rp, col, bp, e2c, e2r)
hidden = int(sys.argv[2]) if len(sys.argv) > 2 else (64 if shape == "reddit" else 128)
Hm = torch.randn(n, hidden, device=dev); W = torch.randn(hidden, classes, device=dev) / 8
L.tune(L.tune_layers(n, [in_dim, hidden, classes]), device=dev)
def t(fn, reps=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3 / reps
print("%s  layer 2 forward (%d -> %d):  A (H W) %.3f ms   |   (A H) W fused %.3f ms   |   SpMM at %d alone %.3f ms" % (
    shape, hidden, classes, t(lambda: TCGNN.forward(L.tall_mm(Hm, W), *meta)), t(lambda: TCGNN.forward_gemm(Hm, W, *meta)), hidden, t(lambda: TCGNN.forward(Hm, *meta))), flush=True)
feats = torch.randn(n, in_dim, device=dev); labels = torch.ones(n, dtype=torch.long, device=dev)
for af in ("0", "1"):
    H.AGGREGATE_FIRST = af
    r = H.time_training("gcn", meta, feats, labels, in_dim, hidden, classes, 2, 10, seed=0)
    print("GCN epoch, conv2 aggregate_first=%s: %.3f ms" % (af, r["train_ms"]), flush=True)
