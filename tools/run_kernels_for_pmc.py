#!/usr/bin/env python3
"""Runs every kernel of the path a few times on one synthetic graph - the process rocprofv3 wraps for a --pmc pass
(tools/collect_profiles.py).  usage: run_kernels_for_pmc.py <shape> <generator> <D> [reps]
Counters are reported per kernel name, so one process serves SpMM, edge-valued SpMM, SDDMM and the fused AGNN pair."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import torch
import TCGNN, tcgnn_graph as G
shape, gen, D = sys.argv[1], sys.argv[2], int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda:0")
n, nnz, _, _ = G.SHAPES[shape]
rp, col = G.GENERATORS[gen.split("+")[0]](n, nnz, seed=0, device=dev)
if gen.endswith("+reorder"):   # the loader-side relabelling (tcgnn_graph.community_order)
    rp, col = G.permute_csr(rp, col, G.community_order(rp, col, seed=0))
E = col.numel(); nw = (n + 15) // 16
bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
fd = os.dup(1); os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r)
os.dup2(fd, 1)
meta = (rp, col, bp, e2c, e2r)
X = torch.randn(n, D, device=dev, generator=torch.Generator(device=dev).manual_seed(0)); att = torch.randn(1, E, device=dev)
w = torch.tensor([0.9], device=dev)
Xs = X / D ** 0.5
_, ef, efm = TCGNN.agnn_fused_forward(Xs, rp, col, w, bp, e2c, e2r)
for _ in range(reps):
    TCGNN.forward(X, *meta)
    TCGNN.forward_AGNN(X, rp, col, att, bp, e2c, e2r)
    TCGNN.forward_ef(X, *meta)
    TCGNN.agnn_fused_forward(Xs, rp, col, w, bp, e2c, e2r)
    TCGNN.agnn_fused_backward(Xs, rp, col, w, ef, efm, bp, e2c, e2r)
torch.cuda.synchronize()
print("E=%d N=%d D=%d" % (E, n, D))
