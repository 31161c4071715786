#!/usr/bin/env python3
"""Runs one of the three kernels a few times on the Reddit-shaped graph (for rocprofv3 --pmc passes).
usage: run_kernel_once.py {spmm|spmm_val|sddmm|agnn_fwd|agnn_bwd} [D] [mode]   (TCGNN_PROFILE_SHAPE=reddit|ogbn-products, TCGNN_PROFILE_GEN=uniform|sbm|sbm_reddit|rmat)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import torch
import TCGNN, tcgnn_graph as G, tcgnn_capi as c
which = sys.argv[1]; D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
if len(sys.argv) > 3: c.lib.tcgnn_set_spmm_mode(int(sys.argv[3]))
dev = torch.device("cuda:0")
n, nnz, _, _ = G.SHAPES[os.environ.get("TCGNN_PROFILE_SHAPE", "reddit")]
rp, col = G.GENERATORS[os.environ.get("TCGNN_PROFILE_GEN", "uniform")](n, nnz, seed=0, device=dev)
E = col.numel(); nw = (n + 15) // 16
bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r)
X = torch.randn(n, D, device=dev); att = torch.randn(1, E, device=dev)
w = torch.tensor([0.9], device=dev)
if which.startswith("agnn"):
    X = X / D ** 0.5
    _, ef, efm = TCGNN.agnn_fused_forward(X, rp, col, w, bp, e2c, e2r)
for _ in range(4):
    if which == "agnn_fwd": TCGNN.agnn_fused_forward(X, rp, col, w, bp, e2c, e2r)
    elif which == "agnn_bwd": TCGNN.agnn_fused_backward(X, rp, col, w, ef, efm, bp, e2c, e2r)
    elif which == "spmm": TCGNN.forward(X, rp, col, bp, e2c, e2r)
    elif which == "spmm_val": TCGNN.forward_AGNN(X, rp, col, att, bp, e2c, e2r)
    else: TCGNN.forward_ef(X, rp, col, bp, e2c, e2r)
torch.cuda.synchronize()
