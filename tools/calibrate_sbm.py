#!/usr/bin/env python3
"""Finds the in-community edge share p_in at which the 50-community SBM of the Reddit shape condenses to as many 16x8 TC blocks as
the REAL Reddit graph (13 566 510, /root/reference/logs/reduce_blocks.csv:18) - VERDICT r02 item 9: locality claims are quoted on a
graph calibrated to the real one, not on the p_in = 0.9 graph (8.23 M blocks, 39 % more condensable).  Bisection on the GPU box;
prints the p_in tcgnn_graph.SBM_REDDIT_P_IN holds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import torch
import TCGNN, tcgnn_graph as G

TARGET = 13566510
dev = torch.device("cuda:0")
n, nnz, _, _ = G.SHAPES["reddit"]

def blocks(p_in):
    rp, col = G.sbm_csr(n, nnz, seed=0, device=dev, p_in=p_in)
    E = col.numel(); nw = (n + 15) // 16
    bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
    fd = os.open(os.devnull, os.O_WRONLY); sv = os.dup(1); os.dup2(fd, 1)
    try:
        TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r)
    finally:
        os.dup2(sv, 1); os.close(fd); os.close(sv)
    return int(bp.sum().item())

lo, hi = 0.0, 0.9
for it in range(12):
    mid = 0.5 * (lo + hi)
    b = blocks(mid)
    print("p_in %.4f -> %d blocks (%+.2f %% of real Reddit)" % (mid, b, 100.0 * (b - TARGET) / TARGET), flush=True)
    if abs(b - TARGET) < 0.005 * TARGET: break
    if b > TARGET: lo = mid
    else: hi = mid
print("uniform (p_in = 0):", blocks(0.0))
