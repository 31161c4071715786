#!/usr/bin/env python3
"""Sweep the column-range size of the range-blocked SpMM (TCGNN_RANGE_KB) on the Reddit-shaped graph."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import TCGNN, tcgnn_graph as G, tcgnn_capi as c
dev = torch.device("cuda:0")
n, nnz, _, _ = G.SHAPES["reddit"]
rp, col = G.synthetic_csr(n, nnz, seed=0, device=dev)
E = col.numel(); nw = (n + 15) // 16
bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r)
meta = (rp, col, bp, e2c, e2r)
g = torch.Generator(device=dev).manual_seed(0)
c.lib.tcgnn_set_spmm_mode(2)
for D in (64, 128, 32, 16):
    X = torch.randn(n, D, device=dev, generator=g)
    row = []
    for kb in (256, 512, 1024, 1536, 2048, 3072, 4096, 8192):
        os.environ["TCGNN_RANGE_KB"] = str(kb)
        TCGNN.forward(X, *meta); TCGNN.kernel_timing(*meta, max_calls=10)
        for _ in range(10): TCGNN.forward(X, *meta)
        row.append("%dK: %.3f" % (kb, np.median(TCGNN.kernel_timing(*meta))))
    print("D=%3d  " % D + "  ".join(row))
