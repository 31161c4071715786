#!/usr/bin/env python3
"""SDDMM kernel time on the Reddit-shaped graph at D = 64 / 16 / 32 / 128 under the environment given (TCGNN_SDDMM_XCD, TCGNN_RANGE_KB), with a checksum of the scores."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import numpy as np, torch
import TCGNN, tcgnn_graph as G
dev = torch.device("cuda:0")
n, nnz, _, _ = G.SHAPES["reddit"]
rp, col = G.synthetic_csr(n, nnz, seed=0, device=dev)
E = col.numel(); nw = (n + 15) // 16
bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r)
meta = (rp, col, bp, e2c, e2r)
g = torch.Generator(device=dev).manual_seed(0)
out = []
for D in (64, 16, 32, 128):
    X = torch.randn(n, D, device=dev, generator=g)
    for _ in range(3): ef = TCGNN.forward_ef(X, *meta)[0]
    TCGNN.kernel_timing(*meta, max_calls=10)
    for _ in range(10): ef = TCGNN.forward_ef(X, *meta)[0]
    km = TCGNN.kernel_timing(*meta); TCGNN.kernel_timing(*meta, max_calls=0)
    out.append("D=%d %.3f (%s) sum %.6e" % (D, float(np.mean(km)), TCGNN.last_kernel(*meta), ef.double().sum().item()))
print(" | ".join(out))
