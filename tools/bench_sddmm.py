#!/usr/bin/env python3
"""A/B of the SDDMM walks (per-window vs range-major persistent) on the Reddit-shaped graph."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
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
for D in (64, 16, 32, 128):
    X = torch.randn(n, D, device=dev, generator=g)
    res = {}
    for mode, kb in ((1, 0), (2, 1024), (2, 2048), (2, 4096), (2, 8192)):
        c.lib.tcgnn_set_spmm_mode(mode)
        if kb: os.environ["TCGNN_RANGE_KB"] = str(kb)
        y = TCGNN.forward_ef(X, *meta)[0]; TCGNN.kernel_timing(*meta, max_calls=8)
        for _ in range(8): y = TCGNN.forward_ef(X, *meta)[0]
        res[(mode, kb)] = (np.median(TCGNN.kernel_timing(*meta)), y)
    ref = res[(1, 0)][1]
    print("D=%3d  " % D + "  ".join("%s %.3f" % ("plain" if m == 1 else "b/%dK" % kb, t) for (m, kb), (t, y) in res.items()) + "  maxdiff %.1e" % max((y - ref).abs().max().item() for _, (t, y) in res.items()))
c.lib.tcgnn_set_spmm_mode(0)
