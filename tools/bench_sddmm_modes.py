#!/usr/bin/env python3
"""SDDMM and the fused AGNN forward on the Reddit shape: automatic walk against the per-window (1) and range-major (2) walks."""
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
for D in (64, 16, 128):
    X = torch.randn(n, D, device=dev)
    for name, fn in (("sddmm", lambda: TCGNN.forward_ef(X, *meta)[0]), ("agnn_fwd", lambda: TCGNN.agnn_fused_forward(X / D ** 0.5, rp, col, torch.tensor([0.9], device=dev), bp, e2c, e2r)[0])):
        tt = {}
        for mode in (0, 1, 2):
            c.lib.tcgnn_set_spmm_mode(mode)
            fn(); TCGNN.kernel_timing(*meta, max_calls=6)
            for _ in range(6): fn()
            tt[mode] = np.median(TCGNN.kernel_timing(*meta))
        c.lib.tcgnn_set_spmm_mode(0); TCGNN.kernel_timing(*meta, max_calls=0)
        print("%-8s D=%d: auto %.3f ms, per-window %.3f, range-major %.3f" % (name, D, tt[0], tt[1], tt[2]), flush=True)
