#!/usr/bin/env python3
"""Three other dense shapes (N = 100 k / 60 k / 400 k), D = 64 and 16: the automatic walk against the forced per-window (1) and
range-blocked (2) walks for SpMM, SDDMM and the fused AGNN forward, with the differences between their results."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import numpy as np, torch
import TCGNN, tcgnn_graph as G, tcgnn_capi as c
dev = torch.device("cuda:0")
for n, nnz in ((100000, 80_000_000), (60000, 20_000_000), (400000, 120_000_000)):
    rp, col = G.synthetic_csr(n, nnz, seed=3, device=dev, skew=0.4)
    E = col.numel(); nw = (n + 15) // 16
    bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
    fd = os.open(os.devnull, os.O_WRONLY); sv = os.dup(1); os.dup2(fd, 1)
    TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r)
    os.dup2(sv, 1)
    meta = (rp, col, bp, e2c, e2r)
    info = TCGNN.plan_info(*meta)
    for D in (64, 16):
        X = torch.randn(n, D, device=dev)
        out = {}
        for mode in (0, 1, 2):
            c.lib.tcgnn_set_spmm_mode(mode)
            TCGNN.forward(X, *meta); TCGNN.kernel_timing(*meta, max_calls=6)
            for _ in range(6): y = TCGNN.forward(X, *meta)[0]
            out[mode] = (np.median(TCGNN.kernel_timing(*meta)), y)
        c.lib.tcgnn_set_spmm_mode(0)
        TCGNN.kernel_timing(*meta, max_calls=0)
        deg = (rp[1:] - rp[:-1]).float()
        scale = deg.sqrt()[:, None] + 1
        print("N=%d nnz=%d D=%d lds_ranges=%d buckets=%d : auto %.3f ms, plain %.3f, blocked %.3f | max diff auto-plain %.2e auto-blocked %.2e (scaled)" % (
            n, E, D, info["lds_ranges"], info["column_buckets"], out[0][0], out[1][0], out[2][0],
            ((out[0][1] - out[1][1]).abs() / scale).max().item(), ((out[0][1] - out[2][1]).abs() / scale).max().item()), flush=True)
        # SDDMM and the fused AGNN forward: automatic choice against the two forced walks
        for name, fn in (("sddmm", lambda: TCGNN.forward_ef(X, *meta)[0]), ("agnn_fwd", lambda: TCGNN.agnn_fused_forward(X / D ** 0.5, rp, col, torch.tensor([0.9], device=dev), bp, e2c, e2r)[0])):
            tt = {}
            for mode in (0, 1, 2):
                c.lib.tcgnn_set_spmm_mode(mode)
                fn(); TCGNN.kernel_timing(*meta, max_calls=6)
                for _ in range(6): fn()
                tt[mode] = np.median(TCGNN.kernel_timing(*meta))
            c.lib.tcgnn_set_spmm_mode(0); TCGNN.kernel_timing(*meta, max_calls=0)
            print("   %-8s D=%d: auto %.3f ms, per-window %.3f, range-major %.3f" % (name, D, tt[0], tt[1], tt[2]), flush=True)
    TCGNN.clear_plan_cache()
