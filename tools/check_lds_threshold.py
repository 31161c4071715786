#!/usr/bin/env python3
"""Where does the LDS-resident SpMM stop paying?  Automatic choice against the forced walks (1 per-window, 2 range-blocked,
3 LDS-resident) over graphs of falling density, D = 64 and 16."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import numpy as np, torch
import TCGNN, tcgnn_graph as G, tcgnn_capi as c
dev = torch.device("cuda:0")
shapes = [(232965, 60_000_000), (232965, 30_000_000), (232965, 15_000_000), (100000, 80_000_000), (100000, 20_000_000), (60000, 20_000_000),
          (400000, 120_000_000), (400000, 60_000_000), (1000000, 120_000_000)]
for n, nnz in shapes:
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
        for mode in (0, 1, 2, 3):
            c.lib.tcgnn_set_spmm_mode(mode)
            try:
                TCGNN.forward(X, *meta); TCGNN.kernel_timing(*meta, max_calls=6)
                for _ in range(6): TCGNN.forward(X, *meta)
                out[mode] = float(np.median(TCGNN.kernel_timing(*meta)))
            except Exception as ex:
                out[mode] = float("nan")
        c.lib.tcgnn_set_spmm_mode(0); TCGNN.kernel_timing(*meta, max_calls=0)
        best = min((v, k) for k, v in out.items() if k and v == v)
        print("N=%7d nnz=%9d deg %5.0f D=%2d auto-lds=%d buckets=%2d : auto %.3f | plain %.3f blocked %.3f lds %.3f  -> best mode %d%s" % (
            n, E, E / n, D, int(info["lds_ranges"] > 0), info["column_buckets"], out[0], out[1], out[2], out[3], best[1],
            "" if out[0] <= 1.08 * best[0] else "   <-- auto is %.2fx the best" % (out[0] / best[0])), flush=True)
    del rp, col, bp, e2c, e2r, X; torch.cuda.empty_cache()
