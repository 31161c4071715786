#!/usr/bin/env python3
"""r06 experiment C: what the out-of-community edges cost the slice-synchronised walk.  ogbn-products shape from the SBM generator with
p_in = 0.9 (the bench graph) / 0.97 / 1.0: SpMM D = 128, per-window against slice-synchronised (kernel ms from the library's events)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import numpy as np, torch
import TCGNN, tcgnn_graph as G, tcgnn_capi as c
dev = torch.device("cuda:0")
n, nnz, _, _ = G.SHAPES["ogbn-products"]
g = torch.Generator(device=dev).manual_seed(0)
D = 128
X = torch.randn(n, D, device=dev, generator=g) / D ** 0.5
for p_in in [float(x) for x in os.environ.get("PINS", "1.0,0.97,0.9").split(",")]:
    rp, col = G.sbm_csr(n, nnz, seed=0, device=dev, p_in=p_in)
    E = col.numel(); nw = (n + 15) // 16
    bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
    TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r)
    meta = (rp, col, bp, e2c, e2r)
    def timed(fn, reps=8, warm=3):
        for _ in range(warm): fn()
        TCGNN.kernel_timing(*meta, max_calls=8 * reps)
        for _ in range(reps): fn()
        t = np.array(TCGNN.kernel_timing(*meta)).reshape(reps, -1).sum(1)
        TCGNN.kernel_timing(*meta, max_calls=0)
        return float(np.median(t))
    info = TCGNN.plan_info(*meta)
    res = []
    for mode in (1, 5):
        c.lib.tcgnn_set_spmm_mode(mode)
        res.append((mode, timed(lambda: TCGNN.forward(X, *meta)), TCGNN.last_kernel(*meta), timed(lambda: TCGNN.forward_ef(X, *meta))))
    c.lib.tcgnn_set_spmm_mode(0)
    print("p_in %.2f: E %d tc_blocks %d | " % (p_in, E, info["tc_blocks"]) + " | ".join("mode %d spmm %.3f (%s) sddmm %.3f" % r for r in res), flush=True)
    TCGNN.clear_plan_cache(); del rp, col, bp, e2c, e2r, meta
    torch.cuda.empty_cache()
