#!/usr/bin/env python3
"""r06 experiment F: the LDS-resident kernel FORCED (mode 3) on the ogbn-products shape with communities, against the slice-synchronised
walk (automatic) - D = 128 and 64.  BLOCKS=50,200"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import numpy as np, torch
import TCGNN, tcgnn_graph as G, tcgnn_capi as c
dev = torch.device("cuda:0")
n, nnz, _, _ = G.SHAPES["ogbn-products"]
g = torch.Generator(device=dev).manual_seed(0)
for blocks in [int(x) for x in os.environ.get("BLOCKS", "50,200").split(",")]:
    rp, col = G.sbm_csr(n, nnz, seed=0, device=dev, blocks=blocks)
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
    for D in (128, 64):
        X = torch.randn(n, D, device=dev, generator=g) / D ** 0.5
        res = []
        ref = None
        for mode in (0, 1, 3):
            c.lib.tcgnn_set_spmm_mode(mode)
            try:
                Y = TCGNN.forward(X, *meta)[0]
                t = timed(lambda: TCGNN.forward(X, *meta))
                if ref is None: ref = Y.clone()
                res.append("mode %d %.3f ms (%s) maxdiff %.1e" % (mode, t, TCGNN.last_kernel(*meta), (Y - ref).abs().max().item()))
            except Exception as exc:
                res.append("mode %d failed: %s" % (mode, str(exc)[:150]))
        c.lib.tcgnn_set_spmm_mode(0)
        print("blocks %d D=%d: %s" % (blocks, D, " | ".join(res)), flush=True)
        del X
    TCGNN.clear_plan_cache(); del rp, col, bp, e2c, e2r, meta
    torch.cuda.empty_cache()
