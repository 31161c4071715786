#!/usr/bin/env python3
"""r06 experiment D: where the slice-synchronised walk starts to pay - ogbn-products shape, SBM with 25 / 50 / 100 / 200 communities
(98 k / 49 k / 24 k / 12 k rows each: 25 / 12.5 / 6.3 / 3.1 MB of fp16 image at D = 128), per-window (mode 1) against forced (mode 5) and
what the automatic mode takes.  BLOCKS=25,50,100,200 DIMS=128"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import numpy as np, torch
import TCGNN, tcgnn_graph as G, tcgnn_capi as c
dev = torch.device("cuda:0")
n, nnz, _, _ = G.SHAPES["ogbn-products"]
g = torch.Generator(device=dev).manual_seed(0)
for blocks in [int(x) for x in os.environ.get("BLOCKS", "25,50,100,200").split(",")]:
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
    att = torch.randn(1, E, device=dev, generator=g); w = torch.tensor([0.9], device=dev)
    for D in [int(x) for x in os.environ.get("DIMS", "128").split(",")]:
        X = torch.randn(n, D, device=dev, generator=g) / D ** 0.5
        res = []
        for mode in (1, 5, 0):
            c.lib.tcgnn_set_spmm_mode(mode)
            t1 = timed(lambda: TCGNN.forward(X, *meta)); k1 = TCGNN.last_kernel(*meta)
            t2 = timed(lambda: TCGNN.forward_ef(X, *meta)); k2 = TCGNN.last_kernel(*meta)
            t3 = timed(lambda: TCGNN.forward_AGNN(X, rp, col, att, bp, e2c, e2r))
            t4 = timed(lambda: TCGNN.agnn_fused_forward(X, rp, col, w, bp, e2c, e2r))
            _, eff, efm = TCGNN.agnn_fused_forward(X, rp, col, w, bp, e2c, e2r)
            t5 = timed(lambda: TCGNN.agnn_fused_backward(X, rp, col, w, eff, efm, bp, e2c, e2r))
            res.append("mode %d spmm %.3f (%s) sddmm %.3f (%s) val %.3f fused %.3f / %.3f" % (mode, t1, k1.replace("_kernel", ""), t2, k2.replace("_kernel", ""), t3, t4, t5))
        c.lib.tcgnn_set_spmm_mode(0)
        print("blocks %3d D=%d: " % (blocks, D) + " | ".join(res), flush=True)
        del X
    TCGNN.clear_plan_cache(); del rp, col, bp, e2c, e2r, meta
    torch.cuda.empty_cache()
