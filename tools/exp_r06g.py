#!/usr/bin/env python3
"""r06 experiment G: the slice-synchronised walk's automatic rule on SMALLER sparse community graphs than the ones it was calibrated on
(tools/exp_r06d.py): N = 400 k / 16 M edges / 200 communities of 2 k rows, N = 40 k / 1.6 M / 16, N = 1 M / 40 M / 50.  Per-window (mode 1),
forced (mode 5), automatic (mode 0); SpMM, edge-valued, SDDMM, fused forward / backward; D = 128, 64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import numpy as np, torch
import TCGNN, tcgnn_graph as G, tcgnn_capi as c
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
w = torch.tensor([0.9], device=dev)
for n, nnz, blocks in ((400_000, 16_000_000, 200), (40_000, 1_600_000, 16), (1_000_000, 40_000_000, 50)):
    rp, col = G.sbm_csr(n, nnz, seed=0, device=dev, blocks=blocks)
    E = col.numel(); nw = (n + 15) // 16
    bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
    TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r)
    meta = (rp, col, bp, e2c, e2r)
    att = torch.randn(1, E, device=dev, generator=g)
    def timed(fn, reps=10, warm=5):
        for _ in range(warm): fn()
        TCGNN.kernel_timing(*meta, max_calls=8 * reps)
        for _ in range(reps): fn()
        t = np.array(TCGNN.kernel_timing(*meta)).reshape(reps, -1).sum(1)
        TCGNN.kernel_timing(*meta, max_calls=0)
        return float(np.median(t))
    for D in (128, 64):
        X = torch.randn(n, D, device=dev, generator=g) / D ** 0.5
        res = []
        for mode in (1, 5, 0):
            c.lib.tcgnn_set_spmm_mode(mode)
            t1 = timed(lambda: TCGNN.forward(X, *meta)); k1 = TCGNN.last_kernel(*meta)
            t2 = timed(lambda: TCGNN.forward_AGNN(X, rp, col, att, bp, e2c, e2r))
            t3 = timed(lambda: TCGNN.forward_ef(X, *meta))
            t4 = timed(lambda: TCGNN.agnn_fused_forward(X, rp, col, w, bp, e2c, e2r))
            _, eff, efm = TCGNN.agnn_fused_forward(X, rp, col, w, bp, e2c, e2r)
            t5 = timed(lambda: TCGNN.agnn_fused_backward(X, rp, col, w, eff, efm, bp, e2c, e2r))
            res.append("mode %d: spmm %.3f (%s) val %.3f sddmm %.3f fused %.3f / %.3f" % (mode, t1, k1.replace("_kernel", "").replace("spmm_", ""), t2, t3, t4, t5))
        c.lib.tcgnn_set_spmm_mode(0)
        print("N %d E %d blocks %d D=%d | %s" % (n, E, blocks, D, " | ".join(res)), flush=True)
        del X
    TCGNN.clear_plan_cache(); del rp, col, bp, e2c, e2r, meta, att
    torch.cuda.empty_cache()
