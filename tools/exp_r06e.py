#!/usr/bin/env python3
"""r06 experiment E: column-range size of the range-major SDDMM with XCD affinity and own-order items (Reddit shape, D = 64 / 128 / 16).
GENS=sbm_reddit,uniform  KBS=default,8192,2048,1024"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import numpy as np, torch
import TCGNN, tcgnn_graph as G
dev = torch.device("cuda:0")
n, nnz, _, _ = G.SHAPES["reddit"]
g = torch.Generator(device=dev).manual_seed(0)
for gen in os.environ.get("GENS", "sbm_reddit,uniform").split(","):
    rp, col = G.GENERATORS[gen](n, nnz, seed=0, device=dev)
    E = col.numel(); nw = (n + 15) // 16
    bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
    TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r)
    meta = (rp, col, bp, e2c, e2r)
    def timed(fn, reps=12, warm=40):
        for _ in range(warm): fn()
        TCGNN.kernel_timing(*meta, max_calls=8 * reps)
        for _ in range(reps): fn()
        t = np.array(TCGNN.kernel_timing(*meta)).reshape(reps, -1).sum(1)
        TCGNN.kernel_timing(*meta, max_calls=0)
        return float(np.median(t))
    for D in [int(x) for x in os.environ.get("DIMS", "64,128,16").split(",")]:
        X = torch.randn(n, D, device=dev, generator=g) / D ** 0.5
        res = []
        for kb in os.environ.get("KBS", "default,8192,2048,1024").split(","):
            if kb == "default": os.environ.pop("TCGNN_RANGE_KB", None)
            else: os.environ["TCGNN_RANGE_KB"] = kb
            res.append("%s: %.3f" % (kb, timed(lambda: TCGNN.forward_ef(X, *meta))))
        os.environ.pop("TCGNN_RANGE_KB", None)
        print("%s D=%d sddmm by range KB: %s" % (gen, D, "  ".join(res)), flush=True)
        del X
    TCGNN.clear_plan_cache(); del rp, col, bp, e2c, e2r, meta
    torch.cuda.empty_cache()
