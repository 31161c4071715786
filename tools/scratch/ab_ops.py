#!/usr/bin/env python3
"""Steady-state kernel time (the library's own HIP events) of every operator on one Reddit-shaped graph; run once per library
(TCGNN_LIB_PATH) on ONE box and compare.   GEN=uniform|sbm_reddit|rmat DIMS=64,41 python tools/scratch/ab_ops.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import numpy as np, torch
import TCGNN, tcgnn_graph as G
dev = torch.device("cuda:0")
shape = os.environ.get("SHAPE", "reddit")
n, nnz, _, _ = G.SHAPES[shape]
rp, col = G.GENERATORS[os.environ.get("GEN", "uniform")](n, nnz, seed=0, device=dev)
E = col.numel(); nw = (n + 15) // 16
bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
fd = os.open(os.devnull, os.O_WRONLY); sv = os.dup(1); os.dup2(fd, 1)
TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r)
os.dup2(sv, 1)
meta = (rp, col, bp, e2c, e2r)
g = torch.Generator(device=dev).manual_seed(0)
w = torch.tensor([0.9], device=dev)
def timed(fn, reps=20):
    fn(); fn()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.08:
        for _ in range(8): fn()
        torch.cuda.synchronize()
    TCGNN.kernel_timing(*meta, max_calls=8 * reps)
    for _ in range(reps): fn()
    t = np.array(TCGNN.kernel_timing(*meta)); TCGNN.kernel_timing(*meta, max_calls=0)
    return float(t.reshape(reps, -1).sum(1).mean()), TCGNN.last_kernel(*meta)
out = []
for D in tuple(int(x) for x in os.environ.get("DIMS", "64").split(",")):
    X = torch.randn(n, D, device=dev, generator=g) / D ** 0.5
    dY = torch.randn(n, D, device=dev, generator=g)
    att = torch.randn(1, E, device=dev, generator=g)
    Yf, ef, efm = TCGNN.agnn_fused_forward(X, rp, col, w, bp, e2c, e2r)
    legs = [("spmm", lambda: TCGNN.forward(X, *meta)), ("spmm_val", lambda: TCGNN.forward_AGNN(X, rp, col, att, bp, e2c, e2r)),
            ("sddmm", lambda: TCGNN.forward_ef(X, *meta)), ("fused_fwd", lambda: TCGNN.agnn_fused_forward(X, rp, col, w, bp, e2c, e2r)),
            ("fused_bwd", lambda: TCGNN.agnn_fused_backward(dY, rp, col, w, ef, efm, bp, e2c, e2r))]
    for name, fn in legs:
        ms, k = timed(fn)
        out.append("%s D=%d %.3f (%s)" % (name, D, ms, k[:40]))
print(os.environ.get("GEN", "uniform"), shape, " | ".join(out))
