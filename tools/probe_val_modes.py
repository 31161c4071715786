#!/usr/bin/env python3
"""Edge-valued SpMM, SDDMM and the fused AGNN forward under the per-window (1) and range-blocked (2) gather walks and the automatic
choice (0), on one synthetic graph.  usage: probe_val_modes.py <shape> <generator> <D>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import torch, TCGNN, tcgnn_graph as G, tcgnn_capi as c
dev = torch.device("cuda:0")
shape, gen, D = sys.argv[1], sys.argv[2], int(sys.argv[3])
n, nnz, _, _ = G.SHAPES[shape]
rp, col = G.GENERATORS[gen](n, nnz, seed=0, device=dev)
E = col.numel(); nw = (n + 15) // 16
bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
fd = os.dup(1); os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r)
os.dup2(fd, 1)
meta = (rp, col, bp, e2c, e2r)
X = torch.randn(n, D, device=dev); att = torch.randn(1, E, device=dev); w = torch.tensor([0.9], device=dev); Xs = X / D ** 0.5
ops = {"spmm": lambda: TCGNN.forward(X, *meta), "spmm_val": lambda: TCGNN.forward_AGNN(X, rp, col, att, bp, e2c, e2r), "sddmm": lambda: TCGNN.forward_ef(X, *meta),
       "agnn_fwd": lambda: TCGNN.agnn_fused_forward(Xs, rp, col, w, bp, e2c, e2r)}
for name, fn in ops.items():
    for mode in (0, 1, 2):
        c.check(c.lib.tcgnn_set_spmm_mode(mode), "mode")
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize()
        print("%s %s D %d  %-9s mode %d: %.3f ms per call (%s)" % (shape, gen, D, name, mode, (time.perf_counter() - t0) * 100, TCGNN.last_kernel(*meta)), flush=True)
c.lib.tcgnn_set_spmm_mode(0)
