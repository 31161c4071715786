#!/usr/bin/env python3
"""A/B of the SpMM walks (plain per-window vs range-blocked) on the Reddit-shaped graph, interleaved
in one process (kernel time from HIP events), plus an equality check between the two."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import TCGNN, tcgnn_graph as G, tcgnn_capi as c

dev = torch.device("cuda:0")
shape = sys.argv[1] if len(sys.argv) > 1 else "reddit"
n, nnz, _, _ = G.SHAPES[shape]
skew = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0   # > 0: skewed degrees (heavy windows)
rp, col = G.synthetic_csr(n, nnz, seed=0, device=dev, skew=skew)
E = col.numel(); nw = (n + 15) // 16
bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r)
meta = (rp, col, bp, e2c, e2r)
print(TCGNN.plan_info(*meta))
g = torch.Generator(device=dev).manual_seed(0)
att = torch.randn(1, E, device=dev, generator=g)
deg = (rp[1:] - rp[:-1])
print("skew %.1f: max degree %d, mean %.1f" % (skew, int(deg.max()), float(deg.float().mean())))
for D in ((64, 16, 32, 128, 41) if len(sys.argv) <= 3 else tuple(int(x) for x in sys.argv[3].split(","))):
    X = torch.randn(n, D, device=dev, generator=g)
    res = {}
    for name, fn in (("spmm", lambda: TCGNN.forward(X, *meta)[0]), ("spmm_val", lambda: TCGNN.forward_AGNN(X, rp, col, att, bp, e2c, e2r)[0])):
        outs = {}
        modes = (1, 2, 3) if name == "spmm" else (1, 2)
        times = {m: [] for m in modes}
        for rnd in range(3):
            for mode in modes:
                c.check(c.lib.tcgnn_set_spmm_mode(mode), "mode")
                fn(); TCGNN.kernel_timing(*meta, max_calls=8)
                for _ in range(8):
                    y = fn()
                times[mode] += TCGNN.kernel_timing(*meta)
                outs[mode] = y
        TCGNN.kernel_timing(*meta, max_calls=0)
        diff = (outs[1] - outs[2]).abs().max().item()
        scale = outs[1].abs().max().item()
        print("D=%3d %-8s plain %.3f ms (min %.3f)  blocked %.3f ms (min %.3f)  speed-up %.2fx   max|diff| %.2e (scale %.1f)" % (
            D, name, np.median(times[1]), np.min(times[1]), np.median(times[2]), np.min(times[2]), np.median(times[1]) / np.median(times[2]), diff, scale))
        if 3 in modes:
            print("      %-8s LDS-resident ranges %.3f ms (min %.3f)  vs blocked %.2fx   max|diff vs plain| %.2e" % (
                name, np.median(times[3]), np.min(times[3]), np.median(times[2]) / np.median(times[3]), (outs[1] - outs[3]).abs().max().item()))
c.lib.tcgnn_set_spmm_mode(0)
