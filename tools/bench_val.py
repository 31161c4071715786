#!/usr/bin/env python3
"""forward_AGNN (edge-valued SpMM) on a Reddit-shaped graph: the automatic walk (r04: the LDS-resident flat walk of
tcgnn_lds_val.inc once its stream exists) against the forced gather walks; kernel time from HIP events (permutation pass +
main kernel + cold remainder) and per-call wall time.   python tools/bench_val.py [shape] [generator] [D,D,...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import TCGNN, tcgnn_graph as G, tcgnn_capi as c

dev = torch.device("cuda:0")
shape = sys.argv[1] if len(sys.argv) > 1 else "reddit"
gen = sys.argv[2] if len(sys.argv) > 2 else "uniform"
dims = tuple(int(x) for x in sys.argv[3].split(",")) if len(sys.argv) > 3 else (64, 128)
n, nnz, _, _ = G.SHAPES[shape]
rp, col = G.GENERATORS[gen](n, nnz, seed=0, device=dev)
E = col.numel(); nw = (n + 15) // 16
bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r)
meta = (rp, col, bp, e2c, e2r)
g = torch.Generator(device=dev).manual_seed(0)
att = torch.randn(1, E, device=dev, generator=g)
for D in dims:
    X = torch.randn(n, D, device=dev, generator=g)
    fn = lambda: TCGNN.forward_AGNN(X, rp, col, att, bp, e2c, e2r)[0]
    outs = {}
    for mode in (0, 1, 2):
        c.check(c.lib.tcgnn_set_spmm_mode(mode), "mode")
        for _ in range(3):
            y = fn()
        TCGNN.kernel_timing(*meta, max_calls=20)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            y = fn()
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 20 * 1e3
        km = TCGNN.kernel_timing(*meta); TCGNN.kernel_timing(*meta, max_calls=0)
        outs[mode] = y
        print("D=%3d mode %d  %-90s kernel %.3f ms (min %.3f)  call %.3f ms" % (D, mode, TCGNN.last_kernel(*meta), float(np.mean(km)), float(np.min(km)), wall))
    c.lib.tcgnn_set_spmm_mode(0)
    print("      max |auto - per-window| %.3e   plan_bytes %d" % (float((outs[0] - outs[1]).abs().max()), TCGNN.plan_info(*meta)["plan_bytes"]))
