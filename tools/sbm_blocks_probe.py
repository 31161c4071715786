#!/usr/bin/env python3
"""How much is L2 residency of the gathered rows worth at the ogbn-products shape?  SpMM / SDDMM kernel time (HIP events) on
SBM graphs of the same size and degree whose communities shrink from 49 k nodes (12.5 MB of image at D = 128: Infinity Cache only)
to 8 k (2 MB: fits the 4 MB L2 of the XCD whose workgroups gather it, given the XCD-contiguous window order).
usage: sbm_blocks_probe.py [blocks ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import torch
import TCGNN, tcgnn_graph as G
dev = torch.device("cuda:0")
n, nnz, _, _ = G.SHAPES["ogbn-products"]
D = 128
for blocks in [int(b) for b in sys.argv[1:]] or [50, 100, 300, 600]:
    rp, col = G.sbm_csr(n, nnz, seed=0, device=dev, blocks=blocks)
    E = col.numel(); nw = (n + 15) // 16
    bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
    fd = os.dup(1); os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r)
    os.dup2(fd, 1)
    meta = (rp, col, bp, e2c, e2r)
    X = torch.randn(n, D, device=dev)
    res = {}
    for name, fn in (("spmm", lambda: TCGNN.forward(X, *meta)), ("sddmm", lambda: TCGNN.forward_ef(X, *meta))):
        for _ in range(3): fn()
        TCGNN.kernel_timing(*meta, max_calls=16)
        for _ in range(8): fn()
        ms = TCGNN.kernel_timing(*meta)
        res[name] = sum(ms) / len(ms)
    print("blocks %4d  community %6d nodes = %5.1f MB of image  E %d  tc_blocks %d  spmm %.3f ms  sddmm %.3f ms" % (
        blocks, n // blocks, n / blocks * 256 / 1e6, E, TCGNN.plan_info(*meta)["tc_blocks"], res["spmm"], res["sddmm"]), flush=True)
    del X, rp, col, bp, e2c, e2r, meta
    TCGNN.clear_plan_cache(); torch.cuda.empty_cache()
