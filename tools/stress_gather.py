#!/usr/bin/env python3
"""Randomised cross-check of the edge-valued SpMM, the SDDMM and the fused AGNN pair: random generators (uniform with skew,
communities, communities + hubs, R-MAT, hub rows), sizes that are not multiples of 16, random widths, every gather walk (automatic,
per-window, range-blocked) - against fp64 torch evaluations of the same operators (bound: 2^-9 of the sum of |terms|, the operand
rounding) and against each other.  usage: stress_gather.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import numpy as np, torch
import TCGNN, tcgnn_graph as G, tcgnn_capi as c
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
fd = os.open(os.devnull, os.O_WRONLY); sv = os.dup(1)
for k in range(cases):
    n = int(rng.integers(300, 40000)); deg = float(rng.choice([3, 12, 60, 200])); nnz = int(min(n * deg, 6_000_000))
    gen = str(rng.choice(["uniform", "sbm", "sbm_hubs", "rmat", "hubrows"])); seed_ = int(rng.integers(1 << 30))
    if gen == "uniform": rp, col = G.synthetic_csr(n, nnz, seed=seed_, device=dev, skew=float(rng.choice([0.0, 0.5, 0.8])))
    elif gen == "sbm": rp, col = G.sbm_csr(n, max(nnz, 4 * n), seed=seed_, device=dev, blocks=int(rng.choice([3, 10, 40])))
    elif gen == "sbm_hubs": rp, col = G.sbm_csr(n, max(nnz, 4 * n), seed=seed_, device=dev, blocks=int(rng.choice([4, 20])), hubs=int(rng.choice([2, 16])), p_hub=0.15)
    elif gen == "rmat": rp, col = G.rmat_csr(n, max(nnz, 4 * n), seed=seed_, device=dev)
    else:   # a few rows adjacent to every node over a sparse background: runs of eight edges inside eight tile columns
        h = int(rng.integers(1, 20))
        src = torch.cat([torch.arange(h, device=dev).repeat_interleave(n), torch.randint(0, n, (4 * n,), device=dev)])
        dst = torch.cat([torch.arange(n, device=dev).repeat(h), torch.randint(0, n, (4 * n,), device=dev)])
        keep = src != dst
        key = torch.unique(torch.cat([src[keep] * n + dst[keep], dst[keep] * n + src[keep]]))
        rp = torch.zeros(n + 1, dtype=torch.int64, device=dev); rp[1:] = torch.cumsum(torch.bincount(key // n, minlength=n), 0)
        rp, col = rp.to(torch.int32), (key % n).to(torch.int32)
    n = rp.numel() - 1; E = col.numel(); nw = (n + 15) // 16
    D = int(rng.choice([1, 7, 16, 32, 41, 64, 96, 128, 200]))
    bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
    os.dup2(fd, 1); TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r); os.dup2(sv, 1)
    meta = (rp, col, bp, e2c, e2r)
    X = torch.randn(n, D, device=dev) * float(rng.choice([0.01, 1.0, 30.0])); att = torch.randn(E, device=dev)
    rows = torch.repeat_interleave(torch.arange(n, device=dev), (rp[1:] - rp[:-1]).long()); cl = col.long()
    Xd = X.double()
    ef64 = (Xd[rows] * Xd[cl]).sum(1); ef_abs = (Xd[rows].abs() * Xd[cl].abs()).sum(1)
    Y64 = torch.zeros(n, D, dtype=torch.float64, device=dev).index_add_(0, rows, att.double()[:, None] * Xd[cl])
    Yabs = torch.zeros(n, D, dtype=torch.float64, device=dev).index_add_(0, rows, att.double().abs()[:, None] * Xd[cl].abs())
    res = {}
    for mode in (0, 1, 2):
        c.check(c.lib.tcgnn_set_spmm_mode(mode), "mode")
        Yv = TCGNN.forward_AGNN(X, rp, col, att.view(1, -1), bp, e2c, e2r)[0]
        ef = TCGNN.forward_ef(X, *meta)[0]
        assert bool(((Yv.double() - Y64).abs() <= Yabs * 2.0 ** -9 + 1e-30).all()), ("spmm_val", gen, n, E, D, mode)
        assert bool(((ef.double() - ef64).abs() <= ef_abs * 2.0 ** -9 + 1e-30).all()), ("sddmm", gen, n, E, D, mode)
        res[mode] = (Yv, ef)
    c.lib.tcgnn_set_spmm_mode(0)
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[1][1], res[2][1]), ("sddmm differs between walks", gen, n, E, D)
    fused = ""
    if D <= 128 and TCGNN.agnn_fused_supported(X, *meta):
        w = torch.tensor([0.8], device=dev)
        Yf, eff, efm = TCGNN.agnn_fused_forward(X, rp, col, w, bp, e2c, e2r)
        assert torch.equal(eff, res[1][1]), ("fused scores", gen, n, E, D)
        a64 = 0.8 * ef64
        Yf64 = torch.zeros(n, D, dtype=torch.float64, device=dev).index_add_(0, rows, a64[:, None] * Xd[cl])
        Yfabs = torch.zeros(n, D, dtype=torch.float64, device=dev).index_add_(0, rows, (0.8 * ef_abs)[:, None] * Xd[cl].abs())
        assert bool(((Yf.double() - Yf64).abs() <= Yfabs * 2.0 ** -8 + 1e-30).all()), ("fused forward", gen, n, E, D)
        dY = torch.randn(n, D, device=dev)
        Gb, dw = TCGNN.agnn_fused_backward(dY, rp, col, w, eff, efm, bp, e2c, e2r)
        G64 = torch.zeros(n, D, dtype=torch.float64, device=dev).index_add_(0, rows, a64[:, None] * dY.double()[cl])
        Gabs = torch.zeros(n, D, dtype=torch.float64, device=dev).index_add_(0, rows, (0.8 * ef_abs)[:, None] * dY.double().abs()[cl])
        assert bool(((Gb.double() - G64).abs() <= Gabs * 2.0 ** -8 + 1e-30).all()), ("fused backward", gen, n, E, D)
        fused = " + fused pair"
    print("case %2d: %-8s N=%6d E=%8d D=%3d maxdeg %6d ok%s" % (k, gen, n, E, D, int((rp[1:] - rp[:-1]).max()), fused), flush=True)
    TCGNN.clear_plan_cache()
    del rp, col, bp, e2c, e2r, X, att, res; torch.cuda.empty_cache()
print("all %d cases agree" % cases)
