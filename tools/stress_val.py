#!/usr/bin/env python3
"""Randomised cross-check of forward_AGNN on the LDS-resident walk (tcgnn_lds_val.inc, forced: mode 3 + TCGNN_LDS_FLAT=1) against
the per-window gather walk (mode 1) and an fp64 evaluation on the device: random sizes (N % 16 != 0, hubs, near-empty ranges),
random widths, random value scales.  usage: stress_val.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
os.environ["TCGNN_LDS_FLAT"] = "1"
import numpy as np, torch
import TCGNN, tcgnn_graph as G, tcgnn_capi as c
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
fd = os.open(os.devnull, os.O_WRONLY); sv = os.dup(1)
worst = 0.0; taken = 0
for k in range(cases):
    n = int(rng.integers(1500, 60000)); deg = float(rng.choice([8, 40, 150, 400])); skew = float(rng.choice([0.0, 0.4, 0.8]))
    nnz = int(min(n * deg, 12_000_000))
    D = int(rng.choice([64, 128, 64, 41, 112, 100]))   # (the walk exists for whole 64-column groups and a three-plane remainder behind them)
    gen = str(rng.choice(["uniform", "uniform", "sbm", "rmat"]))
    seed_ = int(rng.integers(1 << 30))
    if gen == "uniform": rp, col = G.synthetic_csr(n, nnz, seed=seed_, device=dev, skew=skew)
    elif gen == "sbm": rp, col = G.sbm_csr(n, max(nnz, 4 * n), seed=seed_, device=dev, blocks=int(rng.choice([3, 10, 50])), p_in=float(rng.choice([0.7, 0.95])))
    else: rp, col = G.rmat_csr(n, max(nnz, 4 * n), seed=seed_, device=dev)
    n = rp.numel() - 1; E = col.numel(); nw = (n + 15) // 16
    bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
    os.dup2(fd, 1); TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r); os.dup2(sv, 1)
    meta = (rp, col, bp, e2c, e2r)
    dense = str(rng.choice(["auto", "auto", "1", "200", "5000", "1000000000"]))   # r05: dense entries of the single-edge stream
    os.environ.pop("TCGNN_LDS_DENSE_COLS", None)
    if dense != "auto": os.environ["TCGNN_LDS_DENSE_COLS"] = dense
    X = torch.randn(n, D, device=dev) * float(rng.choice([1e-3, 1.0, 50.0]))
    att = torch.randn(1, E, device=dev) * float(rng.choice([0.01, 1.0, 30.0]))
    c.check(c.lib.tcgnn_set_spmm_mode(3), "mode")
    TCGNN.forward_AGNN(X, rp, col, att, bp, e2c, e2r)
    Y3 = TCGNN.forward_AGNN(X, rp, col, att, bp, e2c, e2r)[0]
    kernel = TCGNN.last_kernel(*meta)
    c.check(c.lib.tcgnn_set_spmm_mode(1), "mode")
    Y1 = TCGNN.forward_AGNN(X, rp, col, att, bp, e2c, e2r)[0]
    c.lib.tcgnn_set_spmm_mode(0)
    rows = torch.repeat_interleave(torch.arange(n, device=dev), (rp[1:] - rp[:-1]).long())
    prod = att.view(-1, 1).double() * X.double()[col.long()]
    Y64 = torch.zeros(n, D, dtype=torch.float64, device=dev).index_add_(0, rows, prod)
    absY = torch.zeros(n, D, dtype=torch.float64, device=dev).index_add_(0, rows, prod.abs())
    del prod
    tol = 2.0 ** -9 * absY + 1e-30            # two 10-bit operands per product
    e3 = ((Y3.double() - Y64).abs() / (tol + 1e-300)).max().item(); e1 = ((Y1.double() - Y64).abs() / (tol + 1e-300)).max().item()
    d13 = ((Y3 - Y1).abs().double() / (absY + 1e-30)).max().item()
    worst = max(worst, e3); taken += "lds_val" in kernel
    print("case %2d: %-8s N=%6d E=%9d D=%3d dense %-10s %-60s err/bound lds %.3f gather %.3f  |lds - gather| / sum|.| %.1e" % (k, gen, n, E, D, dense, kernel[:60], e3, e1, d13), flush=True)
    assert e3 <= 1.0 and e1 <= 1.0, "outside the bound"
    TCGNN.clear_plan_cache()
    del rp, col, bp, e2c, e2r, X, att, Y1, Y3, Y64, absY; torch.cuda.empty_cache()
print("all %d cases inside the bound (%d on spmm_lds_val_kernel); worst error / bound %.3f" % (cases, taken, worst))
