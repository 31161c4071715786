#!/usr/bin/env python3
"""Randomised cross-check of the LDS-resident SpMM (mode 3) against the per-window walk (mode 1) and A @ 1 = degree:
random sizes (N not a multiple of 16, hubs that overflow a wavefront's metadata pad, near-empty ranges), random widths.
usage: stress_lds.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import numpy as np, torch
import TCGNN, tcgnn_graph as G, tcgnn_capi as c
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
fd = os.open(os.devnull, os.O_WRONLY); sv = os.dup(1)
worst = 0.0
for k in range(cases):
    n = int(rng.integers(1500, 60000)); deg = float(rng.choice([2, 8, 40, 150, 400])); skew = float(rng.choice([0.0, 0.4, 0.8]))
    nnz = int(min(n * deg, 12_000_000))
    D = int(rng.choice([1, 7, 16, 32, 41, 48, 64, 80, 96, 100, 128, 160, 200]))
    gen = str(rng.choice(["uniform", "uniform", "sbm", "sbm_hubs", "rmat"]))
    seed_ = int(rng.integers(1 << 30))
    if gen == "uniform": rp, col = G.synthetic_csr(n, nnz, seed=seed_, device=dev, skew=skew)
    elif gen == "sbm": rp, col = G.sbm_csr(n, max(nnz, 4 * n), seed=seed_, device=dev, blocks=int(rng.choice([3, 10, 50])), p_in=float(rng.choice([0.7, 0.95])))
    elif gen == "sbm_hubs": rp, col = G.sbm_csr(n, max(nnz, 4 * n), seed=seed_, device=dev, blocks=int(rng.choice([4, 20])), hubs=int(rng.choice([2, 16, 64])), p_hub=float(rng.choice([0.05, 0.2])))
    else: rp, col = G.rmat_csr(n, max(nnz, 4 * n), seed=seed_, device=dev)
    # the placement and the hot / cold threshold of the cell stream, at random (read when the stream is built)
    place = str(rng.choice(["auto", "auto", "local", "localsplit", "global"])); hot = int(rng.choice([0, 0, 1, 50, 300, 1000, 3000]))
    os.environ.pop("TCGNN_LDS_PLACE", None); os.environ.pop("TCGNN_LDS_HOT_COLS", None)
    if place != "auto": os.environ["TCGNN_LDS_PLACE"] = place
    if hot: os.environ["TCGNN_LDS_HOT_COLS"] = str(hot)
    # r05: flat streams with dense entries - the automatic rule, every overflowing pair dense, some, none; the flat stream forced or not
    dense = str(rng.choice(["auto", "auto", "1", "200", "5000", "1000000000"])); flat = str(rng.choice(["auto", "auto", "1"]))
    os.environ.pop("TCGNN_LDS_DENSE_COLS", None); os.environ.pop("TCGNN_LDS_FLAT", None)
    if dense != "auto": os.environ["TCGNN_LDS_DENSE_COLS"] = dense
    if flat != "auto": os.environ["TCGNN_LDS_FLAT"] = flat
    n = rp.numel() - 1; E = col.numel(); nw = (n + 15) // 16
    bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
    os.dup2(fd, 1); TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r); os.dup2(sv, 1)
    meta = (rp, col, bp, e2c, e2r)
    X = torch.randn(n, D, device=dev)
    out = {}
    for mode in (1, 3):
        c.check(c.lib.tcgnn_set_spmm_mode(mode), "mode")
        out[mode] = TCGNN.forward(X, *meta)[0]
        if mode == 3: lds_kernel = TCGNN.last_kernel(*meta)
        ones = TCGNN.forward(torch.ones(n, D, device=dev), *meta)[0]
        degs = (rp[1:] - rp[:-1]).float()
        assert torch.equal(ones, degs[:, None].expand(-1, D)), ("A @ 1 != degree", mode, n, E, D)
    c.lib.tcgnn_set_spmm_mode(0)
    scale = ((rp[1:] - rp[:-1]).float().sqrt()[:, None] + 1) * X.abs().max()
    err = ((out[1] - out[3]).abs() / scale).max().item()
    worst = max(worst, err)
    maxdeg = int((rp[1:] - rp[:-1]).max())
    print("case %2d: %-8s N=%6d E=%9d D=%3d skew %.1f maxdeg %6d place %-10s hot %4d dense %-10s flat %-4s %-48s: |lds - plain| / scale = %.2e" % (
        k, gen, n, E, D, skew, maxdeg, place, hot, dense, flat, lds_kernel, err), flush=True)
    assert err < 2e-3, "mismatch"
    TCGNN.clear_plan_cache() if hasattr(TCGNN, "clear_plan_cache") else None
    del rp, col, bp, e2c, e2r, X, out; torch.cuda.empty_cache()
print("all %d cases agree; worst scaled difference %.2e" % (cases, worst))
