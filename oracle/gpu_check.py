#!/usr/bin/env python3
"""CHECKER-SIDE UTILITY (lives under oracle/; nothing in the product imports it).  Quick on-GPU diagnostics: max errors of the three kernels against the oracle over a grid of
graph shapes and feature widths, plus first timings.  Not a test (tests/test_gpu_parity.py is);
prints numbers so a failure can be localised from one gpurun call."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import TCGNN
import graphs
from oracle import oracle as O

dev = torch.device("cuda:0")
print("device:", torch.cuda.get_device_name(0))

def meta(rp, col):
    bp, e2c, e2r, cnt = graphs.host_sgt(rp, col)
    t = lambda a: torch.from_numpy(a).to(dev)
    return (bp, e2c, e2r), (t(rp), t(col), t(bp), t(e2c), t(e2r))

def relerr(got, ref):
    return float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref)))) if ref.size else 0.0

worst = 0.0
cases = [(n, rp, c) for n, rp, c in graphs.edge_case_graphs()]
cases.append(("citeseer_shape", *graphs.uniform_graph(3327, 2.8, seed=1)))
cases.append(("dense_n4000_deg200", *graphs.uniform_graph(4000, 200, seed=2)))
cases.append(("powerlaw_n20000_deg60", *graphs.powerlaw_graph(20000, 60, seed=3)))
for name, rp, col in cases:
    (bp, e2c, e2r), (trp, tcol, tbp, te2c, te2r) = meta(rp, col)
    N, E = len(rp) - 1, len(col)
    info = TCGNN.plan_info(trp, tcol, tbp, te2c, te2r)
    for D in (16, 64, 128, 7, 41, 48, 160, 256):
        if N > 5000 and D not in (16, 64, 41, 128): continue
        rng = np.random.default_rng(D + N)
        X = rng.standard_normal((N, D)).astype(np.float32)
        att = rng.standard_normal(E).astype(np.float32)
        tX = torch.from_numpy(X).to(dev)
        Y = TCGNN.forward(tX, trp, tcol, tbp, te2c, te2r)[0].cpu().numpy()
        Yr = O.spmm(X, rp, col, bp, e2c, e2r, round_mode=O.ROUND_TF32)
        e1 = relerr(Y, Yr)
        Yv = TCGNN.forward_AGNN(tX, trp, tcol, torch.from_numpy(att).to(dev).unsqueeze(0).contiguous(), tbp, te2c, te2r)[0].cpu().numpy()
        Yvr = O.spmm_val(X, rp, col, att, bp, e2c, e2r, round_mode=O.ROUND_TF32)
        e2 = relerr(Yv, Yvr)
        ef = TCGNN.forward_ef(tX, trp, tcol, tbp, te2c, te2r)[0].cpu().numpy()
        efr = O.sddmm(X, rp, col, bp, e2c, e2r, round_mode=O.ROUND_TF32)
        e3 = relerr(ef, efr)
        worst = max(worst, e1, e2, e3)
        flag = "" if max(e1, e2, e3) < 1e-3 else "   <-- FAIL"
        print("%-26s N=%6d E=%8d D=%3d waves=%d wb=%7d | spmm %.2e  spmm_val %.2e  sddmm %.2e%s" % (name, N, E, D, info["waves_per_window"], info["wide_blocks"], e1, e2, e3, flag))
print("WORST relative error:", worst)

# ---- device SGT vs host SGT
for name, rp, col in cases:
    (bp, e2c, e2r), (trp, tcol, *_r) = meta(rp, col)
    N = len(rp) - 1
    gbp = torch.full((len(bp) + 2,), -7, dtype=torch.int32, device=dev)
    ge2c = torch.zeros(len(col), dtype=torch.int32, device=dev); ge2r = torch.zeros(len(col), dtype=torch.int32, device=dev)
    TCGNN.preprocess_gpu(tcol, trp, N, 16, 8, gbp, ge2c, ge2r)
    ok = (gbp[:len(bp)].cpu().numpy() == bp).all() and (ge2c.cpu().numpy() == e2c).all() and (ge2r.cpu().numpy() == e2r).all()
    print("preprocess_gpu %-26s %s  guard=%s" % (name, "ok" if ok else "MISMATCH", gbp[len(bp):].tolist()))

# ---- first timings on a mid-size graph
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
rp, col = graphs.uniform_graph(232965 // 4, 200, seed=5)
(bp, e2c, e2r), (trp, tcol, tbp, te2c, te2r) = meta(rp, col)
N, E = len(rp) - 1, len(col)
print("timing graph: N=%d E=%d info=%s" % (N, E, TCGNN.plan_info(trp, tcol, tbp, te2c, te2r)))
for D in (16, 64, 128):
    tX = torch.randn(N, D, device=dev)
    att = torch.randn(1, E, device=dev)
    t1 = timeit(lambda: TCGNN.forward(tX, trp, tcol, tbp, te2c, te2r))
    t2 = timeit(lambda: TCGNN.forward_ef(tX, trp, tcol, tbp, te2c, te2r))
    t3 = timeit(lambda: TCGNN.forward_AGNN(tX, trp, tcol, att, tbp, te2c, te2r))
    print("D=%3d  spmm %.3f ms (%.1f GTEPS)  sddmm %.3f ms (%.1f GTEPS)  spmm_val %.3f ms (%.1f GTEPS)" % (D, t1, E / t1 / 1e6, t2, E / t2 / 1e6, t3, E / t3 / 1e6))
sys.exit(0 if worst < 1e-3 else 1)
