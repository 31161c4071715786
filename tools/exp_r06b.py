#!/usr/bin/env python3
"""r06 experiment B: the slice-synchronised range walk on the ogbn-products shape (kernel time from the library's HIP events).
GEN=sbm|uniform|rmat, DIMS=128,64, KBS=2048,1024,3072 (phase sizes tried in mode 5)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import numpy as np, torch
import TCGNN, tcgnn_graph as G, tcgnn_capi as c
dev = torch.device("cuda:0")
gen = os.environ.get("GEN", "sbm")
n, nnz, _, _ = G.SHAPES[os.environ.get("SHAPE", "ogbn-products")]
rp, col = G.GENERATORS[gen](n, nnz, seed=0, device=dev)
E = col.numel(); nw = (n + 15) // 16
bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r)
meta = (rp, col, bp, e2c, e2r)
g = torch.Generator(device=dev).manual_seed(0)
w = torch.tensor([0.9], device=dev)
def timed(fn, reps=8, warm=3):
    for _ in range(warm): fn()
    TCGNN.kernel_timing(*meta, max_calls=8 * reps)
    for _ in range(reps): out = fn()
    t = np.array(TCGNN.kernel_timing(*meta)).reshape(reps, -1).sum(1)
    TCGNN.kernel_timing(*meta, max_calls=0)
    return float(np.median(t)), out
print("%s %s: N %d E %d" % (os.environ.get("SHAPE", "ogbn-products"), gen, n, E), flush=True)
att = torch.randn(1, E, device=dev, generator=g)
for D in [int(x) for x in os.environ.get("DIMS", "128,64").split(",")]:
    X = torch.randn(n, D, device=dev, generator=g) / D ** 0.5
    ref = {}
    for mode, kb in [(1, None), (0, None)] + [(5, k) for k in os.environ.get("KBS", "2048,1024,3072").split(",")]:
        if kb: os.environ["TCGNN_RANGE_KB"] = kb
        else: os.environ.pop("TCGNN_RANGE_KB", None)
        c.lib.tcgnn_set_spmm_mode(mode)
        t1, Y = timed(lambda: TCGNN.forward(X, *meta)[0]); k1 = TCGNN.last_kernel(*meta)
        t2, Yv = timed(lambda: TCGNN.forward_AGNN(X, rp, col, att, bp, e2c, e2r)[0]); k2 = TCGNN.last_kernel(*meta)
        t3, ef = timed(lambda: TCGNN.forward_ef(X, *meta)[0]); k3 = TCGNN.last_kernel(*meta)
        t4, (Yf, eff, efm) = timed(lambda: TCGNN.agnn_fused_forward(X, rp, col, w, bp, e2c, e2r)); k4 = TCGNN.last_kernel(*meta)
        t5, (Gb, dw) = timed(lambda: TCGNN.agnn_fused_backward(X, rp, col, w, eff, efm, bp, e2c, e2r))
        if mode == 1: ref = dict(Y=Y.clone(), Yv=Yv.clone(), ef=ef.clone())
        print("  D=%3d mode %d kb %s: spmm %.3f (%s) | spmm_val %.3f (%s) | sddmm %.3f (%s) | maxdiff Y %.2e Yv %.2e ef-equal %s | fused fwd %.3f bwd %.3f (%s) dw %.6e" % (
            D, mode, kb, t1, k1, t2, k2, t3, k3, (Y - ref["Y"]).abs().max().item(), (Yv - ref["Yv"]).abs().max().item(), torch.equal(ef, ref["ef"]), t4, t5, k4, float(dw)), flush=True)
    c.lib.tcgnn_set_spmm_mode(0)
    os.environ.pop("TCGNN_RANGE_KB", None)
    del X, ref
