#!/usr/bin/env python3
"""r06 experiment A: where the gather walks stand on graphs with communities (kernel time from the library's HIP events).
  part 1  ogbn-products shape, SBM generator (50 communities, 90 % of the edges inside): per-window SpMM at D = 32 / 64 / 128, SDDMM, fused pair
  part 2  Reddit shape, SBM (50 communities of 4.6 k rows: a community's image fits an XCD's L2 at every width): the L2-resident gather rate at 256-byte rows
  part 3  sbm_reddit (headline graph), D = 64: SDDMM and the fused pair under the window-order knobs of r06
PARTS=1,2,3 selects."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import numpy as np, torch
import TCGNN, tcgnn_graph as G, tcgnn_capi as c
dev = torch.device("cuda:0")
parts = [int(x) for x in os.environ.get("PARTS", "1,2,3").split(",")]

def graph(shape, gen):
    n, nnz, _, _ = G.SHAPES[shape]
    rp, col = G.GENERATORS[gen](n, nnz, seed=0, device=dev)
    E = col.numel(); nw = (n + 15) // 16
    bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
    TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r)
    return n, E, (rp, col, bp, e2c, e2r)

def timed(meta, fn, reps=8, warm=3):
    for _ in range(warm): fn()
    TCGNN.kernel_timing(*meta, max_calls=8 * reps)
    for _ in range(reps): out = fn()
    t = np.array(TCGNN.kernel_timing(*meta)).reshape(reps, -1).sum(1)
    TCGNN.kernel_timing(*meta, max_calls=0)
    return float(np.median(t)), out

g = torch.Generator(device=dev).manual_seed(0)
w = torch.tensor([0.9], device=dev)

def knobs(**kv):
    for k, v in kv.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = str(v)

if 1 in parts:
    n, E, meta = graph("ogbn-products", "sbm")
    print("products sbm: N %d E %d info %s" % (n, E, TCGNN.plan_info(*meta)), flush=True)
    for D in (32, 64, 128):
        X = torch.randn(n, D, device=dev, generator=g) / D ** 0.5
        for mode in (0, 1):
            c.lib.tcgnn_set_spmm_mode(mode)
            t, _ = timed(meta, lambda: TCGNN.forward(X, *meta))
            print("  products sbm D=%3d spmm mode %d: %.3f ms (%s)" % (D, mode, t, TCGNN.last_kernel(*meta)), flush=True)
        c.lib.tcgnn_set_spmm_mode(0)
        t, ef = timed(meta, lambda: TCGNN.forward_ef(X, *meta)[0])
        print("  products sbm D=%3d sddmm auto: %.3f ms (%s)" % (D, t, TCGNN.last_kernel(*meta)), flush=True)
        del X, ef
    TCGNN.clear_plan_cache(); del meta; torch.cuda.empty_cache()

if 2 in parts:
    n, E, meta = graph("reddit", "sbm")
    print("reddit sbm: N %d E %d" % (n, E), flush=True)
    for D in (64, 128):
        X = torch.randn(n, D, device=dev, generator=g) / D ** 0.5
        c.lib.tcgnn_set_spmm_mode(1)
        t, _ = timed(meta, lambda: TCGNN.forward(X, *meta))
        print("  reddit sbm D=%3d spmm per-window: %.3f ms = %.1f ps/edge, %.2f TB/s of gathered rows (%s)" % (D, t, t * 1e9 / E, E * 2 * D / t / 1e9, TCGNN.last_kernel(*meta)), flush=True)
        t, _ = timed(meta, lambda: TCGNN.forward_ef(X, *meta)[0])
        print("  reddit sbm D=%3d sddmm per-window: %.3f ms (%s)" % (D, t, TCGNN.last_kernel(*meta)), flush=True)
        c.lib.tcgnn_set_spmm_mode(0)
        t, _ = timed(meta, lambda: TCGNN.agnn_fused_forward(X, meta[0], meta[1], w, *meta[2:]))
        print("  reddit sbm D=%3d fused fwd auto: %.3f ms (%s)" % (D, t, TCGNN.last_kernel(*meta)), flush=True)
        del X
    TCGNN.clear_plan_cache(); del meta; torch.cuda.empty_cache()

if 3 in parts:
    for gen in ("sbm_reddit", "uniform"):
        n, E, meta = graph("reddit", gen)
        rp, col, bp, e2c, e2r = meta
        D = 64
        X = torch.randn(n, D, device=dev, generator=g) / D ** 0.5
        dY = torch.randn(n, D, device=dev, generator=g)
        ref = None
        for xcd, ident in ((None, 0), (2, 0), (0, 1), (None, 1), (2, 1)):
            knobs(TCGNN_SDDMM_XCD=xcd, TCGNN_RM_IDENT=ident)
            t, ef = timed(meta, lambda: TCGNN.forward_ef(X, *meta)[0])
            if ref is None: ref = ef.clone()
            print("  %s sddmm XCD=%s IDENT=%s: %.3f ms  bit-equal %s (%s)" % (gen, xcd, ident, t, torch.equal(ef, ref), TCGNN.last_kernel(*meta)), flush=True)
        knobs(TCGNN_SDDMM_XCD=None, TCGNN_RM_IDENT=None)
        c.lib.tcgnn_set_spmm_mode(1)
        t, _ = timed(meta, lambda: TCGNN.forward_ef(X, *meta)[0])
        print("  %s sddmm per-window: %.3f ms" % (gen, t), flush=True)
        c.lib.tcgnn_set_spmm_mode(0)
        yref = None
        for sliced, rot in ((None, 0), (None, 1), (2, 0), (2, 1), (0, 0)):
            knobs(TCGNN_AGNN_SLICED=sliced, TCGNN_AGNN_ROT=rot)
            TCGNN.clear_plan_cache()   # (the workspace is sized by the walk)
            tf, (Yf, eff, efm) = timed(meta, lambda: TCGNN.agnn_fused_forward(X, rp, col, w, bp, e2c, e2r))
            kf = TCGNN.last_kernel(*meta)
            tb, (Gb, dw) = timed(meta, lambda: TCGNN.agnn_fused_backward(dY, rp, col, w, eff, efm, bp, e2c, e2r))
            if yref is None: yref = (Yf.clone(), Gb.clone())
            print("  %s fused SLICED=%s ROT=%s: fwd %.3f (%s) bwd %.3f (%s)  Y maxdiff %.2e G maxdiff %.2e dw %.6e" % (
                gen, sliced, rot, tf, kf, tb, TCGNN.last_kernel(*meta), (Yf - yref[0]).abs().max().item(), (Gb - yref[1]).abs().max().item(), float(dw)), flush=True)
            att = None
        knobs(TCGNN_AGNN_SLICED=None, TCGNN_AGNN_ROT=None)
        TCGNN.clear_plan_cache(); del meta, X, dY; torch.cuda.empty_cache()
