#!/usr/bin/env python3
"""Fused AGNN products vs the separate calls on the Reddit-shaped graph (kernel time from the library's HIP events).
GEN=uniform|sbm_reddit|rmat|... picks the generator, DIMS=64,48 the widths."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import numpy as np, torch
import TCGNN, tcgnn_graph as G
dev = torch.device("cuda:0")
n, nnz, _, _ = G.SHAPES["reddit"]
rp, col = G.GENERATORS[os.environ.get("GEN", "uniform")](n, nnz, seed=0, device=dev)
E = col.numel(); nw = (n + 15) // 16
bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r)
meta = (rp, col, bp, e2c, e2r)
g = torch.Generator(device=dev).manual_seed(0)
w = torch.tensor([0.9], device=dev)
def timed(fn, reps=8):
    fn(); TCGNN.kernel_timing(*meta, max_calls=4 * reps)
    for _ in range(reps): out = fn()
    t = np.array(TCGNN.kernel_timing(*meta)).reshape(reps, -1).sum(1)
    return float(np.median(t)), out
for D in tuple(int(x) for x in os.environ.get("DIMS", "64,16,32,128").split(",")):
    X = torch.randn(n, D, device=dev, generator=g) / D ** 0.5
    dY = torch.randn(n, D, device=dev, generator=g)
    t_sd, ef = timed(lambda: TCGNN.forward_ef(X, *meta)[0])
    att = (w.view(1, 1) * ef.unsqueeze(0)).contiguous()
    t_sv, Y = timed(lambda: TCGNN.forward_AGNN(X, rp, col, att, bp, e2c, e2r)[0])
    import tcgnn_capi as c
    fused = {}
    for mode in (0, 1, 2):
        c.lib.tcgnn_set_spmm_mode(mode)
        t_ff, (Yf, eff, efm) = timed(lambda: TCGNN.agnn_fused_forward(X, rp, col, w, bp, e2c, e2r))
        t_fb, (Gf, dw) = timed(lambda: TCGNN.agnn_fused_backward(dY, rp, col, w, eff, efm, bp, e2c, e2r))
        fused[mode] = (t_ff, t_fb, Yf, eff, float(dw))
    c.lib.tcgnn_set_spmm_mode(0)
    print("D=%3d  sddmm %.3f + spmm_val %.3f = %.3f ms | fused automatic fwd %.3f bwd %.3f | per-window fwd %.3f bwd %.3f | range-major fwd %.3f bwd %.3f ms | ef equal %s  Y maxdiff %.2e (max |Y| %.2e) dw %.6e %.6e" % (
        D, t_sd, t_sv, t_sd + t_sv, fused[0][0], fused[0][1], fused[1][0], fused[1][1], fused[2][0], fused[2][1], torch.equal(ef, fused[2][3]) and torch.equal(ef, fused[1][3]),
        (Y - fused[2][2]).abs().max().item(), Y.abs().max().item(), fused[1][4], fused[2][4]))
