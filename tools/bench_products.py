#!/usr/bin/env python3
"""BASELINE.json configs[3]: ogbn-products-shaped graph (N = 2 449 029, nnz = 123 718 280), AGNN, hidden 128:
the SDDMM + SpMM pair separately and fused, and the 2-layer AGNN epoch (in 100, classes 47)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import numpy as np, torch
import TCGNN, tcgnn_graph as G, tcgnn_harness as H
dev = torch.device("cuda:0")
shape = sys.argv[1] if len(sys.argv) > 1 else "ogbn-products"
D = int(sys.argv[2]) if len(sys.argv) > 2 else 128
n, nnz, in_dim, classes = G.SHAPES[shape]
rp, col = G.synthetic_csr(n, nnz, seed=0, device=dev)
E = col.numel(); nw = (n + 15) // 16
bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r)
meta = (rp, col, bp, e2c, e2r)
print(shape, TCGNN.plan_info(*meta))
g = torch.Generator(device=dev).manual_seed(0)
w = torch.tensor([0.9], device=dev)
def timed(fn, reps=5):
    fn(); TCGNN.kernel_timing(*meta, max_calls=4 * reps)
    for _ in range(reps): out = fn()
    t = np.array(TCGNN.kernel_timing(*meta)).reshape(reps, -1).sum(1)
    return float(np.median(t)), out
X = torch.randn(n, D, device=dev, generator=g) / D ** 0.5
t_sp, _ = timed(lambda: TCGNN.forward(X, *meta)[0])
t_sd, ef = timed(lambda: TCGNN.forward_ef(X, *meta)[0])
att = (w.view(1, 1) * ef.unsqueeze(0)).contiguous()
t_sv, Y = timed(lambda: TCGNN.forward_AGNN(X, rp, col, att, bp, e2c, e2r)[0])
del att
t_ff, (Yf, eff, efm) = timed(lambda: TCGNN.agnn_fused_forward(X, rp, col, w, bp, e2c, e2r))
t_fb, _ = timed(lambda: TCGNN.agnn_fused_backward(X, rp, col, w, eff, efm, bp, e2c, e2r))
spmm_b = 4 * (n + 1) + 4 * E + 8 * n * D; sddmm_b = 4 * (n + 1) + 8 * E + 4 * n * D
print("D=%d  spmm %.3f ms (%.1f GTEPS, %.3f of HBM roofline) | sddmm %.3f ms (%.1f GTEPS, %.3f) | spmm_val %.3f ms | pair %.3f ms | fused fwd %.3f  bwd %.3f ms | ef equal %s, Y maxdiff %.2e" % (
    D, t_sp, E / t_sp / 1e6, spmm_b / (t_sp * 1e-3) / 8e12, t_sd, E / t_sd / 1e6, sddmm_b / (t_sd * 1e-3) / 8e12, t_sv, t_sd + t_sv, t_ff, t_fb,
    torch.equal(ef, eff), (Y - Yf).abs().max().item()))
del X, Y, Yf, ef, eff
feats = torch.randn(n, in_dim, device=dev, generator=g); labels = torch.ones(n, dtype=torch.long, device=dev)
for model in ("agnn", "gcn"):
    r = H.time_training(model, meta, feats, labels, in_dim, D, classes, 2, 5, seed=0)
    print("%s 2-layer hidden %d: %.3f ms/epoch (loss finite %s)" % (model, D, r["train_ms"], np.isfinite(r["final_loss"])))
