#!/usr/bin/env python3
"""What the range guard sees in a training epoch: for every operand the harness's GCN / AGNN epochs hand to the aggregation
kernels (Reddit shape, the reference's recipe: unscaled randn weights) - max |x|, smallest nonzero |x|, and how many nonzero
elements lie more than 2^28 below the maximum.  usage: probe_training_ranges.py [generator]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import torch
import TCGNN, tcgnn_graph as G, tcgnn_harness as H
gen = sys.argv[1] if len(sys.argv) > 1 else "uniform"
dev = torch.device("cuda:0")
n, nnz, in_dim, classes = G.SHAPES["reddit"]
rp, col = G.GENERATORS[gen](n, nnz, seed=0, device=dev)
E = col.numel(); nw = (n + 15) // 16
bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
fd = os.dup(1); os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r)
os.dup2(fd, 1)
meta = (rp, col, bp, e2c, e2r)
seen = []
def wrap(name, fn, idx=0):
    def f(*a, **k):
        x = a[idx].detach().abs().flatten()
        nz = x[x > 0]
        mx = float(x.max()); mn = float(nz.min()) if nz.numel() else 0.0
        tiny = int((nz < mx * 2.0 ** -28).sum())
        # r05: the rows that hold such elements (the patch's dirty rows) and the edges that touch them (what the patch recomputes)
        xa = a[idx].detach().abs()
        drow = ((xa > 0) & (xa < mx * 2.0 ** -28)).any(1)
        ndirty = int(drow.sum())
        deg = (rp[1:] - rp[:-1]).long()
        dedges = int(2 * deg[drow].sum())
        out = fn(*a, **k)
        seen.append((name, tuple(a[idx].shape), mx, mn, tiny, TCGNN.range_mode(), ndirty, dedges))
        return out
    return f
for name in ("forward", "forward_fused", "forward_gemm", "forward_AGNN", "forward_ef", "agnn_fused_forward", "agnn_fused_backward"):
    setattr(TCGNN, name, wrap(name, getattr(TCGNN, name)))
feats = torch.randn(n, in_dim, device=dev); labels = torch.ones(n, dtype=torch.long, device=dev)
for model in ("gcn", "agnn"):
    seen.clear()
    H.time_training(model, meta, feats, labels, in_dim, 64, classes, 2, 1, seed=0, warmup=2, tune=False)
    print(model, gen)
    for s in seen[-8:]:
        print("   %-20s %-16s max %.3e  min nonzero %.3e  (2^%.1f below)  tiny %d  guard %s  dirty rows %d (~%d edges)" % (s[0], s[1], s[2], s[3], (torch.log2(torch.tensor(s[2] / max(s[3], 1e-45)))).item(), s[4], s[5], s[6], s[7]))
