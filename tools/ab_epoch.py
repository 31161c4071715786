#!/usr/bin/env python3
"""A/B of epoch-level switches in ONE process on ONE box (boxes differ by ~5 %): GCN / AGNN ms per epoch on the Reddit shape with
an environment switch off and on, alternating.   python tools/ab_epoch.py TCGNN_FUSED_LOSS [hidden] [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")):
    sys.path.insert(0, p)
import torch
import TCGNN, tcgnn_graph as G, tcgnn_harness as H

var = sys.argv[1]; hidden = int(sys.argv[2]) if len(sys.argv) > 2 else 64; rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
n, nnz, in_dim, classes = G.SHAPES["reddit"]
rp, col = G.synthetic_csr(n, nnz, seed=0, device=dev)
E = col.numel(); nw = (n + 15) // 16
bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r)
meta = (rp, col, bp, e2c, e2r)
g = torch.Generator(device=dev).manual_seed(0)
feats = torch.randn(n, in_dim, device=dev, generator=g); labels = torch.ones(n, dtype=torch.long, device=dev)
res = {}
for rnd in range(rounds):
    for val in ("0", "1"):
        os.environ[var] = val
        if var == "RANGE_GUARD":          # (process-wide level, not an environment switch after start-up: 0 -> level 1, 1 -> level 2)
            TCGNN.set_range_guard(1 + int(val))
        for model in ("gcn", "agnn"):
            r = H.time_training(model, meta, feats, labels, in_dim, hidden, classes, 2, 10, seed=0)
            res.setdefault((model, val), []).append(r["train_ms"])
for (model, val), v in sorted(res.items()):
    print("%-5s %s=%s  %s  min %.3f ms" % (model, var, val, " ".join("%.3f" % t for t in v), min(v)))
