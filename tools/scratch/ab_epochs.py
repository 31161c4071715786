#!/usr/bin/env python3
"""GCN / AGNN ms per epoch (hidden 64, 25 dry epochs, 20 timed) on one Reddit-shaped graph; run once per library (TCGNN_LIB_PATH)
on ONE box and compare.   GEN=uniform|sbm_reddit|rmat python tools/scratch/ab_epochs.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import torch
import TCGNN, tcgnn_graph as G, tcgnn_harness as H
dev = torch.device("cuda:0")
n, nnz, in_dim, classes = G.SHAPES["reddit"]
rp, col = G.GENERATORS[os.environ.get("GEN", "uniform")](n, nnz, seed=0, device=dev)
E = col.numel(); nw = (n + 15) // 16
bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
fd = os.open(os.devnull, os.O_WRONLY); sv = os.dup(1); os.dup2(fd, 1)
TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r)
os.dup2(sv, 1)
meta = (rp, col, bp, e2c, e2r)
g = torch.Generator(device=dev).manual_seed(0)
feats = torch.randn(n, in_dim, device=dev, generator=g); labels = torch.ones(n, dtype=torch.long, device=dev)
out = []
for model in ("gcn", "agnn"):
    for h in (64, 16):
        r = H.time_training(model, meta, feats, labels, in_dim, h, classes, 2, 20, seed=0, warmup=25)
        out.append("%s h=%d %.3f" % (model, h, r["train_ms"]))
print(os.environ.get("GEN", "uniform"), " | ".join(out))
