#!/usr/bin/env python3
"""The fused AGNN pair's state walk a few times on the Reddit shape (the process a rocprofv3 --pmc pass wraps).  usage: run_state_once.py [D] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import torch
import TCGNN, tcgnn_graph as G
D = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
n, nnz, _, _ = G.SHAPES["reddit"]
rp, col = G.synthetic_csr(n, nnz, seed=0, device=dev)
E = col.numel(); nw = (n + 15) // 16
bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
fd = os.dup(1); os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r)
os.dup2(fd, 1)
meta = (rp, col, bp, e2c, e2r)
g = torch.Generator(device=dev).manual_seed(0)
X = torch.randn(n, D, device=dev, generator=g) / D ** 0.5
dY = torch.randn(n, D, device=dev, generator=g)
w = torch.tensor([0.9], device=dev)
assert TCGNN.agnn_state_supported(X, *meta)
for _ in range(reps):
    Y, st = TCGNN.agnn_state_forward(X, rp, col, w, bp, e2c, e2r)
    Gd, dw = TCGNN.agnn_state_backward(dY, rp, col, w, st, bp, e2c, e2r)
torch.cuda.synchronize()
print("ok", float(dw))
