import os, sys
ROOT = "/root/repo"
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import numpy as np, torch, time
import TCGNN, tcgnn_graph as G, tcgnn_capi as c
dev = torch.device("cuda:0")
n, nnz, _, _ = G.SHAPES["reddit"]
rp, col = G.GENERATORS[os.environ.get("GEN", "uniform")](n, nnz, seed=0, device=dev)
E = col.numel(); nw = (n + 15) // 16
bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
fd = os.open(os.devnull, os.O_WRONLY); sv = os.dup(1); os.dup2(fd, 1)
TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r)
os.dup2(sv, 1)
meta = (rp, col, bp, e2c, e2r)
X = torch.randn(n, 64, device=dev)
TCGNN.forward(X, *meta); TCGNN.forward(X, *meta); torch.cuda.synchronize()
TCGNN.kernel_timing(*meta, max_calls=400)
t0 = time.perf_counter()
for _ in range(400): TCGNN.forward(X, *meta)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
t = np.array(TCGNN.kernel_timing(*meta))
print(os.environ.get("GEN", "uniform"), "wall per call %.3f ms" % (wall / 400 * 1e3), "kernel ms by block of 25 calls:", " ".join("%.3f" % t[i:i+25].mean() for i in range(0, len(t), 25)))
# idle gap then again
time.sleep(1.0)
TCGNN.kernel_timing(*meta, max_calls=100)
for _ in range(100): TCGNN.forward(X, *meta)
torch.cuda.synchronize()
t = np.array(TCGNN.kernel_timing(*meta))
print("after 1 s idle:", " ".join("%.3f" % t[i:i+25].mean() for i in range(0, len(t), 25)))
