import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import torch, TCGNN, tcgnn_graph as G
dev = torch.device("cuda:0")
n, nnz, _, _ = G.SHAPES["reddit"]
rp, col = G.GENERATORS[os.environ.get("GEN", "sbm_reddit")](n, nnz, seed=0, device=dev)
E = col.numel(); nw = (n + 15) // 16
bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
fd = os.open(os.devnull, os.O_WRONLY); sv = os.dup(1); os.dup2(fd, 1)
ts = []
for k in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    if k == 2: torch.cuda.empty_cache()
os.dup2(sv, 1)
print("preprocess_gpu ms:", " ".join("%.1f" % t for t in ts), "| torch reserved GB %.1f" % (torch.cuda.memory_reserved() / 1e9), "free GB %.1f" % (torch.cuda.mem_get_info()[0] / 1e9))
