import os, sys
ROOT = "/root/repo"
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import torch, TCGNN, tcgnn_graph as G, tcgnn_capi as c
dev = torch.device("cuda:0")
n, nnz, blocks = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rp, col = G.sbm_csr(n, nnz, seed=0, device=dev, blocks=blocks)
E = col.numel(); nw = (n + 15) // 16
bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
fd = os.dup(1); os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r)
os.dup2(fd, 1)
meta = (rp, col, bp, e2c, e2r)
for D in (64,):
    X = torch.randn(n, D, device=dev)
    for mode in (0, 1, 2, 3):
        c.check(c.lib.tcgnn_set_spmm_mode(mode), "mode")
        try:
            for _ in range(3): TCGNN.forward(X, *meta)
            torch.cuda.synchronize()
            import time; t0 = time.perf_counter()
            for _ in range(10): TCGNN.forward(X, *meta)
            torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 100
            print("N %d E %d D %d mode %d: %.3f ms per call  (%s)" % (n, E, D, mode, ms, TCGNN.last_kernel(*meta)), flush=True)
        except Exception as exc:
            print("mode", mode, "failed:", str(exc)[:200])
    c.lib.tcgnn_set_spmm_mode(0)
