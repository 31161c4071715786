import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import numpy as np, torch
import TCGNN, tcgnn_graph as G
dev = torch.device("cuda:0")
n, nnz, _, _ = G.SHAPES["reddit"]
for gen in sys.argv[1:]:
    rp, col = G.GENERATORS[gen](n, nnz, seed=0, device=dev)
    E = col.numel(); nw = (n + 15) // 16
    bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
    TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r)
    meta = (rp, col, bp, e2c, e2r)
    res = []
    for D in (64, 128):
        X = torch.randn(n, D, device=dev)
        for x in ("0", "1", "2"):
            os.environ["TCGNN_SDDMM_XCD"] = x
            for _ in range(3): TCGNN.forward_ef(X, *meta)
            TCGNN.kernel_timing(*meta, max_calls=8)
            for _ in range(8): TCGNN.forward_ef(X, *meta)
            km = TCGNN.kernel_timing(*meta); TCGNN.kernel_timing(*meta, max_calls=0)
            res.append("D=%d xcd=%s %.3f" % (D, x, float(np.mean(km))))
    print(gen, " | ".join(res), flush=True)
    TCGNN.clear_plan_cache()
