import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import numpy as np, torch
import TCGNN, tcgnn_graph as G
dev = torch.device("cuda:0")
n, nnz, _, _ = G.SHAPES["reddit"]
rp, col = G.synthetic_csr(n, nnz, seed=0, device=dev)
E = col.numel(); nw = (n + 15) // 16
bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r)
meta = (rp, col, bp, e2c, e2r)
X = torch.randn(n, 64, device=dev)
TCGNN.forward_ef(X, *meta); TCGNN.kernel_timing(*meta, max_calls=10)
for _ in range(10): TCGNN.forward_ef(X, *meta)
print("ABLATE=%s sddmm D=64 kernel ms: %.3f" % (os.environ.get("TCGNN_SDDMM_ABLATE", "0"), np.median(TCGNN.kernel_timing(*meta))))
