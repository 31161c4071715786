#!/usr/bin/env python3
"""Time the LDS-resident range SpMM (mode 3) on the Reddit shape under the TCGNN_LDS_DBG switches (set in the environment; the
phase-timer kernels exist only in a library built with `make -C tc-gnn_atc23_amd/csrc DEBUG_TIMERS=1`)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import numpy as np, torch
import TCGNN, tcgnn_graph as G, tcgnn_capi as c
D = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
n, nnz, _, _ = G.SHAPES["reddit"]
rp, col = G.GENERATORS[os.environ.get("GEN", "uniform")](n, nnz, seed=0, device=dev)   # GEN=uniform|rmat|sbm|sbm_reddit ...
E = col.numel(); nw = (n + 15) // 16
bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
fd = os.open(os.devnull, os.O_WRONLY); sv = os.dup(1); os.dup2(fd, 1)
TCGNN.preprocess_gpu(col, rp, n, 16, 8, bp, e2c, e2r)
os.dup2(sv, 1)
meta = (rp, col, bp, e2c, e2r)
X = torch.randn(n, D, device=dev)
c.lib.tcgnn_set_spmm_mode(int(os.environ.get("MODE", "3")))   # MODE=0: the automatic choice
TCGNN.forward(X, *meta); TCGNN.kernel_timing(*meta, max_calls=10)
for _ in range(int(os.environ.get("WARM", "100"))): TCGNN.forward(X, *meta)   # (past the ~50-launch transient that follows an idle stretch)
TCGNN.kernel_timing(*meta)
for _ in range(30): TCGNN.forward(X, *meta)
t = TCGNN.kernel_timing(*meta)
if int(os.environ.get("TCGNN_LDS_DBG", "0")) & 16:
    y = TCGNN.forward(X, *meta)[0].flatten()[:128].cpu().numpy()
    nr = TCGNN.plan_info(*meta)["lds_ranges"]
    print("cycles per range (%d ranges), rows = wavefronts of workgroup 0, columns = fill issue / multiply / wait / barrier (mean cycles), pad refills (total), longest multiply, longest wait, tiles (total)" % nr)
    y = y.reshape(16, 8); y[:, :4] /= nr
    print(np.round(y).astype(int))
print("dbg=%s D=%d: %.3f ms (min %.3f)  %s  %s" % (os.environ.get("TCGNN_LDS_DBG", "0"), D, np.median(t), np.min(t), TCGNN.last_kernel(*meta), TCGNN.plan_info(*meta)))
