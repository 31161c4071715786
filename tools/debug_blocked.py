import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd"), os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import numpy as np, torch
import TCGNN, tcgnn_capi as c, graphs
dev = torch.device("cuda:0")
rp, col = graphs.uniform_graph(16448, 180, seed=12)
n = len(rp) - 1
bp, e2c, e2r, _ = graphs.host_sgt(rp, col)
meta = [torch.from_numpy(a).to(dev) for a in (rp, col, bp, e2c, e2r)]
for D in (16, 32, 48, 64, 96, 128):
    X = torch.randn(n, D, device=dev)
    out = {}
    for mode in (1, 2):
        c.lib.tcgnn_set_spmm_mode(mode)
        out[mode] = TCGNN.forward(X, *meta)[0].cpu().numpy()
    d = np.abs(out[1] - out[2])
    badrows = np.where(d.max(1) > 1e-3)[0]
    print("D=%d bad rows %d of %d; first bad rows %s; bad windows %d of %d" % (D, len(badrows), n, badrows[:20], len(np.unique(badrows // 16)), (n + 15) // 16))
    if len(badrows):
        r = badrows[0]
        print(" row", r, "plain", out[1][r, :6], "blocked", out[2][r, :6], "ratio", out[2][r, :6] / out[1][r, :6])
        # is the blocked value a partial sum (missing tiles)? compare magnitudes
        print(" mean |plain|", np.abs(out[1]).mean(), "mean |blocked|", np.abs(out[2]).mean())
        w = r // 16
        print(" window", w, "rows bad in window:", [int(x) for x in np.where(d[16*w:16*w+16].max(1) > 1e-3)[0]], "cols bad:", [int(x) for x in np.where(d[16*w:16*w+16].max(0) > 1e-3)[0]])
c.lib.tcgnn_set_spmm_mode(0)
