#!/usr/bin/env python3
"""Times formulations of the dense update X[N,in] @ W[in,out] of the first GCN layer (Reddit: 232965 x 602 x 64, fp32)."""
import os, sys
import torch
import torch.nn.functional as F
dev = torch.device("cuda:0")
N, I, O = 232965, 602, 64
X = torch.randn(N, I, device=dev); W = torch.randn(I, O, device=dev)
Wt = W.t().contiguous()
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
ref = X @ W
cands = {
    "torch.mm(X, W)": lambda: torch.mm(X, W),
    "F.linear(X, Wt)": lambda: F.linear(X, Wt),
    "bmm 64 slabs": lambda: torch.bmm(X[: N // 64 * 64].view(64, N // 64, I), W.expand(64, I, O)),
    "bmm 256 slabs": lambda: torch.bmm(X[: N // 256 * 256].view(256, N // 256, I), W.expand(256, I, O)),
    "(Wt @ X.T).T": lambda: torch.mm(Wt, X.t()).t(),
}
for k, f in cands.items():
    print("%-20s %.3f ms" % (k, t(f)), flush=True)
for pref in ("hipblaslt", "hipblas"):
    try:
        torch.backends.cuda.preferred_blas_library(pref)
        print("preferred %s: mm %.3f ms, linear %.3f ms" % (pref, t(lambda: torch.mm(X, W)), t(lambda: F.linear(X, Wt))), flush=True)
    except Exception as ex:
        print(pref, "failed", ex)

# weight gradient X^T g (602 x 233k x 64) and the second layer's products, under both libraries
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tc-gnn_atc23_amd"))
import tcgnn_layers as L
g = torch.randn(N, O, device=dev); H = torch.randn(N, O, device=dev); W2 = torch.randn(O, 41, device=dev); g2 = torch.randn(N, 41, device=dev)
W2t = W2.t().contiguous()
for pref in ("hipblaslt", "hipblas"):
    torch.backends.cuda.preferred_blas_library(pref)
    print("%s: tall_tn_mm(X, g) %.3f | mm(X.t(), g) %.3f | H@W2 %.3f | linear(H, W2t) %.3f | g2@W2.t() %.3f | tall_tn_mm(H, g2) %.3f" % (
        pref, t(lambda: L.tall_tn_mm(X, g)), t(lambda: torch.mm(X.t(), g)), t(lambda: torch.mm(H, W2)), t(lambda: F.linear(H, W2t)),
        t(lambda: torch.mm(g2, W2.t())), t(lambda: L.tall_tn_mm(H, g2))), flush=True)
