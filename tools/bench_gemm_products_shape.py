#!/usr/bin/env python3
"""The layers' tall products at the ogbn-products shape (N = 2 449 029): both BLAS libraries, both layouts, slab counts of X^T G -
the measurements behind tcgnn_layers choosing per shape at run time instead of by rule."""
import torch, warnings, sys, os
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tc-gnn_atc23_amd"))
warnings.simplefilter("ignore")
dev = torch.device("cuda:0")
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
def split(A, B, parts):
    n = A.shape[0]; m = n // parts * parts
    out = torch.bmm(A[:m].reshape(parts, m // parts, -1).transpose(1, 2), B[:m].reshape(parts, m // parts, -1)).sum(0)
    return out + A[m:].t() @ B[m:] if m < n else out
N = 2449029
for K, M in ((100, 128), (128, 47), (128, 128), (47, 128)):
    A = torch.randn(N, K, device=dev); W = torch.randn(K, M, device=dev); Wt = W.t().contiguous(); G = torch.randn(N, M, device=dev)
    res = {}
    for lib in ("hipblaslt", "hipblas"):
        torch.backends.cuda.preferred_blas_library(lib)
        res[lib] = (t(lambda: torch.mm(A, W)), t(lambda: F.linear(A, Wt)), t(lambda: F.linear(G, W)), t(lambda: split(A, G, 64)), t(lambda: split(A, G, 256)))
    torch.backends.cuda.preferred_blas_library("hipblaslt")
    print("N=%d K=%d M=%d | A W: lt mm %.3f, lt linear %.3f, roc linear %.3f | G W^T: lt %.3f roc %.3f | A^T G: lt64 %.3f lt256 %.3f roc64 %.3f roc256 %.3f" % (
        N, K, M, res["hipblaslt"][0], res["hipblaslt"][1], res["hipblas"][1], res["hipblaslt"][2], res["hipblas"][2], res["hipblaslt"][3], res["hipblaslt"][4], res["hipblas"][3], res["hipblas"][4]), flush=True)
    del A, W, Wt, G
