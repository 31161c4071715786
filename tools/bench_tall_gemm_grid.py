#!/usr/bin/env python3
"""A[N, K] @ W[K, M] for tall A: torch's default library against rocBLAS with W K-contiguous, over the shapes the layers see."""
import torch, warnings
import torch.nn.functional as F
dev = torch.device("cuda:0")
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
warnings.simplefilter("ignore")
print("N,K,M,default_mm_ms,rocblas_linear_ms,ratio")
for N in (50000, 100000, 400000, 1900000):
    for K in (16, 41, 64, 96, 128, 602):
        for M in (7, 16, 41, 64, 128):
            if N * K > 6e8: continue
            A = torch.randn(N, K, device=dev); W = torch.randn(K, M, device=dev); Wt = W.t().contiguous()
            torch.backends.cuda.preferred_blas_library("hipblaslt"); a = t(lambda: torch.mm(A, W))
            torch.backends.cuda.preferred_blas_library("hipblas"); b = t(lambda: F.linear(A, Wt))
            torch.backends.cuda.preferred_blas_library("hipblaslt")
            print("%d,%d,%d,%.4f,%.4f,%.2f" % (N, K, M, a, b, a / b), flush=True)
            del A, W, Wt
