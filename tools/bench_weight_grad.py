#!/usr/bin/env python3
"""X^T G for tall operands (the weight gradients): slab counts of the batched split-K form, under both BLAS libraries."""
import warnings
import torch
dev = torch.device("cuda:0")
warnings.simplefilter("ignore")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
def split(A, B, parts):
    n = A.shape[0]; m = n // parts * parts
    out = torch.bmm(A[:m].reshape(parts, m // parts, -1).transpose(1, 2), B[:m].reshape(parts, m // parts, -1)).sum(0)
    return out + A[m:].t() @ B[m:] if m < n else out
for (N, K, M) in ((232965, 602, 64), (232965, 64, 41), (232965, 64, 64), (410236, 96, 16)):
    A = torch.randn(N, K, device=dev); B = torch.randn(N, M, device=dev)
    for lib in ("hipblaslt", "hipblas"):
        torch.backends.cuda.preferred_blas_library(lib)
        print("%dx%dx%d %-9s" % (N, K, M, lib), " ".join("%d: %.3f" % (p, t(lambda: split(A, B, p))) for p in (16, 32, 64, 128, 256, 512)), "| mm %.3f" % t(lambda: A.t() @ B), flush=True)
torch.backends.cuda.preferred_blas_library("hipblaslt")
