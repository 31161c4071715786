"""Alternatives for the tall reductions dW = X^T G at the Reddit GCN shapes (fp32)."""
import torch
N = 232965
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps
X = torch.randn(N, 602, device="cuda"); g = torch.randn(N, 64, device="cuda")
H = torch.randn(N, 64, device="cuda"); g2 = torch.randn(N, 41, device="cuda")
ref1 = torch.mm(X.t(), g); ref2 = torch.mm(H.t(), g2)
def splitk(A, B, P):
    n = A.shape[0]; m = n // P * P
    out = torch.bmm(A[:m].view(P, m // P, -1).transpose(1, 2), B[:m].view(P, m // P, -1)).sum(0)
    if m < n: out = out + torch.mm(A[m:].t(), B[m:])
    return out
for name, A, B, ref in (("XtG", X, g, ref1), ("HtG2", H, g2, ref2)):
    print(name, "mm(A.t,B) %.3f   (B.t@A).t %.3f" % (t(lambda: torch.mm(A.t(), B)), t(lambda: torch.mm(B.t(), A).t())), end="  ")
    for P in (16, 64, 256, 1024):
        y = splitk(A, B, P)
        print("splitk%d %.3f (err %.1e)" % (P, t(lambda: splitk(A, B, P)), ((y - ref).abs().max() / ref.abs().max()).item()), end="  ")
    print()
W = torch.randn(602, 64, device="cuda")
print("XW mm %.3f  addmm-free F.linear(X, Wt) %.3f" % (t(lambda: torch.mm(X, W)), t(lambda: torch.nn.functional.linear(X, W.t().contiguous()))))
