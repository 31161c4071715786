"""Times the dense updates around the SpMM at the Reddit GCN shapes (torch.mm, fp32) - is the BLAS choice sane?"""
import torch
N = 232965
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps
X = torch.randn(N, 602, device="cuda"); W = torch.randn(602, 64, device="cuda"); g = torch.randn(N, 64, device="cuda")
W2 = torch.randn(64, 41, device="cuda"); g2 = torch.randn(N, 41, device="cuda"); H = torch.randn(N, 64, device="cuda")
for lib in ("default", "hipblaslt", "hipblas"):
    if lib != "default": torch.backends.cuda.preferred_blas_library(lib)
    print(lib, "XW %.3f  XtG %.3f  GWt %.3f  HW2 %.3f  HtG2 %.3f  G2W2t %.3f ms" % (
        t(lambda: torch.mm(X, W)), t(lambda: torch.mm(X.t(), g)), t(lambda: torch.mm(g, W.t())),
        t(lambda: torch.mm(H, W2)), t(lambda: torch.mm(H.t(), g2)), t(lambda: torch.mm(g2, W2.t()))))
print("read X once at 6.5 TB/s: %.3f ms" % (X.numel() * 4 / 6.5e12 * 1e3))
lab = torch.ones(N, dtype=torch.long, device="cuda"); out = torch.randn(N, 41, device="cuda", requires_grad=True)
import torch.nn.functional as F
def lossfb():
    l = F.nll_loss(F.log_softmax(out, dim=1), lab); l.backward(); out.grad = None
print("log_softmax+nll fwd+bwd %.3f ms" % t(lossfb))
