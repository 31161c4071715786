"""Layer library over the TCGNN operator API - the host-side mirror of the reference's gnn_conv.py.

Same class names, constructor arguments, forward signatures and gradient formulas as
gnn_conv.py:26-247, so a model written against the reference runs against this file:

    TCGNNFunction_SAG   Y = A X                         bwd: dX = A dY                 (:26-49)
    TCGNNFunction       Y = A (X W)                     bwd: G = A dY; dX = G W^T; dW = X^T G   (:52-85)
    TCGNNFunction_GIN   Y = (A X) W                     bwd: dW = (A X)^T dY; dX = A (dY W^T)   (:87-113)
    TCGNNFunction_AGNN  H = X W; ef = sddmm(H); att = (ef[:,None] @ a)^T; Y = A_att H         (:115-158)
                        bwd: G = A_att dY; dX = G W^T; dW = X^T G;
                             da = (sddmm(dY)[None,:] @ col[:,None].float())^T
    SAG / GCNConv / GINConv / AGNNConv modules, n_heads = 1                                    (:10, :167-247)

The reference's backward uses A, not A^T (it assumes a symmetric graph) and does not propagate
through ef into H; both are reproduced because the golden fixtures captured from gnn_conv.py
(tests/golden/layers_n200.npz) pin exactly that.

The operators come from `backend()`: the TCGNN module of this package (HIP kernels).  Tests on a
machine without a GPU may install another object with the same three functions via set_backend().
"""
import math
import os
import sys
import time
import warnings

import torch
import torch.nn.functional as F

n_heads = 1  # gnn_conv.py:10
USE_FUSED_AGNN = True  # tests switch it off to compare with the separate calls

_backend = None


def set_backend(module):
    global _backend
    _backend = module


def backend():
    global _backend
    if _backend is None:
        import TCGNN  # the drop-in extension module of this package; raises if the library is missing
        _backend = TCGNN
    return _backend


_SPLIT_ROWS = 1 << 15   # below this the plain products are fine

# The layers' own GEMMs are tall and thin (233k x 602 x 64, 2.4M x 100 x 128 ...) and neither BLAS library's heuristics are
# reliable there: on MI355X rocBLAS with the small operand K-contiguous runs Reddit's X W in 0.23 ms against hipBLASLt's
# 0.37 but loses 0.99 : 0.93 at the ogbn-products shape; the weight gradient as a 256-slab batched product takes 0.18 ms
# in rocBLAS at Reddit's shape and 1.35 against hipBLASLt's 0.67 at products'; K = 16 is 3-5x slower in rocBLAS
# (tools/bench_tall_gemm_grid.py, bench_weight_grad.py, bench_gemm_products_shape.py).  So there is a table (product, shape) ->
# candidate, filled ONLY by an explicit tune() call (r1 VERDICT: measuring inside autograd Functions put hidden synchronisations
# in the first epochs and made the choice differ from run to run); without it every product takes candidate 0, torch's default.
# All candidates are fp32 products of the same operands (they differ in summation order only).  A rocBLAS candidate switches
# torch's process-wide BLAS preference for the duration of its call and restores it (torch offers no per-call selector).
_blas_switch = None   # (set, rocblas, default) once probed; False: this torch build has no such switch
_tuned = {}           # (product, N, K, M, dtype) -> index of the winning candidate


def _probe_blas_switch():
    global _blas_switch
    try:
        # (the setter's one-off "experimental feature" notice is written to fd 2 by the C++ logger: silenced for the probe)
        sys.stderr.flush()
        saved, null = os.dup(2), os.open(os.devnull, os.O_WRONLY)
        try:
            os.dup2(null, 2)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                default = torch.backends.cuda.preferred_blas_library()
                rocblas = torch.backends.cuda.preferred_blas_library("hipblas")
                torch.backends.cuda.preferred_blas_library(default)
        finally:
            os.dup2(saved, 2); os.close(saved); os.close(null)
        setter = getattr(torch._C, "_set_blas_preferred_backend", None) or torch.backends.cuda.preferred_blas_library
        _blas_switch = (setter, rocblas, default)
    except Exception:
        _blas_switch = False


if torch.cuda.is_available():
    _probe_blas_switch()   # here, not inside an autograd Function: C++ warnings raised there surface after the filter is gone


def _with_rocblas(fn, *args):
    """fn(*args) with torch's BLAS preference set to rocBLAS for the call (host-side state, restored on exit; the epochs of
    small graphs are launch-bound, so the switch is two plain calls, not a context manager)."""
    if _blas_switch is None:
        _probe_blas_switch()
    if not _blas_switch:
        return fn(*args)
    setter, rocblas, default = _blas_switch
    setter(rocblas)
    try:
        return fn(*args)
    finally:
        setter(default)


def _slabs_tn(A, B, parts, rocblas):
    """A^T B as `parts` slab products (one batched GEMM) and a fixed-order sum - deterministic for a given `parts`."""
    n = A.shape[0]
    m = n // parts * parts
    a3 = A[:m].reshape(parts, m // parts, -1).transpose(1, 2)
    b3 = B[:m].reshape(parts, m // parts, -1)
    out = (_with_rocblas(torch.bmm, a3, b3) if rocblas else torch.bmm(a3, b3)).sum(0)
    if m < n:
        out = out + torch.mm(A[m:].t(), B[m:])
    return out


_CANDIDATES = {
    # A [N, K] @ W [K, M]
    "mm": (lambda A, W: torch.mm(A, W),
           lambda A, W: _with_rocblas(F.linear, A, W.t().contiguous())),
    # A [N, K] @ Bt [M, K]^T
    "nt": (lambda A, Bt: F.linear(A, Bt),
           lambda A, Bt: _with_rocblas(F.linear, A, Bt)),
    # A [N, K]^T @ B [N, M]
    "tn": (lambda A, B: _slabs_tn(A, B, 64, False),
           lambda A, B: _slabs_tn(A, B, 256, False),
           lambda A, B: _slabs_tn(A, B, 128, False),
           lambda A, B: _slabs_tn(A, B, 256, True)),
}


def _tall(product, A, B):
    """The candidate `tune()` recorded for this (product, shape), else candidate 0 (torch's default library and layout).
    Nothing is measured here: a call inside an autograd Function never synchronises and never decides anything, so a
    process that does not call tune() runs the same code on every run."""
    cands = _CANDIDATES[product]
    key = (product, A.shape[0], A.shape[1], B.shape[0] if product == "nt" else B.shape[1], A.dtype)
    return cands[_tuned.get(key, 0)](A, B)


def tune(shapes, device=None, dtype=torch.float32, margin=0.05):
    """Fill the candidate table for the tall products of a model - an explicit step, outside autograd and outside any graph
    capture (the harness and bench.py call it before their dry epochs).  shapes: iterable of (product, N, K, M) with product
    in {"mm", "nt", "tn"}; tune_layers() lists them for a layer stack.  Every candidate runs on random operands of that
    shape (once to warm up, then twice under HIP events); a candidate replaces the default only if it is more than `margin`
    faster, so near-ties (run-to-run noise) always resolve to the default.  Returns {key: (winner, times in ms)}."""
    device = torch.device("cuda") if device is None else torch.device(device)
    assert not torch.cuda.is_current_stream_capturing(), "tune() synchronises: call it before capturing a graph"
    report = {}
    g = torch.Generator(device=device).manual_seed(0)
    for product, n, k, m in shapes:
        key = (product, int(n), int(k), int(m), dtype)
        if n < _SPLIT_ROWS or key in report:
            continue
        A = torch.randn(n, k, device=device, dtype=dtype, generator=g)
        B = torch.randn((m, k) if product == "nt" else ((k, m) if product == "mm" else (n, m)), device=device, dtype=dtype, generator=g)
        times = []
        for fn in _CANDIDATES[product]:
            fn(A, B)   # warm-up (library initialisation, workspace)
            t0, t1, t2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            t0.record(); fn(A, B); t1.record(); fn(A, B); t2.record(); t2.synchronize()
            times.append(min(t0.elapsed_time(t1), t1.elapsed_time(t2)))
        best = min(range(len(times)), key=times.__getitem__)
        if times[best] > (1.0 - margin) * times[0]:
            best = 0
        _tuned[key] = best
        report[key] = (best, times)
        del A, B
    return report


def tune_layers(num_nodes, dims, first_layer_needs_input_grad=False):
    """The (product, N, K, M) list of a layer stack with widths dims = [in, hidden, ..., classes]: X W forward, dY W^T and
    X^T G backward per layer (gnn_conv.py:59-68, 83-84)."""
    shapes = []
    for i, (a, b) in enumerate(zip(dims[:-1], dims[1:])):
        shapes.append(("mm", num_nodes, a, b))
        shapes.append(("tn", num_nodes, a, b))
        if i > 0 or first_layer_needs_input_grad:
            shapes.append(("nt", num_nodes, b, a))
    return shapes


def tall_tn_mm(A, B):
    """A^T B for tall operands ([N, K]^T [N, M], N >> K, M): the weight-gradient products of gnn_conv.py:84,111,147.
    The BLAS libraries run this shape as one long reduction per output tile (0.59 ms for 233k x 602 x 64 on MI355X); cut into
    slabs it takes 0.18-0.24 ms, which slab count and library being measured per shape (see above)."""
    if not A.is_cuda or A.shape[0] < _SPLIT_ROWS:
        return torch.mm(A.t(), B)
    return _tall("tn", A, B)


def tall_nt_mm(A, Bt):
    """A Bt^T for a tall A ([N, K] [M, K]^T, N >> K, M): dY W^T of gnn_conv.py:83; library measured per shape."""
    if not A.is_cuda or A.shape[0] < _SPLIT_ROWS:
        return F.linear(A, Bt)
    return _tall("nt", A, Bt)


def tall_mm(A, B):
    """A B for a tall A ([N, K] [K, M]): the dense updates X W of gnn_conv.py:59-68; library and layout measured per shape."""
    if not A.is_cuda or A.shape[0] < _SPLIT_ROWS:
        return torch.mm(A, B)
    return _tall("mm", A, B)


class _DenseUpdate(torch.autograd.Function):
    """X W with the tall products above in both directions (autograd's own backward of torch.mm takes the plain X^T G)."""

    @staticmethod
    def forward(ctx, X, W):
        ctx.save_for_backward(X, W)
        return tall_mm(X, W)

    @staticmethod
    def backward(ctx, g):
        X, W = ctx.saved_tensors
        g = g.contiguous()
        d_x = tall_nt_mm(g, W) if ctx.needs_input_grad[0] else None
        d_w = tall_tn_mm(X, g) if ctx.needs_input_grad[1] else None
        return d_x, d_w


def dense_update(X, W):
    """Differentiable X W for callers outside the layer Functions below (tcgnn_shard.ShardedGCN)."""
    return _DenseUpdate.apply(X, W)


class TCGNNFunction_SAG(torch.autograd.Function):
    """Pure neighbour aggregation."""

    @staticmethod
    def forward(ctx, X, row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow):
        ctx.meta = (row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow)
        return backend().forward(X, *ctx.meta)[0]

    @staticmethod
    def backward(ctx, d_output):
        d_input = backend().forward(d_output.contiguous(), *ctx.meta)[0] if ctx.needs_input_grad[0] else None
        return (d_input,) + (None,) * 5


class TCGNNFunction(torch.autograd.Function):
    """GCN layer: dense update first, aggregation second."""

    @staticmethod
    def forward(ctx, X, weights, row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow, fuse_relu=False, aggregate_first=False):
        """fuse_relu (not in the reference): the ReLU that follows the layer (main_tcgnn.py:100-139) runs in the
        SpMM kernel's stores, and its backward mask is applied to dY while dY is staged - relu(layer(x)) without the two
        element-wise passes over N x D.  Same values as F.relu(layer(x)), bit for bit.
        aggregate_first (f3, opt-in): A (X W) evaluated as (A X) W in ONE launch (backend().forward_gemm: the dense update in the
        aggregation kernel's epilogue) - the same matrix, rounded at a different point (X, not X W, meets the 10-bit operand
        rounding).  The backward pass is unchanged: it only needs X, W and dY."""
        ctx.meta = (row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow)
        ctx.fused = bool(fuse_relu) and hasattr(backend(), "forward_fused")
        if aggregate_first and not fuse_relu and hasattr(backend(), "forward_gemm") and max(weights.shape) <= 128:
            ctx.fused = False
            ctx.masked = False
            ctx.save_for_backward(X, weights)
            return backend().forward_gemm(X, weights, *ctx.meta)[0]
        if ctx.fused:
            Y = backend().forward_fused(tall_mm(X, weights), *ctx.meta, relu=True)[0]
            ctx.save_for_backward(X, weights, Y)
            return Y
        Y = backend().forward(tall_mm(X, weights), *ctx.meta)[0]
        # a backend without the fused entry point: the same ReLU as a plain step - and, autograd being off inside a Function,
        # its backward mask applied by hand below (ctx.masked), or dX / dW would silently miss it
        ctx.masked = bool(fuse_relu)
        if ctx.masked:
            Y = torch.relu(Y)
            ctx.save_for_backward(X, weights, Y)
        else:
            ctx.save_for_backward(X, weights)
        return Y

    @staticmethod
    def backward(ctx, d_output):
        if ctx.fused:
            X, weights, Y = ctx.saved_tensors
            g = backend().forward_fused(d_output.contiguous(), *ctx.meta, gate=Y)[0]
        elif getattr(ctx, "masked", False):
            X, weights, Y = ctx.saved_tensors
            g = backend().forward((d_output * (Y > 0)).contiguous(), *ctx.meta)[0]
        else:
            X, weights = ctx.saved_tensors
            g = backend().forward(d_output.contiguous(), *ctx.meta)[0]
        # the input features of the first layer need no gradient: skip their N x in_dim product
        d_input = tall_nt_mm(g, weights) if ctx.needs_input_grad[0] else None
        return (d_input, tall_tn_mm(X, g)) + (None,) * 7


class TCGNNFunction_GIN(torch.autograd.Function):
    """GIN layer: aggregation first, dense update second."""

    @staticmethod
    def forward(ctx, X, weights, row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow):
        ctx.meta = (row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow)
        if not any(ctx.needs_input_grad[:2]) and hasattr(backend(), "forward_gemm") and max(weights.shape) <= 128:
            # inference: the dense update runs in the aggregation kernel's epilogue (f3), A X never reaches memory.  Training
            # keeps the two steps: the weight gradient is (A X)^T dY (gnn_conv.py:111) and needs A X
            return backend().forward_gemm(X, weights, *ctx.meta)[0]
        agg = backend().forward(X, *ctx.meta)[0]
        ctx.save_for_backward(agg, weights)
        return tall_mm(agg, weights)

    @staticmethod
    def backward(ctx, d_output):
        agg, weights = ctx.saved_tensors
        d_weights = tall_tn_mm(agg, d_output.contiguous())
        d_input = None
        if ctx.needs_input_grad[0]:
            d_input = backend().forward(tall_nt_mm(d_output.contiguous(), weights), *ctx.meta)[0]
        return (d_input, d_weights) + (None,) * 5


class TCGNNFunction_AGNN(torch.autograd.Function):
    """AGNN layer: edge scores by SDDMM, then edge-weighted aggregation.

    When the backend offers the fused products (TCGNN.agnn_fused_*: one gather of the neighbour rows
    for both, no [E]-sized attention / gradient tensors) they are used; the values are those of the
    separate calls below, which remain the path for anything the fused kernels do not cover."""

    @staticmethod
    def forward(ctx, X, weights, attention_w, row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow):
        meta = (row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow)
        H = tall_mm(X, weights)
        b = backend()
        ctx.meta = meta
        ctx.fused = bool(USE_FUSED_AGNN and attention_w.numel() == 1 and hasattr(b, "agnn_fused_forward")
                         and b.agnn_fused_supported(H, *meta))
        if ctx.fused:
            w1 = attention_w.detach().reshape(1).contiguous()
            out, ef, ef_absmax = b.agnn_fused_forward(H, row_pointers, column_index, w1, blockPartition, edgeToColumn, edgeToRow)
            ctx.save_for_backward(X, weights, w1, ef, ef_absmax)
            return out
        ef = b.forward_ef(H, *meta)[0]
        # reference: mm(ef[:, None], attention_w).T -> [n_heads, E]; a k = 1 matmul is one product per
        # element, so the broadcast below is value-identical and avoids a degenerate GEMM launch
        att = (attention_w.reshape(-1, 1) * ef.unsqueeze(0)).contiguous()
        out = b.forward_AGNN(H, row_pointers, column_index, att, blockPartition, edgeToColumn, edgeToRow)[0]
        ctx.save_for_backward(X, weights, att)
        return out

    @staticmethod
    def backward(ctx, d_output):
        row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow = ctx.meta
        d_output = d_output.contiguous()
        b = backend()
        if ctx.fused:
            X, weights, w1, ef, ef_absmax = ctx.saved_tensors
            g, d_w = b.agnn_fused_backward(d_output, row_pointers, column_index, w1, ef, ef_absmax, blockPartition, edgeToColumn, edgeToRow)
            d_attention_w = d_w.reshape(1, n_heads)
        else:
            X, weights, att = ctx.saved_tensors
            g = b.forward_AGNN(d_output, row_pointers, column_index, att, blockPartition, edgeToColumn, edgeToRow)[0]
            d_att = b.forward_ef(d_output, *ctx.meta)[0]
            # reference: mm(d_att[None, :].expand(n_heads, -1), column_index[:, None].float()).T, i.e. the
            # dot product <d_att, column_index> per head.  As an [n_heads, E] x [E] matrix-vector product:
            # the 1 x E x 1 GEMM form falls off rocBLAS' fast paths at E ~ 1e8 (30 s per call measured).
            d_attention_w = torch.mv(d_att[None, :].expand(n_heads, -1), column_index.float()).reshape(1, n_heads)
        d_input = tall_nt_mm(g, weights) if ctx.needs_input_grad[0] else None
        d_weights = tall_tn_mm(X, g)
        return (d_input, d_weights, d_attention_w) + (None,) * 5


class SAG(torch.nn.Module):
    """Holds the graph metadata; profile() times `num_rounds` bare aggregations (gnn_conv.py:179-190,
    the single-kernel benchmark of 2_tcgnn_single_kernel.py) and prints the line 1_log2csv.py scrapes."""

    def __init__(self, row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow):
        super().__init__()
        self.row_pointers = row_pointers
        self.column_index = column_index
        self.blockPartition = blockPartition
        self.edgeToColumn = edgeToColumn
        self.edgeToRow = edgeToRow
        # build the device plan now, like the rest of the preprocessing: not inside the first timed call
        prefetch = getattr(backend(), "plan_info", None)
        if prefetch is not None and row_pointers.is_cuda:
            prefetch(row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow)

    def forward(self, X):
        return TCGNNFunction_SAG.apply(X, self.row_pointers, self.column_index, self.blockPartition, self.edgeToColumn, self.edgeToRow)

    def profile(self, X, num_rounds=200):
        torch.cuda.synchronize()
        start = time.perf_counter()
        for _ in range(num_rounds):
            self.forward(X)
        torch.cuda.synchronize()
        avg_ms = (time.perf_counter() - start) * 1e3 / num_rounds
        print("=> SAG profiling avg (ms): {:.3f}".format(avg_ms))
        print()
        return avg_ms


class GCNConv(torch.nn.Module):
    def __init__(self, input_dim, output_dim):
        super().__init__()
        self.weights = torch.nn.Parameter(torch.randn(input_dim, output_dim))  # unscaled, as gnn_conv.py:195

    def reset_parameters(self):
        bound = 1.0 / math.sqrt(self.weights.size(1))
        self.weights.data.uniform_(-bound, bound)

    def forward(self, X, row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow, fuse_relu=False, aggregate_first=False):
        if aggregate_first and not fuse_relu:
            return TCGNNFunction.apply(X, self.weights, row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow, False, True)
        if fuse_relu and hasattr(backend(), "forward_fused"):
            return TCGNNFunction.apply(X, self.weights, row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow, True)
        y = TCGNNFunction.apply(X, self.weights, row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow)
        return torch.relu(y) if fuse_relu else y


# GINConv's one-launch inference form (tcgnn_spmm_gemm); see GINConv.forward
GIN_FUSED_INFERENCE = os.environ.get("TCGNN_GIN_FUSED_INFERENCE", "1") != "0"


class GINConv(torch.nn.Module):
    def __init__(self, input_dim, output_dim):
        super().__init__()
        self.weights = torch.nn.Parameter(torch.randn(input_dim, output_dim))

    def forward(self, X, row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow):
        # inference (no gradient will be asked for: grad mode off, or neither operand requires one) takes the one-launch form
        # (A X) W of tcgnn_spmm_gemm - decided HERE, from the grad mode: inside a Function `needs_input_grad` only mirrors
        # `requires_grad`, so under model.eval() + torch.no_grad() with ordinary Parameters it never fired (r2 ADVICE).
        # forward_gemm is inference-only: it has no backward.
        # The shortcut sums in another order than forward + torch.mm (fp32 matrix pipe behind the aggregation; on 64-wide inputs two
        # passes added atomically): equal within accumulation noise, not bit for bit - `GIN_FUSED_INFERENCE = False` (or
        # TCGNN_GIN_FUSED_INFERENCE=0) keeps eval on the training path's arithmetic (ADVICE r03;
        # tests/test_gpu_parity.py::test_gin_layer_eval_and_train_forward_agree).
        b = backend()
        if (GIN_FUSED_INFERENCE and (not torch.is_grad_enabled() or not (X.requires_grad or self.weights.requires_grad)) and X.is_cuda
                and hasattr(b, "forward_gemm") and max(self.weights.shape) <= getattr(b, "GEMM_FUSED_MAX_DIM", 128)):
            return b.forward_gemm(X, self.weights.detach(), row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow)[0]
        return TCGNNFunction_GIN.apply(X, self.weights, row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow)


class AGNNConv(torch.nn.Module):
    def __init__(self, input_dim, output_dim):
        super().__init__()
        self.weights = torch.nn.Parameter(torch.randn(input_dim, output_dim))
        self.attention_w = torch.nn.Parameter(torch.randn(1, n_heads))
        self.reset_parameters()

    def reset_parameters(self):
        bound = 1.0 / math.sqrt(self.weights.size(1))
        self.weights.data.uniform_(-bound, bound)

    def forward(self, X, row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow):
        return TCGNNFunction_AGNN.apply(X, self.weights, self.attention_w, row_pointers, column_index, blockPartition,
                                        edgeToColumn, edgeToRow)
