"""`TCGNN` - drop-in replacement of the reference's PyTorch extension module of the same name.

Same seven names, same positional signatures, same return shapes as the pybind11 module built from
TCGNN_conv/TCGNN.cpp:260-272, so the reference's gnn_conv.py / main_tcgnn.py import and call it
unchanged (`import TCGNN`):

    preprocess(edgeList, nodePointer, num_nodes, blockSize_h, blockSize_w,
               blockPartition, edgeToColumn, edgeToRow) -> None        TCGNN.cpp:172
    preprocess_gpu(... same, CUDA tensors ...)              -> None        TCGNN.cpp:229
    forward(input, nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow) -> [Y]    :63
    forward_ef(...same six...)                                                    -> [ef]   :126
    forward_AGNN(input, nodePointer, edgeList, edgeAttention, blockPartition,
                 edgeToColumn, edgeToRow)                                         -> [Y]    :93
    backward = forward, backward_ef = forward_ef                                            :270-271

Behind it sits the C ABI of include/tcgnn.h (libtcgnn_hip.so, hand-written gfx950 kernels).  Torch
is only plumbing here: device memory, the current HIP stream, tensor lifetime.  There is no CPU or
eager fallback; a missing library fails at import.

Differences from the reference that a caller can observe (all supersets):
  * any embedding_dim is computed in full (the reference leaves columns >= 16*min(D//16, 8) zero),
  * launch errors raise RuntimeError instead of printf + exit(-1) (TCGNN_kernel.cu:211-217),
  * kernels run on torch's current stream (the reference uses the legacy default stream and, for
    forward_AGNN, a leaked stream per call, TCGNN_kernel.cu:245-255),
  * preprocess never writes past the end of blockPartition when num_nodes % blockSize_h == 0.
"""
import collections
import os
import sys

import torch

import tcgnn_capi as _c

__all__ = ["preprocess", "preprocess_gpu", "forward", "forward_ef", "forward_AGNN", "backward", "backward_ef",
           "plan_info", "kernel_timing", "last_kernel", "clear_plan_cache", "set_plan_cache_size", "agnn_fused_supported", "agnn_fused_forward", "agnn_fused_backward",
           "forward_fused", "forward_gemm"]

_plan_cache_size = max(1, int(os.environ.get("TCGNN_PLAN_CACHE_SIZE", "8")))
_plans = collections.OrderedDict()  # key -> (handle, tensors kept alive, device index)
_retired = []                       # evicted plans waiting for the kernels that may still read them: (events, handle, tensors)
_workspaces = {}                    # (device index, stream id) -> uint8 tensor


def set_plan_cache_size(n):
    """Plans (packed tile streams, ~3x the CSR's bytes each) kept per process; the least recently used one beyond this is
    retired.  A mini-batch loop over k graphs wants n >= k.  Also the environment variable TCGNN_PLAN_CACHE_SIZE."""
    global _plan_cache_size
    _plan_cache_size = max(1, int(n))
    _evict()


def _reap(block=False):
    """Destroy retired plans whose last possible reader has finished (stream-ordered: an event per stream this module has
    launched on for that device, recorded at eviction time; nothing is synchronised unless block=True)."""
    keep = []
    for events, handle, tensors in _retired:
        if block:
            for e in events:
                e.synchronize()
        if all(e.query() for e in events):
            _c.lib.tcgnn_plan_destroy(handle)
        else:
            keep.append((events, handle, tensors))
    _retired[:] = keep


def _evict():
    while len(_plans) > _plan_cache_size:
        _, (old, keep, dev_index) = _plans.popitem(last=False)
        events = []
        for (d, stream_id) in list(_workspaces):
            if d == dev_index:   # the streams this module has launched kernels on, on the EVICTED plan's device
                with torch.cuda.device(d):
                    e = torch.cuda.Event()
                    e.record(torch.cuda.ExternalStream(stream_id, device=d) if stream_id else torch.cuda.default_stream(d))
                    events.append(e)
        _retired.append((events, old, keep))
    if _retired:
        _reap()


# ---------------------------------------------------------------- argument checks (TCGNN.cpp:54-56)

def _check_input(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA tensor" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)


def _check_int(t, name):
    if t.dtype != torch.int32:  # libtorch's data<int>() raises the same way in the reference
        raise RuntimeError("expected scalar type Int but found %s (%s)" % (str(t.dtype).replace("torch.", "").capitalize(), name))


def _check_float(t, name):
    if t.dtype != torch.float32:
        raise RuntimeError("expected scalar type Float but found %s (%s)" % (str(t.dtype).replace("torch.", "").capitalize(), name))


def _stream_handle(device):
    return torch.cuda.current_stream(device).cuda_stream


# ---------------------------------------------------------------- plan cache

def _plan_for(nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow):
    """The packed tile stream is a pure function of the five metadata tensors; it is built on the
    device the first time they are seen and reused while they are unchanged (storage address,
    length and in-place version counter).  The cache keeps the tensors alive, so an address can not
    be recycled under a live entry."""
    tensors = (nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow)
    key = tuple((t.data_ptr(), t.numel(), t._version) for t in tensors) + (nodePointer.device.index,)
    hit = _plans.get(key)
    if hit is not None:
        _plans.move_to_end(key)
        return hit[0]
    dev = nodePointer.device
    for t, n in zip(tensors, ("nodePointer", "edgeList", "blockPartition", "edgeToColumn", "edgeToRow")):
        _check_int(t, n)
        if t.device != dev:
            raise RuntimeError("%s is on %s but nodePointer is on %s" % (n, t.device, dev))
    N = nodePointer.numel() - 1
    E = edgeList.numel()
    if N < 0:
        raise RuntimeError("nodePointer must hold num_nodes + 1 entries")
    if edgeToColumn.numel() < E or edgeToRow.numel() < E:
        raise RuntimeError("edgeToColumn / edgeToRow are shorter than edgeList")
    handle = _c._vp()
    with torch.cuda.device(dev):
        st = _c.lib.tcgnn_plan_create(nodePointer.data_ptr(), edgeList.data_ptr(), blockPartition.data_ptr(),
                                      edgeToColumn.data_ptr(), edgeToRow.data_ptr(), N, E, blockPartition.numel(),
                                      _stream_handle(dev), _c.ctypes.byref(handle))
    _c.check(st, "tcgnn_plan_create")
    _plans[key] = (handle, tensors, dev.index)
    _evict()
    return handle


def clear_plan_cache():
    if torch.cuda.is_available():
        for d in {v[2] for v in _plans.values()}:
            torch.cuda.synchronize(d)
    while _plans:
        _, (old, _keep, _d) = _plans.popitem()
        _c.lib.tcgnn_plan_destroy(old)
    _reap(block=True)
    _workspaces.clear()


def plan_info(nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow):
    """Not part of the reference API: statistics of the packed tile stream (dict)."""
    info = _c.PlanInfo()
    _c.check(_c.lib.tcgnn_plan_get_info(_plan_for(nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow),
                                        _c.ctypes.byref(info)), "tcgnn_plan_get_info")
    return {f: getattr(info, f) for f, _ in info._fields_}


def prepare(widths, nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow, edge_valued=False):
    """Not part of the reference API: build, now, what the hot path would otherwise build at its first call of each feature width
    in `widths` (tcgnn_plan_prepare: the cell streams of the LDS-resident kernel where the plan's time model picks it; with
    edge_valued=True also tcgnn_plan_prepare_val: the single-edge stream forward_AGNN's LDS-resident walk reads).  After it no
    forward / backward (/ forward_AGNN) call of those widths synchronises or allocates inside the library, and a call captured into
    a HIP graph takes the walk it would take outside one.  The harness calls it with the model's widths before the dry epochs."""
    plan = _plan_for(nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow)
    dev = nodePointer.device
    with torch.cuda.device(dev):
        for d in sorted({int(w) for w in widths if int(w) >= 1}):
            _c.check(_c.lib.tcgnn_plan_prepare(plan, d, _stream_handle(dev)), "tcgnn_plan_prepare")
            if edge_valued:
                _c.check(_c.lib.tcgnn_plan_prepare_val(plan, d, _stream_handle(dev)), "tcgnn_plan_prepare_val")


def set_plan_modes(nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow, spmm_mode=None, range_guard=None):
    """Not part of the reference API: the walk (tcgnn_plan_set_spmm_mode) and the range-guard level (tcgnn_plan_set_range_guard) of
    THIS graph's plan only; -1 hands a setting back to the process-wide value.  Two graphs of one process may differ."""
    plan = _plan_for(nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow)
    if spmm_mode is not None:
        _c.check(_c.lib.tcgnn_plan_set_spmm_mode(plan, int(spmm_mode)), "tcgnn_plan_set_spmm_mode")
    if range_guard is not None:
        _c.check(_c.lib.tcgnn_plan_set_range_guard(plan, int(range_guard)), "tcgnn_plan_set_range_guard")


def range_mode(device=None):
    """Not part of the reference API: which way the range guard sent the LAST call staged on this device's current stream -
    (wide_x, wide_val): 1 = the fp32 fallback ran (a matrix with a wide dynamic range, include/tcgnn.h "Operand range"), 0 = the
    MFMA path.  Reads the workspace header back (synchronises the stream): a test / diagnosis aid."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    ws = _workspaces.get((dev.index, _stream_handle(dev)))
    if ws is None:
        return (0, 0)
    off = (-ws.data_ptr()) % 256
    a, b = _c._i32(0), _c._i32(0)
    _c.check(_c.lib.tcgnn_range_mode(ws.data_ptr() + off, _stream_handle(dev), _c.ctypes.byref(a), _c.ctypes.byref(b)), "tcgnn_range_mode")
    return (a.value, b.value)


def set_range_guard(level):
    """Not part of the reference API: the range guard's level - 0 off, 1 the SpMM operators only, 2 (default since r04) also SDDMM and
    the fused AGNN pair when a matrix has a few lost elements (patched behind the MFMA kernels), 3 strict: any wide matrix in fp32
    (include/tcgnn.h: tcgnn_set_range_guard)."""
    _c.check(_c.lib.tcgnn_set_range_guard(int(level)), "tcgnn_set_range_guard")


def kernel_timing(nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow, max_calls=None):
    """Not part of the reference API.  kernel_timing(meta..., max_calls=K) arms HIP-event timing of
    the main kernel for the next K calls on this graph; kernel_timing(meta...) (no max_calls) waits
    for them and returns their durations in ms."""
    plan = _plan_for(nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow)
    if max_calls is not None:
        _c.check(_c.lib.tcgnn_plan_set_timing(plan, int(max_calls)), "tcgnn_plan_set_timing")
        return None
    buf = (_c.ctypes.c_float * 4096)()
    n = _c._i32(0)
    _c.check(_c.lib.tcgnn_plan_read_timing(plan, buf, 4096, _c.ctypes.byref(n)), "tcgnn_plan_read_timing")
    return [buf[i] for i in range(n.value)]


def last_kernel(nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow):
    """Not part of the reference API: name of the main kernel the most recent call on this graph launched."""
    return _c.lib.tcgnn_plan_last_kernel(_plan_for(nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow)).decode()


def _workspace(plan, D, device):
    need = _c.lib.tcgnn_workspace_bytes(plan, D)
    key = (device.index, _stream_handle(device))
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < need + 256:
        ws = torch.empty(int(need * 1.25) + 256, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    off = (-ws.data_ptr()) % 256
    return ws.data_ptr() + off, ws.numel() - off


# ---------------------------------------------------------------- sparse-graph translation

def _report(tc_blocks):
    # the reference prints exactly this from C (TCGNN.cpp:225); 1_log2csv.py-style scrapers and the
    # committed logs (logs/RTX3090_GCN.log:1-2) rely on the two lines
    sys.stdout.write("TC_Blocks:\t%d\nExp_Edges:\t%d\n" % (tc_blocks, tc_blocks * 8 * 16))
    sys.stdout.flush()


def preprocess(edgeList, nodePointer, num_nodes, blockSize_h, blockSize_w, blockPartition, edgeToColumn, edgeToRow):
    """Host SGT: fills blockPartition / edgeToColumn / edgeToRow in place (CPU int32 tensors)."""
    names = ("edgeList", "nodePointer", "blockPartition", "edgeToColumn", "edgeToRow")
    for t, n in zip((edgeList, nodePointer, blockPartition, edgeToColumn, edgeToRow), names):
        if not isinstance(t, torch.Tensor):
            raise TypeError("%s must be a torch.Tensor" % n)
        if t.is_cuda:
            raise RuntimeError("%s must be a CPU tensor (use preprocess_gpu for device tensors)" % n)
        _check_int(t, n)
        if not t.is_contiguous():
            raise RuntimeError("%s must be contiguous" % n)
    num_nodes = int(num_nodes)
    if nodePointer.numel() < num_nodes + 1:
        raise RuntimeError("nodePointer holds %d entries, need num_nodes + 1 = %d" % (nodePointer.numel(), num_nodes + 1))
    E = int(nodePointer[num_nodes])
    if edgeList.numel() < E or edgeToColumn.numel() < E or edgeToRow.numel() < E:
        raise RuntimeError("edgeList / edgeToColumn / edgeToRow hold fewer than nodePointer[num_nodes] = %d entries" % E)
    n = _c._i64(0)
    st = _c.lib.tcgnn_preprocess(edgeList.data_ptr(), nodePointer.data_ptr(), num_nodes, int(blockSize_h), int(blockSize_w),
                                 blockPartition.data_ptr(), blockPartition.numel(), edgeToColumn.data_ptr(),
                                 edgeToRow.data_ptr(), _c.ctypes.byref(n), 0)
    _c.check(st, "tcgnn_preprocess")
    _report(n.value)


def preprocess_gpu(edgeList, nodePointer, num_nodes, blockSize_h, blockSize_w, blockPartition, edgeToColumn, edgeToRow):
    """Device SGT: same outputs as preprocess, all tensors on the GPU."""
    names = ("edgeList", "nodePointer", "blockPartition", "edgeToColumn", "edgeToRow")
    for t, n in zip((edgeList, nodePointer, blockPartition, edgeToColumn, edgeToRow), names):
        _check_input(t, n)
        _check_int(t, n)
    num_nodes = int(num_nodes)
    if nodePointer.numel() < num_nodes + 1:
        raise RuntimeError("nodePointer holds %d entries, need num_nodes + 1 = %d" % (nodePointer.numel(), num_nodes + 1))
    E = edgeList.numel()
    if edgeToColumn.numel() < E or edgeToRow.numel() < E:
        raise RuntimeError("edgeToColumn / edgeToRow are shorter than edgeList")
    n = _c._i64(0)
    dev = edgeList.device
    with torch.cuda.device(dev):
        # the translation's scratch (sort keys, positions, flags, ranks, rocPRIM's own) comes from torch's caching allocator: the library call
        # allocates nothing and synchronises once (include/tcgnn.h, tcgnn_preprocess_gpu_ws); a second translation of a graph this size
        # finds the block in the cache
        need = _c._sz(0)
        _c.check(_c.lib.tcgnn_preprocess_gpu_workspace_bytes(num_nodes, E, int(blockSize_h), _c.ctypes.byref(need)), "tcgnn_preprocess_gpu_workspace_bytes")
        ws = torch.empty(max(int(need.value), 256), dtype=torch.uint8, device=dev)
        st = _c.lib.tcgnn_preprocess_gpu_ws(edgeList.data_ptr(), nodePointer.data_ptr(), num_nodes, E, int(blockSize_h),
                                            int(blockSize_w), blockPartition.data_ptr(), blockPartition.numel(),
                                            edgeToColumn.data_ptr(), edgeToRow.data_ptr(), ws.data_ptr(), ws.numel(), _c.ctypes.byref(n), _stream_handle(dev))
        del ws   # (the call synchronised the stream: nothing still reads it)
    _c.check(st, "tcgnn_preprocess_gpu_ws")
    _report(n.value)


# ---------------------------------------------------------------- the hot path

def _six(input, nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow):
    _check_input(input, "input")
    _check_input(nodePointer, "nodePointer")
    _check_input(edgeList, "edgeList")
    _check_input(blockPartition, "blockPartition")
    _check_input(edgeToColumn, "edgeToColumn")
    _check_input(edgeToRow, "edgeToRow")
    _check_float(input, "input")
    if input.dim() != 2:
        raise RuntimeError("input must be [num_nodes, embedding_dim]")
    N = nodePointer.numel() - 1
    if input.size(0) != N:
        raise RuntimeError("input has %d rows but nodePointer describes %d nodes" % (input.size(0), N))
    if input.device != nodePointer.device:
        raise RuntimeError("input and nodePointer are on different devices")


def forward(input, nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow):
    """SpMM  Y = A_bin @ input  (GCN / GIN / SAG aggregation, forward and backward)."""
    _six(input, nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow)
    dev = input.device
    N, D = input.shape
    out = torch.empty_like(input)
    if N == 0 or D == 0:
        return [out]
    with torch.cuda.device(dev):
        plan = _plan_for(nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow)
        ws, ws_bytes = _workspace(plan, D, dev)
        st = _c.lib.tcgnn_spmm(plan, input.data_ptr(), out.data_ptr(), D, ws, ws_bytes, _stream_handle(dev))
    _c.check(st, "tcgnn_spmm")
    return [out]


def forward_fused(input, nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow, relu=False, gate=None):
    """Not in the reference module: `forward` with the layer's element-wise steps fused in.
    relu=True: max(A @ input, 0) - the ReLU the reference applies after the layer (main_tcgnn.py:100-139) runs in the
    kernel's stores.  gate (same shape as input): A @ (input * (gate > 0)) - with gate = the forward output, the ReLU
    backward mask is applied to dY while it is staged.  Bit-identical to the unfused compositions."""
    _six(input, nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow)
    if gate is not None:
        _check_input(gate, "gate")
        _check_float(gate, "gate")
        if gate.shape != input.shape or gate.device != input.device:
            raise RuntimeError("gate must have the shape and device of input")
    dev = input.device
    N, D = input.shape
    out = torch.empty_like(input)
    if N == 0 or D == 0:
        return [out]
    with torch.cuda.device(dev):
        plan = _plan_for(nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow)
        ws, ws_bytes = _workspace(plan, D, dev)
        st = _c.lib.tcgnn_spmm_fused(plan, input.data_ptr(), gate.data_ptr() if gate is not None else None, out.data_ptr(), D,
                                     1 if relu else 0, ws, ws_bytes, _stream_handle(dev))
    _c.check(st, "tcgnn_spmm_fused")
    return [out]


GEMM_FUSED_MAX_DIM = 128


def forward_gemm(input, weights, nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow, relu=False):
    """Not in the reference module: [(A @ input) @ weights] in ONE launch - the GIN order of
    gnn_conv.py:92-97 (`X' = TCGNN.forward(X, ...)[0]; X' = torch.mm(X', weights)`) without the N x D_in round trip: the
    aggregated rows go from the accumulators through LDS into the fp32 matrix pipe against W.  input [N, D_in], weights
    [D_in, D_out], both <= 128 wide.  relu=True fuses max(., 0) where the kernel writes the product in one pass; where it
    accumulates over column passes (the LDS-resident kernel on a 64-column input) the ReLU runs as a separate step here."""
    _six(input, nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow)
    _check_input(weights, "weights")
    _check_float(weights, "weights")
    N, D = input.shape
    if weights.dim() != 2 or weights.shape[0] != D or weights.device != input.device:
        raise RuntimeError("weights must be [input.size(1), D_out] on the input's device")
    Dout = weights.shape[1]
    if D > GEMM_FUSED_MAX_DIM or Dout > GEMM_FUSED_MAX_DIM or D == 0 or Dout == 0:
        raise RuntimeError("forward_gemm covers 1 <= D_in, D_out <= %d (got %d -> %d): compose forward() with torch.mm" % (GEMM_FUSED_MAX_DIM, D, Dout))
    dev = input.device
    out = torch.empty(N, Dout, dtype=torch.float32, device=dev)
    if N == 0:
        return [out]
    with torch.cuda.device(dev):
        plan = _plan_for(nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow)
        ws, ws_bytes = _workspace(plan, D, dev)
        st = _c.lib.tcgnn_spmm_gemm(plan, input.data_ptr(), weights.data_ptr(), out.data_ptr(), D, Dout, 1 if relu else 0, ws, ws_bytes,
                                    _stream_handle(dev))
        if st == 6 and relu:   # TCGNN_ERR_UNSUPPORTED: the product is accumulated over column passes - ReLU as its own step
            st = _c.lib.tcgnn_spmm_gemm(plan, input.data_ptr(), weights.data_ptr(), out.data_ptr(), D, Dout, 0, ws, ws_bytes, _stream_handle(dev))
            _c.check(st, "tcgnn_spmm_gemm")
            return [torch.relu_(out)]
    _c.check(st, "tcgnn_spmm_gemm")
    return [out]


def forward_AGNN(input, nodePointer, edgeList, edgeAttention, blockPartition, edgeToColumn, edgeToRow):
    """SpMM with edge values  Y = A_val @ input,  A_val[row(e), col(e)] = edgeAttention[0, e]."""
    _six(input, nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow)
    _check_input(edgeAttention, "edgeAttention")
    _check_float(edgeAttention, "edgeAttention")
    dev = input.device
    N, D = input.shape
    E = edgeList.numel()
    # [n_heads, E]; every head's launch in the reference reads row 0 and overwrites the same output
    # (TCGNN_kernel.cu:253-268, :529), so only row 0 is meaningful
    if edgeAttention.numel() < E:
        raise RuntimeError("edgeAttention holds %d values for %d edges" % (edgeAttention.numel(), E))
    out = torch.empty_like(input)
    if N == 0 or D == 0:
        return [out]
    with torch.cuda.device(dev):
        plan = _plan_for(nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow)
        ws, ws_bytes = _workspace(plan, D, dev)
        st = _c.lib.tcgnn_spmm_val(plan, input.data_ptr(), edgeAttention.data_ptr(), out.data_ptr(), D, ws, ws_bytes,
                                   _stream_handle(dev))
    _c.check(st, "tcgnn_spmm_val")
    return [out]


def forward_ef(input, nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow):
    """SDDMM  ef[e] = <input[row(e)], input[col(e)]>  for every CSR edge, fp32 [E]."""
    _six(input, nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow)
    dev = input.device
    N, D = input.shape
    E = edgeList.numel()
    out = torch.empty(E, dtype=torch.float32, device=dev)
    if E == 0:
        return [out]
    if D == 0:
        return [out.zero_()]
    with torch.cuda.device(dev):
        plan = _plan_for(nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow)
        ws, ws_bytes = _workspace(plan, D, dev)
        st = _c.lib.tcgnn_sddmm(plan, input.data_ptr(), out.data_ptr(), D, ws, ws_bytes, _stream_handle(dev))
    _c.check(st, "tcgnn_sddmm")
    return [out]


# ---- additions (not in the reference module): the two products of an AGNN layer in one pass ------------

def agnn_fused_supported(input, nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow):
    """True if agnn_fused_forward / agnn_fused_backward cover this graph and width (canonical CSR, D <= 128, E >= 8)."""
    if not (input.is_cuda and input.dim() == 2 and input.dtype == torch.float32) or input.shape[0] == 0 or input.shape[1] == 0:
        return False
    with torch.cuda.device(input.device):
        plan = _plan_for(nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow)
    return bool(_c.lib.tcgnn_agnn_supported(plan, input.shape[1]))


def _weight_scalar(attention_w, dev):
    _check_input(attention_w, "attention_w")
    _check_float(attention_w, "attention_w")
    if attention_w.numel() != 1 or attention_w.device != dev:
        raise RuntimeError("attention_w must hold one value (n_heads = 1) on the input's device")


def agnn_fused_forward(input, nodePointer, edgeList, attention_w, blockPartition, edgeToColumn, edgeToRow):
    """[Y, ef, ef_absmax] with ef = forward_ef(input), Y = forward_AGNN(input, attention_w * ef): what
    gnn_conv.py:125-132 computes with two calls (two gathers of the neighbour rows), here in one pass.
    ef_absmax (1 + N int32 words on the device: max |ef| and the per-row scale exponents) must be handed to agnn_fused_backward."""
    _six(input, nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow)
    dev = input.device
    _weight_scalar(attention_w, dev)
    N, D = input.shape
    out = torch.empty_like(input)
    ef = torch.empty(edgeList.numel(), dtype=torch.float32, device=dev)
    absmax = torch.zeros(1 + input.shape[0], dtype=torch.int32, device=dev)   # word 0: max |ef|; words 1 .. N: per-row exponents of the edge weights (include/tcgnn.h)
    with torch.cuda.device(dev):
        plan = _plan_for(nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow)
        ws, ws_bytes = _workspace(plan, D, dev)
        st = _c.lib.tcgnn_agnn_pair_forward(plan, input.data_ptr(), attention_w.data_ptr(), ef.data_ptr(), absmax.data_ptr(), absmax.numel(),
                                            out.data_ptr(), D, ws, ws_bytes, _stream_handle(dev))
    _c.check(st, "tcgnn_agnn_pair_forward")
    return [out, ef, absmax]


def agnn_fused_backward(d_output, nodePointer, edgeList, attention_w, ef, ef_absmax, blockPartition, edgeToColumn, edgeToRow):
    """[G, d_w] with G = forward_AGNN(d_output, attention_w * ef) and d_w = <forward_ef(d_output), edgeList.float()>
    (gnn_conv.py:143 and :150-153), one pass."""
    _six(d_output, nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow)
    dev = d_output.device
    _weight_scalar(attention_w, dev)
    _check_input(ef, "ef")
    _check_float(ef, "ef")
    if ef.numel() != edgeList.numel() or ef_absmax.numel() != 1 + d_output.shape[0] or ef_absmax.dtype != torch.int32 or not ef_absmax.is_cuda:
        raise RuntimeError("ef / ef_absmax are not what agnn_fused_forward returned for this graph")
    N, D = d_output.shape
    out = torch.empty_like(d_output)
    d_w = torch.empty(1, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        plan = _plan_for(nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow)
        ws, ws_bytes = _workspace(plan, D, dev)
        st = _c.lib.tcgnn_agnn_pair_backward(plan, d_output.data_ptr(), attention_w.data_ptr(), ef.data_ptr(), ef_absmax.data_ptr(), ef_absmax.numel(),
                                             out.data_ptr(), d_w.data_ptr(), D, ws, ws_bytes, _stream_handle(dev))
    _c.check(st, "tcgnn_agnn_pair_backward")
    return [out, d_w]


backward = forward        # TCGNN.cpp:270
backward_ef = forward_ef  # TCGNN.cpp:271
