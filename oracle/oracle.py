"""ctypes front-end of the CPU oracle (oracle/tcgnn_oracle.c) and loader of the compiled reference.

TEST INFRASTRUCTURE.  Importable only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under tc-gnn_atc23_amd/ imports this module.

Functions mirror the reference's extension API (TCGNN_conv/TCGNN.cpp:260-272) on numpy arrays:
    preprocess(col, rowptr, N, bh, bw, bp, e2c, e2r)  -> tc_blocks   (TCGNN.cpp:172-226)
    spmm(X, rowptr, col, bp, e2c, e2r, ...)           -> Y           (TCGNN_kernel.cu:336-454)
    spmm_val(X, rowptr, col, att, bp, e2c, e2r, ...)  -> Y           (TCGNN_kernel.cu:459-578)
    sddmm(X, rowptr, col, bp, e2c, e2r, ...)          -> ef          (TCGNN_kernel.cu:584-727)
plus fp64 evaluations of the mathematical contract and the row-parallel CSR CPU baseline.
"""
import ctypes
import glob
import importlib.util
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ROUND_NONE, ROUND_TF32, ROUND_FP16 = 0, 1, 2


def build(force=False):
    """Compile liboracle.so with gcc (and, when /root/reference exists, oracle/_ref)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "tcgnn_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        i32p = ctypes.POINTER(ctypes.c_int32)
        f32p = ctypes.POINTER(ctypes.c_float)
        f64p = ctypes.POINTER(ctypes.c_double)
        i64p = ctypes.POINTER(ctypes.c_int64)
        c_i32, c_i64 = ctypes.c_int32, ctypes.c_int64
        _LIB.oracle_preprocess.argtypes = [i32p, i32p, c_i32, c_i32, c_i32, i32p, c_i64, i32p, i32p, i64p]
        _LIB.oracle_spmm.argtypes = [i32p] * 5 + [c_i32, c_i64, c_i32, c_i32, f32p, f32p, f32p, c_i32, c_i32, c_i32, c_i32]
        _LIB.oracle_sddmm.argtypes = [i32p] * 5 + [c_i32, c_i64, c_i32, c_i32, f32p, f32p, c_i32, c_i32, c_i32]
        _LIB.oracle_spmm_windows.argtypes = [i32p] * 5 + [c_i32, c_i32, f32p, f32p, f32p, c_i32, i32p, c_i32]
        _LIB.oracle_sddmm_windows.argtypes = [i32p] * 5 + [c_i32, c_i64, c_i32, f32p, f32p, c_i32, i32p, c_i32]
        _LIB.oracle_spmm_f64.argtypes = [i32p, i32p, c_i32, c_i32, f32p, f32p, f64p, f64p]
        _LIB.oracle_sddmm_f64.argtypes = [i32p, i32p, c_i32, c_i32, f32p, f64p, f64p]
        _LIB.oracle_csr_spmm.argtypes = [i32p, i32p, c_i32, c_i32, f32p, f32p, c_i32]
        _LIB.oracle_round_tf32.argtypes = [ctypes.c_float]
        _LIB.oracle_round_tf32.restype = ctypes.c_float
        _LIB.oracle_round_fp16.argtypes = [ctypes.c_float]
        _LIB.oracle_round_fp16.restype = ctypes.c_float
    return _LIB


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def preprocess(col, rowptr, num_nodes, bh, bw, bp, e2c, e2r):
    """In-place like TCGNN.preprocess; bp/e2c/e2r must be C-contiguous int32 numpy arrays.
    Returns the TC-block count the reference prints."""
    for a in (bp, e2c, e2r):
        assert a.dtype == np.int32 and a.flags.c_contiguous
    col, pc = _i32(col)
    rowptr, pr = _i32(rowptr)
    p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    n = ctypes.c_int64(0)
    rc = lib().oracle_preprocess(pc, pr, int(num_nodes), int(bh), int(bw), p(bp), bp.shape[0], p(e2c), p(e2r), ctypes.byref(n))
    if rc:
        raise MemoryError("oracle_preprocess")
    return n.value


def spmm(X, rowptr, col, bp, e2c, e2r, att=None, round_mode=ROUND_TF32, scale_exp_x=0, scale_exp_a=0, ref_quirks=False):
    X, px = _f32(X)
    rowptr, pr = _i32(rowptr); col, pc = _i32(col); bp, pb = _i32(bp); e2c, p2c = _i32(e2c); e2r, p2r = _i32(e2r)
    N, D = rowptr.shape[0] - 1, X.shape[1]
    Y = np.empty((N, D), dtype=np.float32)
    pa = None
    if att is not None:
        att, pa = _f32(np.asarray(att).reshape(-1)[: col.shape[0]])
    rc = lib().oracle_spmm(pr, pc, pb, p2c, p2r, N, col.shape[0], bp.shape[0], D, px, pa,
                           Y.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), round_mode, scale_exp_x, scale_exp_a, int(ref_quirks))
    if rc:
        raise MemoryError("oracle_spmm")
    return Y


def spmm_val(X, rowptr, col, att, bp, e2c, e2r, **kw):
    return spmm(X, rowptr, col, bp, e2c, e2r, att=att, **kw)


def sddmm(X, rowptr, col, bp, e2c, e2r, round_mode=ROUND_TF32, scale_exp_x=0, ref_quirks=False):
    X, px = _f32(X)
    rowptr, pr = _i32(rowptr); col, pc = _i32(col); bp, pb = _i32(bp); e2c, p2c = _i32(e2c); e2r, p2r = _i32(e2r)
    N, D = rowptr.shape[0] - 1, X.shape[1]
    ef = np.empty(col.shape[0], dtype=np.float32)
    lib().oracle_sddmm(pr, pc, pb, p2c, p2r, N, col.shape[0], bp.shape[0], D, px,
                       ef.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), round_mode, scale_exp_x, int(ref_quirks))
    return ef


def spmm_windows(X, rowptr, col, bp, e2c, e2r, windows, att=None, round_mode=ROUND_TF32):
    """The oracle's thread-block body (oracle_spmm) for the listed row windows only -> (rows, Y_rows): the row numbers of those
    windows and their result rows.  For graphs at BASELINE size, where the whole oracle product is minutes of CPU work."""
    X, px = _f32(X)
    rowptr, pr = _i32(rowptr); col, pc = _i32(col); bp, pb = _i32(bp); e2c, p2c = _i32(e2c); e2r, p2r = _i32(e2r)
    windows, pw = _i32(windows)
    N, D = rowptr.shape[0] - 1, X.shape[1]
    rows = (windows[:, None].astype(np.int64) * 16 + np.arange(16)[None, :]).reshape(-1)
    rows = rows[rows < N]
    Y = np.zeros((N, D), dtype=np.float32)
    pa = None
    if att is not None:
        att, pa = _f32(np.asarray(att).reshape(-1)[: col.shape[0]])
    rc = lib().oracle_spmm_windows(pr, pc, pb, p2c, p2r, N, D, px, pa, Y.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), round_mode, pw, windows.shape[0])
    if rc:
        raise MemoryError("oracle_spmm_windows")
    return rows, Y[rows]


def sddmm_windows(X, rowptr, col, bp, e2c, e2r, windows, round_mode=ROUND_TF32):
    """oracle_sddmm's warp body for the listed row windows only -> (edges, ef_edges): the CSR positions of those windows' edges
    and their scores."""
    X, px = _f32(X)
    rowptr, pr = _i32(rowptr); col, pc = _i32(col); bp, pb = _i32(bp); e2c, p2c = _i32(e2c); e2r, p2r = _i32(e2r)
    windows, pw = _i32(windows)
    N, D = rowptr.shape[0] - 1, X.shape[1]
    ef = np.zeros(col.shape[0], dtype=np.float32)
    lib().oracle_sddmm_windows(pr, pc, pb, p2c, p2r, N, col.shape[0], D, px, ef.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), round_mode, pw, windows.shape[0])
    lo = rowptr[np.minimum(windows.astype(np.int64) * 16, N)].astype(np.int64)
    hi = rowptr[np.minimum(windows.astype(np.int64) * 16 + 16, N)].astype(np.int64)
    edges = np.concatenate([np.arange(a, b, dtype=np.int64) for a, b in zip(lo, hi)]) if len(windows) else np.zeros(0, np.int64)
    return edges, ef[edges]


def spmm_f64(X, rowptr, col, att=None):
    """fp64 evaluation of Y = A*X from the CSR; returns (Y, sum|a||x|)."""
    X, px = _f32(X); rowptr, pr = _i32(rowptr); col, pc = _i32(col)
    N, D = rowptr.shape[0] - 1, X.shape[1]
    Y = np.empty((N, D)); A = np.empty((N, D))
    pa = None
    if att is not None:
        att, pa = _f32(np.asarray(att).reshape(-1)[: col.shape[0]])
    f64 = ctypes.POINTER(ctypes.c_double)
    lib().oracle_spmm_f64(pr, pc, N, D, px, pa, Y.ctypes.data_as(f64), A.ctypes.data_as(f64))
    return Y, A


def sddmm_f64(X, rowptr, col):
    X, px = _f32(X); rowptr, pr = _i32(rowptr); col, pc = _i32(col)
    N, D = rowptr.shape[0] - 1, X.shape[1]
    ef = np.empty(col.shape[0]); a = np.empty(col.shape[0])
    f64 = ctypes.POINTER(ctypes.c_double)
    lib().oracle_sddmm_f64(pr, pc, N, D, px, ef.ctypes.data_as(f64), a.ctypes.data_as(f64))
    return ef, a


def csr_spmm(X, rowptr, col, threads=0, out=None):
    """Row-parallel CSR gather-add (the DGL-CPU-style baseline). Timed by bench.py."""
    X, px = _f32(X); rowptr, pr = _i32(rowptr); col, pc = _i32(col)
    N, D = rowptr.shape[0] - 1, X.shape[1]
    Y = out if out is not None else np.empty((N, D), dtype=np.float32)
    lib().oracle_csr_spmm(pr, pc, N, D, px, Y.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), int(threads))
    return Y


def num_threads():
    return lib().oracle_num_threads()


def round_tf32(x):
    return np.vectorize(lambda v: lib().oracle_round_tf32(float(v)), otypes=[np.float32])(np.asarray(x, dtype=np.float32))


def round_fp16(x):
    return np.vectorize(lambda v: lib().oracle_round_fp16(float(v)), otypes=[np.float32])(np.asarray(x, dtype=np.float32))


# ------------------------------------------------------------------ compiled reference (oracle/_ref)

def tile_counts(rowptr, col, tile_h=16, tile_w=8):
    """(sliding, condensed) tile counts as 3_cnt_TC_blk_SpMM.py:56-81 computes them (pure Python, small graphs):
    per window of tile_h rows, the sorted set of neighbour ids; condensed = ceil(|set| / tile_w) (:66);
    sliding = greedy walk placing a tile of tile_w ids at the first uncovered id (:72-81)."""
    rowptr = np.asarray(rowptr); col = np.asarray(col)
    n = len(rowptr) - 1
    sliding = condensed = 0
    for w0 in range(0, n, tile_h):
        ids = sorted(set(int(c) for c in col[rowptr[w0]:rowptr[min(w0 + tile_h, n)]]))
        condensed += (len(ids) + tile_w - 1) // tile_w
        i = j = 0
        while i < len(ids) and j < len(ids):
            end = ids[i] + tile_w
            while j < len(ids) and ids[j] < end:
                j += 1
            i = j
            sliding += 1
    return sliding, condensed


def ref_available():
    return bool(glob.glob(os.path.join(_HERE, "_ref", "TCGNN_ref*.so")))


def load_ref():
    """Import the reference's own pybind module (built by oracle/build_ref.sh from the unmodified
    /root/reference/TCGNN_conv/TCGNN.cpp).  Its CUDA launcher symbols are undefined, so the module
    must be dlopen'ed lazily; only `preprocess` is callable."""
    import torch  # noqa: F401  (libtorch must be resident before the extension loads)
    hits = glob.glob(os.path.join(_HERE, "_ref", "TCGNN_ref*.so"))
    if not hits:
        raise FileNotFoundError("oracle/_ref not built (run oracle/build_ref.sh where /root/reference exists)")
    if "TCGNN_ref" in sys.modules:
        return sys.modules["TCGNN_ref"]
    old = sys.getdlopenflags()
    sys.setdlopenflags(os.RTLD_LAZY | os.RTLD_LOCAL)
    try:
        spec = importlib.util.spec_from_file_location("TCGNN_ref", hits[0])
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.setdlopenflags(old)
    sys.modules["TCGNN_ref"] = mod
    return mod
