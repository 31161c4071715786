"""Builds oracle/cpu_baseline.c FOR THE MACHINE IT RUNS ON (gcc -O3 -march=native -fopenmp, into a temporary directory) and times
Y = A X on the host cores.  Measurement infrastructure for bench.py's `cpu_baseline` leg only - never imported by the product."""
import ctypes
import os
import subprocess
import tempfile
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None
_flags = None


def lib():
    global _lib, _flags
    if _lib is None:
        out = os.path.join(tempfile.mkdtemp(prefix="tcgnn_cpu_baseline_"), "libcpu_baseline.so")
        for flags in (["-O3", "-march=native", "-fopenmp"], ["-O3", "-fopenmp"]):
            r = subprocess.run(["gcc", *flags, "-fPIC", "-shared", "-std=gnu11", os.path.join(_HERE, "cpu_baseline.c"), "-o", out], capture_output=True, text=True)
            if r.returncode == 0:
                _flags = " ".join(flags)
                break
        else:
            raise RuntimeError("gcc could not build oracle/cpu_baseline.c: " + r.stderr[-500:])
        _lib = ctypes.CDLL(out)
        _lib.cpu_csr_spmm.restype = ctypes.c_int
        _lib.cpu_csr_spmm.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int32] * 2 + [ctypes.c_void_p] * 3 + [ctypes.c_int32] * 2
    return _lib


def build_flags():
    lib()
    return _flags


def csr_spmm(X, rowptr, col, threads=0, reps=1):
    """-> (Y, [seconds per pass]).  X / Y used by the timed passes are fresh allocations first-touched in parallel."""
    L = lib()
    Xsrc = np.ascontiguousarray(X, dtype=np.float32)
    rp = np.ascontiguousarray(rowptr, dtype=np.int32); cl = np.ascontiguousarray(col, dtype=np.int32)
    n, D = len(rp) - 1, Xsrc.shape[1]
    Xw = np.empty((n, D), dtype=np.float32); Y = np.empty((n, D), dtype=np.float32)       # untouched pages: placed by the first call
    args = (rp.ctypes.data, cl.ctypes.data, n, D, Xw.ctypes.data, Xsrc.ctypes.data, Y.ctypes.data, int(threads))
    assert L.cpu_csr_spmm(*args, 1) == 0
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        L.cpu_csr_spmm(*args, 0)
        times.append(time.perf_counter() - t0)
    return Y, times
