"""Host sparse-graph translation through the C ABI, the exported symbols, and the TCGNN module's
CPU-side behaviour (no GPU needed: nothing here launches a kernel)."""
import ctypes
import glob
import os
import re

import numpy as np
import pytest
import torch

import graphs
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
SGT_FIXTURES = sorted(glob.glob(os.path.join(GOLD, "sgt_*.npz")))


def test_library_exports_every_symbol_the_header_declares():
    import tcgnn_capi as c
    hdr = open(os.path.join(ROOT, "include", "tcgnn.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(tcgnn_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(c.SIGNATURES), (declared ^ set(c.SIGNATURES))
    for name in declared:
        assert hasattr(c.lib, name)
    assert c.lib.tcgnn_abi_version() == 1
    assert c.lib.tcgnn_status_string(0) == b"ok"


@pytest.mark.parametrize("path", SGT_FIXTURES, ids=[os.path.basename(p)[4:-4] for p in SGT_FIXTURES])
def test_host_sgt_equals_reference_fixture(path):
    f = np.load(path)
    rp, col, guard = f["rowptr"], f["col"], int(f["guard"])
    n = len(rp) - 1
    nw = (n + 15) // 16
    bp, e2c, e2r, tc = graphs.host_sgt(rp, col, guard=guard)
    assert np.array_equal(bp[:nw], f["bp_with_guard"][:nw])   # identical for every real window
    assert not bp[nw:].any()                                    # and nothing written past the end
    assert np.array_equal(e2c, f["e2c"]) and np.array_equal(e2r, f["e2r"])
    assert tc == int(f["tc_blocks"])                            # the count the reference prints


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_host_sgt_threads_and_unsorted_rows(threads):
    import tcgnn_capi as c
    rng = np.random.default_rng(threads)
    rp, col = graphs.powerlaw_graph(5000, 30, seed=threads)
    col = col.copy()
    for r in range(0, 5000, 3):  # shuffle some rows and plant duplicates: the reference accepts both
        s, e = rp[r], rp[r + 1]
        col[s:e] = rng.permutation(col[s:e])
        if e - s > 2:
            col[s] = col[s + 1]
    n = 5000
    nw = (n + 15) // 16
    bp = np.zeros(nw, np.int32); e2c = np.zeros(len(col), np.int32); e2r = np.zeros(len(col), np.int32)
    cnt = ctypes.c_int64()
    st = c.lib.tcgnn_preprocess(col.ctypes.data, rp.ctypes.data, n, 16, 8, bp.ctypes.data, nw, e2c.ctypes.data, e2r.ctypes.data,
                                ctypes.byref(cnt), threads)
    assert st == 0
    bo = np.zeros(nw, np.int32); co = np.zeros(len(col), np.int32); ro = np.zeros(len(col), np.int32)
    tco = O.preprocess(col, rp, n, 16, 8, bo, co, ro)
    assert np.array_equal(bp, bo) and np.array_equal(e2c, co) and np.array_equal(e2r, ro) and cnt.value == tco


def test_host_sgt_other_tile_shapes():
    """blockSize_h / blockSize_w are run-time arguments of preprocess (main_tcgnn.py:51-52)."""
    import tcgnn_capi as c
    rp, col = graphs.uniform_graph(333, 7, seed=8)
    for bh, bw in ((16, 16), (8, 8), (32, 4)):
        nw = (333 + bh - 1) // bh
        bp = np.zeros(nw, np.int32); e2c = np.zeros(len(col), np.int32); e2r = np.zeros(len(col), np.int32)
        cnt = ctypes.c_int64()
        assert c.lib.tcgnn_preprocess(col.ctypes.data, rp.ctypes.data, 333, bh, bw, bp.ctypes.data, nw, e2c.ctypes.data,
                                      e2r.ctypes.data, ctypes.byref(cnt), 2) == 0
        bo = np.zeros(nw, np.int32); co = np.zeros(len(col), np.int32); ro = np.zeros(len(col), np.int32)
        tco = O.preprocess(col, rp, 333, bh, bw, bo, co, ro)
        assert np.array_equal(bp, bo) and np.array_equal(e2c, co) and np.array_equal(e2r, ro) and cnt.value == tco


def test_c_abi_reports_errors_instead_of_aborting():
    import tcgnn_capi as c
    rp = np.array([0, 2, 1], dtype=np.int32)  # decreasing row pointer
    col = np.zeros(2, np.int32); bp = np.zeros(1, np.int32); e2c = np.zeros(2, np.int32); e2r = np.zeros(2, np.int32)
    st = c.lib.tcgnn_preprocess(col.ctypes.data, rp.ctypes.data, 2, 16, 8, bp.ctypes.data, 1, e2c.ctypes.data, e2r.ctypes.data, None, 1)
    assert st != 0 and c.lib.tcgnn_last_error()
    assert c.lib.tcgnn_preprocess(None, None, 2, 16, 8, None, 0, None, None, None, 1) == 1   # TCGNN_ERR_INVALID_ARG
    assert c.lib.tcgnn_preprocess(col.ctypes.data, rp.ctypes.data, 2, 0, 8, bp.ctypes.data, 1, e2c.ctypes.data, e2r.ctypes.data, None, 1) == 1
    assert c.lib.tcgnn_workspace_bytes(None, 64) == 0
    assert c.lib.tcgnn_plan_destroy(None) == 0


def test_module_preprocess_signature_prints_and_mutates_in_place(capfd):
    """TCGNN.preprocess(edgeList, nodePointer, N, bh, bw, bp, e2c, e2r) -> None, prints the two
    lines the reference prints (TCGNN.cpp:225)."""
    import TCGNN
    f = np.load(os.path.join(GOLD, "sgt_uniform_n1000.npz"))
    rp, col = torch.from_numpy(f["rowptr"]), torch.from_numpy(f["col"])
    n = 1000
    bp = torch.zeros((n + 15) // 16, dtype=torch.int); e2c = torch.zeros(col.numel(), dtype=torch.int); e2r = torch.zeros(col.numel(), dtype=torch.int)
    assert TCGNN.preprocess(col, rp, n, 16, 8, bp, e2c, e2r) is None
    out = capfd.readouterr().out
    assert out == "TC_Blocks:\t%d\nExp_Edges:\t%d\n" % (int(f["tc_blocks"]), int(f["tc_blocks"]) * 128)
    assert np.array_equal(e2c.numpy(), f["e2c"]) and np.array_equal(bp.numpy(), f["bp_with_guard"][: bp.numel()])
    assert set(["preprocess", "preprocess_gpu", "forward", "forward_ef", "forward_AGNN", "backward", "backward_ef"]) <= set(dir(TCGNN))
    assert TCGNN.backward is TCGNN.forward and TCGNN.backward_ef is TCGNN.forward_ef


def test_module_argument_checks_match_the_reference_messages():
    import TCGNN
    x = torch.zeros(4, 16)
    i = torch.zeros(5, dtype=torch.int)
    with pytest.raises(RuntimeError, match="input must be a CUDA tensor"):       # CHECK_CUDA, TCGNN.cpp:54
        TCGNN.forward(x, i, i, i, i, i)
    with pytest.raises(RuntimeError, match="input must be a CUDA tensor"):
        TCGNN.forward_ef(x, i, i, i, i, i)
    with pytest.raises(RuntimeError, match="input must be a CUDA tensor"):
        TCGNN.forward_AGNN(x, i, i, x, i, i, i)
    with pytest.raises(RuntimeError, match="Int"):
        TCGNN.preprocess(i.long(), i, 4, 16, 8, i, i, i)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        TCGNN.preprocess_gpu(i, i, 4, 16, 8, i, i, i)


def test_host_sgt_throughput_sanity():
    """Not a benchmark: guards against an accidental O(tiles x edges) regression (the reference's
    host path is ~0.2 us/edge serial, logs/RTX3090_GCN.log:49)."""
    import time
    rng = np.random.default_rng(0)
    n, deg = 40000, 50
    col = np.sort(rng.integers(0, n, size=(n, deg), dtype=np.int32), axis=1).reshape(-1)
    rp = (np.arange(n + 1, dtype=np.int64) * deg).astype(np.int32)
    t = time.perf_counter()
    graphs.host_sgt(rp, col)
    assert (time.perf_counter() - t) / len(col) < 0.5e-6
