"""Seeded synthetic CSR graphs shared by the tests, tools/ and bench.py (no reference data needed)."""
import numpy as np
import scipy.sparse as sp


def csr_from_edges(src, dst, n):
    """COO -> canonical CSR exactly like the reference's dataset.py:94-104 (duplicates merged,
    rows sorted); returns (rowptr int32[n+1], col int32[nnz])."""
    a = sp.coo_matrix((np.ones(len(src), dtype=np.int8), (src, dst)), shape=(n, n)).tocsr()
    a.sum_duplicates()
    a.sort_indices()
    return a.indptr.astype(np.int32), a.indices.astype(np.int32)


def uniform_graph(n, avg_deg, seed, symmetric=True):
    rng = np.random.default_rng(seed)
    m = int(n * avg_deg / (2 if symmetric else 1))
    src = rng.integers(0, n, size=m)
    dst = rng.integers(0, n, size=m)
    if symmetric:
        src, dst = np.concatenate([src, dst]), np.concatenate([dst, src])
    return csr_from_edges(src, dst, n)


def powerlaw_graph(n, avg_deg, seed, alpha=1.8, symmetric=True):
    """Degree-skewed graph: endpoints drawn with probability ~ rank^(-1/alpha)-like weights."""
    rng = np.random.default_rng(seed)
    m = int(n * avg_deg / (2 if symmetric else 1))
    w = (np.arange(1, n + 1, dtype=np.float64)) ** (-1.0 / alpha)
    w /= w.sum()
    perm = rng.permutation(n)
    src = perm[rng.choice(n, size=m, p=w)]
    dst = rng.integers(0, n, size=m)
    if symmetric:
        src, dst = np.concatenate([src, dst]), np.concatenate([dst, src])
    return csr_from_edges(src, dst, n)


def community_graph(n, blocks, avg_deg, p_in, seed):
    """Stochastic block model with consecutively numbered communities: a share p_in of the edges stays inside the endpoint's own
    block (the shape tcgnn_graph.sbm_csr builds at Reddit size: a few column ranges per window are DENSE, the rest sparse)."""
    rng = np.random.default_rng(seed)
    m = int(n * avg_deg / 2)
    src = rng.integers(0, n, size=m)
    size = (n + blocks - 1) // blocks
    inside = rng.random(m) < p_in
    dst_in = np.minimum((src // size) * size + rng.integers(0, size, size=m), n - 1)
    dst = np.where(inside, dst_in, rng.integers(0, n, size=m))
    return csr_from_edges(np.concatenate([src, dst]), np.concatenate([dst, src]), n)


def hub_rows_graph(n, seed, full_rows=24, half_rows=3, background=30000, bg_cols=None):
    """A few rows that are edges to every (or every second) column over a sparse uniform background, symmetrised: inside a hub's
    window a lane's run of edges within its eight tile columns is up to eight long (the edge-valued kernels fetch four values at a
    time), and the hub columns make every other window hold a dense column block."""
    rng = np.random.default_rng(seed)
    half = np.arange(0, n, 2)
    m = n if bg_cols is None else bg_cols   # (bg_cols: the background stays inside the first bg_cols nodes - the column ranges beyond hold hub edges only)
    src = [np.repeat(np.arange(full_rows), n), rng.integers(0, m, background), np.repeat(np.arange(1000, 1000 + half_rows), len(half))]
    dst = [np.tile(np.arange(n), full_rows), rng.integers(0, m, background), np.tile(half, half_rows)]
    s_, d_ = np.concatenate(src), np.concatenate(dst)
    keep = s_ != d_
    s_, d_ = s_[keep], d_[keep]
    return csr_from_edges(np.concatenate([s_, d_]), np.concatenate([d_, s_]), n)


def with_empty_window(rowptr, col, first_row, last_row):
    """Remove every edge of rows [first_row, last_row) (keeps the CSR canonical)."""
    n = len(rowptr) - 1
    a = sp.csr_matrix((np.ones(len(col), dtype=np.int8), col, rowptr), shape=(n, n)).tolil()
    a[first_row:last_row, :] = 0
    a = a.tocsr()
    a.eliminate_zeros()
    a.sort_indices()
    return a.indptr.astype(np.int32), a.indices.astype(np.int32)


def edge_case_graphs():
    """(name, rowptr, col) for the shapes the reference's quirks live at."""
    out = []
    for n, deg in ((1, 1), (15, 4), (16, 5), (17, 3), (32, 6), (40, 4), (1000, 10)):
        rp, c = uniform_graph(n, deg, seed=100 + n, symmetric=n > 1)
        out.append(("uniform_n%d" % n, rp, c))
    rp, c = uniform_graph(48, 3, seed=7)
    rp, c = with_empty_window(rp, c, 16, 32)
    out.append(("empty_middle_window_n48", rp, c))
    rp, c = powerlaw_graph(1000, 12, seed=11)
    out.append(("powerlaw_n1000", rp, c))
    out.append(("no_edges_n20", np.zeros(21, dtype=np.int32), np.zeros(0, dtype=np.int32)))
    return out


def host_sgt(rowptr, col, guard=0):
    """Run the product's host SGT through the C ABI (numpy in / numpy out)."""
    import ctypes
    import tcgnn_capi as c
    n = len(rowptr) - 1
    nw = (n + 15) // 16
    bp = np.zeros(nw + guard, dtype=np.int32)
    e2c = np.zeros(len(col), dtype=np.int32)
    e2r = np.zeros(len(col), dtype=np.int32)
    cnt = ctypes.c_int64(0)
    rp = np.ascontiguousarray(rowptr, dtype=np.int32)
    cl = np.ascontiguousarray(col, dtype=np.int32)
    st = c.lib.tcgnn_preprocess(cl.ctypes.data, rp.ctypes.data, n, 16, 8, bp.ctypes.data, nw, e2c.ctypes.data, e2r.ctypes.data,
                                ctypes.byref(cnt), 0)
    c.check(st, "tcgnn_preprocess")
    return bp, e2c, e2r, cnt.value
