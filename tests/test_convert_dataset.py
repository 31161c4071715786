"""the one-off converter from the graphs' native files to the reference's `.npz` schema
(dataset.py:74-80).  Every format is fabricated here from one small edge set and must come back, through our loader, as the CSR
scipy builds from that edge set directly."""
import gzip
import os
import sys

import numpy as np
import pytest
from scipy.sparse import coo_matrix, save_npz

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tc-gnn_atc23_amd"))
import convert_dataset as C  # noqa: E402
import tcgnn_graph as G      # noqa: E402

N = 37


def edges():
    rng = np.random.default_rng(5)
    s = rng.integers(0, N - 2, size=300); d = rng.integers(0, N - 2, size=300)   # (the last two nodes stay isolated: num_nodes must come from the file)
    return s.astype(np.int64), d.astype(np.int64)


def csr_of(s, d, n):
    m = coo_matrix((np.ones(len(s)), (s, d)), shape=(n, n)).tocsr(); m.sum_duplicates(); m.sort_indices()
    return m.indptr.astype(np.int32), m.indices.astype(np.int32)


def check(out, s, d, n, raw):
    ds = G.TCGNN_dataset(str(out), 8, 3, load_from_txt=False, seed=0)
    rp, col = csr_of(s, d, n)
    assert ds.num_nodes == n and ds.num_edges == raw
    assert np.array_equal(ds.row_pointers.numpy(), rp) and np.array_equal(ds.column_index.numpy(), col)


def test_scipy_npz_the_dgl_reddit_form(tmp_path):
    s, d = edges()
    save_npz(tmp_path / "reddit_graph.npz", coo_matrix((np.ones(len(s)), (s, d)), shape=(N, N)))
    n, e = C.convert(str(tmp_path / "reddit_graph.npz"), str(tmp_path / "o.npz"))
    assert C.detect_format(str(tmp_path / "reddit_graph.npz")) == "scipy-npz" and n == N
    check(tmp_path / "o.npz", s, d, N, e)


@pytest.mark.parametrize("gz", [True, False])
def test_ogb_raw_directory_symmetrized(tmp_path, gz):
    s, d = edges()
    raw = tmp_path / "ogbn_products" / "raw"; raw.mkdir(parents=True)
    opener = (lambda p: gzip.open(str(p) + ".gz", "wt")) if gz else (lambda p: open(p, "w"))
    with opener(raw / "edge.csv") as f:
        f.write("".join("%d,%d\n" % (a, b) for a, b in zip(s, d)))
    with opener(raw / "num-node-list.csv") as f:
        f.write("%d\n" % N)
    n, e = C.convert(str(tmp_path / "ogbn_products"), str(tmp_path / "o.npz"), symmetrize=True)
    assert n == N and e == 2 * len(s)
    check(tmp_path / "o.npz", np.concatenate([s, d]), np.concatenate([d, s]), N, e)
    assert C.convert(str(raw), str(tmp_path / "o2.npz"))[1] == len(s)       # the raw/ directory itself is accepted too


def test_ogb_npz_and_edge_index_dumps(tmp_path):
    s, d = edges()
    np.savez(tmp_path / "data.npz", edge_index=np.stack([s, d]), num_nodes_list=np.array([N]))
    assert C.detect_format(str(tmp_path / "data.npz")) == "ogb-npz"
    n, e = C.convert(str(tmp_path / "data.npz"), str(tmp_path / "o.npz"))
    check(tmp_path / "o.npz", s, d, N, e)
    np.save(tmp_path / "ei.npy", np.stack([s, d], axis=1))                  # [E, 2]: no node count in the file -> max id + 1
    n, e = C.convert(str(tmp_path / "ei.npy"), str(tmp_path / "o3.npz"))
    assert n == int(max(s.max(), d.max())) + 1
    check(tmp_path / "o3.npz", s, d, n, e)
    np.savez(tmp_path / "pyg.npz", edge_index=np.stack([s, d]), num_nodes=N)
    n, e = C.convert(str(tmp_path / "pyg.npz"), str(tmp_path / "o4.npz"), drop_self_loops=True)
    keep = s != d
    check(tmp_path / "o4.npz", s[keep], d[keep], N, e)


def test_snap_edge_list_with_comments_and_matrix_market(tmp_path):
    s, d = edges()
    with open(tmp_path / "amazon0505.txt", "w") as f:
        f.write("# Directed graph (each unordered pair of nodes is saved once)\n# FromNodeId\tToNodeId\n")
        f.write("".join("%d\t%d\n" % (a, b) for a, b in zip(s, d)))
    n, e = C.convert(str(tmp_path / "amazon0505.txt"), str(tmp_path / "o.npz"))
    check(tmp_path / "o.npz", s, d, n, e)
    lower = s >= d                                                          # a symmetric .mtx stores the lower triangle
    ls, ld = s[lower], d[lower]
    with open(tmp_path / "g.mtx", "w") as f:
        f.write("%%MatrixMarket matrix coordinate pattern symmetric\n% a comment\n" + "%d %d %d\n" % (N, N, len(ls)))
        f.write("".join("%d %d\n" % (a + 1, b + 1) for a, b in zip(ls, ld)))
    n, e = C.convert(str(tmp_path / "g.mtx"), str(tmp_path / "o2.npz"))
    off = ls != ld
    check(tmp_path / "o2.npz", np.concatenate([ls, ld[off]]), np.concatenate([ld, ls[off]]), N, e)


def test_refusals(tmp_path):
    s, d = edges()
    np.savez(tmp_path / "done.npz", src_li=s, dst_li=d, num_nodes=N)
    with pytest.raises(ValueError, match="already has"):
        C.convert(str(tmp_path / "done.npz"), str(tmp_path / "o.npz"))
    np.savez(tmp_path / "bad.npz", edge_index=np.stack([s, d]), num_nodes=5)
    with pytest.raises(ValueError, match="beyond num_nodes"):
        C.convert(str(tmp_path / "bad.npz"), str(tmp_path / "o.npz"))
    save_npz(tmp_path / "rect.npz", coo_matrix((np.ones(2), ([0, 1], [2, 3])), shape=(4, 6)))
    with pytest.raises(ValueError, match="not square"):
        C.convert(str(tmp_path / "rect.npz"), str(tmp_path / "o.npz"))


def test_bench_uses_a_real_graph_file_when_one_is_on_the_box(tmp_path, monkeypatch):
    """bench.py looks for the real Reddit / ogbn-products files and measures them when present (it says so in
    `data`); with nothing there it measures the synthetic shape.  The finder and the loader, on fabricated files."""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setenv("TCGNN_DATA_DIR", str(tmp_path))
    monkeypatch.setenv("HOME", str(tmp_path / "nohome"))
    if not any(os.path.exists(os.path.join(r, f)) for r in (os.path.join(ROOT, "dataset"), "/data") for f in bench.REAL_GRAPH_FILES["reddit"]):
        assert bench.find_real_graph("reddit") is None
    s, d = edges()
    (tmp_path / "reddit").mkdir()
    save_npz(tmp_path / "reddit" / "reddit_graph.npz", coo_matrix((np.ones(len(s)), (s, d)), shape=(N, N)))
    found = bench.find_real_graph("reddit")
    assert found == str(tmp_path / "reddit" / "reddit_graph.npz")
    rp, col = bench.load_real_graph(found, "cpu")
    want_rp, want_col = csr_of(s, d, N)
    assert np.array_equal(rp.numpy(), want_rp) and np.array_equal(col.numpy(), want_col)
    np.savez(tmp_path / "reddit.npz", src_li=s, dst_li=d, num_nodes=N)        # the reference's own schema is taken first
    assert bench.find_real_graph("reddit") == str(tmp_path / "reddit.npz")
    rp2, col2 = bench.load_real_graph(str(tmp_path / "reddit.npz"), "cpu")
    assert np.array_equal(rp2.numpy(), want_rp) and np.array_equal(col2.numpy(), want_col)
    raw = tmp_path / "ogbn_products" / "raw"; raw.mkdir(parents=True)          # an OGB raw directory: each undirected edge once
    with gzip.open(str(raw / "edge.csv.gz"), "wt") as f:
        f.write("".join("%d,%d\n" % (a, b) for a, b in zip(s, d)))
    with gzip.open(str(raw / "num-node-list.csv.gz"), "wt") as f:
        f.write("%d\n" % N)
    rp3, col3 = bench.load_real_graph(bench.find_real_graph("ogbn-products"), "cpu")
    w3 = csr_of(np.concatenate([s, d]), np.concatenate([d, s]), N)
    assert np.array_equal(rp3.numpy(), w3[0]) and np.array_equal(col3.numpy(), w3[1])
