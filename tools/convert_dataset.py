#!/usr/bin/env python3
"""One-off converter: the native files of the graphs the reference is benchmarked on -> the `.npz` schema its loader reads
(`src_li`, `dst_li`, `num_nodes`: /root/reference/dataset.py:74-80; ours: tcgnn_graph.TCGNN_dataset).  SURVEY.md 8 row f2.

Nothing here runs on the hot path and nothing is downloaded: point it at files that are already on the box.

  python tools/convert_dataset.py IN OUT.npz [--format auto|scipy-npz|ogb-raw|ogb-npz|edge-index|edge-list|mtx] [--symmetrize] [--no-self-loops]

Formats (`auto` decides from the name and the keys):
  scipy-npz   a matrix written by scipy.sparse.save_npz - DGL's RedditDataset ships `reddit_graph.npz` (and
              `reddit_self_loop_graph.npz`) in this form: row = source, col = destination
  ogb-raw     an OGB node-property dataset directory (or its `raw/`): `edge.csv[.gz]` (one "src,dst" per line) and
              `num-node-list.csv[.gz]`; ogbn-products stores each undirected edge once: pass --symmetrize
  ogb-npz     OGB's large-graph form (`data.npz` of ogbn-papers100M): `edge_index` [2, E], `num_nodes_list`
  edge-index  a .npy / .npz / .pt-less dump of a [2, E] (or [E, 2]) integer array (PyG's `edge_index`)
  edge-list   text, one "src dst" pair per line, `#` / `%` comment lines ignored (SNAP: amazon0505, com-amazon, ...; the
              reference's own txt path, dataset.py:45-60, wants the comments stripped - this does it)
  mtx         Matrix Market coordinate file (SuiteSparse), 1-based, `symmetric` header honoured

The output keeps the RAW pair list (duplicates and all): the loader merges duplicates when it builds the CSR, and the reference's
`num_edges` is the raw count (dataset.py:79).  `--symmetrize` appends the reversed pairs, `--no-self-loops` drops i -> i.
"""
import argparse
import gzip
import io
import os
import sys

import numpy as np


def _open_text(path):
    return io.TextIOWrapper(gzip.open(path, "rb")) if str(path).endswith(".gz") else open(path, "r")


def _first_existing(*paths):
    for p in paths:
        if os.path.exists(p):
            return p
    return None


def _pairs_from_text(path, sep=None, one_based=False):
    """Two integer columns; comment lines (# or %) skipped.  np.loadtxt is slow on 1e8 lines: read in blocks."""
    src, dst = [], []
    with _open_text(path) as f:
        block = []
        for line in f:
            if not line or line[0] in "#%" or not line.strip():
                continue
            block.append(line)
            if len(block) >= 1 << 20:
                a = np.loadtxt(block, dtype=np.int64, delimiter=sep, usecols=(0, 1), ndmin=2)
                src.append(a[:, 0]); dst.append(a[:, 1]); block = []
        if block:
            a = np.loadtxt(block, dtype=np.int64, delimiter=sep, usecols=(0, 1), ndmin=2)
            src.append(a[:, 0]); dst.append(a[:, 1])
    s = np.concatenate(src) if src else np.zeros(0, np.int64)
    d = np.concatenate(dst) if dst else np.zeros(0, np.int64)
    if one_based:
        s, d = s - 1, d - 1
    return s, d


def read_scipy_npz(path):
    from scipy.sparse import load_npz
    m = load_npz(path).tocoo()
    if m.shape[0] != m.shape[1]:
        raise ValueError("adjacency matrix is not square: %s" % (m.shape,))
    return m.row.astype(np.int64), m.col.astype(np.int64), int(m.shape[0])


def read_ogb_raw(path):
    raw = path if os.path.basename(os.path.normpath(path)) == "raw" else os.path.join(path, "raw")
    edge = _first_existing(os.path.join(raw, "edge.csv.gz"), os.path.join(raw, "edge.csv"))
    nnl = _first_existing(os.path.join(raw, "num-node-list.csv.gz"), os.path.join(raw, "num-node-list.csv"))
    if edge is None or nnl is None:
        raise FileNotFoundError("expected edge.csv[.gz] and num-node-list.csv[.gz] under %s" % raw)
    s, d = _pairs_from_text(edge, sep=",")
    with _open_text(nnl) as f:
        counts = [int(x) for x in f.read().split() if x.strip()]
    if len(counts) != 1:
        raise ValueError("%s lists %d graphs; a node-property dataset has one" % (nnl, len(counts)))
    return s, d, counts[0]


def _as_pairs(a):
    a = np.asarray(a)
    if a.ndim != 2 or 2 not in a.shape:
        raise ValueError("edge_index must be [2, E] or [E, 2], got %s" % (a.shape,))
    if a.shape[0] != 2:
        a = a.T
    return a[0].astype(np.int64), a[1].astype(np.int64)


def read_ogb_npz(path):
    obj = np.load(path)
    s, d = _as_pairs(obj["edge_index"])
    n = int(np.asarray(obj["num_nodes_list"]).reshape(-1)[0])
    return s, d, n


def read_edge_index(path):
    if str(path).endswith(".npy"):
        s, d = _as_pairs(np.load(path))
        n = None
    else:
        obj = np.load(path)
        key = "edge_index" if "edge_index" in obj.files else obj.files[0]
        s, d = _as_pairs(obj[key])
        n = int(np.asarray(obj["num_nodes"]).reshape(-1)[0]) if "num_nodes" in obj.files else None
    return s, d, n


def read_edge_list(path):
    s, d = _pairs_from_text(path)
    return s, d, None


def read_mtx(path):
    with _open_text(path) as f:
        header = f.readline().lower()
        if not header.startswith("%%matrixmarket") or "coordinate" not in header:
            raise ValueError("not a Matrix Market coordinate file: %s" % header.strip())
        symmetric = "symmetric" in header
        line = f.readline()
        while line.startswith("%"):
            line = f.readline()
        rows, cols, _ = (int(x) for x in line.split()[:3])
    if rows != cols:
        raise ValueError("adjacency matrix is not square: %d x %d" % (rows, cols))
    s, d = _pairs_from_text(path, one_based=True)      # (the header lines start with %, the size line is dropped below)
    s, d = s[1:], d[1:]                                # size line "rows cols nnz" parsed as a pair
    if symmetric:
        off = s != d
        s, d = np.concatenate([s, d[off]]), np.concatenate([d, s[off]])
    return s, d, rows


READERS = {"scipy-npz": read_scipy_npz, "ogb-raw": read_ogb_raw, "ogb-npz": read_ogb_npz, "edge-index": read_edge_index,
           "edge-list": read_edge_list, "mtx": read_mtx}


def detect_format(path):
    p = str(path)
    if os.path.isdir(p):
        return "ogb-raw"
    if p.endswith(".mtx") or p.endswith(".mtx.gz"):
        return "mtx"
    if p.endswith(".npy"):
        return "edge-index"
    if p.endswith(".npz"):
        files = set(np.load(p).files)
        if {"src_li", "dst_li", "num_nodes"} <= files:
            raise ValueError("%s already has the reference's schema" % p)
        if "format" in files and "shape" in files:
            return "scipy-npz"
        if "num_nodes_list" in files:
            return "ogb-npz"
        return "edge-index"
    return "edge-list"


def convert(path, out, fmt="auto", symmetrize=False, drop_self_loops=False):
    """Returns (num_nodes, raw pair count written)."""
    fmt = detect_format(path) if fmt == "auto" else fmt
    s, d, n = READERS[fmt](path)
    if len(s) and (s.min() < 0 or d.min() < 0):
        raise ValueError("negative node id")
    top = int(max(s.max(), d.max())) + 1 if len(s) else 0
    if n is None:
        n = top                                       # dataset.py:61: max id + 1 when the file does not say
    elif top > n:
        raise ValueError("node id %d beyond num_nodes %d" % (top - 1, n))
    if n >= 2 ** 31 or len(s) * (2 if symmetrize else 1) >= 2 ** 31:
        raise ValueError("the path's column_index / row_pointers are int32 (TCGNN.cpp:165-170): %d nodes, %d pairs do not fit; shard by rows first (tcgnn_shard)" % (n, len(s)))
    if drop_self_loops:
        keep = s != d
        s, d = s[keep], d[keep]
    if symmetrize:
        s, d = np.concatenate([s, d]), np.concatenate([d, s])
    small = n < 2 ** 31
    np.savez(out, src_li=s.astype(np.int32 if small else np.int64), dst_li=d.astype(np.int32 if small else np.int64), num_nodes=np.int64(n))
    return n, len(s)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("input"); ap.add_argument("output")
    ap.add_argument("--format", default="auto", choices=["auto"] + sorted(READERS))
    ap.add_argument("--symmetrize", action="store_true"); ap.add_argument("--no-self-loops", action="store_true")
    a = ap.parse_args(argv)
    n, e = convert(a.input, a.output, a.format, a.symmetrize, a.no_self_loops)
    print("%s: %d nodes, %d pairs -> %s" % (a.input, n, e, a.output))


if __name__ == "__main__":
    sys.exit(main())
