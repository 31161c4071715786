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

  python tools/convert_dataset.py IN OUT_PREFIX --shards P [--symmetrize] [--no-self-loops] [--chunk-pairs K] [--tmp DIR]

Row-sharded output for graphs that cannot exist as ONE int32 CSR (BASELINE.json configs[4], ogbn-papers100M: 3.23 G symmetrised
edges > 2^31; the reference's single-process loader, dataset.py:94-104, has no such form): the int64 pair list
is STREAMED in chunks of K pairs and the global CSR is never materialised -
  pass 1  row degrees (one int64 counter per node) -> P contiguous blocks of whole 16-row windows balanced by nnz
          (tcgnn_shard.partition_rows' rule on the degree prefix sums);
  pass 2  every chunk's pairs are dealt to their source row's owner and appended to P spill files (12 bytes per pair);
  pass 3  one rank at a time: sort by (row, column), merge duplicates (dataset.py:94-99's tocsr()), local int32 row pointers,
          column ids ALREADY in the padded-gather numbering of tcgnn_shard.ShardLayout (owner * H + row inside the owner's block)
          -> OUT_PREFIX.rank<p>of<P>.npz with keys row_pointers, column_index, bounds, rank, world, num_nodes, H, raw_pairs.
`tcgnn_shard.RowShard.from_shard_file` loads one such file per rank; peak host memory is one shard (+ the degree array).
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


def iter_pairs(path, fmt, chunk):
    """Yield (src, dst) int64 chunks of at most `chunk` pairs without holding the whole list where the container allows it:
    .npy and UNCOMPRESSED .npz members are memory-mapped (OGB's data.npz is a plain np.savez); everything else is read by its
    reader and sliced.  -> (iterator factory, num_nodes or None).  The factory can be called again for another pass."""
    def mmap_member(npz_path, key):
        import zipfile
        with zipfile.ZipFile(npz_path) as z:
            info = z.getinfo(key + ".npy")
            if info.compress_type != zipfile.ZIP_STORED:
                return None
            with z.open(info) as f:
                version = np.lib.format.read_magic(f)
                shape, fortran, dtype = np.lib.format.read_array_header_1_0(f) if version == (1, 0) else np.lib.format.read_array_header_2_0(f)
                header_len = f.tell()
            with open(npz_path, "rb") as raw:   # offset of the member's data inside the zip file
                raw.seek(info.header_offset)
                lh = raw.read(30)
                name_len, extra_len = int.from_bytes(lh[26:28], "little"), int.from_bytes(lh[28:30], "little")
            offset = info.header_offset + 30 + name_len + extra_len + header_len
            return np.memmap(npz_path, dtype=dtype, mode="r", offset=offset, shape=shape, order="F" if fortran else "C")

    n = None
    arr = None
    if fmt == "edge-index" and str(path).endswith(".npy"):
        arr = np.load(path, mmap_mode="r")
    elif fmt in ("ogb-npz", "edge-index"):
        obj = np.load(path)
        key = "edge_index" if "edge_index" in obj.files else obj.files[0]
        arr = mmap_member(path, key)
        if arr is None:
            arr = obj[key]
        if "num_nodes_list" in obj.files:
            n = int(np.asarray(obj["num_nodes_list"]).reshape(-1)[0])
        elif "num_nodes" in obj.files:
            n = int(np.asarray(obj["num_nodes"]).reshape(-1)[0])
    if arr is not None:
        if arr.ndim != 2 or 2 not in arr.shape:
            raise ValueError("edge_index must be [2, E] or [E, 2], got %s" % (arr.shape,))
        by_rows = arr.shape[0] == 2
        total = arr.shape[1] if by_rows else arr.shape[0]

        def it():
            for a in range(0, total, chunk):
                blk = np.asarray(arr[:, a: a + chunk] if by_rows else arr[a: a + chunk].T)
                yield blk[0].astype(np.int64), blk[1].astype(np.int64)
        return it, n
    s, d, n = READERS[fmt](path)

    def it2():
        for a in range(0, len(s), chunk):
            yield s[a: a + chunk], d[a: a + chunk]
    return it2, n


def convert_sharded(path, out_prefix, shards, fmt="auto", symmetrize=False, drop_self_loops=False, chunk=1 << 26, tmp=None, blk_h=16):
    """See the module docstring.  -> list of (file, rows, nnz) per rank."""
    import tempfile
    fmt = detect_format(path) if fmt == "auto" else fmt
    factory, n = iter_pairs(path, fmt, chunk)
    # ---- pass 1: degrees (duplicates included: the balance is by raw pairs, the merge happens per shard)
    top = -1
    raw = 0
    deg = None
    for s, d in factory():
        if len(s) == 0:
            continue
        if s.min() < 0 or d.min() < 0:
            raise ValueError("negative node id")
        top = max(top, int(s.max()), int(d.max()))
        need = (n if n is not None else top + 1)
        if deg is None or need > len(deg):
            grown = np.zeros(max(need, top + 1), np.int64)
            if deg is not None:
                grown[: len(deg)] = deg
            deg = grown
        keep = s != d if drop_self_loops else slice(None)
        deg += np.bincount(s[keep], minlength=len(deg))
        if symmetrize:
            deg += np.bincount(d[keep], minlength=len(deg))
        raw += len(s)
    if n is None:
        n = top + 1
    elif top >= n:
        raise ValueError("node id %d beyond num_nodes %d" % (top, n))
    if deg is None:
        deg = np.zeros(n, np.int64)
    deg = deg[:n] if len(deg) >= n else np.concatenate([deg, np.zeros(n - len(deg), np.int64)])
    rp = np.zeros(n + 1, np.int64); np.cumsum(deg, out=rp[1:])
    del deg
    nw = (n + blk_h - 1) // blk_h
    win_end = rp[np.minimum(np.arange(1, nw + 1, dtype=np.int64) * blk_h, n)]
    bounds = [0]
    for p in range(1, shards):    # (tcgnn_shard.partition_rows, restated on the prefix sums so that this tool imports nothing from the package)
        w = int(np.searchsorted(win_end, rp[n] * p / shards, side="left")) + 1
        w = min(max(w, bounds[-1] // blk_h), nw)
        bounds.append(min(w * blk_h, n))
    bounds.append(n)
    del rp, win_end
    b = np.asarray(bounds, np.int64)
    rows = b[1:] - b[:-1]
    H = max(blk_h, int((rows.max() + blk_h - 1) // blk_h * blk_h))
    if H * shards >= 2 ** 31:
        raise ValueError("the gathered numbering (%d ranks x %d rows) does not fit int32 column ids" % (shards, H))
    # ---- pass 2: deal the pairs to spill files
    tmpdir = tempfile.mkdtemp(prefix="tcgnn_shards_", dir=tmp)
    spill = [open(os.path.join(tmpdir, "rank%d.bin" % p), "wb") for p in range(shards)]
    rec = np.dtype([("row", np.int32), ("col", np.int64)])
    try:
        for s, d in factory():
            if drop_self_loops:
                keep = s != d
                s, d = s[keep], d[keep]
            if symmetrize:
                s, d = np.concatenate([s, d]), np.concatenate([d, s])
            owner = np.searchsorted(b[1:], s, side="right")
            order = np.argsort(owner, kind="stable")
            s, d, owner = s[order], d[order], owner[order]
            cuts = np.searchsorted(owner, np.arange(shards + 1))
            for p in range(shards):
                lo, hi = cuts[p], cuts[p + 1]
                if hi > lo:
                    out = np.empty(hi - lo, rec)
                    out["row"] = s[lo:hi] - b[p]
                    out["col"] = d[lo:hi]
                    out.tofile(spill[p])
        for f in spill:
            f.close()
        # ---- pass 3: one rank at a time
        written = []
        for p in range(shards):
            pairs = np.fromfile(os.path.join(tmpdir, "rank%d.bin" % p), dtype=rec)
            key = pairs["row"].astype(np.int64) * n + pairs["col"]
            raw_p = len(key)
            del pairs
            key = np.unique(key)                                   # sorted by (row, column), duplicates merged
            if len(key) >= 2 ** 31:
                raise ValueError("rank %d holds %d edges: beyond int32 row pointers - use more shards" % (p, len(key)))
            lrow, gcol = key // n, key % n
            del key
            lrp = np.zeros(int(rows[p]) + 1, np.int32)
            np.cumsum(np.bincount(lrow, minlength=int(rows[p])), out=lrp[1:])
            owner = np.searchsorted(b[1:], gcol, side="right")
            lcol = (owner * H + (gcol - b[owner])).astype(np.int32)   # ShardLayout.remap
            name = "%s.rank%dof%d.npz" % (out_prefix, p, shards)
            np.savez(name, row_pointers=lrp, column_index=lcol, bounds=b, rank=np.int64(p), world=np.int64(shards), num_nodes=np.int64(n), H=np.int64(H),
                     raw_pairs=np.int64(raw_p))
            written.append((name, int(rows[p]), int(len(lcol))))
            del lrow, gcol, owner, lcol, lrp
        return written
    finally:
        for f in spill:
            if not f.closed:
                f.close()
        for p in range(shards):
            try:
                os.remove(os.path.join(tmpdir, "rank%d.bin" % p))
            except OSError:
                pass
        try:
            os.rmdir(tmpdir)
        except OSError:
            pass


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("input"); ap.add_argument("output")
    ap.add_argument("--format", default="auto", choices=["auto"] + sorted(READERS))
    ap.add_argument("--symmetrize", action="store_true"); ap.add_argument("--no-self-loops", action="store_true")
    ap.add_argument("--shards", type=int, default=0, help="write P row-sharded files OUTPUT.rank<p>of<P>.npz instead of one npz (graphs beyond one int32 CSR)")
    ap.add_argument("--chunk-pairs", type=int, default=1 << 26, help="pairs per streamed chunk of the sharded conversion")
    ap.add_argument("--tmp", default=None, help="directory for the sharded conversion's spill files (12 bytes per pair)")
    a = ap.parse_args(argv)
    if a.shards > 0:
        for name, rows, nnz in convert_sharded(a.input, a.output, a.shards, a.format, a.symmetrize, a.no_self_loops, a.chunk_pairs, a.tmp):
            print("%s: %d rows, %d edges" % (name, rows, nnz))
        return 0
    n, e = convert(a.input, a.output, a.format, a.symmetrize, a.no_self_loops)
    print("%s: %d nodes, %d pairs -> %s" % (a.input, n, e, a.output))


if __name__ == "__main__":
    sys.exit(main())
