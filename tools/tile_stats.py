#!/usr/bin/env python3
"""Tile-count table of the sparse-graph translation, in the format of the reference's counting scripts
(3_cnt_TC_blk_SpMM.py / 3_cnt_TC_blk_SDDMM.py: `dataset,origin,reduced,reduction (%)`), extended with
tile fill and blocks per window (logs/16x8_reduction.csv, logs/bpw.csv).

  tools/tile_stats.py --graph_dir tcgnn-ae-graphs citeseer cora        # <graph_dir>/<name>.npz (src_li, dst_li, num_nodes)
  tools/tile_stats.py --synthetic reddit --tile 16x8 --tile 16x16      # seeded shape of tcgnn_graph.SHAPES
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tc-gnn_atc23_amd"))


def csr_of_npz(path):
    # the scripts file an edge (src, dst) under row dst (graph[dst].append(src), 3_cnt_TC_blk_SpMM.py:49)
    from scipy.sparse import coo_matrix
    g = np.load(path, allow_pickle=True)
    n = int(g["num_nodes"])
    m = coo_matrix((np.ones(len(g["src_li"]), dtype=np.int8), (g["dst_li"], g["src_li"])), shape=(n, n)).tocsr()
    return m.indptr.astype(np.int32), m.indices.astype(np.int32)


def main():
    import tcgnn_graph as G
    p = argparse.ArgumentParser()
    p.add_argument("names", nargs="*")
    p.add_argument("--graph_dir", default="tcgnn-ae-graphs/")
    p.add_argument("--synthetic", action="append", default=[])
    p.add_argument("--scale", type=float, default=1.0)
    p.add_argument("--tile", action="append", default=[], help="HxW, default 16x8")
    a = p.parse_args()
    tiles = [tuple(int(v) for v in t.split("x")) for t in (a.tile or ["16x8"])]
    print("dataset,tile,origin,origin_eff,reduced,reduced_eff,reduction (%),origin BPW,reduced BPW")
    jobs = [(n, lambda n=n: csr_of_npz(os.path.join(a.graph_dir, n + ".npz"))) for n in a.names]
    for s in a.synthetic:
        def load(s=s):
            rp, ci, _ = G.synthetic_shape(s, scale=a.scale)[:3]
            return rp, ci
        jobs.append((s + "(synthetic)", load))
    for name, load in jobs:
        rp, ci = load()
        for h, w in tiles:
            st = G.tile_statistics(rp, ci, h, w)
            print("%s,%dx%d,%d,%.4f,%d,%.4f,%.2f,%.1f,%.1f" % (name, h, w, st["sliding_tiles"], st["sliding_fill"], st["condensed_tiles"],
                                                             st["condensed_fill"], st["reduction_pct"], st["sliding_per_window"], st["condensed_per_window"]))


if __name__ == "__main__":
    main()
