#!/usr/bin/env python3
"""CHECKER-SIDE UTILITY (lives under oracle/: it loads the compiled reference; nothing in the product imports it).
Host sparse-graph translation: this repo's tcgnn_preprocess against the reference's own TCGNN.preprocess (the unmodified
TCGNN.cpp compiled into oracle/_ref by oracle/build_ref.sh - only possible where /root/reference exists), same graph, same
machine.  Prints ns per edge for both and checks the outputs are identical."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd"), os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import numpy as np, torch
import graphs
from oracle import oracle as O
import tcgnn_capi as c

assert O.ref_available(), "oracle/_ref is not built (needs /root/reference: run oracle/build_ref.sh)"
ref = O.load_ref()
for name, n, deg in (("citeseer-shape", 3327, 2.8), ("pubmed-shape", 19717, 4.5), ("amazon0505-shape", 410236, 11.9)):
    rp, col = graphs.uniform_graph(n, deg, seed=1)
    E = len(col); nw = (n + 15) // 16
    trp, tcol = torch.from_numpy(rp), torch.from_numpy(col)
    bp_r = torch.zeros(nw + 1, dtype=torch.int32); e2c_r = torch.zeros(E, dtype=torch.int32); e2r_r = torch.zeros(E, dtype=torch.int32)
    bp = np.zeros(nw + 1, np.int32); e2c = np.zeros(E, np.int32); e2r = np.zeros(E, np.int32); tcb = c._i64(0)
    def ours(threads):
        t0 = time.perf_counter()
        c.check(c.lib.tcgnn_preprocess(col.ctypes.data, rp.ctypes.data, n, 16, 8, bp.ctypes.data, nw, e2c.ctypes.data, e2r.ctypes.data, c.ctypes.byref(tcb), threads), "sgt")
        return time.perf_counter() - t0
    tw = time.perf_counter()
    while time.perf_counter() - tw < 2.0: ours(0)          # the container's cores take a second or two to clock up
    fd = os.open(os.devnull, os.O_WRONLY); sv = os.dup(1); sys.stdout.flush(); os.dup2(fd, 1)
    t_ref = 1e9
    for _ in range(2):
        t0 = time.perf_counter(); ref.preprocess(tcol, trp, n, 16, 8, bp_r, e2c_r, e2r_r); t_ref = min(t_ref, time.perf_counter() - t0)
    sys.stdout.flush(); os.dup2(sv, 1)
    t1 = min(ours(1) for _ in range(5))
    best = min(ours(0) for _ in range(5))
    same = np.array_equal(bp[:nw], bp_r.numpy()[:nw]) and np.array_equal(e2c, e2c_r.numpy()) and np.array_equal(e2r, e2r_r.numpy())
    print("%-18s E=%8d  reference %8.1f ns/edge | this repo, 1 thread %6.1f ns/edge, %d threads %6.1f ns/edge | identical %s" % (
        name, E, t_ref / E * 1e9, t1 / E * 1e9, os.cpu_count(), best / E * 1e9, same))
