#!/usr/bin/env python3
"""One rank's share of bench.py's N-GPU weak-scaling workload, on ONE GPU and without a process group: rank 0's rows of the
N x 232 965-node graph (114.6 M edges, columns over all N blocks), its sharded plan and the local SpMM on a replicated X.
Shows which kernel the shard takes and what it costs per GPU at N = 1, 2, 4, 8 before the driver runs the real thing.
TCGNN_SHARD_LOCALITY=f (0..1) puts that fraction of the edges inside the rank's own column block (a partitioned graph; the
driver's workload is f = 0: columns uniform over all blocks).
usage: bench_shard_shape.py [N ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import numpy as np, torch
import tcgnn_graph as G, tcgnn_shard as S, tcgnn_capi as c

dev = torch.device("cuda:0")
n0, nnz0, _, _ = G.SHAPES["reddit"]
D = 64
for world in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    n_global = n0 * world
    g = torch.Generator(device=dev).manual_seed(0)
    rows = torch.randint(0, n0, (nnz0,), device=dev, generator=g)
    cols = torch.randint(0, n_global, (nnz0,), device=dev, generator=g)
    f_local = float(os.environ.get("TCGNN_SHARD_LOCALITY", "0"))
    if f_local > 0:   # rank 0's own block is columns [0, n0)
        own = torch.rand(nnz0, device=dev, generator=g) < f_local
        cols = torch.where(own, torch.randint(0, n0, (nnz0,), device=dev, generator=g), cols)
    keys = torch.unique(rows.long() * n_global + cols.long())
    rows, cols = keys // n_global, keys % n_global
    counts = torch.bincount(rows, minlength=n0)
    lrp = torch.zeros(n0 + 1, dtype=torch.int64, device=dev); lrp[1:] = torch.cumsum(counts, 0)
    E = int(keys.numel())
    shard = S.RowShard(rank=0, world_size=world, device=dev, bounds=[p * n0 for p in range(world + 1)],
                       local=(lrp.cpu().numpy().astype(np.int32), cols.cpu().numpy()))
    del rows, cols, keys
    info = c.PlanInfo(); c.check(c.lib.tcgnn_plan_get_info(shard.ops.plan, c.ctypes.byref(info)), "info")
    xg = torch.randn(shard.layout.num_cols, D, device=dev, generator=g)
    for _ in range(3): y = shard.ops.spmm(xg)
    shard.ops.set_timing(10)
    for _ in range(10): y = shard.ops.spmm(xg)
    t = shard.ops.read_timing(); shard.ops.set_timing(0)
    assert torch.isfinite(y).all()
    deg = torch.from_numpy(np.diff(shard.local_row_pointers)).float().to(dev)
    ones = torch.ones(shard.layout.num_cols, 16, device=dev)
    assert torch.equal(shard.ops.spmm(ones)[:, 0], deg), "A @ 1 != degree"
    print("N=%d  rows %d  cols %d  nnz %d  lds_ranges %d  buckets %d : kernel %.3f ms (min %.3f) = %.1f GTEPS per GPU; X16 %.0f MB" % (
        world, n0, shard.layout.num_cols, E, info.lds_ranges, info.column_buckets, np.median(t), np.min(t), E / np.median(t) / 1e6, shard.layout.num_cols * 128 / 1e6), flush=True)
    shard.ops.close(); del shard, xg, y
    torch.cuda.empty_cache()
