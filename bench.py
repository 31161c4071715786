#!/usr/bin/env python3
"""bench.py - the headline measurement of BASELINE.json:
    "SpMM/SDDMM GTEPS + GCN/AGNN ms/epoch, Reddit h=64, 1xMI355X".

    python bench.py --gpus N --steps K --warmup W          (N = 1: plain python; N > 1: either under torchrun - RANK / WORLD_SIZE in
                                                            the environment - or plain: it then starts its own N ranks, self_launch)

One STEP = one pass of the hot path over one batch of synthetic input = one `TCGNN.forward` call
(fp16 staging pass + the SpMM kernel) on the Reddit-shaped graph at D = 64, inputs resident in HBM.
`value` = traversed edges per second of the whole job in GTEPS (E * K * N / t).  The other
quantities the metric names are measured in the same run and reported under `extra`
(SDDMM and SpMM-AGNN GTEPS, GCN / AGNN ms per epoch, host and device SGT times).

workload (config.workload): BASELINE.json configs[2] "Reddit GCN 2-layer hidden=64" -
N = 232 965 nodes, nnz = 114 615 892, seeded synthetic symmetric graph (no dataset
or network on the box), features randn, labels ones (dataset.py:115,122).

N > 1 (weak scaling): the papers100M pattern of configs[4] in miniature - the graph has N x 232 965
nodes, rank p owns the rows of its 232 965 nodes (114.6 M nnz with columns over ALL N x 232 965
nodes), a step = all-gather of the X row blocks over RCCL + the local SpMM (the replicated-X variant, no collective in the
step, is timed next to it and reported under `extra`).  Per-GPU work is fixed.

roofline: the SpMM kernel is HBM-bound by its algorithmic bytes B = 4(N+1) + 4E + 8ND
(SURVEY.md 8d); `achieved` = B / (mean kernel time from HIP events on the launch stream, recorded
inside the timed steps), `peak` = 8 TB/s (MI355X_MICROARCH.md).  `traffic` is the PMC-measured
HBM bytes per launch when profiles/ holds a measurement for this workload, else null.

`roofline.traffic`, `mfma_busy`, `mfma_useful_frac` come from profiles/r*/traffic.json ONLY when that file was collected from the
sources this process runs (tcgnn_capi.build_id(), tools/collect_profiles.py) - else null with the reason in `traffic_source`;
`mfma_useful_tflops` / `mfma_peak_frac` (2 E D over the kernel time of this run against 2.5 PFLOP/s) are always live.
`datasets` repeats those figures per dataset: the headline graph, its R-MAT and community (SBM) variants, and the ogbn-products
shape at D = 128 (BASELINE.json configs[3]).

cpu_baseline: the row-parallel CSR gather-add (the DGL-CPU-style aggregation) of oracle/cpu_baseline.c, built on the box with
-march=native, and torch.sparse.mm next to it, on the host cores, same graph and D, rank 0, N = 1 only.
"""
import argparse
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"
GRAPH_NOTES = {
    "uniform": "seeded uniform symmetric, canonical CSR",
    "sbm_reddit": "seeded 50-community SBM calibrated to real Reddit's TC-block count (22.5 % of the edges inside a community; "
                  "tcgnn_graph.SBM_REDDIT_P_IN), symmetric, canonical CSR",
    "rmat": "seeded R-MAT 0.57/0.19/0.19/0.05, symmetric, canonical CSR",
    "sbm": "seeded 50-community SBM, 90 % of the edges inside a community, symmetric, canonical CSR",
}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--dim", type=int, default=64, help="hidden width D of the timed SpMM")
    p.add_argument("--shape", type=str, default="reddit")
    p.add_argument("--graph", type=str, default="sbm_reddit",
                   help="generator of the headline graph (tcgnn_graph.GENERATORS).  Default since r05: sbm_reddit, the 50-community graph "
                        "calibrated to real Reddit's TC-block count (13.63 M 16x8 blocks against 13.57 M, /root/reference/logs/reduce_blocks.csv:18); "
                        "r01-r04 reported the uniform graph, whose figures stay in `summary` (uniform_*)")
    p.add_argument("--scale", type=float, default=1.0, help="shrink the graph (debug only; the reported config is scale 1)")
    p.add_argument("--epochs", type=int, default=10, help="timed epochs of the GCN / AGNN legs")
    p.add_argument("--no-extra", action="store_true", help="skip SDDMM / epoch / CPU legs (profiling runs)")
    p.add_argument("--no-cpu", action="store_true")
    p.add_argument("--exchange", choices=["auto", "always", "never"], default="always",
                   help="N > 1: all-gather X inside every step (always: the sharded-GCN step - the SpMM input is the previous layer's "
                        "row-sharded output, so a step without the gather is not the workload), replicate X and time the local SpMM "
                        "only (never), or exchange only when the global feature matrix does not fit one GPU (auto)")
    p.add_argument("--all-generators", action="store_true", help="also the r02 graph variants (SBM with hubs, under random ids, relabelled)")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--plan-only", action="store_true",
                   help="no GPU, no process group: print the per-rank memory plan of the row-sharded workload (--shape, default "
                        "ogbn-papers100M, --dim, --gpus ranks) against 288 GB of HBM per GPU and exit")
    p.add_argument("--shard-files", default=None,
                   help="--plan-only: template of the shard files written by tools/convert_dataset.py --shards (\"g.rank{rank}of{world}.npz\"): rows and "
                        "edges per rank are read from them instead of assumed even")
    return p.parse_args()


LEG_SETTLE_S = 0.06   # untimed calls in front of every secondary kernel leg (timed_leg)
EPOCH_WARMUP = 25     # dry epochs in front of every timed training leg (the harness's own default, 9, is main_tcgnn.py's habit: 26 ms of a GCN)


def settle(fn, seconds=0.03, max_calls=256):
    """Untimed calls of the step for ~30 ms before the W warm-up steps.  The first few dozen launches after set-up run up to
    15 % slower than the steady state (clocks ramp, and the Infinity Cache has yet to hold the step's working set); a short
    --steps/--warmup run would time that transient (10 steps after 2 warm-ups: 0.67 ms per step against 0.58).  Neither a
    cache-resident matrix product nor a stream of large copies reproduces the effect, so it is the step itself that runs.
    Returns the number of calls made (a profiler's per-kernel average of the same command includes them: the JSON line
    therefore also carries the mean over ALL launches next to the mean over the K timed ones)."""
    import torch
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds and n + 8 <= max_calls:
        for _ in range(8):
            fn()
        n += 8
        torch.cuda.synchronize()
    return n


def sync_time(fn, steps, warmup, barrier):
    for _ in range(warmup):
        fn()
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize(); barrier()
    return time.perf_counter() - t0


def spmm_bytes(n, e, d):
    return 4 * (n + 1) + 4 * e + 8 * n * d


def sddmm_bytes(n, e, d):
    return 4 * (n + 1) + 8 * e + 4 * n * d


def load_profile(kernel, workload, val=None, bwd=None):
    """PMC numbers of (kernel, workload) from the newest profiles/r*/traffic.json - ONLY if that file was collected from the
    sources this process runs (tcgnn_capi.build_id(); tools/collect_profiles.py writes it).  -> (row or None, note).
    A stale or missing profile yields None and says why: the line never quotes counters of another kernel version."""
    import glob
    import tcgnn_capi
    bid = tcgnn_capi.build_id()
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")))
    if not files:
        return None, "no profiles/r*/traffic.json"
    try:
        with open(files[-1]) as f:
            doc = json.load(f)
    except (OSError, ValueError) as exc:
        return None, "unreadable %s: %s" % (os.path.relpath(files[-1], ROOT), exc)
    rel = os.path.relpath(files[-1], ROOT)
    if doc.get("build_id") != bid:
        return None, "stale: %s was collected from sources %s, this build is %s (re-run tools/collect_profiles.py)" % (rel, doc.get("build_id"), bid)
    for row in doc.get("rows", []):
        name = row.get("kernel", "")
        if row.get("workload") != workload or name.split("<")[0] != kernel.split("<")[0]:
            continue
        # spmm_kernel / spmm_blocked_kernel exist with and without edge values (last template argument)
        if val is not None and name.split("<")[0] in ("spmm_kernel", "spmm_blocked_kernel") and name.rstrip(">").split(",")[-1].strip() != ("true" if val else "false"):
            continue
        # spmm_sync_kernel<NT, MAXW, VAL, NBUF>
        if val is not None and name.split("<")[0] == "spmm_sync_kernel" and name.rstrip(">").split(",")[2].strip() != ("true" if val else "false"):
            continue
        # agnn_kernel<NT, WAVES, BWD, MAXW>: forward and backward are different instantiations (the edge-valued SpMM on the sliced walk
        # runs the backward one with its score half off: the collector's row averages over both uses)
        if bwd is not None and name.split("<")[0] == "agnn_kernel" and name.rstrip(">").split(",")[2].strip() != ("true" if bwd else "false"):
            continue
        return row, "%s (build %s)" % (rel, bid)
    return None, "%s has no row for %s on %s" % (rel, kernel, workload)


MFMA_PEAK = 2.5e15   # fp16 dense, MI355X_MICROARCH.md


def profile_fields(kernel, workload, flops, kernel_ms, val=False, bwd=None):
    """traffic + MFMA figures for one kernel on one dataset: PMC-measured where a fresh profile exists, live otherwise.
    A leg that runs two kernels ("spmm_lds_kernel + spmm_kernel (cold remainder)") reports the sum of their traffic and the
    MFMA figures of the first."""
    names = [k.split("(")[0].strip() for k in kernel.split(" + ")]
    row, note = load_profile(names[0], workload, val, bwd)
    if row and len(names) > 1:
        row = dict(row)
        for extra_name in names[1:]:
            r2, _ = load_profile(extra_name, workload, val, bwd)
            if r2 is None and extra_name == "agnn_slice_sum_kernel":   # (the pass that adds the sliced walk's addends: not profiled, ~1 GB of streaming)
                note += "; without agnn_slice_sum_kernel"
                continue
            if r2 is None:
                row, note = None, note + "; no row for " + extra_name
                break
            row["hbm_bytes_per_launch"] = row.get("hbm_bytes_per_launch", 0) + r2.get("hbm_bytes_per_launch", 0)
            row["mfma_useful_frac"] = None   # (2 E D over one kernel's MFMA count is meaningless when two kernels share the edges)
    out = {"traffic": row.get("hbm_bytes_per_launch") if row else None, "traffic_source": note,
           "mfma_busy": row.get("mfma_busy") if row else None, "mfma_useful_frac": row.get("mfma_useful_frac") if row else None,
           "l2_hit_rate": row.get("l2_hit_rate") if row else None,
           # useful flops (2 E D) over the kernel time measured in THIS run, against the dense fp16 MFMA peak
           "mfma_useful_tflops": round(flops / (kernel_ms * 1e-3) / 1e12, 2) if kernel_ms and kernel_ms == kernel_ms else None,
           "mfma_peak_frac": round(flops / (kernel_ms * 1e-3) / MFMA_PEAK, 5) if kernel_ms and kernel_ms == kernel_ms else None}
    return out


# The reference's own result tables, RTX 3090: single SpMM kernel at D = 16 over 200 rounds
# (/root/reference/logs/profile.csv:2-15) and 2-layer GCN hidden 16, ms per epoch (/root/reference/logs/RTX3090_GCN.csv:2-15)
REF_RTX3090_KERNEL_MS = {"citeseer": 0.040, "cora": 0.066, "pubmed": 0.147, "ppi": 0.537, "PROTEINS_full": 0.115, "OVCAR-8H": 3.199, "Yeast": 2.786,
                         "DD": 0.814, "SW-620H": 3.197, "amazon0505": 3.682, "artist": 1.643, "com-amazon": 1.744, "soc-BlogCatalog": 1.898,
                         "amazon0601": 1.985}
REF_RTX3090_GCN_EPOCH_MS = {"citeseer": 3.031, "cora": 2.971, "pubmed": 2.793, "ppi": 4.833, "PROTEINS_full": 2.722, "OVCAR-8H": 66.381, "Yeast": 61.057,
                            "DD": 11.429, "SW-620H": 68.017, "amazon0505": 23.806, "artist": 4.994, "com-amazon": 17.365, "soc-BlogCatalog": 10.130,
                            "amazon0601": 20.310}


class quiet_stdout:
    """C-level and Python-level stdout to /dev/null for the duration (the SGT prints the reference's two lines, SAG.profile its
    scraped line): the JSON line must stay the only thing bench.py itself prints."""

    def __enter__(self):
        sys.stdout.flush()
        self.devnull = os.open(os.devnull, os.O_WRONLY); self.saved = os.dup(1); os.dup2(self.devnull, 1)

    def __exit__(self, *exc):
        sys.stdout.flush(); os.dup2(self.saved, 1); os.close(self.saved); os.close(self.devnull)
        return False


def artifact_shapes(seed, epochs=20):
    """the reference's two committed tables re-measured in THIS run, through the harness flow the tables were
    produced by (tcgnn_harness = main_tcgnn.py: `--single_kernel` -> SAG.profile, 200 rounds at D = 16, wall time per call incl.
    launch; `--model gcn --hidden 16` -> ms per epoch after 9 dry epochs).  The artifact graphs are not on the box: same-SIZE
    seeded uniform graphs (tcgnn_graph.SHAPES), which condense worse than the real ones.  -> list of rows."""
    import TCGNN
    import tcgnn_graph as G
    import tcgnn_harness as H
    rows = []
    for name in REF_RTX3090_KERNEL_MS:
        n, nnz, dim, classes = G.SHAPES[name]
        base = ["--synthetic", name, "--classes", str(classes), "--gpu_preprocess", "--seed", str(seed)]
        row = {"shape": name, "N": n, "nnz_target": nnz}
        try:
            with quiet_stdout():
                k = H.run(H.build_parser().parse_args(base + ["--dim", "16", "--hidden", "16", "--single_kernel"]), quiet=True)
                # (the better of two 200-round averages: the first run after torch.cuda.empty_cache() pays the allocator's
                #  hipMallocs inside its timed rounds - seen once as 0.127 ms against 0.019 on the PROTEINS_full shape)
                k2 = H.run(H.build_parser().parse_args(base + ["--dim", "16", "--hidden", "16", "--single_kernel"]), quiet=True)
                k_runs = [round(k["sag_ms"], 4), round(k2["sag_ms"], 4)]
                if k2["sag_ms"] < k["sag_ms"]: k = k2
                e = H.run(H.build_parser().parse_args(base + ["--dim", str(dim), "--hidden", "16", "--model", "gcn", "--epochs", str(epochs)]), quiet=True)
                # (the better of two runs here too: these epochs are ~60 launches of a few microseconds, the host sets their time, and one
                #  stall of the host inside twenty of them - seen once: 2.89 ms against 0.61 on the PROTEINS_full shape - is not the GPU's)
                e2 = H.run(H.build_parser().parse_args(base + ["--dim", str(dim), "--hidden", "16", "--model", "gcn", "--epochs", str(epochs)]), quiet=True)
                e_runs = [round(e["train_ms"], 3), round(e2["train_ms"], 3)]
                if e2["train_ms"] < e["train_ms"]: e = e2
            # (both runs are recorded; the headline figure of a row is their minimum - `method`)
            row.update({"spmm_d16_ms_runs": k_runs, "gcn_h16_ms_per_epoch_runs": e_runs, "method": "min of 2 runs",
                        "nnz": k["nnz"], "spmm_d16_ms": round(k["sag_ms"], 4), "rtx3090_spmm_d16_ms": REF_RTX3090_KERNEL_MS[name],
                        "spmm_speedup_vs_rtx3090": round(REF_RTX3090_KERNEL_MS[name] / k["sag_ms"], 2),
                        "gcn_h16_ms_per_epoch": round(e["train_ms"], 3), "rtx3090_gcn_h16_ms_per_epoch": REF_RTX3090_GCN_EPOCH_MS[name],
                        "gcn_speedup_vs_rtx3090": round(REF_RTX3090_GCN_EPOCH_MS[name] / e["train_ms"], 2)})
        except Exception as exc:   # an extra: must never take the headline down
            row["error"] = str(exc)[:200]
        rows.append(row)
        TCGNN.clear_plan_cache()
        torch.cuda.empty_cache()
    return rows


# "If the real Reddit / OGB files happen to be present on the GPU box, use them and say so."  Neither box has a
# network, so this normally finds nothing and the seeded synthetic graph of the same shape is measured; the line's `data` field says
# which it was.  Looked for (first hit wins) under $TCGNN_DATA_DIR, ~/.dgl, <repo>/dataset, ~/dataset, /data:
REAL_GRAPH_FILES = {
    "reddit": ["reddit.npz", "reddit/reddit.npz", "reddit/reddit_graph.npz", "reddit_graph.npz"],
    "ogbn-products": ["ogbn-products.npz", "ogbn_products.npz", "ogbn_products/raw", "products/raw"],
}


def find_real_graph(shape):
    roots = [os.environ.get("TCGNN_DATA_DIR"), os.path.expanduser("~/.dgl"), os.path.join(ROOT, "dataset"), os.path.expanduser("~/dataset"), "/data"]
    for root in roots:
        if not root:
            continue
        for rel in REAL_GRAPH_FILES.get(shape, []):
            path = os.path.join(root, rel)
            if os.path.exists(path):
                return path
    return None


def load_real_graph(path, device):
    """-> (row_pointers, column_index) int32 on `device`, canonical CSR (duplicates merged, columns sorted) as dataset.py:94-104
    builds it; the reference's own npz schema is read directly, native files through tools/convert_dataset.py's readers (an OGB
    raw directory stores each undirected edge once: symmetrised)."""
    from scipy.sparse import coo_matrix
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import convert_dataset as C
    if os.path.isfile(path) and path.endswith(".npz") and {"src_li", "dst_li", "num_nodes"} <= set(np.load(path).files):
        obj = np.load(path)
        src, dst, n = obj["src_li"].astype(np.int64), obj["dst_li"].astype(np.int64), int(obj["num_nodes"])
    else:
        fmt = C.detect_format(path)
        src, dst, n = C.READERS[fmt](path)
        if n is None:
            n = int(max(src.max(), dst.max())) + 1
        if fmt == "ogb-raw":
            src, dst = np.concatenate([src, dst]), np.concatenate([dst, src])
    csr = coo_matrix((np.ones(len(src), dtype=np.int8), (src, dst)), shape=(n, n)).tocsr()
    csr.sum_duplicates(); csr.sort_indices()
    return torch.from_numpy(csr.indptr.astype(np.int32)).to(device), torch.from_numpy(csr.indices.astype(np.int32)).to(device)


def single_gpu(args):
    import TCGNN
    import tcgnn_graph as G
    import tcgnn_harness as H
    dev = torch.device("cuda:0")
    n, nnz_target, in_dim, classes = G.SHAPES[args.shape]
    if args.scale != 1.0:
        n, nnz_target = int(n * args.scale), int(nnz_target * args.scale * args.scale)
    D = args.dim
    t0 = time.perf_counter()
    data = "synthetic"
    real = find_real_graph(args.shape) if args.scale == 1.0 else None
    if real:
        try:
            rp_d, col_d = load_real_graph(real, dev)
            n = rp_d.numel() - 1
            data = "real: " + real
        except Exception as exc:   # a file that is there but unreadable must not take the measurement down
            sys.stderr.write("bench: %s found but not loaded (%s); measuring the synthetic graph of the same shape\n" % (real, str(exc)[:200]))
            real = None
    if not real:
        rp_d, col_d = G.GENERATORS[args.graph](n, nnz_target, seed=args.seed, device=dev)
    torch.cuda.synchronize()
    gen_s = time.perf_counter() - t0
    E = col_d.numel()
    nw = (n + 15) // 16

    # ---- sparse-graph translation: host (as main_tcgnn.py:51 does) and device, timed once each
    rp_h, col_h = rp_d.cpu(), col_d.cpu()
    bp_h = torch.zeros(nw, dtype=torch.int32); e2c_h = torch.zeros(E, dtype=torch.int32); e2r_h = torch.zeros(E, dtype=torch.int32)
    e2c_h.fill_(1); e2r_h.fill_(1)  # touch the pages before timing
    devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(1); sys.stdout.flush(); os.dup2(devnull, 1)
    try:
        t0 = time.perf_counter()
        TCGNN.preprocess(col_h, rp_h, n, 16, 8, bp_h, e2c_h, e2r_h)
        host_sgt_ms = (time.perf_counter() - t0) * 1e3
        bp = torch.zeros(nw, dtype=torch.int32, device=dev); e2c = torch.zeros(E, dtype=torch.int32, device=dev); e2r = torch.zeros(E, dtype=torch.int32, device=dev)
        TCGNN.preprocess_gpu(col_d, rp_d, n, 16, 8, bp, e2c, e2r)  # warm (rocPRIM temp allocation)
        dev_sgt_runs = []
        for _ in range(3):   # (r06: the scratch comes from torch's caching allocator - tcgnn_preprocess_gpu_ws allocates nothing - so the MEDIAN of three is
            #  reported and all three are recorded; r05 took the best of three because one run in a dozen read 95 ms against 5: ~10 hipMalloc / hipFree per call)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            TCGNN.preprocess_gpu(col_d, rp_d, n, 16, 8, bp, e2c, e2r)
            torch.cuda.synchronize(); dev_sgt_runs.append((time.perf_counter() - t0) * 1e3)
        dev_sgt_ms = float(np.median(dev_sgt_runs))
    finally:
        sys.stdout.flush(); os.dup2(saved, 1); os.close(saved); os.close(devnull)
    sgt_equal = bool(torch.equal(bp.cpu(), bp_h) and torch.equal(e2c.cpu(), e2c_h) and torch.equal(e2r.cpu(), e2r_h))
    meta = (rp_d, col_d, bp, e2c, e2r)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    info = TCGNN.plan_info(*meta)   # first sight of the five arrays: packs the tile stream (+ the cell stream of dense graphs)
    torch.cuda.synchronize(); plan_ms = (time.perf_counter() - t0) * 1e3

    g = torch.Generator(device=dev).manual_seed(args.seed)
    X = torch.randn(n, D, device=dev, generator=g)
    step = lambda: TCGNN.forward(X, *meta)
    noop = lambda: None

    # ---- the timed K steps, kernel events recorded inside them
    TCGNN.kernel_timing(*meta, max_calls=args.steps + args.warmup + 256)   # event pairs around every launch from here on
    settle(step)
    for _ in range(args.warmup):
        step()
    elapsed = sync_time(step, args.steps, 0, noop)
    kernel_ms_all = TCGNN.kernel_timing(*meta)
    kernel_ms = kernel_ms_all[-args.steps:]                                  # the K timed steps
    TCGNN.kernel_timing(*meta, max_calls=0)
    k_mean = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
    k_all = float(np.mean(kernel_ms_all)) if kernel_ms_all else k_mean
    ms_per_step = elapsed * 1e3 / args.steps
    gteps = E / (elapsed / args.steps) / 1e9
    workload = "%s N=%d nnz=%d, SpMM D=%d (GCN aggregation, fwd = bwd)" % (args.shape + ("-shape synthetic graph" if data == "synthetic" else " (real graph)"), n, E, D)
    roof_b = spmm_bytes(n, E, D)
    kname = TCGNN.last_kernel(*meta)   # what the launcher actually ran for this plan and width (tcgnn_plan_last_kernel)
    out = {
        "metric": "SpMM/SDDMM GTEPS + GCN/AGNN ms/epoch, Reddit h=64, 1xMI355X",
        "value": round(gteps, 3), "unit": "GTEPS (SpMM, edges/s/1e9)", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        # the same rate over EVERY launch of the process (settle + warm-up + timed), from the per-launch kernel events plus the
        # measured staging / launch share of a step: what a cold-start average would report
        "value_all_launches": round(E / ((float(np.mean(kernel_ms_all)) + (ms_per_step - k_mean)) * 1e-3) / 1e9, 3) if kernel_ms_all else None,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16 x f16 -> f32 (MFMA), f32 I/O", "data": data,
        "config": {"workload": workload, "graph": GRAPH_NOTES.get(args.graph, args.graph) if data == "synthetic" else data, "tc_blocks_16x8": info["tc_blocks"],
                   "reddit_real_tc_blocks_16x8": 13566510, "wide_blocks_16x32": info["wide_blocks"],
                   "waves_per_window": info["waves_per_window"], "lds_column_ranges": info.get("lds_ranges", 0), "parallelism": "1 GPU"},
        "roofline": {"bound": "hbm", "kernel": kname,
                     # (`achieved` / `frac` over EVERY launch of the process - settle + warm-up + timed - the population a
                     #  profiler's per-kernel average covers; the K timed, post-settle launches alone are `frac_steady`)
                     "achieved": round(roof_b / (k_all * 1e-3) / 1e9, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": round(roof_b / (k_all * 1e-3) / HBM_PEAK, 5),
                     "frac_steady": round(roof_b / (k_mean * 1e-3) / HBM_PEAK, 5),
                     **profile_fields(kname, "%s_%s_d%d" % (args.shape.replace("ogbn-", ""), args.graph, D), 2.0 * E * D, k_mean),
                     "algorithmic_bytes": roof_b, "kernel_ms_mean": round(k_mean, 4), "kernel_ms_min": round(float(np.min(kernel_ms)), 4) if kernel_ms else None,
                     "kernel_launches_timed": len(kernel_ms),
                     # every launch of the process (settle + W + K): the population a profiler's per-kernel average covers
                     "kernel_ms_mean_all_launches": round(float(np.mean(kernel_ms_all)), 4) if kernel_ms_all else None,
                     "kernel_launches_all": len(kernel_ms_all)},
    }
    extra = {"graph_gen_s": round(gen_s, 2), "host_sgt_ms": round(host_sgt_ms, 1), "host_sgt_ns_per_edge": round(host_sgt_ms * 1e6 / E, 2),
             "device_sgt_ms": round(dev_sgt_ms, 1), "device_sgt_ms_runs": [round(x, 1) for x in dev_sgt_runs], "device_sgt_equals_host_sgt": sgt_equal, "plan_create_ms": round(plan_ms, 1), "plan_bytes": info["plan_bytes"],
             "staging_plus_launch_ms_per_step": round(ms_per_step - k_mean, 4)}

    def timed_leg(meta_, E_, fn, bytes_, reps=20):
        # (r05: every leg settles like the headline step - ~60 ms of untimed calls.  Three warm-up calls left the 10-20 timed ones inside
        #  the transient that follows every idle stretch, here the graph generation before a leg: the first 25 launches run ~12 % slow,
        #  the next 25 ~3 %, tools/scratch/series.py - the uniform graph read 0.555 ms as a leg and 0.479 in steady state)
        settle(fn, seconds=LEG_SETTLE_S)
        TCGNN.kernel_timing(*meta_, max_calls=reps)
        el = sync_time(fn, reps, 0, noop)
        km = TCGNN.kernel_timing(*meta_)
        TCGNN.kernel_timing(*meta_, max_calls=0)
        kmean = float(np.mean(km))
        return {"gteps": round(E_ / (el / reps) / 1e9, 3), "ms_per_call": round(el * 1e3 / reps, 4), "kernel_ms": round(kmean, 4),
                "hbm_frac": round(bytes_ / (kmean * 1e-3) / HBM_PEAK, 5), "kernel": TCGNN.last_kernel(*meta_)}

    def dataset_legs(shape, gen, d, ops, seed, reorder=False):
        """One graph of a named shape from one generator (uniform / rmat / sbm): the kernels named in `ops` at width d, each with
        its HBM fraction by algorithmic bytes, the PMC traffic / MFMA figures of a fresh profile, and the useful MFMA rate."""
        n_, nnz_, _, _ = G.SHAPES[shape]
        rp_, col_ = G.GENERATORS[gen](n_, nnz_, seed=seed, device=dev)
        reorder_ms = None
        if reorder:   # the relabelling a loader would apply once (tcgnn_harness --reorder): communities made contiguous
            torch.cuda.synchronize(); t0_ = time.perf_counter()
            rp_, col_ = G.permute_csr(rp_, col_, G.community_order(rp_, col_, seed=seed))
            torch.cuda.synchronize(); reorder_ms = (time.perf_counter() - t0_) * 1e3
        E_ = col_.numel()
        bp_ = torch.zeros((n_ + 15) // 16, dtype=torch.int32, device=dev); e2c_ = torch.zeros(E_, dtype=torch.int32, device=dev); e2r_ = torch.zeros(E_, dtype=torch.int32, device=dev)
        devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(1); sys.stdout.flush(); os.dup2(devnull, 1)
        try:
            TCGNN.preprocess_gpu(col_, rp_, n_, 16, 8, bp_, e2c_, e2r_)
        finally:
            sys.stdout.flush(); os.dup2(saved, 1); os.close(saved); os.close(devnull)
        m_ = (rp_, col_, bp_, e2c_, e2r_)
        info_ = TCGNN.plan_info(*m_)
        X_ = torch.randn(n_, d, device=dev, generator=g)
        wl = "%s_%s_d%d" % (shape.replace("ogbn-", ""), gen, d)
        if reorder:
            wl += "_reordered"
        row = {"dataset": "%s shape, %s generator%s" % (shape, gen, ", nodes relabelled by tcgnn_graph.community_order" if reorder else ""), "workload": wl, "N": n_, "nnz": int(E_), "D": d, "tc_blocks_16x8": info_["tc_blocks"],
               "max_degree": int((rp_[1:] - rp_[:-1]).max())}
        if reorder_ms is not None:
            row["reorder_ms"] = round(reorder_ms, 1)
        if "spmm" in ops:
            leg = timed_leg(m_, E_, lambda: TCGNN.forward(X_, *m_), spmm_bytes(n_, E_, d), reps=20)
            leg.update(profile_fields(leg["kernel"], wl, 2.0 * E_ * d, leg["kernel_ms"]))
            row["spmm"] = leg
        if "spmm_val" in ops:
            att_ = torch.randn(1, E_, device=dev, generator=g)
            leg = timed_leg(m_, E_, lambda: TCGNN.forward_AGNN(X_, rp_, col_, att_, bp_, e2c_, e2r_), spmm_bytes(n_, E_, d) + 4 * E_, reps=20)
            leg.update(profile_fields(leg["kernel"], wl, 2.0 * E_ * d, leg["kernel_ms"], val=True, bwd=True))
            row["spmm_val"] = leg
            del att_
        if "sddmm" in ops:
            leg = timed_leg(m_, E_, lambda: TCGNN.forward_ef(X_, *m_), sddmm_bytes(n_, E_, d), reps=20)
            leg.update(profile_fields(leg["kernel"], wl, 2.0 * E_ * d, leg["kernel_ms"]))
            row["sddmm"] = leg
        if "agnn" in ops:
            w_ = torch.tensor([0.9], device=dev)
            Xs_ = X_ / d ** 0.5
            _, ef_, efm_ = TCGNN.agnn_fused_forward(Xs_, rp_, col_, w_, bp_, e2c_, e2r_)
            pb = sddmm_bytes(n_, E_, d) + 4 * n_ * d
            leg = timed_leg(m_, E_, lambda: TCGNN.agnn_fused_forward(Xs_, rp_, col_, w_, bp_, e2c_, e2r_), pb, reps=20)
            leg.update(profile_fields(leg["kernel"], wl, 4.0 * E_ * d, leg["kernel_ms"], bwd=False))
            row["agnn_fused_fwd"] = leg
            leg = timed_leg(m_, E_, lambda: TCGNN.agnn_fused_backward(Xs_, rp_, col_, w_, ef_, efm_, bp_, e2c_, e2r_), pb, reps=20)
            leg.update(profile_fields(leg["kernel"], wl, 4.0 * E_ * d, leg["kernel_ms"], bwd=True))
            row["agnn_fused_bwd"] = leg
            del ef_, efm_, Xs_
        if "agnn_epoch" in ops or "gcn_epoch" in ops:
            _, _, in_dim_, classes_ = G.SHAPES[shape]
            feats_ = torch.randn(n_, in_dim_, device=dev, generator=g); labels_ = torch.ones(n_, dtype=torch.long, device=dev)
            for model_ in ("gcn", "agnn"):
                if model_ + "_epoch" in ops:
                    r_ = H.time_training(model_, m_, feats_, labels_, in_dim_, d, classes_, 2, max(3, args.epochs // 2), seed=args.seed, warmup=EPOCH_WARMUP)
                    row[model_ + "_ms_per_epoch"] = round(r_["train_ms"], 3)
            del feats_, labels_
        del X_, rp_, col_, bp_, e2c_, e2r_, m_
        TCGNN.clear_plan_cache()
        return row

    if not args.no_extra:
        kernel_leg = lambda fn, bytes_, reps=20: timed_leg(meta, E, fn, bytes_, reps)
        att = torch.randn(1, E, device=dev, generator=g)
        extra["sddmm_d%d" % D] = kernel_leg(lambda: TCGNN.forward_ef(X, *meta), sddmm_bytes(n, E, D))
        extra["spmm_agnn_d%d" % D] = kernel_leg(lambda: TCGNN.forward_AGNN(X, rp_d, col_d, att, bp, e2c, e2r), spmm_bytes(n, E, D) + 4 * E)
        # the two products of an AGNN layer in one pass (what TCGNNFunction_AGNN calls): pair time vs the two legs above
        wdev = torch.tensor([0.9], device=dev)
        _, ef_s, efm_s = TCGNN.agnn_fused_forward(X, rp_d, col_d, wdev, bp, e2c, e2r)
        pair_bytes = sddmm_bytes(n, E, D) + 4 * n * D
        extra["agnn_fused_fwd_d%d" % D] = kernel_leg(lambda: TCGNN.agnn_fused_forward(X, rp_d, col_d, wdev, bp, e2c, e2r), pair_bytes)
        extra["agnn_fused_bwd_d%d" % D] = kernel_leg(lambda: TCGNN.agnn_fused_backward(X, rp_d, col_d, wdev, ef_s, efm_s, bp, e2c, e2r), pair_bytes)
        del ef_s, efm_s
        for d2 in (16, 128):
            X2 = torch.randn(n, d2, device=dev, generator=g)
            extra["spmm_d%d" % d2] = kernel_leg(lambda: TCGNN.forward(X2, *meta), spmm_bytes(n, E, d2), reps=20)
            extra["sddmm_d%d" % d2] = kernel_leg(lambda: TCGNN.forward_ef(X2, *meta), sddmm_bytes(n, E, d2), reps=20)
            del X2
        del att
        # ---- the same SpMM on a degree-skewed graph of the same size (real Reddit: max degree 21 657 at a mean of 492;
        #      skew 0.6 gives a ~24 k hub): heavy windows must not stall the persistent walks
        try:
            rp_s, col_s = G.synthetic_csr(n, nnz_target, seed=args.seed + 1, device=dev, skew=0.6)
            Es = col_s.numel()
            bp_s = torch.zeros(nw, dtype=torch.int32, device=dev); e2c_s = torch.zeros(Es, dtype=torch.int32, device=dev); e2r_s = torch.zeros(Es, dtype=torch.int32, device=dev)
            devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(1); sys.stdout.flush(); os.dup2(devnull, 1)
            try:
                TCGNN.preprocess_gpu(col_s, rp_s, n, 16, 8, bp_s, e2c_s, e2r_s)
            finally:
                sys.stdout.flush(); os.dup2(saved, 1); os.close(saved); os.close(devnull)
            meta_s = (rp_s, col_s, bp_s, e2c_s, e2r_s)
            settle(lambda: TCGNN.forward(X, *meta_s), seconds=LEG_SETTLE_S)
            TCGNN.kernel_timing(*meta_s, max_calls=10)
            el = sync_time(lambda: TCGNN.forward(X, *meta_s), 10, 0, noop)
            km = TCGNN.kernel_timing(*meta_s); TCGNN.kernel_timing(*meta_s, max_calls=0)
            extra["spmm_d%d_skewed_graph" % D] = {"gteps": round(Es / (el / 10) / 1e9, 3), "kernel_ms": round(float(np.mean(km)), 4),
                                                  "max_degree": int((rp_s[1:] - rp_s[:-1]).max()), "nnz": int(Es)}
            del rp_s, col_s, bp_s, e2c_s, e2r_s, meta_s
        except Exception as exc:   # the extra leg must never take the headline down
            extra["spmm_d%d_skewed_graph" % D] = {"error": str(exc)[:200]}
        # ---- end-to-end epochs (main_tcgnn.py:146-181): 2 layers, hidden = D, EPOCH_WARMUP dry epochs (steady state, as the kernel legs)
        feats = torch.randn(n, in_dim, device=dev, generator=g)
        labels = torch.ones(n, dtype=torch.long, device=dev)
        for model in ("gcn", "agnn"):
            r = H.time_training(model, meta, feats, labels, in_dim, D, classes, 2, args.epochs, seed=args.seed, warmup=EPOCH_WARMUP)
            extra["%s_ms_per_epoch" % model] = round(r["train_ms"], 3)
            extra["%s_final_loss_finite" % model] = bool(np.isfinite(r["final_loss"]))
            try:   # the same epoch captured once in a HIP graph and replayed (tcgnn_harness --hip_graph)
                rg = H.time_training(model, meta, feats, labels, in_dim, D, classes, 2, args.epochs, seed=args.seed, warmup=3, hip_graph=True)
                extra["%s_ms_per_epoch_hip_graph" % model] = round(rg["train_ms"], 3)
            except Exception as exc:
                extra["%s_ms_per_epoch_hip_graph" % model] = "failed: %s" % str(exc)[:120]
        # ---- the north star's grid, "hidden=16/64/128": the same two epochs at the other two widths (1_bench_gcn.py:6,33-38 and
        #      1_bench_agnn.py:29-41 sweep --hidden the same way)
        for h in (16, 128):
            if h == D:
                continue
            for model in ("gcn", "agnn"):
                try:
                    r = H.time_training(model, meta, feats, labels, in_dim, h, classes, 2, max(3, args.epochs // 2), seed=args.seed, warmup=EPOCH_WARMUP)
                    extra["%s_h%d_ms_per_epoch" % (model, h)] = round(r["train_ms"], 3)
                except Exception as exc:
                    extra["%s_h%d_ms_per_epoch" % (model, h)] = "failed: %s" % str(exc)[:120]
        del feats

    # ---- per-dataset list (north star: "MFMA utilisation and HBM GB/s ... on each dataset"): the headline graph, the same shape
    #      from the R-MAT and community generators (condensing and cache behaviour depend on locality), and
    #      BASELINE.json configs[3], ogbn-products AGNN hidden = 128
    datasets = [{"dataset": "%s shape, %s generator (headline)" % (args.shape, args.graph), "workload": "%s_%s_d%d" % (args.shape.replace("ogbn-", ""), args.graph, D),
                 "N": n, "nnz": int(E), "D": D, "tc_blocks_16x8": info["tc_blocks"],
                 "spmm": {"kernel": kname, "kernel_ms": round(k_mean, 4), "gteps": round(gteps, 3), "hbm_frac": out["roofline"]["frac"],
                          **{k: out["roofline"][k] for k in ("traffic", "mfma_busy", "mfma_useful_frac", "mfma_useful_tflops", "mfma_peak_frac")}}}]
    if not args.no_extra:
        # (r06, VERDICT r05 item 7: the headline graph's row carries every operator with its PMC figures, not the SpMM alone - the legs
        #  were timed above, the counters come from the same profile file)
        wl0 = datasets[0]["workload"]
        for key, src, val_, bwd_, fl in (("sddmm", "sddmm_d%d" % D, False, None, 2.0), ("spmm_val", "spmm_agnn_d%d" % D, True, True, 2.0),
                                         ("agnn_fused_fwd", "agnn_fused_fwd_d%d" % D, False, False, 4.0), ("agnn_fused_bwd", "agnn_fused_bwd_d%d" % D, False, True, 4.0)):
            leg = extra.get(src)
            if isinstance(leg, dict) and leg.get("kernel"):
                datasets[0][key] = dict(leg, **profile_fields(leg["kernel"], wl0, fl * E * D, leg["kernel_ms"], val=val_, bwd=bwd_))
        datasets[0]["gcn_ms_per_epoch"] = extra.get("gcn_ms_per_epoch"); datasets[0]["agnn_ms_per_epoch"] = extra.get("agnn_ms_per_epoch")
    if not args.no_extra and args.scale == 1.0:
        TCGNN.clear_plan_cache()
        every = ("spmm", "spmm_val", "sddmm", "agnn")
        # (the three generators SURVEY.md 8d names, plus the community graph calibrated to real Reddit's TC-block count -
        #  tcgnn_graph.SBM_REDDIT_P_IN; the hub / shuffled / relabelled variants of r02 are behind --all-generators)
        more = ((args.shape, "sbm_hubs", D, every), (args.shape, "sbm_shuffled", D, every), (args.shape, "sbm_shuffled+reorder", D, every + ("gcn_epoch", "agnn_epoch"))) if args.all_generators else ()
        # (the headline graph is args.graph - measured above; the uniform graph r01-r04 reported takes its place in this list)
        second = "uniform" if args.graph != "uniform" else "sbm_reddit"
        for shape, gen, d, ops in ((args.shape, second, D, every + ("gcn_epoch", "agnn_epoch")), (args.shape, "sbm", D, every + ("gcn_epoch", "agnn_epoch")),
                                   (args.shape, "rmat", D, every + ("gcn_epoch", "agnn_epoch")), *more,
                                   ("ogbn-products", "uniform", 128, every + ("gcn_epoch", "agnn_epoch")),
                                   ("ogbn-products", "sbm", 128, every + ("gcn_epoch", "agnn_epoch")),   # (r06: config 4's model end to end on the graph with communities)
                                   ("ogbn-products", "rmat", 128, every)):
            try:
                datasets.append(dataset_legs(shape, gen.split("+")[0], d, ops, args.seed, reorder=gen.endswith("+reorder")))
            except Exception as exc:   # an extra dataset must never take the headline down
                datasets.append({"dataset": "%s shape, %s generator" % (shape, gen), "error": str(exc)[:300]})
            torch.cuda.empty_cache()
    out["datasets"] = datasets
    if not args.no_extra and args.scale == 1.0:
        # ---- row (g): the reference's published RTX 3090 tables, same harness flow, same-size graphs
        shapes = artifact_shapes(args.seed)
        extra["artifact_shapes"] = shapes
        extra["artifact_shapes_note"] = ("reference: /root/reference/logs/profile.csv:2-15 (single SpMM kernel, D=16, 200 rounds) and logs/RTX3090_GCN.csv:2-15 "
                                         "(2-layer GCN hidden 16), both RTX 3090; here: same-size seeded uniform graphs through tcgnn_harness (the artifact .npz files are not on the box)")
        good = [r for r in shapes if "error" not in r]
        if good:
            extra["artifact_shapes_beating_rtx3090"] = {"spmm_d16": sum(r["spmm_speedup_vs_rtx3090"] > 1 for r in good),
                                                        "gcn_h16_epoch": sum(r["gcn_speedup_vs_rtx3090"] > 1 for r in good), "of": len(shapes)}
        cs = [r for r in good if r["shape"] == "citeseer"]
        if cs:   # BASELINE.json configs[1]: "Citeseer GCN hidden=16 TC-SpMM single-kernel" - the one published number for a named config
            # (vs_baseline stays null: BASELINE.md publishes no GTEPS and nothing at Reddit size.  The one published number for a named
            #  config is the reference's single-kernel time on citeseer, 0.040 ms on an RTX 3090, logs/profile.csv:2)
            extra["citeseer_spmm_d16_speedup_vs_rtx3090"] = cs[0]["spmm_speedup_vs_rtx3090"]
    if not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(rp_h.numpy(), col_h.numpy(), n, E, D, args.seed)
    out["extra"] = extra
    return out


def cpu_baseline(rp, col, n, E, D, seed):
    """Y = A X on the host cores, bounded to roughly 10-30 s of CPU work: the row-parallel CSR gather-add of oracle/cpu_baseline.c
    built on this machine with -march=native (equal-nnz static row blocks, parallel first touch, prefetch) - the number used -
    and torch's own sparse-CSR product next to it.  Rank 0, N = 1 only; a reported baseline, never the target."""
    from oracle import cpu_baseline as CB
    threads = os.cpu_count() or 1
    X = np.random.default_rng(seed).standard_normal((n, D)).astype(np.float32)
    _, probe = CB.csr_spmm(X, rp, col, threads=threads, reps=1)                       # build + place pages + cost probe
    reps = int(max(2, min(20, 12.0 / max(probe[0], 1e-3))))
    _, times = CB.csr_spmm(X, rp, col, threads=threads, reps=reps)
    best, mean = min(times), float(np.mean(times))
    cpu = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu = [l.split(":", 1)[1].strip() for l in f if l.startswith("model name")][0]
    except (OSError, IndexError):
        pass
    out = {"value": round(E / mean / 1e9, 4), "unit": "GTEPS (SpMM, edges/s/1e9)", "cores": threads, "kind": "port",
           "sample": "full %d-edge graph, D=%d, %d timed passes after a first-touch pass (mean %.3f s, min %.3f s)" % (E, D, reps, mean, best),
           "what": "oracle/cpu_baseline.c: row-parallel CSR gather-add (the DGL-CPU-style aggregation), gcc %s, equal-nnz static row blocks, "
                   "parallel first touch of X and Y, software prefetch; fp32" % CB.build_flags(),
           "value_best_pass": round(E / best / 1e9, 4), "cpu_model": cpu, "used": "oracle/cpu_baseline.c (the faster of the two CPU paths timed here is reported as value)"}
    try:   # torch's sparse CSR @ dense on the same cores, for reference
        torch.set_num_threads(threads)
        A = torch.sparse_csr_tensor(torch.from_numpy(rp.astype(np.int64)), torch.from_numpy(col.astype(np.int64)), torch.ones(E), size=(n, n))
        Xt = torch.from_numpy(X)
        t0 = time.perf_counter(); torch.sparse.mm(A, Xt); first = time.perf_counter() - t0
        k = int(max(1, min(5, 6.0 / max(first, 1e-3))))
        ts = []
        for _ in range(k):
            t0 = time.perf_counter(); torch.sparse.mm(A, Xt); ts.append(time.perf_counter() - t0)
        out["torch_sparse_csr_mm_gteps"] = round(E / float(np.mean(ts)) / 1e9, 4)
        out["torch_sparse_csr_mm_note"] = "torch.sparse.mm(CSR fp32, dense) with %d threads, %d passes after 1 warm-up" % (threads, k)
        if out["torch_sparse_csr_mm_gteps"] > out["value"]:
            out["value"], out["used"] = out["torch_sparse_csr_mm_gteps"], "torch.sparse.mm (faster than oracle/cpu_baseline.c here)"
        del A, Xt
    except Exception as exc:
        out["torch_sparse_csr_mm_note"] = "failed: %s" % str(exc)[:160]
    # ---- the GCN epoch of the reference's DGL baseline (dgl_baseline/gcn.py + train.py) restated on the host cores
    #      (oracle/dgl_gcn_cpu.py; DGL itself is absent and unpinned): BASELINE.json configs[0] (Cora shape, hidden 16) and the
    #      headline graph at hidden D.  Bounded: a few epochs each.
    try:
        from oracle import dgl_gcn_cpu as B
        import tcgnn_graph as G
        cn, cnnz, cdim, ccls = G.SHAPES["cora"]
        crp, ccol = G.synthetic_csr(cn, cnnz, seed=seed)
        cx = np.random.default_rng(seed).standard_normal((cn, cdim)).astype(np.float32)
        r0 = B.time_training(crp.numpy(), ccol.numpy(), cx, np.ones(cn, np.int64), 16, ccls, epochs=50, threads=min(threads, 16), symmetric=True)
        out["gcn_epoch_ms_cora_shape_h16"] = round(r0["train_ms"], 3)
        _, _, in_dim, classes = G.SHAPES["reddit"] if n > 100000 else (0, 0, 64, 8)
        fx = np.random.default_rng(seed + 1).standard_normal((n, in_dim)).astype(np.float32)
        r1 = B.time_training(rp, col, fx, np.ones(n, np.int64), D, classes, epochs=2, dry_runs=1, threads=threads, symmetric=True)
        out["gcn_epoch_ms_same_graph_h%d" % D] = round(r1["train_ms"], 1)
        out["gcn_epoch_note"] = "GraphConv(norm=both)+bias stack, CrossEntropy, Adam(1e-2, wd 5e-4): %d-thread Cora shape over 50 epochs; headline graph over 2 epochs after 1 dry run" % min(threads, 16)
    except Exception as exc:   # the epoch leg is an extra: the SpMM baseline above stands on its own
        out["gcn_epoch_note"] = "failed: %s" % str(exc)[:200]
    return out


def multi_gpu(args):
    import tcgnn_graph as G
    import tcgnn_harness as H
    import tcgnn_shard as S
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist.init_process_group("nccl", device_id=dev)
    n0, nnz0, _, _ = G.SHAPES[args.shape]
    if args.scale != 1.0:
        n0, nnz0 = int(n0 * args.scale), int(nnz0 * args.scale * args.scale)
    D = args.dim
    n_global = n0 * world
    # this rank's rows only: n0 rows, ~nnz0 edges, columns uniform over all n_global nodes
    g = torch.Generator(device=dev).manual_seed(args.seed * 1000 + rank)
    rows = torch.randint(0, n0, (nnz0,), device=dev, generator=g)
    cols = torch.randint(0, n_global, (nnz0,), device=dev, generator=g)
    keys = torch.unique(rows.long() * n_global + cols.long())
    rows, cols = keys // n_global, keys % n_global
    counts = torch.bincount(rows, minlength=n0)
    lrp = torch.zeros(n0 + 1, dtype=torch.int64, device=dev); lrp[1:] = torch.cumsum(counts, 0)
    E_local = int(keys.numel())
    bounds = [p * n0 for p in range(world + 1)]
    shard = S.RowShard(rank=rank, world_size=world, device=dev, bounds=bounds,
                       local=(lrp.cpu().numpy().astype(np.int32), cols.cpu().numpy()),
                       always_collective=True)   # (a forced world of one still issues every RCCL call of the N-rank step)
    del rows, cols, keys
    x_local = torch.randn(n0, D, device=dev, generator=g)
    import tcgnn_capi as _c
    _info = _c.PlanInfo()
    _c.check(_c.lib.tcgnn_plan_get_info(shard.ops.plan, _c.ctypes.byref(_info)), "tcgnn_plan_get_info")
    shard_kernel = "spmm_lds_kernel" if _info.lds_ranges > 0 else ("spmm_blocked_kernel / spmm_kernel" if _info.column_buckets > 0 else "spmm_kernel")
    # North star: "an RCCL all-reduce of the dense feature-update only where the graph is too large for one
    # 288 GB HBM".  When the global X fits one GPU it is replicated (gathered once, outside the timed region)
    # and a step is the local SpMM on this rank's row windows; otherwise every step all-gathers X.
    global_x_bytes = n_global * D * 4
    exchange = args.exchange == "always" or (args.exchange == "auto" and global_x_bytes > 64 * (1 << 30))
    xg = shard.gather(x_local).clone()       # one collective at set-up; the replicated matrix in gathered numbering
    step = (lambda: shard.spmm(x_local)) if exchange else (lambda: shard.ops.spmm(xg))
    barrier = lambda: dist.barrier()
    n_settle = 16   # untimed steps before the warm-up (a FIXED count: with a collective inside, every rank must make the same calls)
    for _ in range(n_settle):
        step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    shard.ops.set_timing(args.steps)
    elapsed = sync_time(step, args.steps, 0, barrier)
    kernel_ms = shard.ops.read_timing()
    # the other variant, for the exchange fraction
    other = (lambda: shard.ops.spmm(xg)) if exchange else (lambda: shard.spmm(x_local))
    t_other = sync_time(other, max(3, args.steps // 5), 2, barrier) * args.steps / max(3, args.steps // 5)
    t_nox = t_other if exchange else elapsed
    t_withx = elapsed if exchange else t_other
    # ---- the exchange with fp16 on the wire (each rank converts its own rows; the kernels' image, not fp32 X, is gathered)
    t_wire16 = float("nan")
    try:
        w16 = lambda: shard.spmm(x_local, wire="fp16")
        t_wire16 = sync_time(w16, max(3, args.steps // 5), 2, barrier) * args.steps / max(3, args.steps // 5)
    except Exception as exc:   # an extra: must never take the headline down
        if rank == 0:
            print("fp16-exchange leg failed: %s" % str(exc)[:300], file=sys.stderr)
    # ---- the exchange in 32-column chunks through a ring of two buffers (RowShard.spmm_chunked: what config 5 needs to fit)
    t_chunk = float("nan")
    try:
        wc = lambda: shard.spmm_chunked(x_local, chunk=32, wire="fp32")
        t_chunk = sync_time(wc, max(3, args.steps // 5), 2, barrier) * args.steps / max(3, args.steps // 5)
    except Exception as exc:   # an extra: must never take the headline down
        if rank == 0:
            print("chunked-exchange leg failed: %s" % str(exc)[:300], file=sys.stderr)
    # ---- the exchange overlapped with the own-block product (RowShard.spmm_overlapped: the gather on a side stream)
    t_overlap = float("nan")
    try:
        wo = lambda: shard.spmm_overlapped(x_local)
        t_overlap = sync_time(wo, max(3, args.steps // 5), 2, barrier) * args.steps / max(3, args.steps // 5)
    except Exception as exc:   # an extra: must never take the headline down
        if rank == 0:
            print("overlapped-exchange leg failed: %s" % str(exc)[:300], file=sys.stderr)
    # ---- one sharded GCN training epoch (2 layers, hidden D, main_tcgnn.py:146-181 on the shard): X W locally, all-gather +
    #      local SpMM forward and backward in both layers, one all-reduce of the weight gradients
    gcn_ms = float("nan")
    try:
        _, _, in_dim, classes = G.SHAPES[args.shape]
        feats = torch.randn(n0, in_dim, device=dev, generator=g)
        labels = torch.ones(n0, dtype=torch.long, device=dev)
        model = S.ShardedGCN(in_dim, D, classes, num_layers=2, seed=args.seed).to(dev)
        opt = H.make_adam(model.parameters())
        ep = lambda: S.sharded_train_step(model, shard, feats, labels, opt, n_global)
        gcn_ms = sync_time(ep, max(2, args.epochs // 2), 3, barrier) * 1e3 / max(2, args.epochs // 2)
        del feats, model, opt
    except Exception as exc:   # the extra leg must never take the headline down
        if rank == 0:
            print("sharded GCN leg failed: %s" % str(exc)[:300], file=sys.stderr)
    # ---- STRONG scaling next to the weak line above (r06): the single-GPU headline graph itself - args.graph at the
    #      shape's own size, every rank builds the same seeded CSR - cut into N row blocks balanced by nnz (tcgnn_shard.partition_rows); X is
    #      replicated (it fits one GPU: no collective in the step), a step is every rank's SpMM over its block, the time is
    #      the slowest rank's.  value_strong = the WHOLE graph's edges over that time: N = 1 would read the single-GPU headline.
    t_strong, e_strong, strong_kernel = float("nan"), 0.0, ""
    try:
        rp_g, col_g = G.GENERATORS[args.graph](n0, nnz0, seed=args.seed, device=dev)
        shard_s = S.RowShard(rp_g.cpu().numpy(), col_g.cpu().numpy(), rank=rank, world_size=world, device=dev, always_collective=False)
        del rp_g, col_g
        xs = torch.randn(n0, D, device=dev, generator=torch.Generator(device=dev).manual_seed(args.seed))   # (the same matrix on every rank)
        xg_s = shard_s.place_replicated(xs)
        step_s = lambda: shard_s.ops.spmm(xg_s)
        for _ in range(n_settle):
            step_s()
        torch.cuda.synchronize()
        t_strong = sync_time(step_s, args.steps, args.warmup, barrier)
        e_strong = float(shard_s.ops.nnz)
        strong_kernel = str(_c.lib.tcgnn_plan_last_kernel(shard_s.ops.plan).decode()) if hasattr(_c.lib, "tcgnn_plan_last_kernel") else ""
        del shard_s, xs, xg_s
    except Exception as exc:   # an extra: must never take the headline down
        if rank == 0:
            print("strong-scaling leg failed: %s" % str(exc)[:300], file=sys.stderr)
    stats = torch.tensor([elapsed, t_nox, float(E_local), float(np.mean(kernel_ms)), t_withx, gcn_ms, t_wire16, t_overlap, t_chunk, t_strong, e_strong], dtype=torch.float64, device=dev)
    mx = stats.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    sm = stats.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    out = None
    if rank == 0:
        t, t_local, t_x = float(mx[0]), float(mx[1]), float(mx[4])
        E_total = float(sm[2])
        k_mean = float(mx[3])
        roof_b = spmm_bytes(n0, E_local, D) + 4 * (n_global - n0) * D  # + the remote X rows it must read once
        out = {
            "metric": "SpMM/SDDMM GTEPS + GCN/AGNN ms/epoch, Reddit h=64, 1xMI355X",
            "value": round(E_total * args.steps / t / 1e9, 3), "unit": "GTEPS (SpMM, edges/s/1e9)", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(t * 1e3 / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16 x f16 -> f32 (MFMA), f32 I/O", "data": "synthetic",
            "config": {"workload": "row-sharded %s-shape graph: %d nodes total, %d rows and ~%d nnz per GPU, SpMM D=%d, %s"
                                   % (args.shape, n_global, n0, E_local, D, "X all-gathered every step" if exchange else "X replicated (fits one GPU), no collective in the step"),
                       "parallelism": "row-window sharding x%d%s" % (world, ", RCCL all_gather_into_tensor of X" if exchange else "")},
            "roofline": {"bound": "hbm", "kernel": "%s (slowest rank)" % shard_kernel, "achieved": round(roof_b / (k_mean * 1e-3) / 1e9, 2), "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": round(roof_b / (k_mean * 1e-3) / HBM_PEAK, 5), "traffic": None, "algorithmic_bytes": roof_b,
                         "kernel_ms_mean": round(k_mean, 4)},
            "extra": {"exchange_in_timed_step": exchange, "ms_per_step_without_exchange": round(t_local * 1e3 / args.steps, 4),
                      "ms_per_step_with_exchange": round(t_x * 1e3 / args.steps, 4),
                      "exchange_fraction_if_exchanged": round(max(0.0, 1.0 - t_local / t_x), 4),
                      "gathered_X_bytes": int(shard.layout.num_cols) * D * 4,
                      "gcn_ms_per_epoch_sharded": None if np.isnan(float(mx[5])) else round(float(mx[5]), 3),
                      "ms_per_step_with_fp16_exchange": None if np.isnan(float(mx[6])) else round(float(mx[6]) * 1e3 / args.steps, 4),
                      # the gather on a side stream under the product over the rank's own column block (RowShard.spmm_overlapped)
                      "ms_per_step_with_overlapped_exchange": None if np.isnan(float(mx[7])) else round(float(mx[7]) * 1e3 / args.steps, 4),
                      "exchange_fraction_if_overlapped": None if np.isnan(float(mx[7])) else round(max(0.0, 1.0 - t_local / float(mx[7])), 4),
                      "ms_per_step_with_chunked_exchange_32col_fp32": None if np.isnan(float(mx[8])) else round(float(mx[8]) * 1e3 / args.steps, 4),
                      "own_block_edge_fraction": round(float(getattr(shard, "_own_frac", float("nan"))), 4),
                      # strong scaling: the single-GPU headline graph split over the N ranks, X replicated (see above)
                      "strong_scaling": None if np.isnan(float(mx[9])) else {
                          "workload": "%s-shape graph from the %s generator, %d nodes, %d nnz in all, rows split over %d GPUs by nnz, X replicated" % (args.shape, args.graph, n0, int(sm[10]), world),
                          "value_gteps": round(float(sm[10]) * args.steps / float(mx[9]) / 1e9, 3), "ms_per_step": round(float(mx[9]) * 1e3 / args.steps, 4),
                          "kernel_rank0": strong_kernel, "scaling": "strong"},
                      "value_strong_gteps": None if np.isnan(float(mx[9])) else round(float(sm[10]) * args.steps / float(mx[9]) / 1e9, 3)},
        }
    dist.barrier()
    dist.destroy_process_group()
    return out


HBM_BYTES = 288 * 10 ** 9   # per MI355X (MI355X_MICROARCH.md: 288 GB HBM3E)
XGMI_LINK_BPS = 153e9        # one xGMI link, one direction; 7 links per GPU, point-to-point


def x16_pitch_halves(dpad):
    """csrc x16_pitch: row pitch of the gather walks' fp16 image (power of two up to one 128-byte line, whole lines beyond)."""
    b = dpad * 2
    if b > 128:
        return (b + 127) // 128 * 128 // 2
    p = 32
    while p < b:
        p <<= 1
    return p // 2


def plan_only(args):
    """BASELINE.json configs[4] (ogbn-papers100M GCN hidden 64 over 8 GPUs) has never met hardware here and its graph is not on
    any box; what CAN be checked without either is that every rank's working set fits its 288 GB and what one exchange moves.
    Per rank (tcgnn_shard.RowShard / HipShardOps, include/tcgnn.h): the local int32 CSR, the SGT metadata (blockPartition,
    edgeToColumn, edgeToRow), the packed tile stream of the plan (per 16x32 wide block: 32 ids + 16 mask words + 16 edge offsets =
    256 B; wide blocks from the expected number of distinct columns of a 16-row window), the fp16 image of the GATHERED matrix (the
    kernels' workspace), the fp32 gather buffers of the exchange, and the layer tensors of the 2-layer GCN (features, hidden
    activations, their gradients).  Rows / edges per rank come from shard files when given, else an even split."""
    import math
    import tcgnn_graph as G
    shape = args.shape if args.shape != "reddit" else "ogbn-papers100M"
    n, nnz, in_dim, classes = G.SHAPES[shape]
    world, D = max(1, args.gpus), args.dim
    per = []
    if args.shard_files:
        for r in range(world):
            obj = np.load(args.shard_files.format(rank=r, world=world))
            rp = obj["row_pointers"]
            per.append((len(rp) - 1, int(rp[-1]), int(obj["H"])))
            n = int(obj["num_nodes"])
        nnz = sum(x[1] for x in per)
    else:
        rows = [((n + 15) // 16 * (r + 1) // world - (n + 15) // 16 * r // world) * 16 for r in range(world)]
        rows[-1] -= sum(rows) - n
        H = max(16, (max(rows) + 15) // 16 * 16)
        per = [(rows[r], nnz // world + (1 if r < nnz % world else 0), H) for r in range(world)]
    out_rows = []
    for r, (rows_r, nnz_r, H) in enumerate(per):
        ncols = H * world
        nw = (rows_r + 15) // 16
        k = nnz_r / max(nw, 1)                                            # edges of a window
        distinct = ncols * (1.0 - math.exp(-k / ncols)) if ncols else 0   # expected distinct columns among them (uniform bound: real graphs condense better)
        wide = nw * math.ceil(distinct / 32.0)
        dpad = (D + 15) // 16 * 16
        csr = 4 * (rows_r + 1) + 4 * nnz_r
        sgt = 4 * nw + 8 * nnz_r
        plan = int(wide * 256 + 8 * (nw + 1) + 4 * nw)
        widest = max(D, classes)
        feats = 4 * rows_r * in_dim
        acts = 4 * rows_r * (2 * D + 2 * classes) * 2                     # X W, A(X W) per layer, and their gradients
        # ---- the exchange.  What RowShard.exchange_chunk_for picks at this size (r06: a whole-matrix receive buffer beyond 8 GiB - here
        #      world * H * D * 4 bytes - takes tcgnn_shard.RowShard.spmm_chunked with 64 columns; smaller graphs keep the one gather,
        #      whose total is reported next to it): the matrix crosses the fabric 64
        #      columns at a time as the kernels' fp16 image - every rank converts only its rows - into a ring of TWO image buffers
        #      that tcgnn_spmm_staged multiplies from; no fp32 copy of the gathered matrix exists.  Whole-matrix fp32 gather (r01-r04):
        #      one receive + send buffer pair per width in use and the image of the widest layer - reported next to it.
        chunk = min(64, widest)
        cpitch = x16_pitch_halves((chunk + 15) // 16 * 16)
        image_ring = 2 * (256 + (ncols + 1) * cpitch * 2)
        send_ring = 2 * (H + 1) * cpitch * 2
        chunk_temps = 4 * rows_r * chunk + 2 * 4 * rows_r * widest          # the chunk's columns of X made contiguous; the chunks of Y and their concatenation
        parts = {"csr_bytes": csr, "sgt_metadata_bytes": sgt, "plan_bytes_est": plan, "exchange_image_ring_bytes": image_ring,
                 "exchange_send_ring_bytes": send_ring, "exchange_chunk_temporaries_bytes": chunk_temps, "features_bytes": feats, "layer_tensors_bytes": acts}
        chunked_default = world * H * D * 4 > (8 << 30)
        total = sum(parts.values())
        image_w = 256 + (ncols + 1) * x16_pitch_halves((widest + 15) // 16 * 16) * 2
        gather_all = 4 * (ncols + H) * (D + classes)                      # one fp32 buffer pair per width used (D and the class layer)
        total_whole = csr + sgt + plan + image_w + gather_all + feats + acts
        if not chunked_default:   # the default at this size is the one gather: that is the total that has to fit
            total = total_whole = csr + sgt + plan + image_w + 4 * (ncols + H) * (D + classes) + feats + acts
        out_rows.append({"rank": r, "rows": rows_r, "edges": nnz_r, "gathered_rows": ncols, **parts,
                         "total_bytes": total, "parts_summed": sorted(parts), "frac_of_hbm": round(total / HBM_BYTES, 4), "fits": total < HBM_BYTES,
                         "exchange": ("64-column chunks, fp16 image on the wire, ring of two buffers (RowShard.spmm_chunked: the default at this size, RowShard.exchange_chunk_for)"
                                      if chunked_default else "one fp32 gather of the whole matrix (the default at this size, RowShard.exchange_chunk_for); the parts listed are the chunked exchange's"),
                         "whole_matrix_fp32_exchange": {"image_fp16_widest_layer_bytes": image_w, "gather_buffers_all_widths_bytes": gather_all,
                                                        "total_bytes": total_whole, "frac_of_hbm": round(total_whole / HBM_BYTES, 4)}})
    H = per[0][2]
    blk32, blk16 = 4 * H * D, 2 * H * x16_pitch_halves((D + 15) // 16 * 16)
    doc = {"plan_only": True, "workload": "%s GCN hidden=%d, rows sharded over %d GPUs (BASELINE.json configs[4])" % (shape, D, world),
           "nodes": n, "edges": nnz, "world": world, "D": D, "hbm_bytes_per_gpu": HBM_BYTES, "all_fit": all(r["fits"] for r in out_rows),
           "max_frac_of_hbm": max(r["frac_of_hbm"] for r in out_rows),
           "int32_csr_possible_unsharded": nnz < 2 ** 31,
           "exchange_per_spmm": {"fp32_block_bytes": blk32, "fp16_block_bytes": blk16,
                                 # every GPU pushes its block to its (world - 1) peers at once, one xGMI link each
                                 "fp32_ms_link_bound": round(blk32 / XGMI_LINK_BPS * 1e3, 2) if world > 1 else 0.0,
                                 "fp16_ms_link_bound": round(blk16 / XGMI_LINK_BPS * 1e3, 2) if world > 1 else 0.0},
           "source": args.shard_files or "even split of the published node / symmetrised edge counts (SURVEY.md 8a)", "per_rank": out_rows}
    return doc


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start one process per GPU ourselves (torch.distributed.run, rendezvous on
    127.0.0.1 at a free port) with the same arguments; rank 0 of that job prints the JSON line, last, on the stdout we share
    with it.  The driver's torchrun form (RANK / WORLD_SIZE already in the environment) never comes through here."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    nproc = max(1, args.gpus)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver (RCCL / device-tensor sharing)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // nproc)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    return subprocess.call(cmd, env=env)


# ---- the line the driver parses.  r03's single object had grown to 20 KB and the driver's parser got nothing out of it
#      (BENCH_r03.json "parsed": null): the LAST stdout line is now a bounded summary, everything else goes to bench_detail.json.
LINE_LIMIT = 4000   # bytes; tests/test_bench_line.py holds the line to < 4096
TOP_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
CONFIG_KEYS = ("workload", "graph", "tc_blocks_16x8", "parallelism")
ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes", "kernel_ms_mean", "kernel_ms_min",
             "kernel_ms_mean_all_launches", "mfma_busy", "l2_hit_rate")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "cpu_model", "gcn_epoch_ms_cora_shape_h16", "gcn_epoch_ms_same_graph_h64")


def _leg(d, key="kernel_ms"):
    return d.get(key) if isinstance(d, dict) else None


def compact_line(out, limit=LINE_LIMIT):
    """The bounded last line: the contract's keys, `roofline`, `cpu_baseline` and a flat `summary` of scalars (most important
    first; trailing ones are dropped until the line fits `limit`).  Strings are clipped; nothing nested below two levels."""
    clip = lambda v, n=160: (v[:n] if isinstance(v, str) else v)
    line = {k: clip(out[k]) for k in TOP_KEYS if k in out}
    line["config"] = {k: clip(out.get("config", {}).get(k)) for k in CONFIG_KEYS if k in out.get("config", {})}
    line["roofline"] = {k: clip(out.get("roofline", {}).get(k), 80) for k in ROOF_KEYS if k in out.get("roofline", {})}
    if "cpu_baseline" in out:
        line["cpu_baseline"] = {k: clip(out["cpu_baseline"].get(k), 120) for k in CPU_KEYS if k in out["cpu_baseline"]}
    ex = out.get("extra", {})
    summary = []   # (name, value) in order of importance
    D = None
    for k in ex:
        m = re.match(r"sddmm_d(\d+)$", k)
        if m and "agnn_fused_fwd_d" + m.group(1) in ex:
            D = m.group(1)
    if D:
        for name, key in (("sddmm_d%s_ms", "sddmm_d%s"), ("sddmm_d%s_frac", None), ("spmm_agnn_d%s_ms", "spmm_agnn_d%s"), ("agnn_fused_fwd_d%s_ms", "agnn_fused_fwd_d%s"),
                          ("agnn_fused_bwd_d%s_ms", "agnn_fused_bwd_d%s")):
            if key is None:
                summary.append((name % D, _leg(ex.get("sddmm_d%s" % D), "hbm_frac")))
            else:
                summary.append((name % D, _leg(ex.get(key % D))))
    for k in ("gcn_ms_per_epoch", "agnn_ms_per_epoch"):
        summary.append((k, ex.get(k)))
    for row in out.get("datasets", []):   # the graph calibrated to real Reddit's TC-block count, next to the uniform headline
        if isinstance(row, dict) and row.get("workload", "").endswith("_sbm_reddit_d%s" % (D or "")) and isinstance(row.get("spmm"), dict):
            summary += [("sbm_reddit_spmm_ms", row["spmm"].get("kernel_ms")), ("sbm_reddit_spmm_frac", row["spmm"].get("hbm_frac")),
                        ("sbm_reddit_spmm_kernel", clip(row["spmm"].get("kernel"), 40)), ("sbm_reddit_tc_blocks_16x8", row.get("tc_blocks_16x8")),
                        ("sbm_reddit_gcn_ms_per_epoch", row.get("gcn_ms_per_epoch")), ("sbm_reddit_agnn_ms_per_epoch", row.get("agnn_ms_per_epoch"))]
        if isinstance(row, dict) and row.get("workload", "").endswith("_uniform_d%s" % (D or "")) and isinstance(row.get("spmm"), dict) and "(headline)" not in row.get("dataset", ""):
            # the graph r01-r04 reported as the headline: kept for continuity
            summary += [("uniform_spmm_ms", row["spmm"].get("kernel_ms")), ("uniform_spmm_gteps", row["spmm"].get("gteps")), ("uniform_spmm_frac", row["spmm"].get("hbm_frac")),
                        ("uniform_spmm_kernel", clip(row["spmm"].get("kernel"), 40)), ("uniform_sddmm_ms", _leg(row.get("sddmm"))), ("uniform_spmm_agnn_ms", _leg(row.get("spmm_val"))),
                        ("uniform_agnn_fused_fwd_ms", _leg(row.get("agnn_fused_fwd"))), ("uniform_agnn_fused_bwd_ms", _leg(row.get("agnn_fused_bwd"))),
                        ("uniform_gcn_ms_per_epoch", row.get("gcn_ms_per_epoch")), ("uniform_agnn_ms_per_epoch", row.get("agnn_ms_per_epoch"))]
        if isinstance(row, dict) and row.get("workload", "").endswith("_rmat_d%s" % (D or "")) and isinstance(row.get("spmm"), dict):
            summary += [("rmat_spmm_ms", row["spmm"].get("kernel_ms")), ("rmat_spmm_agnn_ms", _leg(row.get("spmm_val")))]
        if isinstance(row, dict) and row.get("workload") == "products_sbm_d128":
            # r06: the community graph at ogbn-products size - communities of 12.5 MB, three times an XCD's L2: the
            # slice-synchronised range walk (spmm_sync_kernel and its SDDMM / fused-forward forms)
            summary += [("products_sbm_%s_ms" % op, _leg(row.get(op))) for op in ("spmm", "sddmm", "spmm_val", "agnn_fused_fwd", "agnn_fused_bwd")]
            summary += [("products_sbm_agnn_h128_ms_per_epoch", row.get("agnn_ms_per_epoch")), ("products_sbm_gcn_h128_ms_per_epoch", row.get("gcn_ms_per_epoch"))]
            summary += [("products_sbm_spmm_kernel", clip(_leg(row.get("spmm"), "kernel"), 40)), ("products_sbm_spmm_traffic", _leg(row.get("spmm"), "traffic")),
                        ("products_sbm_sddmm_traffic", _leg(row.get("sddmm"), "traffic")), ("products_sbm_spmm_frac", _leg(row.get("spmm"), "hbm_frac")),
                        ("products_sbm_sddmm_frac", _leg(row.get("sddmm"), "hbm_frac"))]
        if isinstance(row, dict) and row.get("workload") == "products_uniform_d128":
            summary += [("products_d128_%s_ms" % op, _leg(row.get(op))) for op in ("spmm", "sddmm", "spmm_val")]
            summary += [("products_d128_sddmm_frac", _leg(row.get("sddmm"), "hbm_frac")), ("products_agnn_h128_ms_per_epoch", row.get("agnn_ms_per_epoch")),
                        ("products_gcn_h128_ms_per_epoch", row.get("gcn_ms_per_epoch"))]
    for k in ("spmm_d16", "spmm_d128", "sddmm_d16", "sddmm_d128"):
        summary.append((k + "_ms", _leg(ex.get(k))))
    for k in ("gcn_h16_ms_per_epoch", "gcn_h128_ms_per_epoch", "agnn_h16_ms_per_epoch", "agnn_h128_ms_per_epoch"):
        summary.append((k, ex.get(k)))
    sk = [k for k in ex if k.endswith("_skewed_graph")]
    if sk:
        summary.append(("skewed_spmm_ms", _leg(ex[sk[0]])))
    summary += [("value_all_launches", out.get("value_all_launches")), ("host_sgt_ms", ex.get("host_sgt_ms")), ("device_sgt_ms", ex.get("device_sgt_ms")),
                ("plan_create_ms", ex.get("plan_create_ms")),
                ("device_sgt_equals_host_sgt", ex.get("device_sgt_equals_host_sgt")), ("plan_bytes", ex.get("plan_bytes")),
                ("citeseer_spmm_d16_speedup_vs_rtx3090", ex.get("citeseer_spmm_d16_speedup_vs_rtx3090"))]
    beat = ex.get("artifact_shapes_beating_rtx3090")
    if isinstance(beat, dict):
        summary.append(("artifact_shapes_beating_rtx3090", "%s/%s spmm_d16, %s/%s gcn_h16" % (beat.get("spmm_d16"), beat.get("of"), beat.get("gcn_h16_epoch"), beat.get("of"))))
    for k in ("exchange_in_timed_step", "ms_per_step_without_exchange", "exchange_fraction_if_exchanged", "gcn_ms_per_epoch_sharded",
              "ms_per_step_with_fp16_exchange", "ms_per_step_with_overlapped_exchange", "ms_per_step_with_chunked_exchange_32col_fp32", "value_strong_gteps"):   # (the N > 1 line)
        if k in ex:
            summary.append((k, ex[k]))
    summary = [(k, v) for k, v in summary if v is not None]
    if out.get("detail"):
        line["detail"] = out["detail"]
    line["summary"] = dict(summary)
    while len(json.dumps(line)) > limit and line["summary"]:
        line["summary"].pop(next(reversed(line["summary"])))
    return line


def write_detail(out):
    """Everything the run measured (per-dataset rows, the artifact-shape table, notes) as one JSON document: bench_detail.json
    next to this file and, on a gpurun box, under gpurun_out/ so that it is merged back.  -> the path(s) written, for the line."""
    written = []
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if d != ROOT and not os.path.isdir(d):
            continue
        try:
            with open(os.path.join(d, "bench_detail.json"), "w") as f:
                json.dump(out, f, indent=1)
            written.append(os.path.relpath(os.path.join(d, "bench_detail.json"), ROOT))
        except OSError:
            pass
    return written


def main():
    args = parse()
    if args.plan_only:
        print(json.dumps(plan_only(args)), flush=True)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1 or os.environ.get("TCGNN_BENCH_FORCE_SHARDED"):   # (the env switch runs the sharded path with a world of 1)
        if "RANK" not in os.environ:
            sys.exit(self_launch(args))
        out = multi_gpu(args)
    else:
        out = single_gpu(args)
    if out is not None:
        written = write_detail(out)
        if written:
            out["detail"] = written[0]
        # RCCL prints a version banner through C stdio, which is block-buffered when stdout is a pipe or a file and would come
        # out AFTER this line at exit: drain it first so that the JSON line is the last thing on stdout
        try:
            import ctypes
            sys.stdout.flush()
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(compact_line(out)), flush=True)


if __name__ == "__main__":
    main()
