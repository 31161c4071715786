"""Graph ingestion for the TC-GNN path - the host-side mirror of the reference's dataset.py.

`TCGNN_dataset(path, dim, num_class, load_from_txt)` accepts the same inputs as dataset.py:13-122:
an .npz with `src_li`, `dst_li`, `num_nodes` (dataset.py:74-80) or a whitespace edge list, and
exposes the same attributes (`num_nodes`, `num_edges`, `num_features`, `num_classes`,
`column_index`, `row_pointers`, `x`, `y`, `degrees`).  Facts reproduced on purpose, pinned by
tests/golden/dataset_toy.npz:
  * the CSR is scipy's canonical form (duplicates merged, columns sorted): dataset.py:94-104,
  * `num_edges` is the RAW pair count, so it can exceed nnz (dataset.py:79); the metadata tensors
    main_tcgnn.py allocates from it (main_tcgnn.py:45-46) are then longer than column_index,
  * features are randn(N, dim), labels all ones (dataset.py:115,122).
Unlike the reference nothing is moved to the GPU behind the caller's back; call `.to(device)`.

Also here: seeded synthetic generators for the shapes BASELINE.json names (no datasets and no
network exist on either box), built with torch so the Reddit-sized graph can be made on the GPU.
"""
import numpy as np
import torch
from scipy.sparse import coo_matrix


def _func(x):  # config.py:5-9: degree clamp
    return x if x > 0 else 1


class TCGNN_dataset(torch.nn.Module):
    def __init__(self, path, dim, num_class, load_from_txt=True, verbose=False, seed=None):
        super().__init__()
        self.num_features = dim
        self.num_classes = num_class
        self.verbose_flag = verbose
        if load_from_txt:
            pairs = np.loadtxt(path, dtype=np.int64, ndmin=2)
            src, dst = pairs[:, 0], pairs[:, 1]
            self.num_nodes = int(max(src.max(), dst.max())) + 1 if len(src) else 0
        else:
            if not str(path).endswith(".npz"):
                raise ValueError("graph file must be a .npz file")
            obj = np.load(path)
            src, dst = obj["src_li"], obj["dst_li"]
            self.num_nodes = int(obj["num_nodes"])
        self.num_edges = len(src)  # raw count, before duplicates are merged
        self.edge_index = np.stack([src, dst])
        self.avg_degree = self.num_edges / max(self.num_nodes, 1)
        self.avg_edgeSpan = float(np.mean(np.abs(np.subtract(src, dst)))) if len(src) else 0.0
        csr = coo_matrix((np.ones(self.num_edges, dtype=np.int64), self.edge_index),
                         shape=(self.num_nodes, self.num_nodes)).tocsr()
        csr.sum_duplicates()
        csr.sort_indices()
        self.column_index = torch.from_numpy(csr.indices.astype(np.int32))
        self.row_pointers = torch.from_numpy(csr.indptr.astype(np.int32))
        deg = (self.row_pointers[1:] - self.row_pointers[:-1]).clamp(min=1).float()
        self.degrees = torch.sqrt(deg)
        gen = torch.Generator().manual_seed(seed) if seed is not None else None
        self.x = torch.randn(self.num_nodes, dim, generator=gen)
        self.y = torch.ones(self.num_nodes).long()
        if verbose:
            print("# nodes: {}".format(self.num_nodes))
            print("# avg_degree: {:.2f}".format(self.avg_degree))
            print("# avg_edgeSpan: {}".format(int(self.avg_edgeSpan)))

    def to(self, device):
        for name in ("column_index", "row_pointers", "degrees", "x", "y"):
            setattr(self, name, getattr(self, name).to(device))
        return self


# ---------------------------------------------------------------- synthetic shapes

SHAPES = {  # name: (N, nnz target, in_dim, classes) - SURVEY.md 8(d)
    "cora": (2708, 10556, 1433, 7),
    "citeseer": (3327, 9228, 3703, 6),
    "pubmed": (19717, 88648, 500, 3),
    "reddit": (232965, 114615892, 602, 41),
    "ogbn-products": (2449029, 123718280, 100, 47),
    # BASELINE.json configs[4]: symmetrised edge count; never built as ONE CSR (3.23 G > 2^31: row shards only -
    # tools/convert_dataset.py --shards, tcgnn_shard.RowShard.from_shard_file, bench.py --plan-only)
    "ogbn-papers100M": (111059956, 3231371744, 128, 172),
    # The reference's artifact graphs (1_bench_gcn.py / 2_tcgnn_single_kernel.py dataset lists give dim and classes;
    # the .npz files are not in the tree).  N and nnz are the node / edge counts of the TC-GNN paper's dataset table,
    # used for same-SIZE synthetic stand-ins (tools/bench_artifact_shapes.py) - real graphs have more locality.
    "ppi": (56944, 818716, 50, 121),
    "PROTEINS_full": (43471, 162088, 29, 2),
    "OVCAR-8H": (1890931, 3946402, 66, 2),
    "Yeast": (1714644, 3636546, 74, 2),
    "DD": (334925, 1686092, 89, 2),
    "SW-620H": (1889971, 3944206, 66, 2),
    "amazon0505": (410236, 4878875, 96, 22),
    "artist": (50515, 1638396, 100, 12),
    "com-amazon": (334863, 1851744, 96, 22),
    "soc-BlogCatalog": (88784, 2093195, 128, 39),
    "amazon0601": (403394, 3387388, 96, 22),
}


def _symmetric_csr(n, nnz_target, draw, g, dev, rounds=6):
    """Canonical symmetric CSR without self loops from an endpoint sampler: draw(m) -> (a, b) int64 tensors of m candidate
    pairs.  Pairs are symmetrised, de-duplicated and topped up until nnz is within ~0.1 % of the target (or the sampler
    stops producing new pairs: heavy-tailed samplers saturate their hubs)."""
    half = int(nnz_target) // 2
    keys = torch.empty(0, dtype=torch.int64, device=dev)
    want, boost = half, 1.002
    for _ in range(rounds):
        m = int(want * boost) + 16
        a, b = draw(m)
        lo, hi = torch.minimum(a, b), torch.maximum(a, b)
        keep = lo != hi
        before = keys.numel()
        keys = torch.unique(torch.cat([keys, lo[keep] * n + hi[keep]]))
        if keys.numel() >= half:
            break
        gained = keys.numel() - before
        boost = min(64.0, max(boost, 1.05 * m / max(gained, 1)))   # duplicates ate (1 - gained / m) of the draw: oversample accordingly
        want = half - keys.numel()
    if keys.numel() > half:
        keys = keys[torch.randperm(keys.numel(), generator=g, device=dev)[:half]]
    lo, hi = keys // n, keys % n
    full = torch.sort(torch.cat([lo * n + hi, hi * n + lo]))[0]
    rows, cols = full // n, (full % n).to(torch.int32)
    counts = torch.bincount(rows, minlength=n)
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(counts, 0)
    return rowptr.to(torch.int32), cols.contiguous()


def synthetic_csr(num_nodes, nnz_target, seed=0, device="cpu", skew=0.0):
    """Seeded symmetric graph without self loops, canonical CSR, nnz within ~0.1 % of the target.
    skew = 0: uniform endpoints; skew > 0: one endpoint drawn as floor(N * u^(1+skew)) of a random
    permutation (heavier tail).  Returns int32 (row_pointers, column_index) on `device`."""
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)
    n = int(num_nodes)

    def draw(m):
        u = torch.rand(m, generator=g, device=dev, dtype=torch.float64)
        if skew > 0:
            u = u ** (1.0 + skew)
        a = (u * n).long().clamp_(max=n - 1)
        b = torch.randint(0, n, (m,), generator=g, device=dev)
        if skew > 0:
            perm = torch.randperm(n, generator=g, device=dev)
            a = perm[a]
        return a, b
    return _symmetric_csr(n, nnz_target, draw, g, dev)


def rmat_csr(num_nodes, nnz_target, seed=0, device="cpu", abcd=(0.57, 0.19, 0.19, 0.05), shuffle=False):
    """Seeded R-MAT graph (a, b, c, d = 0.57, 0.19, 0.19, 0.05), symmetrised, self loops removed, de-duplicated,
    canonical CSR.  Endpoints are drawn in the 2^s x 2^s square (s = ceil(log2 N)) one bit per level and pairs with an
    endpoint >= N are discarded, which keeps the recursive-quadrant distribution for any N.  Low ids are the hubs and
    neighbouring ids share neighbourhoods: this is the locality the condensing step and the caches see; shuffle=True
    relabels the nodes with a random permutation (a dataset whose ids carry no structure)."""
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)
    n = int(num_nodes)
    s = max(1, (n - 1).bit_length())
    a_, b_, c_, _ = abcd
    perm = torch.randperm(n, generator=g, device=dev) if shuffle else None

    def draw(m):
        m2 = int(m * 1.3) + 16                               # head room for the pairs that fall outside [0, N)
        src = torch.zeros(m2, dtype=torch.int64, device=dev)
        dst = torch.zeros(m2, dtype=torch.int64, device=dev)
        for _ in range(s):
            u = torch.rand(m2, generator=g, device=dev)
            right = ((u >= a_) & (u < a_ + b_)) | (u >= a_ + b_ + c_)     # quadrants b and d: destination bit set
            down = u >= a_ + b_                                           # quadrants c and d: source bit set
            src = (src << 1) | down.long()
            dst = (dst << 1) | right.long()
        ok = (src < n) & (dst < n)
        src, dst = src[ok], dst[ok]
        if perm is not None:
            src, dst = perm[src], perm[dst]
        return src, dst
    return _symmetric_csr(n, nnz_target, draw, g, dev, rounds=14)


def sbm_csr(num_nodes, nnz_target, seed=0, device="cpu", blocks=50, p_in=0.9, shuffle=False, hubs=0, p_hub=0.0):
    """Seeded stochastic-block-model graph - the "community" variant SURVEY.md 8d asks for next to the uniform one because
    condensing depends on locality: `blocks` equal communities of consecutive ids; an edge stays inside its first endpoint's
    community with probability p_in, else its second endpoint is uniform.  Same post-processing as synthetic_csr.
    shuffle=True relabels the nodes at random (communities present, but invisible to a window of 16 consecutive ids)."""
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)
    n = int(num_nodes)
    size = (n + blocks - 1) // blocks
    perm = torch.randperm(n, generator=g, device=dev) if shuffle else None

    def draw(m):
        a = torch.randint(0, n, (m,), generator=g, device=dev)
        inside = torch.rand(m, generator=g, device=dev) < p_in
        blk0 = (a // size) * size
        width = torch.clamp(blk0 + size, max=n) - blk0
        b_in = blk0 + (torch.rand(m, generator=g, device=dev, dtype=torch.float64) * width).long().clamp_(max=size - 1)
        b_in = torch.minimum(b_in, blk0 + width - 1)
        b = torch.where(inside, b_in, torch.randint(0, n, (m,), generator=g, device=dev))
        if hubs > 0 and p_hub > 0:   # communities AND hubs: a fraction p_hub of the edges starts at one of `hubs` nodes spread over the graph and ends anywhere
            to_hub = torch.rand(m, generator=g, device=dev) < p_hub
            hub_id = torch.randint(0, hubs, (m,), generator=g, device=dev) * (n // hubs) + 7
            a = torch.where(to_hub, hub_id.clamp_(max=n - 1), a)
            b = torch.where(to_hub, torch.randint(0, n, (m,), generator=g, device=dev), b)
        if perm is not None:
            a, b = perm[a], perm[b]
        return a, b
    return _symmetric_csr(n, nnz_target, draw, g, dev, rounds=10)


def sbm_hubs_csr(num_nodes, nnz_target, seed=0, device="cpu"):
    """The 50-community graph with 64 hubs that hold 8 % of the edge endpoints (communities and a power-law tail at once)."""
    return sbm_csr(num_nodes, nnz_target, seed=seed, device=device, hubs=64, p_hub=0.08)


def sbm_shuffled_csr(num_nodes, nnz_target, seed=0, device="cpu"):
    """The 50-community graph under random node ids: the structure is there, the numbering hides it (what community_order undoes)."""
    return sbm_csr(num_nodes, nnz_target, seed=seed, device=device, shuffle=True)


# In-community edge share at which the 50-community graph of the Reddit shape condenses to as many 16x8 TC blocks as the REAL Reddit
# graph: 13 626 429 against 13 566 510 (/root/reference/logs/reduce_blocks.csv:18), +0.44 % (tools/calibrate_sbm.py, bisection on
# the GPU box).  The p_in = 0.9 graph SURVEY.md 8d names condenses to 8.23 M - 39 % more condensable than the real graph - and the
# uniform one to 14.11 M (+4 %): locality claims are quoted on THIS graph.
SBM_REDDIT_P_IN = 0.225


def sbm_reddit_csr(num_nodes, nnz_target, seed=0, device="cpu"):
    """The 50-community graph calibrated to real Reddit's condensability (SBM_REDDIT_P_IN)."""
    return sbm_csr(num_nodes, nnz_target, seed=seed, device=device, p_in=SBM_REDDIT_P_IN)


GENERATORS = {"uniform": synthetic_csr, "rmat": rmat_csr, "sbm": sbm_csr, "sbm_reddit": sbm_reddit_csr, "sbm_hubs": sbm_hubs_csr, "sbm_shuffled": sbm_shuffled_csr}


def community_order(row_pointers, column_index, sweeps=24, seed=0, verbose=False):
    """A node order that puts communities next to each other: `order[k]` = old id of the node that becomes node k.

    Why: the windows of the sparse-graph translation are 16 CONSECUTIVE rows, and everything downstream - how well they condense
    (dataset.py's graphs are used as numbered), which column ranges a workgroup of the LDS-resident SpMM streams, what the
    gather walks find in L2 - follows the numbering (the community graph runs 1.35-1.7x faster than the
    uniform one of the same size).  A dataset whose ids hide its communities gets them back with a relabelling, done once
    when the graph is loaded.  Not in the reference (its datasets are used as numbered).

    How: label propagation, on whatever device the CSR lives on, torch only.  Every node starts with its own label and in each
    sweep half of the nodes (a hash of node and sweep: updating all at once oscillates on bipartite-like structure) adopt the most
    frequent label among their neighbours - one sort of E (row, label) keys, run lengths, a segmented maximum with hashed
    tie-breaks; neighbours of more than 8x the mean degree do not vote (hubs would carry one label everywhere).  Stops when under
    0.1 % of the nodes change.  The order sorts the nodes by (size rank of their label, label,
    old id): big communities first, members of a community contiguous and in their old relative order.  Deterministic for a seed."""
    dev = column_index.device
    n = int(row_pointers.numel()) - 1
    if n <= 0:
        return torch.zeros(0, dtype=torch.int64, device=dev)
    deg = (row_pointers[1:] - row_pointers[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(n, device=dev), deg)
    col = column_index.long()
    # hubs do not vote: a node adjacent to a large part of the graph carries whatever label it holds into every community and one
    # label takes over (with 64 hubs on the 50-community graph plain propagation ends with ONE label).  They still receive.
    votes = deg[col] <= 8.0 * float(deg.float().mean())
    if not bool(votes.all()):
        rows, col = rows[votes], col[votes]
    label = torch.arange(n, device=dev)
    ids = torch.arange(n, device=dev)
    for s in range(sweeps):
        key = rows * n + label[col]
        key = torch.sort(key)[0]
        uniq, counts = torch.unique_consecutive(key, return_counts=True)
        urow, ulab = uniq // n, uniq % n
        noise = (ulab * 2654435761 + (s + seed) * 40503) % 251            # tie-break: a hash of the label, new every sweep
        score = (counts << 40) + (noise << 32) + ulab                      # (the label itself last: exactly one winner per node, whatever the device's write order)
        best = torch.zeros(n, dtype=torch.int64, device=dev).scatter_reduce(0, urow, score, "amax", include_self=True)
        win = score == best[urow]
        proposal = label.clone()
        proposal[urow[win]] = ulab[win]
        active = ((ids * 40503 + (s + seed) * 2654435761) >> 7) % 2 == 0 if s + 1 < sweeps else torch.ones(n, dtype=torch.bool, device=dev)
        new = torch.where(active, proposal, label)
        changed = int((new != label).sum())
        label = new
        if verbose:
            print("community_order: sweep %d, %d labels, %d nodes changed" % (s, int(torch.unique(label).numel()), changed))
        if changed < max(1, n // 1000) and s >= 2:
            break
    lab, inv, cnt = torch.unique(label, return_inverse=True, return_counts=True)
    size_rank = torch.empty_like(cnt)
    size_rank[torch.argsort(cnt, descending=True, stable=True)] = torch.arange(cnt.numel(), device=dev)
    return torch.argsort(size_rank[inv] * n + ids, stable=True)


def permute_csr(row_pointers, column_index, order):
    """The same graph with node order[k] renamed k: (row_pointers, column_index), canonical int32 CSR on the input's device.
    Features and labels follow with x[order], y[order]."""
    dev = column_index.device
    n = int(row_pointers.numel()) - 1
    newid = torch.empty(n, dtype=torch.int64, device=dev)
    newid[order] = torch.arange(n, device=dev)
    deg = (row_pointers[1:] - row_pointers[:-1]).long()
    rows = newid[torch.repeat_interleave(torch.arange(n, device=dev), deg)]
    key = torch.sort(rows * n + newid[column_index.long()])[0]
    rp = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    rp[1:] = torch.cumsum(torch.bincount(key // n, minlength=n), 0)
    return rp.to(torch.int32), (key % n).to(torch.int32)


def synthetic_shape(name, seed=0, device="cpu", scale=1.0, generator="uniform"):
    """(row_pointers, column_index, in_dim, classes) for a named shape; `scale` shrinks N and nnz
    together (tests use small scales); generator: "uniform", "rmat" or "sbm" (GENERATORS)."""
    n, nnz, dim, classes = SHAPES[name]
    n2 = max(16, int(n * scale))
    nnz2 = max(2, int(nnz * scale * scale)) if scale < 1.0 else nnz
    nnz2 = min(nnz2, n2 * (n2 - 1) // 2)
    rp, col = GENERATORS[generator](n2, nnz2, seed=seed, device=device)
    return rp, col, dim, classes


def tile_statistics(row_pointers, column_index, tile_h=16, tile_w=8, threads=0):
    """Sliding-window vs condensed tile counts of a CSR graph - the numbers 3_cnt_TC_blk_SpMM.py:38-94
    (16x8) and 3_cnt_TC_blk_SDDMM.py (16x16) print as `dataset,origin,reduced,reduction (%)`, plus the
    tile fill ("eff" of logs/16x8_reduction.csv) and blocks per window.  Host tensors / arrays; runs in
    the threaded host library (tcgnn_tile_stats), no Python loop over windows."""
    import tcgnn_capi as C
    rp = np.ascontiguousarray(torch.as_tensor(row_pointers).cpu().numpy(), dtype=np.int32)
    ci = np.ascontiguousarray(torch.as_tensor(column_index).cpu().numpy(), dtype=np.int32)
    if rp.ndim != 1 or rp.size < 1:
        raise ValueError("row_pointers must hold N + 1 entries")
    if ci.size < int(rp[-1]):
        raise ValueError("column_index is shorter than row_pointers[-1]")
    st = C.TileStats()
    C.check(C.lib.tcgnn_tile_stats(ci.ctypes.data, rp.ctypes.data, rp.size - 1, tile_h, tile_w, st, threads), "tcgnn_tile_stats")
    out = {f: int(getattr(st, f)) for f, _ in C.TileStats._fields_}
    area = float(tile_h * tile_w)
    out["reduction_pct"] = 100.0 * (out["sliding_tiles"] - out["condensed_tiles"]) / out["sliding_tiles"] if out["sliding_tiles"] else 0.0
    out["sliding_fill"] = out["edges"] / (out["sliding_tiles"] * area) if out["sliding_tiles"] else 0.0
    out["condensed_fill"] = out["edges"] / (out["condensed_tiles"] * area) if out["condensed_tiles"] else 0.0
    out["sliding_per_window"] = out["sliding_tiles"] / out["windows"] if out["windows"] else 0.0
    out["condensed_per_window"] = out["condensed_tiles"] / out["windows"] if out["windows"] else 0.0
    return out
