"""The seeded synthetic graph generators SURVEY.md 8(d) prescribes (no datasets exist on either box): uniform, R-MAT
(0.57 / 0.19 / 0.19 / 0.05) and the 50-block "community" variant.  CPU: structure, determinism, and the locality the
condensing step sees (tile counts through tcgnn_tile_stats)."""
import numpy as np
import pytest
import torch

import tcgnn_graph as G

N, NNZ = 20000, 800000


def _facts(rp, col):
    n = rp.numel() - 1
    deg = (rp[1:] - rp[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(n), deg)
    c = col.long()
    k1 = torch.sort(rows * n + c)[0]
    k2 = torch.sort(c * n + rows)[0]
    sorted_rows = bool(((c[1:] > c[:-1]) | (rows[1:] != rows[:-1])).all())
    return n, deg, rows, c, bool(torch.equal(k1, k2)), sorted_rows


@pytest.mark.parametrize("name", sorted(G.GENERATORS))
def test_generators_make_canonical_symmetric_graphs_of_the_requested_size(name):
    rp, col = G.GENERATORS[name](N, NNZ, seed=3)
    n, deg, rows, c, symmetric, canonical = _facts(rp, col)
    assert n == N and rp.dtype == torch.int32 and col.dtype == torch.int32 and int(rp[-1]) == col.numel()
    assert symmetric and canonical and int((rows == c).sum()) == 0
    assert abs(col.numel() - NNZ) <= 2e-3 * NNZ
    rp2, col2 = G.GENERATORS[name](N, NNZ, seed=3)
    assert torch.equal(rp, rp2) and torch.equal(col, col2)
    rp3, col3 = G.GENERATORS[name](N, NNZ, seed=4)
    assert not torch.equal(col, col3[: col.numel()]) if col3.numel() >= col.numel() else True


def test_sbm_keeps_nine_edges_in_ten_inside_the_community_and_condenses_better():
    rp, col = G.sbm_csr(N, NNZ, seed=1, blocks=50, p_in=0.9)
    n, deg, rows, c, _, _ = _facts(rp, col)
    size = (N + 49) // 50
    inside = float(((rows // size) == (c // size)).float().mean())
    assert 0.88 <= inside <= 0.93                                   # p_in + (1 - p_in) / blocks, before de-duplication bites
    u = G.tile_statistics(*G.synthetic_csr(N, NNZ, seed=1))
    s = G.tile_statistics(rp, col)
    sh = G.tile_statistics(*G.sbm_csr(N, NNZ, seed=1, shuffle=True))
    assert s["condensed_tiles"] < 0.75 * u["condensed_tiles"]       # windows of consecutive ids share neighbours
    assert sh["condensed_tiles"] > 0.95 * u["condensed_tiles"]      # the same communities under random labels: invisible


def test_sbm_with_hubs_keeps_its_communities_and_grows_a_tail():
    rp, col = G.sbm_hubs_csr(N, NNZ, seed=1)
    n, deg, rows, c, _, _ = _facts(rp, col)
    size = (N + 49) // 50
    assert float(((rows // size) == (c // size)).float().mean()) > 0.7    # 8 % of the endpoints moved to the hubs, the rest as sbm_csr
    plain = (G.sbm_csr(N, NNZ, seed=1)[0][1:] - G.sbm_csr(N, NNZ, seed=1)[0][:-1]).max()
    assert int(deg.max()) > 8 * int(plain)                                 # 64 hubs hold a twelfth of the edges


def test_community_order_finds_the_communities_a_shuffled_numbering_hides():
    n, nnz, blocks = 12000, 900000, 12
    rp, col = G.sbm_csr(n, nnz, seed=5, blocks=blocks, shuffle=True)
    order = G.community_order(rp, col, seed=5)
    assert order.shape == (n,) and torch.equal(torch.sort(order)[0], torch.arange(n))          # a permutation
    assert torch.equal(order, G.community_order(rp, col, seed=5))                                # deterministic
    rp2, col2 = G.permute_csr(rp, col, order)
    n2, deg2, rows2, c2, symmetric, canonical = _facts(rp2, col2)
    assert n2 == n and symmetric and canonical and col2.numel() == col.numel()
    deg = (rp[1:] - rp[:-1]).long()
    assert torch.equal(deg2, deg[order])                                                         # the same graph, renamed
    k = 777                                                                                       # ... and the same neighbours
    old_nb = col[rp[order[k]]:rp[order[k] + 1]].long()
    newid = torch.empty(n, dtype=torch.long); newid[order] = torch.arange(n)
    assert torch.equal(torch.sort(newid[old_nb])[0], col2[rp2[k]:rp2[k + 1]].long())
    near = lambda rows, c: float(((rows - c).abs() <= n // 16).float().mean())
    _, _, rows, c, _, _ = _facts(rp, col)
    assert near(rows, c) < 0.2 and near(rows2, c2) > 0.8                                          # 2/16 for a random numbering
    plain = G.tile_statistics(*G.sbm_csr(n, nnz, seed=5, blocks=blocks))["condensed_tiles"]
    assert G.tile_statistics(rp2, col2)["condensed_tiles"] < 1.05 * plain < 0.8 * G.tile_statistics(rp, col)["condensed_tiles"]


def test_community_order_is_not_taken_over_by_hubs():
    """Hubs adjacent to every community would carry one label everywhere (plain label propagation ends with a single label on
    this graph); they receive labels but do not vote."""
    n, nnz = 12000, 900000
    kw = dict(seed=6, blocks=12, hubs=8, p_hub=0.08)
    rp, col = G.sbm_csr(n, nnz, shuffle=True, **kw)
    rp2, col2 = G.permute_csr(rp, col, G.community_order(rp, col, seed=6))
    ordered = G.tile_statistics(*G.sbm_csr(n, nnz, **kw))["condensed_tiles"]
    assert G.tile_statistics(rp2, col2)["condensed_tiles"] < 1.05 * ordered < 0.85 * G.tile_statistics(rp, col)["condensed_tiles"]


def test_rmat_is_heavy_tailed_and_follows_its_quadrant_weights():
    rp, col = G.rmat_csr(N, NNZ, seed=2)
    n, deg, rows, c, _, _ = _facts(rp, col)
    assert int(deg.max()) > 20 * float(deg.float().mean())
    assert int(deg[: N // 100].sum()) > 0.10 * col.numel()            # the first 1 % of the ids holds > 10 % of the edge ends
    # top-level quadrants of the 2^s square: the symmetrised weights are a : (b + c) / 2 : (b + c) / 2 : d; de-duplication
    # flattens the dense corner, so only the ordering and a loose band are asserted
    half = 1 << ((N - 1).bit_length() - 1)
    q00 = float(((rows < half) & (c < half)).float().mean())
    q11 = float(((rows >= half) & (c >= half)).float().mean())
    q01 = float(((rows < half) & (c >= half)).float().mean())
    assert q00 > q01 > q11 and q00 > 0.4
    t = G.tile_statistics(rp, col)
    u = G.tile_statistics(*G.synthetic_csr(N, NNZ, seed=2))
    assert t["condensed_tiles"] < u["condensed_tiles"]


def test_named_shapes_accept_a_generator():
    rp, col, dim, classes = G.synthetic_shape("reddit", seed=0, scale=0.01, generator="sbm")
    assert rp.numel() - 1 == int(232965 * 0.01) and dim == 602 and classes == 41 and col.numel() > 0
