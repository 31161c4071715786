"""The oracle against the reference's golden vectors and against the mathematical contract (CPU)."""
import glob
import os

import numpy as np
import pytest

import graphs
from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SGT_FIXTURES = sorted(glob.glob(os.path.join(GOLD, "sgt_*.npz")))


def test_fixtures_present():
    assert len(SGT_FIXTURES) >= 10


@pytest.mark.parametrize("path", SGT_FIXTURES, ids=[os.path.basename(p)[4:-4] for p in SGT_FIXTURES])
def test_oracle_preprocess_equals_reference_fixture(path):
    """bit-exact vs what the reference's own compiled preprocess produced (incl. the one-past-the-end
    store for N % 16 == 0 and blockPartition = 1 for an edgeless window, seen through guard slots)."""
    f = np.load(path)
    rp, col, guard = f["rowptr"], f["col"], int(f["guard"])
    n = len(rp) - 1
    nw = (n + 15) // 16
    bp = np.full(nw + guard, -7, dtype=np.int32)
    e2c = np.zeros(len(col), dtype=np.int32)
    e2r = np.zeros(len(col), dtype=np.int32)
    tc = O.preprocess(col, rp, n, 16, 8, bp, e2c, e2r)
    assert np.array_equal(bp, f["bp_with_guard"])
    assert np.array_equal(e2c, f["e2c"])
    assert np.array_equal(e2r, f["e2r"])
    assert tc == int(f["tc_blocks"])


@pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_preprocess_equals_live_reference():
    import torch
    ref = O.load_ref()
    rp, col = graphs.powerlaw_graph(3000, 20, seed=77)
    n = len(rp) - 1
    nw = (n + 15) // 16
    bp_r = torch.zeros(nw + 1, dtype=torch.int32); e2c_r = torch.zeros(len(col), dtype=torch.int32); e2r_r = torch.zeros(len(col), dtype=torch.int32)
    ref.preprocess(torch.from_numpy(col), torch.from_numpy(rp), n, 16, 8, bp_r, e2c_r, e2r_r)
    bp = np.zeros(nw + 1, dtype=np.int32); e2c = np.zeros(len(col), dtype=np.int32); e2r = np.zeros(len(col), dtype=np.int32)
    O.preprocess(col, rp, n, 16, 8, bp, e2c, e2r)
    assert np.array_equal(bp, bp_r.numpy()) and np.array_equal(e2c, e2c_r.numpy()) and np.array_equal(e2r, e2r_r.numpy())


def test_block_count_formula_of_the_tile_study():
    """3_cnt_TC_blk_SpMM.py:64-67 counts condensed tiles as ceil(|unique cols of 16 rows| / 8):
    an independent restatement of blockPartition for non-empty windows."""
    rp, col = graphs.uniform_graph(500, 9, seed=3)
    n = len(rp) - 1
    nw = (n + 15) // 16
    bp = np.zeros(nw, dtype=np.int32); e2c = np.zeros(len(col), dtype=np.int32); e2r = np.zeros(len(col), dtype=np.int32)
    O.preprocess(col, rp, n, 16, 8, bp, e2c, e2r)
    for w in range(nw):
        u = np.unique(col[rp[16 * w]: rp[min(16 * w + 16, n)]])
        assert bp[w] == max(1, (len(u) + 7) // 8)


def _meta(rp, col):
    n = len(rp) - 1
    nw = (n + 15) // 16
    bp = np.zeros(nw, dtype=np.int32); e2c = np.zeros(len(col), dtype=np.int32); e2r = np.zeros(len(col), dtype=np.int32)
    O.preprocess(col, rp, n, 16, 8, bp, e2c, e2r)
    return bp, e2c, e2r


@pytest.mark.parametrize("D", [16, 32, 64, 128])
def test_kernels_restatement_equals_contract_fp64(D):
    """With rounding off, the tile-by-tile restatement must equal Y = A X and ef = <x_r, x_c>."""
    rp, col = graphs.powerlaw_graph(700, 14, seed=D)
    bp, e2c, e2r = _meta(rp, col)
    rng = np.random.default_rng(D)
    X = rng.standard_normal((len(rp) - 1, D)).astype(np.float32)
    att = rng.standard_normal(len(col)).astype(np.float32)
    Y64, absY = O.spmm_f64(X, rp, col)
    assert np.all(np.abs(O.spmm(X, rp, col, bp, e2c, e2r, round_mode=O.ROUND_NONE) - Y64) <= 1e-6 * (absY + 1))
    Yv64, absYv = O.spmm_f64(X, rp, col, att)
    assert np.all(np.abs(O.spmm_val(X, rp, col, att, bp, e2c, e2r, round_mode=O.ROUND_NONE) - Yv64) <= 1e-6 * (absYv + 1))
    ef64, absef = O.sddmm_f64(X, rp, col)
    assert np.all(np.abs(O.sddmm(X, rp, col, bp, e2c, e2r, round_mode=O.ROUND_NONE) - ef64) <= 1e-6 * (absef + 1))


def test_tf32_rounding_error_is_bounded_by_operand_ulp():
    """TF32 mode differs from fp64 by at most 2^-11 per operand (relative to sum |a||x|)."""
    rp, col = graphs.uniform_graph(600, 12, seed=9)
    bp, e2c, e2r = _meta(rp, col)
    X = np.random.default_rng(1).standard_normal((600, 32)).astype(np.float32)
    Y64, absY = O.spmm_f64(X, rp, col)
    Yt = O.spmm(X, rp, col, bp, e2c, e2r, round_mode=O.ROUND_TF32)
    assert np.all(np.abs(Yt - Y64) <= (2.0 ** -11) * absY * 1.01 + 1e-6)
    ef64, absef = O.sddmm_f64(X, rp, col)
    eft = O.sddmm(X, rp, col, bp, e2c, e2r, round_mode=O.ROUND_TF32)
    assert np.all(np.abs(eft - ef64) <= (2.0 ** -10) * absef * 1.01 + 1e-6)


def test_rounding_helpers_known_answers():
    # tf32 = 10 explicit mantissa bits, ties away from zero (cvt.rna)
    assert O.round_tf32(np.float32(1.0 + 2.0 ** -11)) == np.float32(1.0 + 2.0 ** -10)   # tie -> away
    assert O.round_tf32(np.float32(-(1.0 + 2.0 ** -11))) == np.float32(-(1.0 + 2.0 ** -10))
    assert O.round_tf32(np.float32(1.0 + 2.0 ** -12)) == np.float32(1.0)
    # fp16 = nearest even
    assert O.round_fp16(np.float32(1.0 + 2.0 ** -11)) == np.float32(1.0)                 # tie -> even
    assert O.round_fp16(np.float32(1.0 + 3 * 2.0 ** -11)) == np.float32(1.0 + 2.0 ** -9)
    assert O.round_fp16(np.float32(65519.0)) == np.float32(65504.0)
    assert np.isinf(O.round_fp16(np.float32(65520.0)))
    assert O.round_fp16(np.float32(2.0 ** -24)) == np.float32(2.0 ** -24)                # smallest subnormal
    assert O.round_fp16(np.float32(2.0 ** -25)) == np.float32(0.0)
    x = np.random.default_rng(0).standard_normal(2000).astype(np.float32)
    assert np.array_equal(O.round_fp16(x), x.astype(np.float16).astype(np.float32))


def test_reference_quirks_are_reproducible():
    """ref_quirks=True shows the out-of-domain behaviour the product deliberately does not copy."""
    rp, col = graphs.uniform_graph(64, 6, seed=2)
    bp, e2c, e2r = _meta(rp, col)
    X = np.random.default_rng(3).standard_normal((64, 41)).astype(np.float32)
    Yq = O.spmm(X, rp, col, bp, e2c, e2r, round_mode=O.ROUND_NONE, ref_quirks=True)
    Y = O.spmm(X, rp, col, bp, e2c, e2r, round_mode=O.ROUND_NONE)
    assert np.array_equal(Yq[:, :32], Y[:, :32])          # floor(41/16) = 2 column tiles computed
    assert not Yq[:, 32:].any() and Y[:, 32:].any()       # the rest stays zero in the reference
    X200 = np.random.default_rng(4).standard_normal((64, 200)).astype(np.float32)
    assert not O.spmm(X200, rp, col, bp, e2c, e2r, round_mode=0, ref_quirks=True)[:, 128:].any()  # 8 warps only


def test_csr_baseline_is_the_same_operator():
    rp, col = graphs.uniform_graph(900, 10, seed=5)
    X = np.random.default_rng(5).standard_normal((900, 24)).astype(np.float32)
    Y64, absY = O.spmm_f64(X, rp, col)
    assert np.all(np.abs(O.csr_spmm(X, rp, col, threads=2) - Y64) <= 1e-6 * (absY + 1))


@pytest.mark.parametrize("maker", [lambda: graphs.uniform_graph(1000, 20, seed=3), lambda: graphs.powerlaw_graph(700, 12, seed=4),
                                   lambda: graphs.uniform_graph(47, 5, seed=5)], ids=["uniform_n1000", "powerlaw_n700", "ragged_last_window_n47"])
def test_window_list_entry_points_are_the_same_thread_block_bodies(maker):
    """oracle_spmm_windows / oracle_sddmm_windows (what the GPU tests at BASELINE size evaluate on sampled row windows) run the
    SAME per-window bodies as oracle_spmm / oracle_sddmm: bit-equal on the listed windows, nothing written elsewhere."""
    rp, col = maker()
    n = len(rp) - 1
    bp, e2c, e2r, _ = graphs.host_sgt(rp, col)
    rng = np.random.default_rng(n)
    X = rng.standard_normal((n, 24)).astype(np.float32)
    att = rng.standard_normal(len(col)).astype(np.float32)
    nw = (n + 15) // 16
    wins = np.sort(rng.choice(nw, size=min(5, nw), replace=False)).astype(np.int32)
    if nw - 1 not in wins:
        wins = np.append(wins, np.int32(nw - 1)).astype(np.int32)     # the ragged last window
    for mode in (O.ROUND_TF32, O.ROUND_NONE):
        Y = O.spmm(X, rp, col, bp, e2c, e2r, round_mode=mode); Yv = O.spmm_val(X, rp, col, att, bp, e2c, e2r, round_mode=mode)
        ef = O.sddmm(X, rp, col, bp, e2c, e2r, round_mode=mode)
        rows, Ys = O.spmm_windows(X, rp, col, bp, e2c, e2r, wins, round_mode=mode)
        assert rows.max() < n and np.array_equal(Ys, Y[rows])
        _, Yvs = O.spmm_windows(X, rp, col, bp, e2c, e2r, wins, att=att, round_mode=mode)
        assert np.array_equal(Yvs, Yv[rows])
        edges, efs = O.sddmm_windows(X, rp, col, bp, e2c, e2r, wins, round_mode=mode)
        assert np.array_equal(efs, ef[edges])
        want = np.concatenate([np.arange(rp[min(16 * w, n)], rp[min(16 * w + 16, n)]) for w in wins])
        assert np.array_equal(edges, want)
