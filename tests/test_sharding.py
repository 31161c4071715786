"""Row-window sharding + the X all-gather, world_size 2 over gloo on CPU.  The three operators are
supplied by an oracle-backed stand-in for HipShardOps (same constructor, same methods), so what is
tested here is the partitioning, the gathered numbering, the collective and the autograd wiring."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import graphs
from oracle import oracle as O


class OracleShardOps:
    """CPU stand-in: embeds the rank's rows in a square matrix over the gathered numbering."""

    def __init__(self, lrp, lcol, layout, rank, device):
        self.rows, self.num_cols, self.row_off = len(lrp) - 1, layout.num_cols, rank * layout.H
        rp = np.zeros(self.num_cols + 1, dtype=np.int32)
        rp[self.row_off + 1: self.row_off + self.rows + 1] = lrp[1:]
        rp[self.row_off + self.rows + 1:] = lrp[-1]
        self.rp, self.col = rp, np.ascontiguousarray(lcol, dtype=np.int32)
        nw = (self.num_cols + 15) // 16
        self.bp = np.zeros(nw, np.int32); self.e2c = np.zeros(len(lcol), np.int32); self.e2r = np.zeros(len(lcol), np.int32)
        O.preprocess(self.col, self.rp, self.num_cols, 16, 8, self.bp, self.e2c, self.e2r)

    def _sl(self, y):
        return torch.from_numpy(np.ascontiguousarray(y[self.row_off: self.row_off + self.rows]))

    def spmm(self, Xg):
        return self._sl(O.spmm(Xg.numpy(), self.rp, self.col, self.bp, self.e2c, self.e2r, round_mode=O.ROUND_NONE))

    def spmm_val(self, Xg, val):
        return self._sl(O.spmm_val(Xg.numpy(), self.rp, self.col, val.numpy(), self.bp, self.e2c, self.e2r, round_mode=O.ROUND_NONE))

    def sddmm(self, Xg):
        return torch.from_numpy(O.sddmm(Xg.numpy(), self.rp, self.col, self.bp, self.e2c, self.e2r, round_mode=O.ROUND_NONE))


def test_partition_rows_is_window_aligned_and_balanced():
    import tcgnn_shard as S
    rp, col = graphs.powerlaw_graph(5000, 30, seed=1)
    for world in (1, 2, 3, 8):
        b = S.partition_rows(rp, world)
        assert b[0] == 0 and b[-1] == 5000 and len(b) == world + 1
        assert all(x <= y for x, y in zip(b, b[1:]))
        assert all(x % 16 == 0 for x in b[:-1])
        nnz = [int(rp[b[p + 1]] - rp[b[p]]) for p in range(world)]
        heaviest_window = max(int(rp[min(w + 16, 5000)] - rp[w]) for w in range(0, 5000, 16))
        assert max(nnz) - min(nnz) <= 2 * heaviest_window + 1
    lay = S.ShardLayout(S.partition_rows(rp, 3))
    g = np.arange(5000)
    m = lay.remap(g)
    assert len(np.unique(m)) == 5000 and m.max() < lay.num_cols and lay.H % 16 == 0
    assert np.all(np.diff(m) > 0)   # monotone: canonical rows stay canonical


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tc-gnn_atc23_amd"), os.path.join(root, "tests")):
        sys.path.insert(0, p)
    import tcgnn_shard as S
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rp, col = graphs.powerlaw_graph(700, 16, seed=5)          # symmetric: A = A^T
        n, D = 700, 24
        rng = np.random.default_rng(0)
        X = rng.standard_normal((n, D)).astype(np.float32)
        att = rng.standard_normal(len(col)).astype(np.float32)
        shard = S.RowShard(rp, col, ops_factory=OracleShardOps)
        b0, b1 = shard.layout.bounds[rank], shard.layout.bounds[rank + 1]
        x_local = torch.from_numpy(X[b0:b1])
        bp = np.zeros((n + 15) // 16, np.int32); e2c = np.zeros(len(col), np.int32); e2r = np.zeros(len(col), np.int32)
        O.preprocess(col, rp, n, 16, 8, bp, e2c, e2r)
        Yfull = O.spmm(X, rp, col, bp, e2c, e2r, round_mode=O.ROUND_NONE)
        ok = {}
        ok["spmm"] = np.allclose(shard.spmm(x_local).numpy(), Yfull[b0:b1], atol=1e-5)
        e0, e1 = rp[b0], rp[b1]
        Yv = O.spmm_val(X, rp, col, att, bp, e2c, e2r, round_mode=O.ROUND_NONE)
        ok["spmm_val"] = np.allclose(shard.spmm_val(x_local, torch.from_numpy(att[e0:e1])).numpy(), Yv[b0:b1], atol=1e-5)
        ef = O.sddmm(X, rp, col, bp, e2c, e2r, round_mode=O.ROUND_NONE)
        ok["sddmm"] = np.allclose(shard.sddmm(x_local).numpy(), ef[e0:e1], atol=1e-4)
        # the exchange overlapped with the own-block product: A_local = [A_own | A_rest], same result up to summation order
        ok["overlapped"] = np.allclose(shard.spmm_overlapped(x_local).numpy(), Yfull[b0:b1], atol=1e-5) and 0.0 < shard._own_frac < 1.0
        # replicated placement (no exchange) gives the same gathered matrix as the collective
        ok["replicated"] = torch.equal(shard.place_replicated(torch.from_numpy(X)), shard.gather(x_local))
        # local-only construction (each rank materialises just its rows)
        lrp = (rp[b0: b1 + 1] - rp[b0]).astype(np.int32)
        shard2 = S.RowShard(bounds=shard.layout.bounds, local=(lrp, col[e0:e1]), ops_factory=OracleShardOps)
        ok["local_ctor"] = np.allclose(shard2.spmm(x_local).numpy(), Yfull[b0:b1], atol=1e-5)
        # autograd through the exchange: loss = sum(W-weighted aggregate); dX_local = (A dY)[rows]
        xl = x_local.clone().requires_grad_(True)
        w = torch.nn.Parameter(torch.ones(D, 1))
        y = shard.aggregate(xl) @ w
        G = rng.standard_normal((n, 1)).astype(np.float32)
        (y * torch.from_numpy(G[b0:b1])).sum().backward()
        dY = G @ np.ones((1, D), np.float32)
        ok["backward"] = np.allclose(xl.grad.numpy(), O.spmm(dY, rp, col, bp, e2c, e2r, round_mode=O.ROUND_NONE)[b0:b1], atol=1e-4)
        S.allreduce_gradients([w])
        ok["allreduce"] = np.allclose(w.grad.numpy()[:, 0], (Yfull * G).sum(0), rtol=1e-4, atol=1e-3)
        np.save(os.path.join(out_dir, "rank%d.npy" % rank), np.array([int(v) for v in ok.values()]))
        with open(os.path.join(out_dir, "rank%d.txt" % rank), "w") as f:
            f.write(repr(ok))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_aggregation_matches_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        flags = np.load(os.path.join(tmp_path, "rank%d.npy" % r))
        assert flags.all(), open(os.path.join(tmp_path, "rank%d.txt" % r)).read()


def _gcn_worker(rank, world, port, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tc-gnn_atc23_amd"), os.path.join(root, "tests")):
        sys.path.insert(0, p)
    import tcgnn_shard as S
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rp, col = graphs.powerlaw_graph(500, 12, seed=9)          # symmetric
        n, in_dim, hidden, classes = 500, 20, 16, 5
        rng = np.random.default_rng(1)
        X = (rng.standard_normal((n, in_dim)) * 0.1).astype(np.float32)
        y = rng.integers(0, classes, size=n)
        shard = S.RowShard(rp, col, ops_factory=OracleShardOps)
        b0, b1 = shard.layout.bounds[rank], shard.layout.bounds[rank + 1]
        model = S.ShardedGCN(in_dim, hidden, classes, num_layers=2, dropout=0.0, seed=3)
        model.weights[0].data.mul_(0.1); model.weights[1].data.mul_(0.1)
        w0 = [w.detach().clone() for w in model.weights]
        opt = torch.optim.SGD(model.parameters(), lr=0.5)
        loss = S.sharded_train_step(model, shard, torch.from_numpy(X[b0:b1]), torch.from_numpy(y[b0:b1]), opt, n)
        # the same step on the whole graph in one process, dense A, autograd
        A = np.zeros((n, n), np.float32)
        for r in range(n):
            A[r, col[rp[r]:rp[r + 1]]] = 1.0
        At = torch.from_numpy(A)
        W = [w.clone().requires_grad_(True) for w in w0]
        h = torch.relu(At @ (torch.from_numpy(X) @ W[0]))
        logp = torch.log_softmax(At @ (h @ W[1]), dim=1)
        ref_loss = -logp.gather(1, torch.from_numpy(y).view(-1, 1)).mean()
        ref_loss.backward()
        ok = {"loss": abs(float(loss) - float(ref_loss)) < 1e-5 * max(1.0, abs(float(ref_loss)))}
        for k in range(2):
            ok["w%d" % k] = torch.allclose(model.weights[k].detach(), w0[k] - 0.5 * W[k].grad, rtol=1e-4, atol=1e-6)
        np.save(os.path.join(out_dir, "gcn_rank%d.npy" % rank), np.array([int(v) for v in ok.values()]))
        with open(os.path.join(out_dir, "gcn_rank%d.txt" % rank), "w") as f:
            f.write(repr(ok) + " loss %.6f ref %.6f" % (float(loss), float(ref_loss)))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_gcn_step_equals_single_process_step(tmp_path):
    """Forward loss and the SGD-updated weights of one sharded training step (all-gather in both directions of both
    layers, summed weight gradients) equal the same step taken on the whole graph in one process."""
    port = _free_port()
    mp.spawn(_gcn_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        flags = np.load(os.path.join(tmp_path, "gcn_rank%d.npy" % r))
        assert flags.all(), open(os.path.join(tmp_path, "gcn_rank%d.txt" % r)).read()


def test_bench_self_launch_builds_a_torchrun_command_on_localhost(monkeypatch):
    """`python bench.py --gpus N` without RANK in the environment starts N ranks itself: the command is
    torch.distributed.run with one process per GPU, a rendezvous on 127.0.0.1 and bench.py's own arguments."""
    import importlib
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_a_world_of_one_can_still_issue_the_collectives(tmp_path):
    """RowShard(always_collective=True): the gather and the gradient / loss all-reduces run even with one rank (how a 1-GPU box
    rehearses the RCCL calls); the results are those of the collective-free path."""
    import tcgnn_shard as S
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        rp, col = graphs.powerlaw_graph(300, 10, seed=2)
        X = torch.from_numpy(np.random.default_rng(0).standard_normal((300, 8)).astype(np.float32))
        a = S.RowShard(rp, col, ops_factory=OracleShardOps, always_collective=False)
        b = S.RowShard(rp, col, ops_factory=OracleShardOps, always_collective=True)
        assert b.gather(X).shape[0] == b.layout.num_cols and torch.equal(a.spmm(X), b.spmm(X)) and torch.equal(a.sddmm(X), b.sddmm(X))
    finally:
        dist.destroy_process_group()
