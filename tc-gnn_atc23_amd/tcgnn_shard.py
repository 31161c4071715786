"""Row-window sharding of the aggregation path across the GPUs of one node (RCCL over xGMI).

The reference is single-GPU (main_tcgnn.py:141 hard-codes cuda:0; no collective anywhere).  The
multi-GPU form below is the one SURVEY.md 8(e) derives from the path itself:

  * units = 16-row windows.  Rank p owns a contiguous block of windows, balanced by nnz
    (`partition_rows`), i.e. the rows [b_p, b_{p+1}) of A, a local int32 CSR and local
    blockPartition / edgeToColumn / edgeToRow (condensing is per window, so sharding never changes
    them), and computes Y[b_p : b_{p+1}].
  * the one real exchange step: SpMM needs every X row its columns reference, so X's row blocks are
    all-gathered (`all_gather_into_tensor`, one process per GPU, backend "nccl" = RCCL).  xGMI is
    point-to-point - each GPU pushes its block to its 7 peers at once - so the message is kept as
    ONE large gather per SpMM, not bucketed.  Blocks are padded to a common height H so the gather
    output IS the global matrix: global row id = rank * H + local row, and column ids are remapped
    to that numbering once at set-up (no compaction copy per step).
  * SDDMM shards the same way; ef stays sharded by edge range, no reduction.
  * graphs that fit one GPU can replicate X and skip the exchange (`exchange=False`).
  * backward of the aggregation uses A, not A^T, exactly like the reference's single-GPU layers
    (gnn_conv.py:46,80: symmetric graphs), so it is the same gather + SpMM on dY.

Operators come from a backend: `HipShardOps` (C ABI, tcgnn_plan_create_sharded) on GPUs, or any
object with the same three methods - the gloo world_size-2 tests on CPU pass an oracle-backed one.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from tcgnn_layers import dense_update

BLK_H = 16


def partition_rows(row_pointers, world_size):
    """Boundaries b[0..world] (multiples of 16 except the last) splitting rows into contiguous
    blocks of whole row windows with nnz as even as the windows allow."""
    rp = np.asarray(row_pointers, dtype=np.int64)
    n = len(rp) - 1
    nw = (n + BLK_H - 1) // BLK_H
    win_end = rp[np.minimum(np.arange(1, nw + 1) * BLK_H, n)]
    total = rp[n]
    bounds = [0]
    for p in range(1, world_size):
        target = total * p / world_size
        w = int(np.searchsorted(win_end, target, side="left")) + 1
        w = max(w, bounds[-1] // BLK_H)
        w = min(w, nw)
        bounds.append(min(w * BLK_H, n))
    bounds.append(n)
    return [int(b) for b in bounds]


class ShardLayout:
    """Numbering of the all-gathered feature matrix: rank p's rows live at [p*H, p*H + rows_p)."""

    def __init__(self, bounds):
        self.bounds = list(bounds)
        self.world = len(bounds) - 1
        rows = [bounds[p + 1] - bounds[p] for p in range(self.world)]
        self.H = max(BLK_H, (max(rows) + BLK_H - 1) // BLK_H * BLK_H)
        self.rows = rows
        self.num_cols = self.H * self.world

    def remap(self, global_ids):
        g = np.asarray(global_ids, dtype=np.int64)
        owner = np.searchsorted(np.asarray(self.bounds[1:], dtype=np.int64), g, side="right")
        return (owner * self.H + (g - np.asarray(self.bounds, dtype=np.int64)[owner])).astype(np.int32)


def local_csr(row_pointers, column_index, layout, rank):
    """This rank's rows as a local CSR whose column ids use the gathered numbering."""
    rp = np.asarray(row_pointers, dtype=np.int64)
    b0, b1 = layout.bounds[rank], layout.bounds[rank + 1]
    lrp = (rp[b0: b1 + 1] - rp[b0]).astype(np.int32)
    cols = layout.remap(np.asarray(column_index)[rp[b0]: rp[b1]])
    # remapping is monotone inside an owner block and blocks are ordered, so rows stay sorted
    return lrp, cols


def _host_staged(group, t):
    """gloo moves device tensors only for broadcast / all_reduce: with that backend (two test ranks sharing one GPU - RCCL
    refuses a duplicate device - or a debugging run) the gathers below go through host memory.  Never taken under RCCL."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_gather_rows(recv, send, group=None):
    """recv[world * n] <- every rank's send[n] (flat, contiguous views): all_gather_into_tensor, host-staged under gloo."""
    if _host_staged(group, send):
        r = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_gather_into_tensor(r, send.cpu(), group=group)
        recv.copy_(r)
    else:
        dist.all_gather_into_tensor(recv, send, group=group)


class HipShardOps:
    """The three kernels on this rank's row shard through the C ABI (GPU)."""

    def __init__(self, lrp, lcol, layout, rank, device):
        import tcgnn_capi as c
        self.c, self.dev = c, device
        self.rows, self.num_cols, self.row_off = len(lrp) - 1, layout.num_cols, rank * layout.H
        n, nnz = self.rows, len(lcol)
        nw = (n + BLK_H - 1) // BLK_H
        bp = np.zeros(max(nw, 1), np.int32); e2c = np.zeros(max(nnz, 1), np.int32); e2r = np.zeros(max(nnz, 1), np.int32)
        lrp = np.ascontiguousarray(lrp, np.int32); lcol = np.ascontiguousarray(lcol, np.int32)
        c.check(c.lib.tcgnn_preprocess(lcol.ctypes.data, lrp.ctypes.data, n, 16, 8, bp.ctypes.data, nw, e2c.ctypes.data, e2r.ctypes.data, None, 0),
                "tcgnn_preprocess")
        self.meta = [torch.from_numpy(a).to(device) for a in (lrp, lcol, bp[:nw], e2c[:nnz], e2r[:nnz])]
        self.nnz = nnz
        self.plan = c._vp()
        with torch.cuda.device(device):
            st = c.lib.tcgnn_plan_create_sharded(*[t.data_ptr() for t in self.meta], n, self.num_cols, self.row_off, nnz, nw,
                                                 torch.cuda.current_stream(device).cuda_stream, c.ctypes.byref(self.plan))
        c.check(st, "tcgnn_plan_create_sharded")
        self._ws = None

    def _workspace(self, D):
        need = self.c.lib.tcgnn_workspace_bytes(self.plan, D)
        if self._ws is None or self._ws.numel() < need + 256:
            self._ws = torch.empty(need + 256, dtype=torch.uint8, device=self.dev)
        off = (-self._ws.data_ptr()) % 256
        return self._ws.data_ptr() + off, self._ws.numel() - off

    def _check(self, Xg):
        assert Xg.is_cuda and Xg.is_contiguous() and Xg.dtype == torch.float32 and Xg.shape[0] == self.num_cols

    def spmm(self, Xg):
        self._check(Xg)
        D = Xg.shape[1]
        Y = torch.empty(self.rows, D, device=self.dev)
        ws, nb = self._workspace(D)
        with torch.cuda.device(self.dev):
            self.c.check(self.c.lib.tcgnn_spmm(self.plan, Xg.data_ptr(), Y.data_ptr(), D, ws, nb, torch.cuda.current_stream(self.dev).cuda_stream), "tcgnn_spmm")
        return Y

    def spmm_fp16_exchange(self, x_local, layout, rank, group=None, always_collective=False, slot=0):
        """Y_local = A_local @ X with fp16 on the wire: this rank converts ITS rows to the kernels' scaled fp16 image (global
        max|X| from a one-word all-reduce), the image slices are all-gathered (half the bytes of the fp32 gather, and no rank
        stages the whole gathered matrix again) and the SpMM reads the image directly (tcgnn_spmm_staged)."""
        image = self.exchange_fp16(x_local, layout, rank, group, always_collective, slot)
        return self.spmm_from_image(image, x_local.shape[1])

    def spmm_from_image(self, image, D):
        """Y_local = A_local @ X from a gathered fp16 image (exchange_fp16)."""
        c, dev = self.c, self.dev
        with torch.cuda.device(dev):
            Y = torch.empty(self.rows, D, device=dev)
            if getattr(image, "_tcgnn_planar", False):   # (r06: the image was staged in the LDS-resident kernel's own layout)
                c.check(c.lib.tcgnn_spmm_staged_planar(self.plan, image.data_ptr(), Y.data_ptr(), D, torch.cuda.current_stream(dev).cuda_stream), "tcgnn_spmm_staged_planar")
            else:
                c.check(c.lib.tcgnn_spmm_staged(self.plan, image.data_ptr(), Y.data_ptr(), D, torch.cuda.current_stream(dev).cuda_stream), "tcgnn_spmm_staged")
        return Y

    def exchange_fp16(self, x_local, layout, rank, group=None, always_collective=False, slot=0):
        """The exchange half of spmm_fp16_exchange: abs-max word (all-reduce MAX), this rank's rows -> scaled fp16, all-gather of
        the image slices into image buffer `slot` (a ring of two lets the exchange of one column chunk run under the multiply
        of the chunk before: RowShard.spmm_chunked).  -> the image (header + rows), ready for spmm_from_image."""
        c, dev = self.c, self.dev
        rows, D = x_local.shape
        H, world = layout.H, layout.world
        with torch.cuda.device(dev):
            planar = bool(c.lib.tcgnn_spmm_staged_layout(self.plan, D, torch.cuda.current_stream(dev).cuda_stream))
        if planar:
            return self._exchange_fp16_planar(x_local, layout, group, always_collective, slot)
        pitch = c.lib.tcgnn_x16_pitch(D)
        # ONE buffer pair per ring slot, whatever the chunk's width (keyed by width, the 64 / 64 / 44 chunks of a 172-class
        # layer held a third full image); a wider chunk than any seen so far replaces the pair
        need_image, need_send = 256 + (self.num_cols + 1) * pitch * 2, (H + 1) * pitch * 2
        if not hasattr(self, "_wire"):
            self._wire = {}
        raw = self._wire.get(slot)
        if raw is None or raw[0].numel() < need_image + 256 or raw[1].numel() < need_send:
            self._wire.pop(slot, None)
            raw = (torch.zeros(need_image + 256, dtype=torch.uint8, device=dev), torch.zeros(need_send, dtype=torch.uint8, device=dev))
            self._wire[slot] = raw
        off = (-raw[0].data_ptr()) % 256
        image = raw[0][off: off + need_image]
        body = image[256:].view(torch.float16).view(self.num_cols + 1, pitch)
        send = raw[1][:need_send].view(torch.float16).view(H + 1, pitch)
        word = image[:4].view(torch.int32)
        if not hasattr(self, "_wire_pitch"):
            self._wire_pitch = {}
        if self._wire_pitch.get(slot) != pitch:   # (another width used this slot: rows nobody sends - padding, the sentinel - must be zero)
            image.zero_(); send.zero_()
            self._wire_pitch[slot] = pitch
        collective = world > 1 or always_collective   # (a world of one can still run the collectives: bench.py's RCCL rehearsal)
        st = torch.cuda.current_stream(dev).cuda_stream
        x_local = x_local.contiguous()
        with torch.cuda.device(dev):
            word.zero_()
            c.check(c.lib.tcgnn_stage_absmax(x_local.data_ptr(), rows * D, word.data_ptr(), st), "tcgnn_stage_absmax")
            if collective:
                dist.all_reduce(word, op=dist.ReduceOp.MAX, group=group)   # bit patterns of non-negative floats order like ints
            c.check(c.lib.tcgnn_stage_rows(x_local.data_ptr(), rows, D, word.data_ptr(), send.data_ptr(), st), "tcgnn_stage_rows")
            if collective:
                all_gather_rows(body[: world * H].view(-1), send[:H].view(-1), group)
            else:
                body[:H].copy_(send[:H])
        return image

    def _exchange_fp16_planar(self, x_local, layout, group, always_collective, slot):
        """exchange_fp16 where the LDS-resident kernel takes this shard at this width (r06): the image is PLANAR -
        ceil(D / 16) planes of (num_cols + 1) 32-byte records, include/tcgnn.h - a rank's rows are one contiguous run of every plane, and
        the exchange is one all-gather per plane into that plane's first world * H records.  Same ring of two buffers, same abs-max word."""
        c, dev = self.c, self.dev
        rows, D = x_local.shape
        H, world = layout.H, layout.world
        P = (D + 15) // 16
        need_image, need_send = 256 + P * (self.num_cols + 1) * 32, P * H * 32
        if not hasattr(self, "_wire"):
            self._wire = {}
        raw = self._wire.get(slot)
        if raw is None or raw[0].numel() < need_image + 256 or raw[1].numel() < need_send:
            self._wire.pop(slot, None)
            raw = (torch.zeros(need_image + 256, dtype=torch.uint8, device=dev), torch.zeros(need_send, dtype=torch.uint8, device=dev))
            self._wire[slot] = raw
        off = (-raw[0].data_ptr()) % 256
        image = raw[0][off: off + need_image]
        planes = image[256:].view(torch.float16).view(P, self.num_cols + 1, 16)
        send = raw[1][:need_send].view(torch.float16).view(P, H, 16)
        word = image[:4].view(torch.int32)
        if not hasattr(self, "_wire_pitch"):
            self._wire_pitch = {}
        if self._wire_pitch.get(slot) != ("planar", P):   # (another layout used this slot: the header, the sentinel records and the padding must be zero)
            image.zero_(); send.zero_()
            self._wire_pitch[slot] = ("planar", P)
        collective = world > 1 or always_collective
        st = torch.cuda.current_stream(dev).cuda_stream
        x_local = x_local.contiguous()
        with torch.cuda.device(dev):
            word.zero_()
            c.check(c.lib.tcgnn_stage_absmax(x_local.data_ptr(), rows * D, word.data_ptr(), st), "tcgnn_stage_absmax")
            if collective:
                dist.all_reduce(word, op=dist.ReduceOp.MAX, group=group)
            c.check(c.lib.tcgnn_stage_rows_planar(x_local.data_ptr(), rows, D, word.data_ptr(), send.data_ptr(), H, st), "tcgnn_stage_rows_planar")
            for p in range(P):
                if collective:
                    all_gather_rows(planes[p, : world * H].reshape(-1), send[p].reshape(-1), group)
                else:
                    planes[p, :H].copy_(send[p])
        image._tcgnn_planar = True
        return image

    def spmm_val(self, Xg, val):
        self._check(Xg)
        D = Xg.shape[1]
        Y = torch.empty(self.rows, D, device=self.dev)
        ws, nb = self._workspace(D)
        with torch.cuda.device(self.dev):
            self.c.check(self.c.lib.tcgnn_spmm_val(self.plan, Xg.data_ptr(), val.contiguous().data_ptr(), Y.data_ptr(), D, ws, nb,
                                                   torch.cuda.current_stream(self.dev).cuda_stream), "tcgnn_spmm_val")
        return Y

    def sddmm(self, Xg):
        self._check(Xg)
        D = Xg.shape[1]
        ef = torch.empty(self.nnz, device=self.dev)
        ws, nb = self._workspace(D)
        with torch.cuda.device(self.dev):
            self.c.check(self.c.lib.tcgnn_sddmm(self.plan, Xg.data_ptr(), ef.data_ptr(), D, ws, nb, torch.cuda.current_stream(self.dev).cuda_stream), "tcgnn_sddmm")
        return ef

    def set_timing(self, max_calls):
        self.c.check(self.c.lib.tcgnn_plan_set_timing(self.plan, int(max_calls)), "tcgnn_plan_set_timing")

    def read_timing(self):
        buf = (self.c.ctypes.c_float * 4096)()
        n = self.c._i32(0)
        self.c.check(self.c.lib.tcgnn_plan_read_timing(self.plan, buf, 4096, self.c.ctypes.byref(n)), "tcgnn_plan_read_timing")
        return [buf[i] for i in range(n.value)]

    def close(self):
        if self.plan:
            torch.cuda.synchronize(self.dev)
            self.c.lib.tcgnn_plan_destroy(self.plan)
            self.plan = None


class _GatherSpmm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_local, shard):
        ctx.shard = shard
        return shard.exchange_spmm(x_local)

    @staticmethod
    def backward(ctx, d_out):
        shard = ctx.shard  # A, not A^T: the reference's symmetric-graph convention (gnn_conv.py:46)
        return shard.exchange_spmm(d_out.contiguous()), None


class RowShard:
    """One rank's share of a graph plus the exchange step."""

    @classmethod
    def from_shard_file(cls, path, rank=None, world_size=None, device=None, group=None, ops_factory=None, always_collective=None):
        """This rank's shard from a file written by `tools/convert_dataset.py --shards P` (BASELINE.json configs[4]: a graph whose
        3.23 G edges cannot exist as one int32 CSR is never materialised whole - dataset.py:94-104 is the single-process form this
        replaces).  `path` is the rank's own file or a template with {rank} / {world} fields ("papers.rank{rank}of{world}.npz").
        The file holds local int32 row pointers, column ids already in the padded-gather numbering, and the global boundaries."""
        rank = dist.get_rank(group) if rank is None else rank
        world = dist.get_world_size(group) if world_size is None else world_size
        obj = np.load(path.format(rank=rank, world=world))
        if int(obj["rank"]) != rank or int(obj["world"]) != world:
            raise ValueError("%s is shard %d of %d, this process is rank %d of %d" % (path, int(obj["rank"]), int(obj["world"]), rank, world))
        bounds = [int(x) for x in obj["bounds"]]
        shard = cls(rank=rank, world_size=world, device=device, group=group, ops_factory=ops_factory, bounds=bounds,
                    local=(obj["row_pointers"], obj["column_index"]), always_collective=always_collective, local_is_remapped=True)
        if shard.layout.H != int(obj["H"]):
            raise ValueError("%s was written for gather blocks of %d rows, the layout of its boundaries has %d" % (path, int(obj["H"]), shard.layout.H))
        shard.num_nodes_global = int(obj["num_nodes"])
        return shard

    def __init__(self, row_pointers=None, column_index=None, rank=None, world_size=None, device=None, group=None,
                 ops_factory=None, bounds=None, local=None, always_collective=None, local_is_remapped=False, exchange_chunk=None):
        """Either the whole CSR (row_pointers, column_index; every rank holds it or builds it the
        same way) or `local=(local_row_pointers, global_column_ids)` + `bounds` when each rank only
        ever materialises its own rows (graphs too large for one host/GPU)."""
        self.group = group
        # columns per exchange of `aggregate` / `exchange_spmm` (None: by size - exchange_chunk_for; 0: always the whole matrix in one
        # gather, the r01-r04 form; 64: the chunked exchange of spmm_chunked, what a graph of papers100M's size needs to fit - bench.py --plan-only)
        self.exchange_chunk = exchange_chunk if exchange_chunk is not None else (int(os.environ["TCGNN_SHARD_EXCHANGE_CHUNK"]) if os.environ.get("TCGNN_SHARD_EXCHANGE_CHUNK") else None)
        # a world of one normally skips the exchange; with this switch it still issues every collective of the N-rank step
        # (all_gather_into_tensor of X / of the fp16 image slices, the one-word all_reduce(MAX)): how a 1-GPU box rehearses
        # the RCCL calls of the 8-GPU run (bench.py with TCGNN_BENCH_FORCE_SHARDED=1, tests/test_gpu_sharded.py)
        self.always_collective = bool(int(os.environ.get("TCGNN_SHARD_ALWAYS_COLLECTIVE", "0"))) if always_collective is None else bool(always_collective)
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world_size is None else world_size
        self.device = device if device is not None else torch.device("cpu")
        if local is not None:
            assert bounds is not None, "local shards need the global row boundaries"
            self.layout = ShardLayout(bounds)
            lrp = np.ascontiguousarray(local[0], dtype=np.int32)
            lcol = np.ascontiguousarray(local[1], dtype=np.int32) if local_is_remapped else self.layout.remap(local[1])
        else:
            bounds = bounds if bounds is not None else partition_rows(row_pointers, self.world)
            self.layout = ShardLayout(bounds)
            lrp, lcol = local_csr(row_pointers, column_index, self.layout, self.rank)
        self.rows = self.layout.rows[self.rank]
        self.local_row_pointers, self.local_column_index = lrp, lcol
        factory = ops_factory or HipShardOps
        self._factory = factory
        self.ops = factory(lrp, lcol, self.layout, self.rank, self.device)
        self._gbuf = {}

    def gather(self, x_local):
        """[rows_p, D] -> [world * H, D]: pad to H rows, one all_gather_into_tensor (RCCL on GPUs)."""
        D = x_local.shape[1]
        H = self.layout.H
        key = (D, x_local.dtype)
        buf = self._gbuf.get(key)
        if buf is None:
            buf = (torch.zeros(H, D, dtype=x_local.dtype, device=x_local.device),
                   torch.empty(self.world * H, D, dtype=x_local.dtype, device=x_local.device))
            self._gbuf[key] = buf
        send, recv = buf
        send[: self.rows].copy_(x_local)
        if self.world == 1 and not self.always_collective:
            return send
        all_gather_rows(recv, send, self.group)
        return recv

    # ---- the exchange overlapped with the multiply ("report the exchange fraction and minimise it")
    # A rank's columns split into its OWN block - whose X rows it already holds - and the rest: A_local = [A_own | A_rest].  The
    # all-gather is issued on a side stream, A_own @ x_local runs meanwhile on the main one, and A_rest @ gathered follows when the
    # blocks have landed; the two products are added.  Same operand rounding as the one-plan form (a 10-bit mantissa does not
    # depend on the per-matrix power-of-two scale), another summation order: equal to `spmm` within accumulation noise, not bit
    # for bit.  What it can hide is min(own-block multiply, gather): 1/world of the edges on a uniform graph, far more on a
    # partitioned one (90 % of a METIS-style partition's edges are local).
    def _split_ops(self):
        if getattr(self, "_own", None) is None:
            H, r = self.layout.H, self.rank
            lrp, lcol = self.local_row_pointers.astype(np.int64), self.local_column_index.astype(np.int64)
            own = (lcol >= r * H) & (lcol < (r + 1) * H)
            erow = np.repeat(np.arange(len(lrp) - 1), np.diff(lrp))

            def sub(mask, cols):
                cnt = np.bincount(erow[mask], minlength=len(lrp) - 1)
                rp = np.zeros(len(lrp), np.int32); rp[1:] = np.cumsum(cnt)
                return rp, np.ascontiguousarray(cols.astype(np.int32))

            own_layout = ShardLayout([0, H])          # the own block as a one-rank world: H feature rows, row_off 0
            own_layout.H, own_layout.num_cols = H, H
            factory = self._factory
            self._own = factory(*sub(own, lcol[own] - r * H), own_layout, 0, self.device)
            self._rest = factory(*sub(~own, lcol[~own]), self.layout, self.rank, self.device)
            self._own_frac = float(own.mean()) if len(lcol) else 0.0
        return self._own, self._rest

    def spmm_overlapped(self, x_local):
        """Y_local = A_local @ all_gather(X) with the gather on a side stream under the own-block product (see above)."""
        own, rest = self._split_ops()
        D, H = x_local.shape[1], self.layout.H
        # (its OWN buffer pair: `gather` hands its buffers to callers, who may still read them while the side stream writes here)
        key = ("overlapped", D, x_local.dtype)
        buf = self._gbuf.get(key)
        if buf is None:
            buf = (torch.zeros(H, D, dtype=x_local.dtype, device=x_local.device),
                   torch.empty(self.world * H, D, dtype=x_local.dtype, device=x_local.device))
            self._gbuf[key] = buf
        send, recv = buf
        send[: self.rows].copy_(x_local)
        if not x_local.is_cuda:                      # (CPU stand-in of the tests: no streams)
            if self.world > 1 or self.always_collective:
                all_gather_rows(recv, send, self.group)
            else:
                recv = send
            return own.spmm(send) + rest.spmm(recv)
        main = torch.cuda.current_stream(x_local.device)
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=x_local.device)
        side = self._side
        side.wait_stream(main)                       # `send` is ready; `recv`'s previous readers are done
        with torch.cuda.stream(side):
            if self.world > 1 or self.always_collective:
                all_gather_rows(recv, send, self.group)
            else:
                recv[:H].copy_(send)
        y = own.spmm(send)                           # ... while the blocks travel
        main.wait_stream(side)
        # both buffers were allocated on the main stream and are used on the side stream: that is the stream the allocator must
        # be told about (harmless so far only because _gbuf keeps them alive)
        send.record_stream(side)
        recv.record_stream(side)
        return y.add_(rest.spmm(recv))

    # ---- the exchange in COLUMN CHUNKS (r05).  One gather of the whole N x D matrix needs
    # world * H * D fp32 words of receive buffer per width in use - 118 GB of a papers100M rank's 288 (bench.py --plan-only) - plus
    # the kernels' fp16 image of the same matrix.  Here the matrix crosses the fabric `chunk` columns at a time into a RING OF
    # TWO buffers: while the kernel multiplies chunk k out of one buffer the exchange of chunk k + 1 fills the other (side
    # stream).  With fp16 on the wire (the default from 64 columns up, where a row of the image is a whole 128-byte line) every
    # rank converts only ITS rows, the gathered buffer IS the kernels' image (tcgnn_spmm_staged) and no fp32 copy of the gathered
    # matrix exists at all: 2 x 14.2 GB instead of 118 + 42.6 GB per papers100M rank.  Results: a column of Y depends on that
    # column of X alone, and rounding to a 10-bit mantissa does not depend on the per-chunk power-of-two scale - so every chunk
    # equals the same columns of the whole-matrix call on the same walk, bit for bit, AS LONG AS no element falls below fp16's
    # normal range under either scale (each chunk carries its own abs-max, so which elements of a matrix spanning
    # more than 2^28 become subnormal or flush differs from the whole-matrix call).  And the range guard of tcgnn_spmm does NOT
    # apply on the fp16 wire: a staged image has no fp32 matrix to fall back to (include/tcgnn.h, tcgnn_spmm_staged: "never
    # wide") - a caller whose activations span that much keeps wire="fp32", whose chunks go through tcgnn_spmm and its guard.
    def spmm_chunked(self, x_local, chunk=64, wire="auto"):
        D = x_local.shape[1]
        chunk = max(16, int(chunk))
        if wire == "auto":
            wire = "fp16" if (min(chunk, D) >= 64 and hasattr(self.ops, "exchange_fp16")) else "fp32"
        if wire == "fp16" and not hasattr(self.ops, "exchange_fp16"):
            wire = "fp32"
        collective = self.world > 1 or self.always_collective
        starts = list(range(0, D, chunk))
        gpu = x_local.is_cuda
        outs = [None] * len(starts)

        def exchange(k):   # chunk k -> ring slot k & 1
            xc = x_local[:, starts[k]: starts[k] + chunk].contiguous()
            if wire == "fp16":
                return self.ops.exchange_fp16(xc, self.layout, self.rank, self.group, self.always_collective, slot=k & 1)
            H, dc = self.layout.H, xc.shape[1]
            key = ("chunk", dc, xc.dtype, k & 1)
            buf = self._gbuf.get(key)
            if buf is None:
                buf = (torch.zeros(H, dc, dtype=xc.dtype, device=xc.device), torch.empty(self.world * H, dc, dtype=xc.dtype, device=xc.device))
                self._gbuf[key] = buf
            send, recv = buf
            send[: self.rows].copy_(xc)
            if not collective:
                return send
            all_gather_rows(recv, send, self.group)
            return recv

        def multiply(k, g):
            dc = min(chunk, D - starts[k])
            return self.ops.spmm_from_image(g, dc) if wire == "fp16" else self.ops.spmm(g)

        if not gpu:           # (CPU stand-in of the tests: no streams)
            for k in range(len(starts)):
                outs[k] = multiply(k, exchange(k))
            return outs[0] if len(outs) == 1 else torch.cat(outs, 1)
        main = torch.cuda.current_stream(x_local.device)
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=x_local.device)
        side = self._side
        side.wait_stream(main)                       # x_local is ready; the ring's previous readers are done
        done = [None, None]                          # per ring slot: the multiply that read it last
        for k in range(len(starts)):
            with torch.cuda.stream(side):
                if done[k & 1] is not None:
                    side.wait_event(done[k & 1])     # the multiply of chunk k - 2 has finished with this slot
                g = exchange(k)
                ready = torch.cuda.Event(); ready.record(side)
            main.wait_event(ready)
            outs[k] = multiply(k, g)
            done[k & 1] = torch.cuda.Event(); done[k & 1].record(main)
            g.record_stream(main)
        main.wait_stream(side)
        return outs[0] if len(outs) == 1 else torch.cat(outs, 1)

    # whole-matrix fp32 gather up to this many bytes of receive buffer; beyond it the exchange goes in 64-column chunks (the
    # default has to be what bench.py --plan-only prices - a papers100M rank's 28 GB gather takes the chunked road, Reddit's 60 MB does not)
    WHOLE_GATHER_MAX_BYTES = 8 << 30

    def exchange_chunk_for(self, D):
        """Columns per exchange of `aggregate` for a width: exchange_chunk when set (argument / TCGNN_SHARD_EXCHANGE_CHUNK; 0 = never chunk),
        else 64 once the whole-matrix receive buffer would exceed WHOLE_GATHER_MAX_BYTES, else None (one gather)."""
        if self.exchange_chunk is not None:
            return self.exchange_chunk or None
        return 64 if self.world * self.layout.H * D * 4 > self.WHOLE_GATHER_MAX_BYTES else None

    def exchange_spmm(self, x_local):
        """What `aggregate` runs in either direction: the whole-matrix gather, or the chunked exchange (exchange_chunk_for)."""
        chunk = self.exchange_chunk_for(x_local.shape[1])
        if chunk:
            return self.spmm_chunked(x_local, chunk)
        return self.ops.spmm(self.gather(x_local))

    def place_replicated(self, x_global):
        """No exchange: scatter a replicated [N, D] matrix into the gathered numbering (graphs
        that fit one GPU)."""
        H, b = self.layout.H, self.layout.bounds
        out = torch.zeros(self.world * H, x_global.shape[1], dtype=x_global.dtype, device=x_global.device)
        for p in range(self.world):
            out[p * H: p * H + b[p + 1] - b[p]] = x_global[b[p]: b[p + 1]]
        return out

    def aggregate(self, x_local):
        """Differentiable Y_local = A_local @ all_gather(X)."""
        return _GatherSpmm.apply(x_local, self)

    def spmm(self, x_local, wire="fp32"):
        """wire="fp16": the exchange carries the kernels' fp16 image instead of fp32 X (HipShardOps.spmm_fp16_exchange)."""
        if wire == "fp16" and hasattr(self.ops, "spmm_fp16_exchange"):
            return self.ops.spmm_fp16_exchange(x_local, self.layout, self.rank, self.group, self.always_collective)
        return self.ops.spmm(self.gather(x_local))

    def spmm_val(self, x_local, val_local):
        return self.ops.spmm_val(self.gather(x_local), val_local)

    def sddmm(self, x_local):
        return self.ops.sddmm(self.gather(x_local))


def allreduce_gradients(params, group=None, always=False):
    """Weight gradients (KBs) are summed with one flat all-reduce; a ring is fine at this size."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not always):
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, group=group)
    off = 0
    for g in grads:
        g.copy_(flat[off: off + g.numel()].view_as(g))
        off += g.numel()


class ShardedGCN(torch.nn.Module):
    """The harness's GCN (tcgnn_harness.Net with GCNConv = main_tcgnn.py:75-139: conv1 -> relu -> dropout -> ... -> conv2 ->
    log_softmax) on a row-sharded graph.  Every rank holds its rows of the features and labels and a full copy of the
    (KB-sized) weights; a layer is the local dense update X_local W followed by `shard.aggregate` (one all-gather of the
    N x D update + the local SpMM); the backward pass of the aggregation is the same exchange on dY (A, not A^T: the
    reference's symmetric-graph convention).  Nothing else crosses the fabric until the weight gradients are summed."""

    def __init__(self, in_dim, hidden, classes, num_layers=2, dropout=0.5, seed=0):
        super().__init__()
        gen = torch.Generator().manual_seed(seed)   # identical initial weights on every rank
        dims = [in_dim] + [hidden] * (num_layers - 1) + [classes]
        self.weights = torch.nn.ParameterList(torch.nn.Parameter(torch.randn(a, b, generator=gen)) for a, b in zip(dims[:-1], dims[1:]))
        self.dropout = dropout

    def forward(self, x_local, shard):
        h = x_local
        last = len(self.weights) - 1
        for k, w in enumerate(self.weights):
            h = shard.aggregate(dense_update(h, w))
            if k != last:
                h = torch.relu(h)
                if k == 0 and self.dropout > 0:
                    h = torch.nn.functional.dropout(h, p=self.dropout, training=self.training)
        return torch.log_softmax(h, dim=1)


def sharded_train_step(model, shard, x_local, y_local, optimizer, num_nodes_global):
    """One epoch of main_tcgnn.py:146-152 on the shard: nll_loss is the mean over ALL nodes, so every rank back-propagates
    its local sum / N and the weight gradients are summed with one all-reduce.  Returns the global loss."""
    model.train()
    optimizer.zero_grad()
    logp = model(x_local, shard)
    loss_local = -logp.gather(1, y_local.view(-1, 1)).sum() / float(num_nodes_global)
    loss_local.backward()
    always = bool(getattr(shard, "always_collective", False))
    allreduce_gradients(list(model.parameters()), shard.group, always)
    optimizer.step()
    loss = loss_local.detach().clone()
    if dist.is_initialized() and (dist.get_world_size(shard.group) > 1 or always):
        dist.all_reduce(loss, group=shard.group)
    return loss
