"""Synchronous data-parallel mini-batch training: one process per GPU, rows sharded contiguously by rank, one
all-reduce(sum) of the gradient-sum buffer per step (RCCL over xGMI through torch.distributed, backend "nccl").

The reference has no counterpart (single process); BASELINE.json's north_star prescribes the scheme.
Every replica holds the full (w0, w, V) and applies the identical update from the identical reduced sums, so replicas
stay in lock step.  The buffer is the engine's own exchange buffer (fmx_grad_buffer), viewed as a torch tensor without
a copy; the all-reduce is enqueued on the engine's HIP stream, so no host synchronisation happens inside a step.

Import order: the torch wheel carries its own copy of the HIP runtime.  When torch is imported first, libfmx.so binds
to that same copy and both share one runtime (what bench.py and the tests' workers do).  Loading libfmx.so first and
torch afterwards leaves two runtimes in the process and torch then finds no GPU, so that order is refused here.
"""
import sys

from . import _lib as _L

if _L._lib is not None and "torch" not in sys.modules:
    raise ImportError("fmwr_amd.distributed (or torch) must be imported before the first fmwr_amd call: libfmx.so is already "
                      "loaded with the system HIP runtime and torch would bring a second one")

import os

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def shard_rows(n_total, rank, world):
    """Contiguous row range [r0, r1) of rank `rank`: rank r gets rows [r*n/N, (r+1)*n/N)."""
    return (n_total * rank) // world, (n_total * (rank + 1)) // world


class GradLayout:
    """Element offsets (elements are fp32, or fp64 with state_fp64) inside the exchange buffer, mirroring
    fm_batch_kernels.hip / fmx_grad_layout: `n_chunks` blocks of `F` consecutive features, each block
    GV [F][kp] | GW [F] | CNT [F] | (has_q: QV [F][kp] | QW [F]), then tail = [G0, Q0, rows, 0].
    has_q: only FTRL with FMX_REDUCE_SUM exchanges the sums of squared gradients.  chunks <= 1: one block."""

    def __init__(self, p, kp, has_q=False, chunks=1):
        self.p, self.kp, self.has_q = p, kp, has_q
        if chunks > 1:
            per = -(-p // chunks)
            self.F = -(-per // 64) * 64
        else:
            self.F = -(-p // 4) * 4
        F = self.F
        self.n_chunks = -(-p // F)
        # offsets relative to the start of a block
        self.gv = 0
        self.gw = F * kp
        self.cnt = self.gw + F
        self.qv = self.cnt + F
        self.qw = self.qv + (F * kp if has_q else 0)
        self.block_elems = self.qw + (F if has_q else 0)
        self.tail = self.n_chunks * self.block_elems
        self.size = self.tail + 4

    def features(self, c):
        """Feature range [f0, f1) of block c."""
        return c * self.F, min(self.p, (c + 1) * self.F)

    def block(self, c):
        """Element range of block c."""
        return c * self.block_elems, (c + 1) * self.block_elems


class _DevBuf:
    def __init__(self, ptr, n, elem_bytes=4):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f%d" % elem_bytes, "data": (ptr, False), "version": 2}


class _DevInts:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (ptr, False), "version": 2}


class EngineStepper:
    """The product stepper: fmx_grad / fmx_grad_buffer / fmx_apply on one GPU (and their chunked forms)."""

    def __init__(self, engine, matrix, device, dense=True):
        """dense=False: the dense exchange buffer ((kp + 2) * p elements: 4.5 GB at p = 33 M, k = 32) is not allocated until a
        dense step asks for it -- a driver that only uses the compact exchange never pays for it."""
        self.e, self.m = engine, matrix
        self.device = torch.device("cuda", device)
        self.stream = torch.cuda.ExternalStream(engine.stream(), device=self.device)
        self._buf = None
        if dense:
            self._dense()

    def _dense(self):
        if self._buf is None:
            ptr, n = self.e.grad_buffer()
            self._buf = torch.as_tensor(_DevBuf(ptr, n, self.e.grad_elem_bytes()), device=self.device)
            self._layout = self.e.grad_layout()
        return self._buf

    @property
    def buf(self):
        return self._dense()

    @property
    def n_chunks(self):
        self._dense(); return self._layout[0]

    @property
    def chunk_features(self):
        self._dense(); return self._layout[1]

    @property
    def chunk_elems(self):
        self._dense(); return self._layout[2]

    @property
    def tail_offset(self):
        self._dense(); return self._layout[3]

    def grad(self, batch, rows_limit=0):
        self.e.grad(self.m, batch, rows_limit)

    def buffer(self):
        return self.buf

    def apply(self):
        self.e.apply(0)  # global row count is read from the reduced buffer tail

    def comm_context(self):
        return torch.cuda.stream(self.stream)

    # chunked exchange
    def grad_begin(self, batch, rows_limit=0):
        self.e.grad_begin(self.m, batch, rows_limit)

    def grad_chunk(self, c):
        self.e.grad_chunk(self.m, c)

    def tail(self):
        return self.buf[self.tail_offset:self.tail_offset + 4]

    def chunk(self, c):
        return self.buf[c * self.chunk_elems:(c + 1) * self.chunk_elems]

    def apply_chunk(self, c, last):
        self.e.apply_chunk(c, 0, last)

    # compact exchange (steps of one sparse tile: records of the occurring features instead of the dense buffer)
    def compact_usable(self):
        self.rec_elems, self.rec_cap, ok = self.e.compact_info(self.m)
        return ok

    def compact_counts(self):
        """records each of this rank's steps publishes (known from ingest)"""
        return [self.e.compact_count(self.m, b) for b in range(self.e.num_batches(self.m))]

    def compact_reserve(self, cap):
        self.e.compact_reserve(cap)
        ptr, _, tail = self.e.compact_records()
        eb = self.e.grad_elem_bytes()
        self.rec = torch.as_tensor(_DevBuf(ptr, cap * self.rec_elems, eb), device=self.device).view(cap, self.rec_elems)
        self.ctail = torch.as_tensor(_DevBuf(tail, 4, eb), device=self.device)

    def grad_compact(self, batch, rows_limit=0):
        self.e.grad_compact(self.m, batch, rows_limit)

    def compact_tail(self):
        return self.ctail

    def compact_send(self, n):
        return self.rec[:n]

    def compact_recv(self, world, n):
        need = world * n * self.rec_elems
        if getattr(self, "_recv", None) is None or self._recv.numel() < need:
            self._recv = torch.empty(max(need, 1), dtype=self.rec.dtype, device=self.device)
        return self._recv[:need].view(world * n, self.rec_elems)

    def apply_compact(self, recv, counts, stride):
        self.e.apply_compact(recv.data_ptr(), counts, stride, 0)  # the global row count travelled in the tail

    # streamed steps: the matrix of the step changes every step (engine.Source.next())
    def set_matrix(self, m):
        self.m = m
        if getattr(self, "rec_elems", None) is None:
            self.rec_elems = self.e.compact_info(m)[0]

    def compact_reserve_grow(self, n):
        """room for n records of this rank (streamed steps: the count is only known when the step arrives)"""
        if getattr(self, "_rec_cap", 0) < n:
            self._rec_cap = n + n // 8 + 1
            self.compact_reserve(self._rec_cap)

    # owner-sharded exchange (include/fmx.h): feature j belongs to rank j mod N
    def owner_configure(self, world, rank):
        self.e.owner_configure(world, rank)
        self.rec_elems = self.e.compact_info(self.m)[0] if self.m is not None else None
        self.row_elems = None

    def num_batches(self):
        return self.e.num_batches(self.m)

    def owner_usable(self):
        self.rec_elems, _, ok = self.e.compact_info(self.m)
        return ok

    def owner_counts(self, batch):
        return self.e.owner_info(self.m, batch)[0]

    def owner_ids(self, batch, n):
        _, ptr = self.e.owner_info(self.m, batch)
        return torch.as_tensor(_DevInts(ptr, max(n, 1)), device=self.device)[:n]

    def _scratch(self, name, shape, dtype):
        n = int(np.prod(shape)) if len(shape) else 1
        t = getattr(self, name, None)
        if t is None or t.numel() < n or t.dtype != dtype:
            t = torch.empty(max(n + n // 8, 1), dtype=dtype, device=self.device)
            setattr(self, name, t)
        return t[:n].view(*shape)

    def state_dtype(self):
        return torch.float64 if self.e.grad_elem_bytes() == 8 else torch.float32

    def ids_buffer(self, n):
        return self._scratch("_req", (n,), torch.int32)

    def rows_buffer(self, which, n):
        if self.row_elems is None:
            self.row_elems = self.e.k_padded() + 4
        return self._scratch("_rows_" + which, (n, self.row_elems), self.state_dtype())

    def rows_pack(self, ids, out):
        self.e.rows_pack(ids.data_ptr(), ids.numel(), out.data_ptr())

    def rows_unpack(self, ids, rows):
        self.e.rows_unpack(ids.data_ptr(), ids.numel(), rows.data_ptr())

    def records_view(self, n):
        """the n records the last grad_compact wrote (owner-major with owner_configure)"""
        ptr, _, tail = self.e.compact_records()
        eb = self.e.grad_elem_bytes()
        self.ctail = torch.as_tensor(_DevBuf(tail, 4, eb), device=self.device)
        return torch.as_tensor(_DevBuf(ptr, max(n, 1) * self.rec_elems, eb), device=self.device).view(-1, self.rec_elems)[:n]

    def records_buffer(self, n):
        return self._scratch("_recs_in", (n, self.rec_elems), self.state_dtype())

    def apply_parts(self, recv, counts, starts):
        self.e.apply_compact_parts(recv.data_ptr(), counts, starts, 0)


class DataParallel:
    """step(batch): local gradient sums -> all-reduce(sum) -> identical update on every replica.

    With a chunked exchange buffer (engine option exchange_chunks = C > 1) the step is pipelined: the forward of the
    whole step runs first, then for each block of features the gradient sums are formed and their all-reduce is
    issued asynchronously, so the exchange of block c travels while block c+1 is being summed, and the update of block
    c runs while block c+1 is still travelling.  Same sums, same order, same result as the unchunked step."""

    def __init__(self, stepper, group=None, exchange="dense"):
        self.s = stepper
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # FMX_DP_FORCE_COLLECTIVES=1: issue every collective even with one rank (they are identities then) -- lets the real backend
        # (RCCL) run the whole exchange on a one-GPU box: tests/test_gpu_distributed.py
        self.collective = self.world > 1 or (dist.is_initialized() and os.environ.get("FMX_DP_FORCE_COLLECTIVES") == "1")
        self.exchange = exchange
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if exchange == "compact":
            self._init_compact()
        elif exchange == "owner":
            self._init_owner()
        elif exchange != "dense":
            raise ValueError("exchange must be 'dense', 'compact' or 'owner'")

    # ---- compact exchange: all-gather of (feature id, sums) records of the features that occur in the step ---------------
    def _init_compact(self):
        """Every rank's record count per step is known from ingest: exchange the tables once, so that no step needs a host
        round trip to size its all-gather.  Falls back to the dense exchange if any rank's tiles are not sparse single-tile steps."""
        s = self.s
        self.bytes_per_step = []
        if getattr(s, "m", True) is None:   # a streamed run: every step brings its own matrix and its own count (train_stream)
            self.counts = None
            return
        ok = torch.tensor([1 if s.compact_usable() else 0], dtype=torch.int64, device=s.device)
        if self.collective:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        if int(ok.item()) == 0:
            self.exchange = "dense"
            return
        mine = torch.tensor(s.compact_counts(), dtype=torch.int64, device=s.device)
        nb = torch.tensor([mine.numel()], dtype=torch.int64, device=s.device)
        if self.collective:
            dist.all_reduce(nb, op=dist.ReduceOp.MAX, group=self.group)
        pad = torch.zeros(int(nb.item()), dtype=torch.int64, device=s.device)
        pad[:mine.numel()] = mine
        table = [torch.zeros_like(pad) for _ in range(self.world)]
        if self.collective:
            dist.all_gather(table, pad, group=self.group)
        else:
            table = [pad]
        self.counts = torch.stack(table).cpu().numpy()       # [world][steps]
        self.max_records = int(self.counts.max()) if self.counts.size else 0
        # every rank sends slices of the same length (the largest count of the step): reserve the global maximum once, so
        # that the engine never has to move its record buffer (the tensor view below points into it)
        s.compact_reserve(max(self.max_records, 1))
        self.bytes_per_step = []                              # filled as steps run: what the step moved per rank

    def _step_compact(self, batch, rows_limit):
        s = self.s
        counts = self.counts[:, batch]
        n = int(counts.max())
        with s.comm_context():
            s.grad_compact(batch, rows_limit)
            if self.collective:
                dist.all_reduce(s.compact_tail(), op=dist.ReduceOp.SUM, group=self.group)
                recv = s.compact_recv(self.world, n)
                if n > 0:
                    send = s.compact_send(n)
                    try:
                        dist.all_gather_into_tensor(recv, send, group=self.group)
                    except (RuntimeError, NotImplementedError):  # a backend without the flat form (gloo with device tensors)
                        dist.all_gather(list(recv.view(self.world, n, -1).unbind(0)), send.contiguous(), group=self.group)
            else:
                recv = s.compact_send(n)
            s.apply_compact(recv, counts, n)
        self.last_exchange_bytes = int(self.world * n * s.rec_elems * recv.element_size()) if n > 0 else 0

    # ---- owner-sharded exchange: records go to the rank that owns the feature (id mod N), the owner updates, rows come back when needed
    def _init_owner(self):
        """Feature j belongs to rank j mod N.  A resident matrix's per-step, per-owner record counts are known from ingest and exchanged
        once; a streamed step's counts are exchanged when the step arrives (stream=True in step())."""
        s = self.s
        s.owner_configure(self.world, self.rank)
        self.owner_table = None
        self.bytes_sent = []
        if s.m is None:
            return
        ok = torch.tensor([1 if s.owner_usable() else 0], dtype=torch.int64, device=s.device)
        if self.collective:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        if int(ok.item()) == 0:
            raise ValueError("the owner-sharded exchange needs steps of one sparse tile on every rank")
        nb = s.num_batches()
        mine = torch.tensor(np.stack([s.owner_counts(b) for b in range(nb)]), dtype=torch.int64, device=s.device)  # [steps][owners]
        steps = torch.tensor([nb], dtype=torch.int64, device=s.device)
        if self.collective:
            dist.all_reduce(steps, op=dist.ReduceOp.MAX, group=self.group)
        pad = torch.zeros(int(steps.item()), self.world, dtype=torch.int64, device=s.device)
        pad[:nb] = mine
        table = [torch.zeros_like(pad) for _ in range(self.world)]
        if self.collective:
            dist.all_gather(table, pad, group=self.group)
        else:
            table = [pad]
        self.owner_table = torch.stack(table).cpu().numpy()   # [rank][step][owner]

    def _gather_counts(self, mine):
        """[rank][owner] counts of one streamed step: a small all-gather on its own stream, so that it waits for nothing but the
        other ranks (the host is about one step ahead of the GPU; the engine's stream is not drained)."""
        if not self.collective:
            return np.asarray(mine, np.int64).reshape(1, -1)
        dev = self.s.device
        if dev.type == "cuda" and dist.get_backend(self.group) != "gloo":
            if getattr(self, "_side", None) is None:
                self._side = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(self._side):
                t = torch.tensor(np.asarray(mine, np.int64), device=dev)
                out = torch.empty(self.world, len(mine), dtype=torch.int64, device=dev)
                try:
                    dist.all_gather_into_tensor(out, t, group=self.group)
                except (RuntimeError, NotImplementedError):
                    dist.all_gather(list(out.unbind(0)), t, group=self.group)
                return out.cpu().numpy()
        t = torch.tensor(np.asarray(mine, np.int64))
        out = [torch.zeros_like(t) for _ in range(self.world)]
        dist.all_gather(out, t, group=self.group)
        return torch.stack(out).numpy()

    def _all_to_all(self, out, inp, out_rows, in_rows):
        """uneven all-to-all along dim 0 (out_rows / in_rows per rank); a backend without it for device tensors (gloo) goes through the host"""
        if not self.collective:
            out.copy_(inp)
            return
        try:
            dist.all_to_all_single(out, inp, [int(x) for x in out_rows], [int(x) for x in in_rows], group=self.group)
        except (RuntimeError, NotImplementedError):
            ho, hi = torch.empty(out.shape, dtype=out.dtype), inp.cpu()
            dist.all_to_all_single(ho, hi, [int(x) for x in out_rows], [int(x) for x in in_rows], group=self.group)
            out.copy_(ho)

    def _step_owner(self, batch, rows_limit, counts=None):
        s = self.s
        me, N = self.rank, self.world
        if counts is None:
            counts = self.owner_table[:, batch, :] if self.owner_table is not None else self._gather_counts(s.owner_counts(batch))
        send, recv = counts[me], counts[:, me]         # records I hold per owner / records every rank holds for me
        n_send, n_recv = int(send.sum()), int(recv.sum())
        with s.comm_context():
            # pull: every row this step reads is the owner's current one
            ids = s.owner_ids(batch, n_send)
            req = s.ids_buffer(n_recv)
            self._all_to_all(req, ids, recv, send)
            rows_out = s.rows_buffer("out", n_recv)
            # (this rank's own features are current where they are: its own slice is neither packed nor stored -- the all-to-all carries it untouched)
            ra, rb = int(recv[:me].sum()), int(recv[:me + 1].sum())
            for a, b in ((0, ra), (rb, n_recv)):
                if b > a:
                    s.rows_pack(req[a:b], rows_out[a:b])
            rows_in = s.rows_buffer("in", n_send)
            self._all_to_all(rows_in, rows_out, send, recv)
            sa, sb = int(send[:me].sum()), int(send[:me + 1].sum())
            for a, b in ((0, sa), (sb, n_send)):
                if b > a:
                    s.rows_unpack(ids[a:b], rows_in[a:b])
            # sums -> the owners
            s.grad_compact(batch, rows_limit)
            recs = s.records_view(n_send)
            if self.collective:
                dist.all_reduce(s.compact_tail(), op=dist.ReduceOp.SUM, group=self.group)
            parts = s.records_buffer(n_recv)
            self._all_to_all(parts, recs, recv, send)
            starts = np.cumsum(recv) - recv
            s.apply_parts(parts, recv, starts)
        eb = parts.element_size()
        away_s, away_r = n_send - int(send[me]), n_recv - int(recv[me])
        self.last_exchange_bytes = int(away_s * 4 + away_r * rows_out.shape[1] * eb + away_s * recs.shape[1] * eb)   # sent to other ranks
        self.last_received_bytes = int(away_r * 4 + away_s * rows_out.shape[1] * eb + away_r * recs.shape[1] * eb)
        self.bytes_sent.append(self.last_exchange_bytes)

    def pull_all(self, p):
        """After training: refresh this rank's copies of every feature it does not own (tests, checkpoints): one pull of all p ids."""
        s = self.s
        N, me = self.world, self.rank
        if N == 1 or self.exchange != "owner":
            return
        send = np.array([len(range(o, p, N)) for o in range(N)], np.int64)   # ids owned by o
        ids_h = np.concatenate([np.arange(o, p, N, dtype=np.int32) for o in range(N)])
        recv = np.full(N, send[me], np.int64)                                  # every rank asks me for all of mine
        with s.comm_context():
            ids = torch.from_numpy(ids_h).to(s.device)
            req = s.ids_buffer(int(recv.sum()))
            self._all_to_all(req, ids, recv, send)
            rows_out = s.rows_buffer("out", int(recv.sum()))
            s.rows_pack(req, rows_out)
            rows_in = s.rows_buffer("in", int(send.sum()))
            self._all_to_all(rows_in, rows_out, send, recv)
            s.rows_unpack(ids, rows_in)

    def _reduce(self, t):
        if self.collective:
            return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return None

    def step(self, batch, rows_limit=0):
        if self.exchange == "owner":
            return self._step_owner(batch, rows_limit)
        if self.exchange == "compact":
            return self._step_compact(batch, rows_limit)
        chunks = getattr(self.s, "n_chunks", 1)
        self.last_exchange_bytes = int(self.s.buffer().numel() * self.s.buffer().element_size())
        if chunks <= 1:
            self.s.grad(batch, rows_limit)
            if self.collective:
                with self.s.comm_context():
                    dist.all_reduce(self.s.buffer(), op=dist.ReduceOp.SUM, group=self.group)
            self.s.apply()
            return
        with self.s.comm_context():  # collectives are ordered against the engine's stream, not torch's default one
            self.s.grad_begin(batch, rows_limit)
            works = [self._reduce(self.s.tail())]  # row count and w0 sums: every block's update needs the global row count
            for c in range(chunks):
                self.s.grad_chunk(c)
                works.append(self._reduce(self.s.chunk(c)))
            for c in range(chunks):
                for w in works[:2] if c == 0 else works[c + 1:c + 2]:
                    if w is not None:
                        w.wait()  # the engine's stream waits for the collective; the host does not (RCCL)
                self.s.apply_chunk(c, c == chunks - 1)


def train_stream(dp, source, steps=None, on_step=None):
    """Streamed data-parallel training: every rank owns one engine.Source over ITS row range of the stream (rank r: rows
    [r T / N, (r + 1) T / N) -- the generators are keyed by the global row id), each global step is one streamed tile per rank:
    next tile -> gradient sums -> exchange -> update.  exchange = "owner" routes the occurring features' records to their owners
    (BASELINE.json configs[3] on N GPUs), "compact" all-gathers them.  Every rank must have the same number of steps.  Returns the
    rows this rank trained on."""
    s = dp.s
    done = 0
    t = 0
    while steps is None or t < steps:
        m = source.next()
        if m is None:
            break
        s.set_matrix(m)
        if dp.exchange == "owner":
            dp._step_owner(0, 0, counts=None if dp.collective else np.asarray(s.owner_counts(0), np.int64).reshape(1, -1))
        elif dp.exchange == "compact":
            n = s.e.compact_count(m, 0)
            dp.counts = dp._gather_counts([n]).reshape(-1, 1)
            s.compact_reserve_grow(int(dp.counts.max()))
            dp._step_compact(0, 0)
        else:
            dp.step(0)
        done += m.n
        t += 1
        if on_step is not None:
            on_step(t)
    return done
