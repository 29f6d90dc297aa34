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
        if exchange == "compact":
            self._init_compact()
        elif exchange != "dense":
            raise ValueError("exchange must be 'dense' or 'compact'")

    # ---- compact exchange: all-gather of (feature id, sums) records of the features that occur in the step ---------------
    def _init_compact(self):
        """Every rank's record count per step is known from ingest: exchange the tables once, so that no step needs a host
        round trip to size its all-gather.  Falls back to the dense exchange if any rank's tiles are not sparse single-tile steps."""
        s = self.s
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

    def _reduce(self, t):
        if self.collective:
            return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return None

    def step(self, batch, rows_limit=0):
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
