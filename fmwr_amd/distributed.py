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

    def __init__(self, engine, matrix, device):
        self.e, self.m = engine, matrix
        self.device = torch.device("cuda", device)
        ptr, n = engine.grad_buffer()
        self.buf = torch.as_tensor(_DevBuf(ptr, n, engine.grad_elem_bytes()), device=self.device)
        self.stream = torch.cuda.ExternalStream(engine.stream(), device=self.device)
        self.n_chunks, self.chunk_features, self.chunk_elems, self.tail_offset = engine.grad_layout()

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


class DataParallel:
    """step(batch): local gradient sums -> all-reduce(sum) -> identical update on every replica.

    With a chunked exchange buffer (engine option exchange_chunks = C > 1) the step is pipelined: the forward of the
    whole step runs first, then for each block of features the gradient sums are formed and their all-reduce is
    issued asynchronously, so the exchange of block c travels while block c+1 is being summed, and the update of block
    c runs while block c+1 is still travelling.  Same sums, same order, same result as the unchunked step."""

    def __init__(self, stepper, group=None):
        self.s = stepper
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def _reduce(self, t):
        if self.world > 1:
            return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return None

    def step(self, batch, rows_limit=0):
        chunks = getattr(self.s, "n_chunks", 1)
        if chunks <= 1:
            self.s.grad(batch, rows_limit)
            if self.world > 1:
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
