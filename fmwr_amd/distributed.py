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
    """Element offsets (elements are fp32, or fp64 with state_fp64) inside the exchange buffer, mirroring fm_batch_kernels.hip:
    GV [p][kp] | GW [p] | CNT [p] | (has_q: QV [p][kp] | QW [p]) | tail = [G0, Q0, rows, 0].
    has_q: only FTRL with FMX_REDUCE_SUM exchanges the sums of squared gradients."""

    def __init__(self, p, kp, has_q=False):
        self.p, self.kp, self.has_q = p, kp, has_q
        self.gv = 0
        self.gw = p * kp
        self.cnt = self.gw + p
        self.qv = self.cnt + p
        self.qw = self.qv + (p * kp if has_q else 0)
        self.tail = self.qw + (p if has_q else 0)
        self.size = self.tail + 4


class _DevBuf:
    def __init__(self, ptr, n, elem_bytes=4):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f%d" % elem_bytes, "data": (ptr, False), "version": 2}


class EngineStepper:
    """The product stepper: fmx_grad / fmx_grad_buffer / fmx_apply on one GPU."""

    def __init__(self, engine, matrix, device):
        self.e, self.m = engine, matrix
        self.device = torch.device("cuda", device)
        ptr, n = engine.grad_buffer()
        self.buf = torch.as_tensor(_DevBuf(ptr, n, engine.grad_elem_bytes()), device=self.device)
        self.stream = torch.cuda.ExternalStream(engine.stream(), device=self.device)

    def grad(self, batch, rows_limit=0):
        self.e.grad(self.m, batch, rows_limit)

    def buffer(self):
        return self.buf

    def apply(self):
        self.e.apply(0)  # global row count is read from the reduced buffer tail

    def comm_context(self):
        return torch.cuda.stream(self.stream)


class DataParallel:
    """step(batch): local gradient sums -> all-reduce(sum) -> identical update on every replica."""

    def __init__(self, stepper, group=None):
        self.s = stepper
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def step(self, batch, rows_limit=0):
        self.s.grad(batch, rows_limit)
        if self.world > 1:
            with self.s.comm_context():
                dist.all_reduce(self.s.buffer(), op=dist.ReduceOp.SUM, group=self.group)
        self.s.apply()
