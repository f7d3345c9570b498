"""Data-parallel harness for the FastGRNN cell: one process per GPU, utterances sharded
over ranks, ONE flattened gradient bucket all-reduced per step (RCCL over xGMI when the
process group's backend is "nccl"; gloo in the CPU tests).

The reference has no distributed code at all; this implements
the partition of SURVEY.md section 8e: batch dimension split in contiguous slices, each rank a full
parameter replica, exchange = all-reduce(sum) of [dW|dU|db_g|db_u|dzeta|dnu]
(20 738 floats = 82 952 B dense at H=128,F=32; 13 314 floats low-rank H=256,r=16),
then divide by world size.  The message is latency-bound (tens of microseconds), so it is
kept as one bucket / one collective on the compute stream.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous [lo, hi) slice of ``n_items`` utterances owned by ``rank``; the first
    ``n_items % world`` ranks take one extra (ragged batches allowed, empty shards too)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(x, rank, world, batch_dim=1):
    """Slice a [T,B,F] (batch_dim=1) or [B,...] (batch_dim=0) tensor for this rank."""
    lo, hi = shard_range(x.shape[batch_dim], rank, world)
    return x.narrow(batch_dim, lo, hi - lo).contiguous()


class GradBucket:
    """Flat fp32 bucket over a fixed parameter list.

    ``all_reduce_()`` packs every ``p.grad`` into one contiguous buffer, runs a single
    ``all_reduce(SUM)``, scales by ``1/divisor`` and scatters the result back into the
    ``.grad`` tensors.  ``divisor`` defaults to the world size (mean over ranks, the usual
    DDP convention); pass ``divisor=1`` to keep the sum (gradient of the summed loss)."""

    def __init__(self, params, world=None, group=None, divisor=None):
        self.params = [p for p in params]
        self.group = group
        self.world = world if world is not None else dist.get_world_size(group)
        self.divisor = float(self.world if divisor is None else divisor)
        self.sizes = [p.numel() for p in self.params]
        self.total = sum(self.sizes)
        p0 = self.params[0]
        self.flat = torch.zeros(self.total, dtype=p0.dtype, device=p0.device)
        self.views = list(self.flat.split(self.sizes))
        self._avg_ok = True

    def pack_(self):
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
        torch._foreach_copy_(self.views, [g.reshape(-1) for g in grads])
        return self.flat

    def unpack_(self):
        missing = [p for p in self.params if p.grad is None]
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                p.grad = v.view_as(p).clone()
        if not missing:                       # the usual case: ONE fused copy kernel, like pack_()
            torch._foreach_copy_([p.grad.view(-1) for p in self.params], self.views)

    def shared_flat_(self):
        """If every ``p.grad`` is a view of one storage, laid out back to back in parameter order -- which
        is how ``kws_amd.fastgrnn_cuda`` allocates the operator's gradient outputs and how autograd then
        adopts them -- return a flat tensor ALIASING them (no copy); else None."""
        grads = [p.grad for p in self.params]
        if any(g is None or not g.is_contiguous() for g in grads):
            return None
        g0 = grads[0]
        base = g0.untyped_storage().data_ptr()
        off = g0.storage_offset()
        for g, n in zip(grads, self.sizes):
            if g.dtype != g0.dtype or g.device != g0.device or g.untyped_storage().data_ptr() != base \
                    or g.storage_offset() != off:
                return None
            off += n
        return torch.empty(0, dtype=g0.dtype, device=g0.device).set_(g0.untyped_storage(), g0.storage_offset(),
                                                                      (self.total,))

    def _collective_(self, t):
        avg_in_collective = (self._avg_ok and self.divisor == float(self.world) and self.world > 1
                             and dist.get_backend(self.group) == "nccl")
        if avg_in_collective:
            try:
                dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group)
            except RuntimeError:              # a backend build without ncclAvg: fall back for good
                self._avg_ok = avg_in_collective = False
        if not avg_in_collective:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            if self.divisor != 1.0:
                t.mul_(1.0 / self.divisor)

    def all_reduce_(self):
        """One collective over all parameter gradients.  Zero-copy when the gradients already share one flat
        buffer (see shared_flat_); otherwise pack (1 kernel) -> collective -> unpack (1 kernel).  With the
        RCCL backend and the default divisor the mean is taken inside the collective (ReduceOp.AVG),
        otherwise SUM + one scale."""
        shared = self.shared_flat_()
        if shared is not None:
            self._collective_(shared)
            return shared
        self.pack_()
        self._collective_(self.flat)
        self.unpack_()
        return self.flat


def data_parallel_step(module, x_local, grad_hs_local, bucket, h0_local=None):
    """One forward+backward of ``module`` (a FastGRNNCUDA-like callable returning all
    hidden states) over this rank's shard followed by the bucketed gradient all-reduce.
    Returns the local hidden states."""
    for p in bucket.params:
        p.grad = None
    hs = module(x_local) if h0_local is None else module(x_local, h0_local)
    hs.backward(grad_hs_local)
    bucket.all_reduce_()
    return hs.detach()
