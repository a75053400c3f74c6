"""Data-parallel step of the MAL trainers.

The reference wraps everything in HF ``accelerate`` (DDP -> bucketed NCCL all-reduce inside
``acc.backward``, manydepth/trainer.py:309-311,469) plus a per-step barrier (:633).  The loss
path itself needs no collective: every rank normalises its losses over its OWN batch and DDP
averages gradients, so DP-G is the mean of G independent per-rank losses, not one big-batch
loss.  What is exchanged is the gradient of the network parameters -- here ONE flat fp32 bucket,
all-reduced once per step over RCCL/xGMI (``backend="nccl"`` is RCCL on ROCm), no barrier.

``FlatGradBucket`` works with any ``torch.distributed`` backend (the CPU tests use gloo).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class FlatGradBucket:
    """All parameter gradients as views of one contiguous buffer, exchanged once per step.

    On the 8-GPU xGMI mesh a single ~165 MB all-reduce (the ~41 M fp32 parameters of RepDepth)
    lets RCCL pick its direct algorithms over all 7 links instead of paying per-bucket latency.

    ``segments`` > 1 overlaps the exchange with the tail of the backward (what DDP's buckets do inside
    ``acc.backward``, manydepth/trainer.py:469): the flat buffer is cut at parameter boundaries into that many
    contiguous pieces of about equal size; ``begin_step()`` arms them, a post-accumulate hook on every parameter counts
    its piece down, and a piece is all-reduced (asynchronously, on the backend's own stream) as soon as it AND every
    piece before it in the ISSUE ORDER are complete -- the pieces are always issued in that one order, so the ranks'
    collectives match whatever order the autograd engine happens to run the leaves in.  The issue order starts as "last
    registered first"; the first armed step records the order in which the backward really completed the pieces
    (pieces holding a parameter that received no gradient last) and rank 0's record becomes everybody's order from the
    second step on (what DDP's bucket rebuild does).  ``finish()`` issues what the backward left and waits.
    The result is bit-identical to one all-reduce of the whole buffer (same element-wise reduction)."""

    def __init__(self, params, process_group=None, segments=1):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatGradBucket: no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=dt, device=dev)
        self.group = process_group
        self._avg = None
        off = 0
        offsets = []
        for p in self.params:
            if p.device != dev or p.dtype != dt:
                raise ValueError("FlatGradBucket: parameters must share device and dtype")
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            offsets.append(off)
            off += p.numel()
        # pieces: [start, end) element ranges cut at parameter boundaries, about n / segments elements each
        segments = max(1, min(int(segments), len(self.params)))
        self.bounds, self._seg_of = [], []
        start, k = 0, 0
        for i, p in enumerate(self.params):
            self._seg_of.append(k)
            end = offsets[i] + p.numel()
            if k < segments - 1 and end >= (k + 1) * n / segments:
                self.bounds.append((start, end))
                start, k = end, k + 1
        self.bounds.append((start, n))
        self._seg_of = [min(sg, len(self.bounds) - 1) for sg in self._seg_of]
        self._count = [self._seg_of.count(k) for k in range(len(self.bounds))]
        self._pending, self._next, self._works, self._armed = None, -1, [], False
        self._order = list(range(len(self.bounds) - 1, -1, -1))  # issue order; re-learnt in the first armed step
        self._seen, self._learnt = [], False
        if len(self.bounds) > 1:
            for i, p in enumerate(self.params):
                p.register_post_accumulate_grad_hook(self._make_hook(self._seg_of[i]))

    @property
    def world_size(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def zero_(self):
        self.flat.zero_()

    def check_views(self):
        """autograd accumulates in place into .grad, so the views survive backward; an optimizer
        that does ``zero_grad(set_to_none=True)`` would drop them -- call ``zero_()`` instead."""
        off = 0
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + off * self.flat.element_size():
                return False
            off += p.numel()
        return True

    def _use_avg(self):
        """does the exchange average inside the collective (``ReduceOp.AVG``: RCCL/NCCL on device buffers)?  Decided once,
        and by ALL ranks together: every rank states what its own backend offers (no probing collective whose failure on one
        rank alone would leave the others inside it) and the minimum over the ranks is taken -- a MIN all-reduce, which every
        backend has -- so that no two ranks can ever issue different collectives.  If AVG then fails, it fails on every rank."""
        if self._avg is None:
            mine = int(self.flat.is_cuda and dist.get_backend(self.group) == "nccl" and hasattr(dist.ReduceOp, "AVG"))
            flag = torch.tensor([mine], dtype=torch.int32, device=self.flat.device if self.flat.is_cuda else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            self._avg = bool(int(flag.item()))
        return self._avg

    def all_reduce_mean(self, async_op=False):
        """mean over ranks (DDP's gradient averaging): ONE collective.  RCCL/NCCL average inside the collective
        (``ReduceOp.AVG``: no second pass over the 165 MB bucket); backends without it (gloo) sum, then scale."""
        ws = self.world_size
        if ws == 1:
            return None
        if self._use_avg():
            work = dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=self.group, async_op=async_op)
            return work if async_op else None
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        if async_op:
            return _Scaled(work, self.flat, 1.0 / ws)
        self.flat.mul_(1.0 / ws)
        return None

    # ---- the exchange overlapped with the backward
    def begin_step(self):
        """zero the gradients and arm the pieces for the backward that follows"""
        self.flat.zero_()
        if len(self.bounds) > 1 and self.world_size > 1:
            self._use_avg()  # the probe collective, if any, happens here on every rank -- not inside a hook
            self._pending = list(self._count)
            self._next = 0  # position in self._order
            self._works = []
            self._seen = []
            self._armed = True

    def _issue(self, k):
        a, b = self.bounds[k]
        piece = self.flat[a:b]
        if self._avg:
            self._works.append(dist.all_reduce(piece, op=dist.ReduceOp.AVG, group=self.group, async_op=True))
        else:
            self._works.append(_Scaled(dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=True), piece,
                                       1.0 / self.world_size))

    def _make_hook(self, k):
        def hook(param):
            if not self._armed:
                return
            self._pending[k] -= 1
            if self._pending[k] < 0:
                # a second backward between begin_step() and finish() (gradient accumulation, retain_graph): the piece has
                # been averaged over the ranks already and this gradient would be added onto the averaged values
                self._armed = False
                raise RuntimeError("FlatGradBucket: a parameter of piece %d received a second gradient between begin_step() "
                                   "and finish(): the overlapped exchange covers exactly ONE backward per step (accumulate "
                                   "locally and call all_reduce_mean() once, or call begin_step() before every backward)" % k)
            if self._pending[k] == 0:
                self._seen.append(k)
            while self._next < len(self._order) and self._pending[self._order[self._next]] <= 0:
                self._issue(self._order[self._next])
                self._next += 1
        return hook

    def finish(self):
        """after backward(): issue the pieces it did not complete, wait for all.  Without armed pieces (one segment, one
        rank) this is the plain single all-reduce.  Returns the number of pieces that were issued from inside the backward."""
        if not self._armed:
            self.all_reduce_mean()
            return 0
        inside = len(self._works)
        self._armed = False
        while self._next < len(self._order):
            self._issue(self._order[self._next])
            self._next += 1
        for w in self._works:
            w.wait()
        self._works = []
        if not self._learnt:  # rank 0's completion order of this first step is everybody's issue order from now on
            self._learnt = True
            order = self._seen + [k for k in self._order if k not in self._seen]
            t = torch.tensor(order, dtype=torch.int64, device=self.flat.device if self.flat.is_cuda else "cpu")
            dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
            self._order = [int(v) for v in t.tolist()]
        return inside


class _Scaled:
    def __init__(self, work, flat, scale):
        self.work, self.flat, self.scale = work, flat, scale

    def wait(self):
        self.work.wait()
        self.flat.mul_(self.scale)


def shard_indices(n_items, rank, world_size):
    """Contiguous per-rank shard of a dataset index range (accelerate's prepared loader with
    shuffle=False, manydepth/trainer.py:304-311): ranks see disjoint, equally sized shards."""
    per = n_items // world_size
    return range(rank * per, (rank + 1) * per)
