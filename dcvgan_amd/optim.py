"""Adam on the HIP kernel, and the data-parallel optimiser wrapper.

``Adam`` has torch.optim.Adam's semantics for the reference's wiring
(train.py:171-176: betas (0.5, 0.999), eps 1e-8, L2 weight decay): parameters
whose ``.grad`` is None are skipped, bias correction uses the per-optimiser step
count.  Only ``.step()`` / ``.zero_grad()`` are needed by a DCVGAN trainer.

``DataParallelAdam`` wraps an ``Adam`` for data-parallel training without any
trainer change: optimisers that are stepped after the same
backward share a ``GradBucket`` (D phase: idis + vdis + gdis, 15.9 MB; G phase:
ggen + cgen, 55.1 MB).  The first ``.step()`` after a backward all-reduces (sum)
the whole bucket — ONE collective per phase, two per iteration — and every
member's Adam kernel applies the 1/world factor as ``grad_scale``.  "After a
backward" is explicit state: a post-accumulate-grad hook on every parameter marks
the bucket dirty, the reduction clears the mark — so the trainer's double
``opt_ggen.step()`` reduces once, and nothing depends on object ids or tensor
version counters.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, List, Optional

import torch

from .native import NativeError, _require, check, lib, ptr, stream_ptr


class Adam:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        self.params: List[torch.nn.Parameter] = [p for p in params]
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.state = {}
        self.grad_scale = 1.0

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @torch.no_grad()
    def step(self):
        L = lib()
        st = stream_ptr()
        ps, gs, ms, vs, ns, keep = [], [], [], [], [], []
        touched = []
        step = None
        for p in self.params:
            g = p.grad
            if g is None:
                continue
            _require(p.data, "Adam parameter")
            if not p.data.is_contiguous():
                raise NativeError("Adam: parameters must be contiguous")
            g = g.contiguous()
            s = self.state.get(p)
            if s is None:
                s = self.state[p] = {"step": 0, "exp_avg": torch.zeros_like(p.data), "exp_avg_sq": torch.zeros_like(p.data)}
            s["step"] += 1
            touched.append(p)
            if step is None:
                step = s["step"]
            if s["step"] != step:   # parameters that joined later keep their own bias correction: single-tensor path
                check(L.dcv_adam_step(ptr(p.data), ptr(g), ptr(s["exp_avg"]), ptr(s["exp_avg_sq"]), p.numel(), self.lr, self.betas[0], self.betas[1],
                                      self.eps, self.weight_decay, s["step"], self.grad_scale, st), "dcv_adam_step")
                continue
            ps.append(p.data.data_ptr()); gs.append(g.data_ptr()); ms.append(s["exp_avg"].data_ptr()); vs.append(s["exp_avg_sq"].data_ptr())
            ns.append(p.numel()); keep.append(g)
        n = len(ps)
        if n:
            arr = lambda xs: (C.c_void_p * n)(*xs)
            check(L.dcv_adam_step_multi(n, arr(ps), arr(gs), arr(ms), arr(vs), (C.c_int64 * n)(*ns), self.lr, self.betas[0], self.betas[1],
                                        self.eps, self.weight_decay, step, self.grad_scale, st), "dcv_adam_step_multi")
        if touched:
            # the kernels wrote through raw pointers: tell autograd (and the packed-weight caches keyed on it) that
            # these tensors changed in place
            torch.autograd.graph.increment_version(touched)


class GradBucket:
    """The gradients that one backward produces and one group of optimiser steps consumes.

    Round 4: the bucket owns ONE persistent flat fp32 buffer (allocated at the first reduction, sized for all members) and every
    member's ``.grad`` IS its slice of it: the post-accumulate-grad hook — which also marks the bucket dirty — re-points ``p.grad`` at
    the slice (copying only when the gradient was produced elsewhere: a parameter used twice in one backward is summed by autograd into
    a tensor of its own; the weight-gradient kernels of single-use parameters write straight into the slice, ``ops._Conv.backward``).
    ``reduce()`` is then one in-place all-reduce (sum) of the buffer — no flatten copy, no re-pointing afterwards.  The slice of a parameter
    whose ``.grad`` is None this backward (the set is the same on every rank: every rank runs the same graph) is zeroed before the collective:
    nobody reads it, but a stale slice would be multiplied by the world size at every reduction and reach inf / NaN inside the communicated
    buffer (RCCL's NaN checks, anomaly tooling).
    `dirty` is set by autograd whenever a member receives a gradient and cleared by `reduce()`."""

    def __init__(self, group=None, bucket_bytes: int = 256 << 20):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bucket_bytes = bucket_bytes
        self.params: List[torch.nn.Parameter] = []
        self.dirty = False
        self.collectives = 0      # counters for tests / bench
        self.reductions = 0
        self.copies = 0           # gradients that had to be copied into their slice (produced outside it)
        self._hooks = []
        self._flat: Optional[torch.Tensor] = None
        self._chunks = []         # [(start, end)] element ranges of the buffer, one collective each (bucket_bytes caps a message)

    def _layout(self):
        """Allocate the flat buffer and hand every member its slice (`p._dcv_grad_slot`, read by the weight-gradient ops)."""
        p0 = self.params[0]
        total, self._chunks, start = 0, [], 0
        for p in self.params:
            if p.dtype != torch.float32:
                raise NativeError("GradBucket: fp32 parameters only")
            n = p.numel()
            if total > start and (total - start + n) * 4 > self.bucket_bytes:
                self._chunks.append((start, total))
                start = total
            p._dcv_grad_off = total
            total += (n + 63) // 64 * 64          # slices start on 256-byte boundaries (16-byte stores of the reduce kernels)
        self._chunks.append((start, total))
        self._flat = torch.zeros(total, dtype=torch.float32, device=p0.device)
        for p in self.params:
            p._dcv_grad_slot = self._flat[p._dcv_grad_off:p._dcv_grad_off + p.numel()].view(p.shape)

    def _mark(self, p):
        self.dirty = True
        if self.world == 1 and not self._force_layout:
            return
        if self._flat is None or self._flat.device != p.device:
            self._layout()
        g, slot = p.grad, p._dcv_grad_slot
        if g is not None and g.data_ptr() != slot.data_ptr():
            slot.copy_(g)
            p.grad = slot
            self.copies += 1

    _force_layout = False

    def add(self, params: Iterable[torch.nn.Parameter]):
        have = {id(p) for p in self.params}
        for p in params:
            if id(p) in have:
                continue              # already a member (a wrapper built around a bucket that was filled by hand)
            self.params.append(p)
            import weakref
            p._dcv_bucket = weakref.ref(self)     # ops.grad_target marks the bucket dirty when it adds a gradient in place (no AccumulateGrad visit, no hook)
            self._hooks.append(p.register_post_accumulate_grad_hook(self._mark))
            self._flat = None         # a new member: the buffer is laid out again at the next gradient (existing .grad slices are copied over)

    @torch.no_grad()
    def reduce(self, force: bool = False):
        """`force`: run the collective even in a world of one (the RCCL smoke test pushes the real 55 MB bucket through the
        `nccl` backend on a single card this way)."""
        if not self.dirty:
            return
        self.dirty = False
        if self.world == 1 and not (force and self.dist.is_initialized()):
            return
        if self._flat is None:      # gradients arrived before the layout existed (world of one, force=True): adopt them now
            self._layout()
        for p in self.params:
            if p.grad is None:
                p._dcv_grad_slot.zero_()      # no gradient this backward: the stale slice must not be summed over the ranks again and again
            elif p.grad.data_ptr() != p._dcv_grad_slot.data_ptr():
                p._dcv_grad_slot.copy_(p.grad)
                p.grad = p._dcv_grad_slot
                self.copies += 1
        self.reductions += 1
        for a, b in self._chunks:
            self.dist.all_reduce(self._flat[a:b], op=self.dist.ReduceOp.SUM, group=self.group)
            self.collectives += 1
        from . import ops
        ops.new_backward_epoch()      # this backward's gradients are in the buffer: the weight-gradient ops may be handed the slices again (also when a plain Adam drives the bucket)


class DataParallelAdam:
    """Wraps an ``Adam``: reduce the shared bucket if a backward has run since the last reduction, then
    step with grad_scale = 1/world."""

    def __init__(self, inner: Adam, bucket: Optional[GradBucket] = None, group=None):
        self.inner = inner
        self.bucket = bucket if bucket is not None else GradBucket(group)
        self.bucket.add(inner.params)
        self.world = self.bucket.world

    @property
    def params(self):
        return self.inner.params

    def zero_grad(self, set_to_none: bool = True):
        self.inner.zero_grad(set_to_none)

    def reduce_gradients(self):
        self.bucket.reduce()

    def step(self):
        self.bucket.reduce()
        self.inner.grad_scale = 1.0 / self.world
        self.inner.step()
        from . import ops
        ops.new_backward_epoch()      # this backward's gradients are consumed: the weight-gradient ops may write into the slices again


def broadcast_module(module: torch.nn.Module, src: int = 0, group=None):
    """Make every rank start from rank `src`'s parameters and buffers."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        ts = list(module.parameters()) + list(module.buffers())
        for t in ts:
            dist.broadcast(t.data, src, group=group)
        torch.autograd.graph.increment_version(ts)   # `.data` writes bypass the version counter


def broadcast_buffers(module: torch.nn.Module, src: int = 0, group=None):
    """BatchNorm running statistics stay per-rank during data-parallel training (each rank = one reference trainer);
    call this before a snapshot / evaluation (trainer.py:70-86, :196) when every rank should save rank `src`'s statistics."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        for b in module.buffers():
            dist.broadcast(b.data, src, group=group)
