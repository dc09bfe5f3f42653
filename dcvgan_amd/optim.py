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


def _copy_into(slot: torch.Tensor, g: torch.Tensor) -> None:
    """slot <- g with the library's strided copy (dcv_axpby) on the device; torch's copy on the host (gloo rehearsals)."""
    if slot.is_cuda and g.dtype == torch.float32 and g.dim() in (2, 4, 5):
        from . import ops
        ops._axpby(g, 1.0, None, 0.0, slot)
    elif slot.is_cuda and g.dtype == torch.float32:
        from . import ops
        ops._axpby(g.reshape(1, -1) if g.is_contiguous() else g.contiguous().reshape(1, -1), 1.0, None, 0.0, slot.reshape(1, -1))
    else:
        slot.copy_(g)


def _zero_slot(p) -> None:
    """p's gradient slice <- 0.  On the device: 0 * p (the parameter is finite where the stale slice need not be) through dcv_axpby, so that no torch fill
    kernel runs beside the library's (DESIGN: packed-FP32 code of torch's elementwise kernels is outside the build's control)."""
    slot = p._dcv_grad_slot
    if slot.is_cuda:
        from . import ops
        ops._axpby(p.detach().reshape(1, -1), 0.0, None, 0.0, slot.reshape(1, -1))
    else:
        slot.zero_()


class GradBucket:
    """The gradients that one backward produces and one group of optimiser steps consumes.

    Round 4: the bucket owns ONE persistent flat fp32 buffer (allocated at the first reduction, sized for all members) and every
    member's ``.grad`` IS its slice of it: the post-accumulate-grad hook — which also marks the bucket dirty — re-points ``p.grad`` at
    the slice (copying only when the gradient was produced elsewhere; the weight-gradient kernels write straight into the slice,
    ``ops._Conv.backward``).  ``reduce()`` is then in-place all-reduces (sum) of the buffer — no flatten copy, no re-pointing afterwards.  The slice of a parameter
    whose ``.grad`` is None this backward (the set is the same on every rank: every rank runs the same graph) is zeroed before the collective:
    nobody reads it, but a stale slice would be multiplied by the world size at every reduction and reach inf / NaN inside the communicated
    buffer (RCCL's NaN checks, anomaly tooling).
    `dirty` is set by autograd whenever a member receives a gradient and cleared by `reduce()`.

    Round 5, ``overlap=True``: the buffer is cut into CHUNKS — one per ``add()`` call, i.e. per model (small neighbours merged up to ``merge_bytes``, large ones cut at
    ``bucket_bytes``) — and a chunk's collective is launched from the hook of the LAST gradient of that chunk to land, on a communication stream behind the events of
    the streams that produced the chunk's gradients, while the backward of the other models is still running (G phase: cgen's 40 MB are reduced under ggen's backward);
    ``reduce()`` then only waits.  Which gradient is a chunk's last one is LEARNED: the set and order of arrivals of the previous backward of this bucket (the same graph
    every iteration); a backward whose arrivals differ from the record simply is not overlapped (the step reduces synchronously, chunk by chunk — the same collectives on
    the same ranges, so both ways give the same bits).  Safety: a chunk is only launched early when none of its members held a gradient at the start of this backward; a
    gradient that arrives for a chunk that is already in flight, or is added in place to it (``ops.grad_target``) before the step has consumed it, raises; the weight-gradient
    op orders itself behind a collective still reading the slice it is about to overwrite (``before_slot_write``)."""

    def __init__(self, group=None, bucket_bytes: int = 256 << 20, overlap: bool = False, merge_bytes: int = 4 << 20):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bucket_bytes, self.merge_bytes, self.overlap = bucket_bytes, merge_bytes, bool(overlap)
        self.params: List[torch.nn.Parameter] = []
        self.dirty = False
        self.collectives = 0      # counters for tests / bench
        self.reductions = 0
        self.copies = 0           # gradients that had to be copied into their slice (produced outside it)
        self.early = 0            # collectives launched from a hook, during the backward
        self._hooks = []
        self._groups = []         # member lists, one per add()
        self._flat: Optional[torch.Tensor] = None
        self._chunks = []         # _Chunk objects: element ranges of the buffer, one collective each
        self._comm = None
        self.timed = None         # a list: (start event, end event, bytes) of every collective launched from reduce() (bench.py's per-phase collective time)

    class _Chunk:
        __slots__ = ("a", "b", "members", "record", "arrived", "events", "work", "task", "fresh")

        def __init__(self, a, b, members):
            self.a, self.b, self.members = a, b, members
            self.record = None        # (frozenset of member ids that received a gradient, id of the last one to arrive) in the previous backward
            self.arrived, self.events, self.work, self.task, self.fresh = [], [], None, -2, True

    def _layout(self):
        """Allocate the flat buffer, hand every member its slice (`p._dcv_grad_slot`, read by the weight-gradient ops) and cut the chunks."""
        p0 = self.params[0]
        total = 0
        spans = []                # (first element, one past the last, members) per add() group
        groups = self._groups if (self.overlap and self._groups) else [self.params]      # without overlap the bucket is ONE message (cut only at bucket_bytes)
        for members in groups:
            g0 = total
            for p in members:
                if p.dtype != torch.float32:
                    raise NativeError("GradBucket: fp32 parameters only")
                p._dcv_grad_off = total
                total += (p.numel() + 63) // 64 * 64      # slices start on 256-byte boundaries (16-byte stores of the reduce kernels)
            spans.append((g0, total, list(members)))
        # chunks: small neighbouring groups merged, a group beyond bucket_bytes cut at member boundaries
        self._chunks = []
        cur_a, cur_members = 0, []
        for (a, b, members) in spans:
            cur_end = cur_members[-1]._dcv_grad_off + (cur_members[-1].numel() + 63) // 64 * 64 if cur_members else cur_a      # one past the open chunk's last member
            if cur_members and ((b - cur_a) * 4 > self.bucket_bytes or ((cur_end - cur_a) * 4 >= self.merge_bytes and (b - a) * 4 >= self.merge_bytes)):
                self._chunks.append(GradBucket._Chunk(cur_a, a, cur_members))
                cur_a, cur_members = a, []
            for p in members:
                if cur_members and (p._dcv_grad_off + (p.numel() + 63) // 64 * 64 - cur_a) * 4 > self.bucket_bytes:
                    self._chunks.append(GradBucket._Chunk(cur_a, p._dcv_grad_off, cur_members))
                    cur_a, cur_members = p._dcv_grad_off, []
                cur_members.append(p)
        if cur_members:
            self._chunks.append(GradBucket._Chunk(cur_a, total, cur_members))
        self._flat = torch.zeros(total, dtype=torch.float32, device=p0.device)
        for c in self._chunks:
            for p in c.members:
                p._dcv_grad_slot = self._flat[p._dcv_grad_off:p._dcv_grad_off + p.numel()].view(p.shape)
                p._dcv_chunk = c

    def _mark(self, p):
        self.dirty = True
        if self.world == 1 and not self._force_layout:
            return
        if self._flat is None or self._flat.device != p.device:
            self._layout()
        g, slot = p.grad, p._dcv_grad_slot
        if g is not None and g.data_ptr() != slot.data_ptr():
            _copy_into(slot, g)
            p.grad = slot
            self.copies += 1
        if self.overlap:
            self._arrival(p)

    _force_layout = False

    # ---- overlap ------------------------------------------------------------------------------------------------------------------
    def _arrival(self, p):
        c = p._dcv_chunk
        task = torch._C._current_graph_task_id()
        if c.task != task:                      # first gradient of this chunk in a new backward
            if c.work is not None:
                raise NativeError("GradBucket(overlap=True): a new backward delivers gradients to a chunk whose collective of the previous backward has not been "
                                  "consumed by an optimiser step (gradient accumulation over several backwards needs overlap=False)")
            c.task, c.arrived, c.events = task, [], []
            # early launch only for a backward that STARTS from empty gradients (zero_grad before it): at the first arrival every other member must still be without one
            c.fresh = all(q.grad is None for q in c.members if q is not p)
        elif c.work is not None:
            raise NativeError("GradBucket(overlap=True): a gradient arrived for a chunk that is already being reduced (the arrival order changed between backwards)")
        c.arrived.append(id(p))
        if p.is_cuda:
            e = torch.cuda.Event()
            e.record(torch.cuda.current_stream(p.device))
            c.events.append(e)
        rec = c.record
        if rec is not None and id(p) == rec[1] and len(c.arrived) == len(rec[0]) and frozenset(c.arrived) == rec[0] and c.fresh \
                and all(q.grad is None for q in c.members if id(q) not in rec[0]):      # (a member that keeps an old gradient and gets none now would be summed again)
            self._launch(c, early=True)

    def note_inplace(self, p):
        """ops.grad_target adds a gradient to p.grad in place (no AccumulateGrad visit, no hook): the bucket is dirty; a chunk that was reduced early and not consumed yet
        must not be added to, and a chunk that receives in-place additions is not 'fresh' (its collective waits for the step)."""
        self.dirty = True
        c = getattr(p, "_dcv_chunk", None)
        if c is None:
            return
        if c.work is not None:
            raise NativeError("GradBucket(overlap=True): a gradient is being added in place to a chunk whose early collective has not been consumed by an optimiser step")
        c.fresh = False

    def before_slot_write(self, p):
        """The weight-gradient op is about to write p's slice: order the current stream behind a collective that is still reading it."""
        c = getattr(p, "_dcv_chunk", None)
        if c is not None and c.work is not None and p.is_cuda:
            c.work.wait()

    def _launch(self, c, early):
        flat = self._flat
        if flat.is_cuda:
            if self._comm is None:
                self._comm = torch.cuda.Stream(flat.device)
            cur = torch.cuda.current_stream(flat.device)
            if early:
                with torch.cuda.stream(self._comm):
                    for e in c.events:
                        self._comm.wait_event(e)
                    for q in c.members:          # slices of members without a gradient this backward: zeroed before they are summed over the ranks
                        if q.grad is None:
                            _zero_slot(q)
                    c.work = self.dist.all_reduce(flat[c.a:c.b], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
                flat.record_stream(self._comm)
            else:
                if self.timed is not None:      # bench.py: HIP events around the collective as the compute stream sees it (issue ... data reduced)
                    e0 = torch.cuda.Event(enable_timing=True); e0.record(cur)
                c.work = self.dist.all_reduce(flat[c.a:c.b], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
                if self.timed is not None:
                    c.work.wait()
                    e1 = torch.cuda.Event(enable_timing=True); e1.record(cur)
                    self.timed.append((e0, e1, (c.b - c.a) * 4))
        else:
            self.dist.all_reduce(flat[c.a:c.b], op=self.dist.ReduceOp.SUM, group=self.group)
            c.work = True                        # CPU (gloo rehearsal): done on return
        self.collectives += 1
        if early:
            self.early += 1

    def add(self, params: Iterable[torch.nn.Parameter]):
        have = {id(p) for p in self.params}
        members = []
        for p in params:
            if id(p) in have:
                continue              # already a member (a wrapper built around a bucket that was filled by hand)
            self.params.append(p)
            members.append(p)
            import weakref
            p._dcv_bucket = weakref.ref(self)     # ops.grad_target marks the bucket dirty when it adds a gradient in place (no AccumulateGrad visit, no hook)
            self._hooks.append(p.register_post_accumulate_grad_hook(self._mark))
            self._flat = None         # a new member: the buffer is laid out again at the next gradient (existing .grad slices are copied over)
        if members:
            self._groups.append(members)

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
        self.reductions += 1
        for c in self._chunks:
            if c.work is None:      # not launched during the backward: now, on the current stream
                for p in c.members:
                    if p.grad is None:
                        _zero_slot(p)                 # no gradient this backward: the stale slice must not be summed over the ranks again and again
                    elif p.grad.data_ptr() != p._dcv_grad_slot.data_ptr():
                        _copy_into(p._dcv_grad_slot, p.grad)
                        p.grad = p._dcv_grad_slot
                        self.copies += 1
                self._launch(c, early=False)
        for c in self._chunks:      # the optimiser steps that follow read the reduced slices on the current stream
            if c.work is not None and c.work is not True:
                c.work.wait()
            # remember this backward's arrivals: the next backward of the same graph launches the chunk from its last arrival's hook
            c.record = (frozenset(c.arrived), c.arrived[-1]) if (self.overlap and c.arrived) else c.record
            c.work, c.arrived, c.events, c.task, c.fresh = None, [], [], -2, True
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
