"""Adam on the HIP kernel, and the data-parallel optimiser wrapper.

``Adam`` has torch.optim.Adam's semantics for the reference's wiring
(train.py:171-176: betas (0.5, 0.999), eps 1e-8, L2 weight decay): parameters
whose ``.grad`` is None are skipped, bias correction uses the per-optimiser step
count.  Only ``.step()`` / ``.zero_grad()`` are needed by a DCVGAN trainer.

``DataParallelAdam`` all-reduces (sum) the gradients of its parameters over the
process group right before the inner step and folds the 1/world factor into the
Adam kernel's ``grad_scale`` — one RCCL collective per optimiser step, no trainer
change.  For the trainer's double ``opt_ggen.step()`` the
reduction is done once per backward (a fresh backward changes the grad tensors'
version counters).
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


class DataParallelAdam:
    """Wraps an ``Adam``: all-reduce(sum) grads over `group`, then step with grad_scale = 1/world."""

    def __init__(self, inner: Adam, group=None, bucket_bytes: int = 64 << 20):
        import torch.distributed as dist
        self.inner, self.group, self.dist = inner, group, dist
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bucket_bytes = bucket_bytes
        self._reduced_versions = None

    @property
    def params(self):
        return self.inner.params

    def zero_grad(self, set_to_none: bool = True):
        self.inner.zero_grad(set_to_none)

    def _grads(self):
        return [p.grad for p in self.inner.params if p.grad is not None]

    @torch.no_grad()
    def reduce_gradients(self):
        grads = self._grads()
        versions = tuple((id(g), g._version) for g in grads)
        if self.world == 1 or versions == self._reduced_versions:
            return
        # one flat bucket per <= bucket_bytes: xGMI rings are per-link bound, so few large messages.  The
        # reduced bucket is not copied back: each parameter's .grad becomes a view into it (one cat kernel
        # and one collective per bucket instead of a copy kernel per parameter).
        owners = [p for p in self.inner.params if p.grad is not None]
        bucket, size = [], 0
        for p in owners + [None]:
            if p is None or (size + p.grad.numel() * 4 > self.bucket_bytes and bucket):
                flat = torch.cat([b.grad.reshape(-1) for b in bucket])
                self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.group)
                off = 0
                for b in bucket:
                    n = b.grad.numel()
                    b.grad = flat[off:off + n].view(b.grad.shape)
                    off += n
                bucket, size = [], 0
            if p is not None:
                bucket.append(p)
                size += p.grad.numel() * 4
        self._reduced_versions = tuple((id(g), g._version) for g in self._grads())

    def step(self):
        self.reduce_gradients()
        self.inner.grad_scale = 1.0 / self.world
        self.inner.step()


def broadcast_module(module: torch.nn.Module, src: int = 0, group=None):
    """Make every rank start from rank `src`'s parameters and buffers."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src, group=group)
