"""GAN losses with the reference's Loss interface (loss.py:9-193), each term one
fused value+gradient HIP kernel, the terms of a loss summed on the device (ops.gan_loss_sum)."""
from __future__ import annotations

from abc import ABCMeta, abstractmethod

import torch

from . import ops, util


class HostMirroredLoss:
    """Gives the 0-d device tensor a loss method returns an asynchronous host copy of its value.

    The reference trainer reads every loss with ``loss.cpu().item()`` AFTER the backward pass and the optimiser steps of the same phase
    (trainer.py:318-328, 355-363).  A plain ``.cpu()`` is a stream-ordered copy: the host would wait for that whole backward + Adam and the
    GPU would then sit idle while the host enqueues the next phase (measured: +2.2 % per iteration).  The value itself exists as soon as the
    loss kernels have run, so it is copied to pinned host memory right then, on a side stream behind an event, and the tensor OBJECT gets a
    ``cpu`` attribute that waits for THAT event only and returns the host value.  The tensor stays a plain torch.Tensor — ``+``, ``.backward()``,
    ``.detach_()`` (trainer.py:324,361) and autograd see nothing unusual; a tensor computed from it carries no mirror and copies the ordinary way."""

    _side = {}

    @staticmethod
    def wrap(t: torch.Tensor) -> torch.Tensor:
        if not t.is_cuda:
            return t
        dev = t.device
        side = HostMirroredLoss._side.get(dev.index)
        if side is None:
            side = HostMirroredLoss._side[dev.index] = torch.cuda.Stream(dev)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(dev))
        host = torch.empty((), dtype=t.dtype, pin_memory=True)
        src = t.detach()
        with torch.cuda.stream(side):
            side.wait_event(ready)
            host.copy_(src, non_blocking=True)
            done = torch.cuda.Event()
            done.record(side)
        src.record_stream(side)
        # the closure must not own `t` (t.__dict__ -> closure -> t would be a reference cycle: the loss tensor, and with it the whole autograd graph of a phase
        # whose backward was gated off, would live until Python's cycle collector happens to run — measured as GB-sized steps in device memory)
        import weakref
        wt = weakref.ref(t)

        def cpu(*args, **kwargs):
            if args or kwargs:
                return torch.Tensor.cpu(wt(), *args, **kwargs)
            done.synchronize()
            return host.clone()
        t.cpu = cpu          # instance attribute: shadows torch.Tensor.cpu for this object only
        return t


class Loss(object):
    __metaclass__ = ABCMeta

    @abstractmethod
    def compute_dis_loss(self, y_real: torch.Tensor, y_fake: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError()

    @abstractmethod
    def compute_gen_loss(self, y_fake_i: torch.Tensor, y_fake_v: torch.Tensor, y_fake_g: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError()


class AdversarialLoss(Loss):
    """BCE-with-logits, sum / numel per term (loss.py:64-131)."""

    def __init__(self):
        super().__init__()
        self.device = util.current_device()

    def compute_dis_loss(self, y_real, y_fake):
        return HostMirroredLoss.wrap(ops.gan_loss_sum([(y_real, ops.KIND_BCE_ONES), (y_fake, ops.KIND_BCE_ZEROS)]))

    def compute_gen_loss(self, y_fake_i, y_fake_v, y_fake_g):
        return HostMirroredLoss.wrap(ops.gan_loss_sum([(y_fake_i, ops.KIND_BCE_ONES), (y_fake_v, ops.KIND_BCE_ONES), (y_fake_g, ops.KIND_BCE_ONES)]))


class HingeLoss(Loss):
    """Hinge for D, softplus(-y) for G; the gradient discriminator's output does
    not enter the generator loss (loss.py:134-193)."""

    def __init__(self):
        super().__init__()
        self.device = util.current_device()

    def compute_dis_loss(self, y_real, y_fake):
        return HostMirroredLoss.wrap(ops.gan_loss_sum([(y_real, ops.KIND_HINGE_REAL), (y_fake, ops.KIND_HINGE_FAKE)]))

    def compute_gen_loss(self, y_fake_i, y_fake_v, y_fake_g):
        return HostMirroredLoss.wrap(ops.gan_loss_sum([(y_fake_i, ops.KIND_SOFTPLUS_NEG), (y_fake_v, ops.KIND_SOFTPLUS_NEG)]))
