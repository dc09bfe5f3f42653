"""GAN losses with the reference's Loss interface (loss.py:9-193), each term one
fused value+gradient HIP kernel (ops.gan_loss)."""
from __future__ import annotations

from abc import ABCMeta, abstractmethod

import torch

from . import ops, util


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
        return ops.gan_loss(y_real, ops.KIND_BCE_ONES) + ops.gan_loss(y_fake, ops.KIND_BCE_ZEROS)

    def compute_gen_loss(self, y_fake_i, y_fake_v, y_fake_g):
        return (ops.gan_loss(y_fake_i, ops.KIND_BCE_ONES) + ops.gan_loss(y_fake_v, ops.KIND_BCE_ONES)
                + ops.gan_loss(y_fake_g, ops.KIND_BCE_ONES))


class HingeLoss(Loss):
    """Hinge for D, softplus(-y) for G; the gradient discriminator's output does
    not enter the generator loss (loss.py:134-193)."""

    def __init__(self):
        super().__init__()
        self.device = util.current_device()

    def compute_dis_loss(self, y_real, y_fake):
        return ops.gan_loss(y_real, ops.KIND_HINGE_REAL) + ops.gan_loss(y_fake, ops.KIND_HINGE_FAKE)

    def compute_gen_loss(self, y_fake_i, y_fake_v, y_fake_g):
        return ops.gan_loss(y_fake_i, ops.KIND_SOFTPLUS_NEG) + ops.gan_loss(y_fake_v, ops.KIND_SOFTPLUS_NEG)
