"""Random sources for latents, Noise layers and Dropout2d masks.

Production: ``PhiloxRng`` — counter-based draws made by the HIP kernels
(dcv_normal_fill / dcv_noise_add / dcv_dropout_mask).  Its seed follows
``torch.manual_seed`` (the reference seeds through torch, train.py:31-45); every
draw advances a stream counter, so runs are reproducible.

Parity tests: ``InjectedRng`` replays tensors recorded from the CPU oracle, because
CPU and device generators can never produce the same stream.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from . import ops


class PhiloxRng:
    def __init__(self, seed: Optional[int] = None):
        self._fixed_seed = seed
        self._seed_seen = None
        self._counter = 0

    def _next(self) -> Tuple[int, int]:
        seed = self._fixed_seed if self._fixed_seed is not None else torch.initial_seed()
        if seed != self._seed_seen:  # (re)seeded through torch.manual_seed
            self._seed_seen, self._counter = seed, 0
        self._counter += 1
        return seed & 0xFFFFFFFFFFFFFFFF, self._counter

    def normal(self, shape: Sequence[int], device) -> torch.Tensor:
        seed, off = self._next()
        return ops.normal(shape, device, seed, off)

    def normal_many(self, count: int, shape: Sequence[int], device) -> torch.Tensor:
        """`count` consecutive draws of one shape, stacked on a new first axis: the same values and the same stream position as `count` calls of normal()."""
        if count <= 0:
            return torch.empty((0,) + tuple(shape), dtype=torch.float32, device=device)
        seed, off = self._next()
        self._counter += count - 1
        return ops.normal_many(count, shape, device, seed, off)

    def noise_add(self, x: torch.Tensor, sigma: float) -> torch.Tensor:
        seed, off = self._next()
        return ops.noise_add(x, sigma, None, seed, off)

    def dropout2d_mask(self, n: int, c: int, p: float, device) -> torch.Tensor:
        seed, off = self._next()
        return ops.dropout2d_mask(n, c, p, device, seed, off)


class InjectedRng:
    """Replays a recorded draw log [(kind, tensor), ...] in order, on `device`."""

    def __init__(self, log: List[Tuple[str, torch.Tensor]]):
        self.log, self.pos = log, 0

    def _take(self, kind, shape, device):
        k, t = self.log[self.pos]
        self.pos += 1
        if k != kind or tuple(t.shape) != tuple(shape):
            raise RuntimeError(f"injected draw {self.pos - 1}: have {k}{tuple(t.shape)}, asked {kind}{tuple(shape)}")
        return t.to(device)

    def normal(self, shape, device):
        return self._take("normal", tuple(shape), device)

    def normal_many(self, count, shape, device):
        return torch.stack([self.normal(shape, device) for _ in range(count)], 0)

    def noise_add(self, x, sigma):
        return ops.noise_add(x, sigma, self._take("normal", tuple(x.shape), x.device))

    def dropout2d_mask(self, n, c, p, device):
        return self._take("dropout2d", (n, c, 1, 1), device)


_default = PhiloxRng()


def default_rng() -> PhiloxRng:
    return _default
