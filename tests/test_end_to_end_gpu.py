"""GPU: bytes in -> parameter update out.  The pieces of the input side are pinned one by one elsewhere (decode kernels byte for byte against
`VideoDataset.__getitem__`, tests/test_dataprep_gpu.py; the prefetcher's hand-over, tests/test_as_trainer_gpu.py); here they run as the chain a training
process runs (/root/reference/src/dataset.py:125-186 -> src/trainer.py:293-297 -> :299-363): pinned host batches of the reference's own mock clips (uint8 RGB
and uint8 depth frames in disk order, tests/golden/dataset_norm.npz `*_in`) -> DevicePrefetcher (side-stream copy, one batch ahead) -> dataprep.decode_* on
the device -> StepRunner with the losses read on the host where trainer.py:326-328,363 reads them — three iterations, against the oracle stepping on the
normalised tensors (the one-line numpy form of dataset.py:127-131, asserted here against the REFERENCE's dataset output `*_out` for the same clips), with the
oracle's draws injected: losses to 1e-3 and every stepcheck bar
(losses 1e-4 against fp64 on this run's own activation pattern, BatchNorm buffers 2e-4, parameter updates and Adam moments 1e-3)."""
import itertools

import numpy as np
import pytest
import torch

from oracle import dcvgan_oracle as O
from oracle import stepcheck as SC
from tests import goldenio as G

pytestmark = pytest.mark.gpu


def test_bytes_to_update_three_iterations():
    from dcvgan_amd import dataprep, layers, native, trainer
    from dcvgan_amd.configs import CONFIGS
    from dcvgan_amd.rng import InjectedRng
    native.lib()
    dev = torch.device("cuda:0")
    fx = G.load("dataset_norm.npz")
    B = 3
    color_u8 = np.stack([fx[f"mock/{i}/color_in"] for i in range(B)])            # (B, T, H, W, 3) uint8, as read from disk
    depth_u8 = np.stack([fx[f"mock/{i}/depth_in"] for i in range(B)])            # (B, T, H, W, 1) uint8
    # dataset.py:127-131,158-168 as one line of numpy, pinned HERE against what the reference's dataset class returned for these very clips ...
    norm = lambda u8: torch.from_numpy(np.ascontiguousarray(u8.transpose(0, 4, 1, 2, 3)).astype(np.float32) / 127.5 - 1.0)
    assert torch.equal(norm(color_u8), torch.from_numpy(np.stack([fx[f"mock/{i}/color_out"] for i in range(B)])))
    assert torch.equal(norm(depth_u8), torch.from_numpy(np.stack([fx[f"mock/{i}/depth_out"] for i in range(B)])))
    # ... and then applied to the same clips with a texture on them: the mock clips are solid colours (test_dataset.py:63-95), on which the discriminators'
    # BatchNorm variances of the real batch are ~0 and their RELATIVE error is the cancellation's, not the path's (measured on the bare clips: running_var 4.5e-4
    # against the 2e-4 bar, everything else inside its bar)
    tex = np.random.default_rng(7)
    color_u8 = (color_u8.astype(np.int32) + tex.integers(0, 97, size=color_u8.shape)).astype(np.uint8)      # wraps mod 256, as bytes do
    depth_u8 = (depth_u8.astype(np.int32) + tex.integers(0, 97, size=depth_u8.shape)).astype(np.uint8)
    xc_ref, xg_ref = norm(color_u8), norm(depth_u8)
    cfg = CONFIGS["isogd-depth"].scaled(batchsize=B, width_div=8)
    ts = (5, 11, 2)

    torch.manual_seed(cfg.seed)
    models = trainer.build_models(cfg, torch.device("cpu"))
    states = {n: {k: v.detach().clone() for k, v in m.state_dict().items()} for n, m in models.items()}
    torch.manual_seed(2)
    so = O.StepOracle(cfg, states)                 # the reference's arithmetic (fp32 torch CPU) on the reference's tensors
    want = [so.step(xc_ref, xg_ref, t) for t in ts]
    for m in models.values():
        m.to(dev)
        for sub in m.modules():
            if hasattr(sub, "device"):
                sub.device = dev
    rng = InjectedRng(so.rng.log)
    for m in models.values():
        m._rng = rng
    opts = trainer.build_optimizers(cfg, models)
    runner = trainer.StepRunner(cfg, models, opts, trainer.build_loss(cfg), sync_losses=True)
    forced = SC.ForcedStepOracle(cfg, so.rng.log)

    # the loader's side: two pinned host batches handed round (DataLoader(pin_memory=True), train.py:101-109), copied one iteration ahead
    pinned = [{"color": torch.from_numpy(color_u8).clone().pin_memory(), "depth": torch.from_numpy(depth_u8).clone().pin_memory()} for _ in range(2)]
    feed = dataprep.DevicePrefetcher(itertools.cycle(pinned), dev)

    class FromBytes:
        """What checked_iteration drives: every step takes the NEXT batch off the prefetcher and decodes it on the device."""
        decoded = []

        @property
        def iteration(self):
            return runner.iteration

        def step(self, _xc, _xg, t):
            b = next(feed)
            assert b["color"].is_cuda and b["color"].dtype == torch.uint8
            xc, xg = dataprep.decode_color(b["color"]), dataprep.decode_depth(b["depth"])
            self.decoded.append((xc, xg))
            return runner.step(xc, xg, t)

    drv = FromBytes()
    worst = 0.0
    for it, t in enumerate(ts):
        res = SC.checked_iteration(drv, models, opts, forced, layers, None, None, xc_ref, xg_ref, t, cfg.lr)
        for k, v in want[it].items():
            assert abs(res["losses"][k] - v) < 1e-3 * max(1.0, abs(v)), (it, k, res["losses"][k], v)
        w, _, _ = SC.assert_iteration(res, cfg.lr, f"bytes-to-update it {it + 1}")
        worst = max(worst, w)
        assert all(r["calls"] == (2 if r["model"] == "ggen" else 1) for r in res["rows"])
    assert rng.pos == len(so.rng.log)                                            # the three iterations consumed exactly the oracle's draws
    xc_d, xg_d = drv.decoded[-1]
    assert torch.equal(xc_d.cpu(), xc_ref) and torch.equal(xg_d.cpu(), xg_ref)  # ... from exactly the reference's normalised tensors
