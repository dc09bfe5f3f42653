"""GPU: an iteration must not leave device tensors in reference cycles, and its memory must be flat.  A cycle is only freed when Python's cyclic collector gets to
it; until then its tensors stay allocated.  Round 6 had one: BnLink.out -> the BatchNorm output -> its grad_fn (the node's ctx) -> ctx.link -> BnLink kept UpBlock 5's
BatchNorm input and output (1.17 GB each at B = 70, twice per iteration) alive between collections — tools/soak.py at the bench batch: 37 GB allocated after 5
iterations, 146 GB after 150, flat at 1.9 GB with the cycle broken.  Full width (the deferred-BatchNorm path needs UpBlock 5's 64 channels), small batch."""
import gc

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["fp32", "bf16cl"])
def test_no_device_tensor_in_a_reference_cycle_and_flat_memory(mode):
    from dcvgan_amd import native, ops_cl, trainer
    from dcvgan_amd.configs import CONFIGS
    native.lib()
    dev = torch.device("cuda:0")
    B = 2
    cfg = CONFIGS["isogd-depth"].scaled(batchsize=B)
    if mode == "bf16cl":
        ops_cl.enable(True)
    try:
        torch.manual_seed(1)
        models = trainer.build_models(cfg, dev)
        runner = trainer.StepRunner(cfg, models, trainer.build_optimizers(cfg, models), trainer.build_loss(cfg))
        g = torch.Generator().manual_seed(2)
        xc = (torch.rand(B, 3, 16, 64, 64, generator=g) * 2 - 1).to(dev)
        xg = (torch.rand(B, cfg.channel, 16, 64, 64, generator=g) * 2 - 1).to(dev)
        for i in range(2):
            runner.step(xc, xg, i)
        torch.cuda.synchronize()
        gc.collect()
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            torch.empty(1, device=dev)                      # (an allocation lets the allocator take back blocks freed on other streams)
            base = torch.cuda.memory_allocated()
            out = None
            for i in range(3):
                out = runner.step(xc, xg, 2 + i)
            torch.cuda.synchronize()
            torch.empty(1, device=dev)
            held = torch.cuda.memory_allocated() - base
            gc.set_debug(gc.DEBUG_SAVEALL)
            gc.collect()
            stuck = [o for o in gc.garbage if torch.is_tensor(o) and o.is_cuda]
            desc = [(tuple(t.shape), type(t.grad_fn).__name__) for t in stuck[:6]]
        finally:
            gc.set_debug(0)
            gc.garbage.clear()
            if was_enabled:
                gc.enable()
        assert not stuck, f"{len(stuck)} device tensors were only reachable from reference cycles: {desc}"
        assert abs(held) < 4e6, f"{held / 1e6:.1f} MB more allocated after three further iterations"
        assert all(float(v) == float(v) for v in out.values())
    finally:
        if mode == "bf16cl":
            ops_cl.enable(False)
