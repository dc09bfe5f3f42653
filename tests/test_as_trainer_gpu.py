"""GPU: the two objects an UNCHANGED reference trainer meets when it drives these modules: loss.HostMirroredLoss — the per-object `cpu`
attribute behind `loss.cpu().item()` (trainer.py:326-328,363) — and dataprep.DevicePrefetcher — the batch already on the device when the trainer says
`.to(self.device)` (trainer.py:293-297)."""
import gc

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _cfg_models(B=2, div=8):
    from dcvgan_amd import trainer
    from dcvgan_amd.configs import CONFIGS
    from dcvgan_amd.rng import PhiloxRng
    cfg = CONFIGS["surreal-depth1"].scaled(batchsize=B, width_div=div)      # num_gen_update = 2: every second iteration's D phase is gated off
    torch.manual_seed(3)
    models = trainer.build_models(cfg, DEV)
    r = PhiloxRng(5)
    for m in models.values():
        m._rng = r
    return cfg, models


def test_host_mirror_equals_tensor_cpu_and_survives_the_trainers_uses():
    from dcvgan_amd import loss as L
    from dcvgan_amd import trainer
    cfg, models = _cfg_models()
    lossf = trainer.build_loss(cfg)
    y_r = torch.randn(2, 4, 4, device=DEV, requires_grad=True); y_f = torch.randn(2, 4, 4, device=DEV, requires_grad=True)
    l = lossf.compute_dis_loss(y_r, y_f)
    assert "cpu" in l.__dict__, "compute_dis_loss returns a host-mirrored tensor"
    mirrored = l.cpu().item()
    plain = torch.Tensor.cpu(l).item()                      # the ordinary stream-ordered copy of the same tensor
    assert mirrored == plain
    assert l.cpu().item() == plain                          # a second read
    assert torch.Tensor.cpu(l, memory_format=torch.preserve_format).item() == plain and l.cpu(memory_format=torch.preserve_format).item() == plain   # arguments: torch's own path
    # trainer.py:315: loss_dis = loss_idis + loss_vdis + loss_gdis -> a NEW tensor: no mirror, ordinary semantics, and backward works through it
    l2 = lossf.compute_dis_loss(y_r * 2, y_f)
    s = l + l2
    assert "cpu" not in s.__dict__
    assert abs(s.cpu().item() - (plain + torch.Tensor.cpu(l2).item())) <= 1e-6 * max(1.0, abs(plain))
    s.backward()
    assert y_r.grad is not None and torch.isfinite(y_r.grad).all()
    # trainer.py:324: a gated-off phase calls loss_dis.detach_() on the sum; :326-328 still read the three mirrored members afterwards
    l3 = lossf.compute_dis_loss(y_r.detach().requires_grad_(True), y_f.detach())
    want = torch.Tensor.cpu(l3).item()
    tot = l3 + l3
    tot.detach_()
    l3.detach_()
    assert l3.cpu().item() == want and not l3.requires_grad
    # generator loss (trainer.py:352,363)
    lg = lossf.compute_gen_loss(torch.randn(2, 4, 4, device=DEV), torch.randn(2, 4, 4, 4, device=DEV), torch.randn(2, 3, 4, 4, device=DEV))
    assert lg.cpu().item() == torch.Tensor.cpu(lg).item()
    assert isinstance(L.HostMirroredLoss._side, dict)


def test_gated_off_iterations_do_not_leak_device_memory():
    """The first form of the mirror closed a reference cycle through the tensor's `cpu` attribute: the loss — and on gated-off iterations the whole un-run
    autograd graph of the phase — lived until Python's cycle collector ran (tools/soak.py: GB-sized steps).  20 iterations whose D phase is gated off on
    every second one, collector disabled: allocated device memory after iteration 6 and after iteration 20 must agree."""
    from dcvgan_amd import trainer
    cfg, models = _cfg_models(B=2, div=8)
    opts = trainer.build_optimizers(cfg, models)
    runner = trainer.StepRunner(cfg, models, opts, trainer.build_loss(cfg), sync_losses=True)
    xc = torch.rand(2, 3, 16, 64, 64, device=DEV) * 2 - 1; xg = torch.rand(2, 1, 16, 64, 64, device=DEV) * 2 - 1
    gc.collect(); gc.disable()
    try:
        marks = []
        for i in range(20):
            out = runner.step(xc, xg, i % 16)
            assert all(v == v for v in out.values())
            if i in (5, 19):
                torch.cuda.synchronize()
                marks.append(torch.cuda.memory_allocated(DEV))
        assert marks[1] <= marks[0] + (1 << 20), marks          # flat (the caching allocator's live bytes, not reserved ones)
    finally:
        gc.enable()


def test_device_prefetcher_hands_out_the_batches_bit_for_bit_also_to_a_lane_stream():
    from dcvgan_amd.dataprep import DevicePrefetcher
    g = torch.Generator().manual_seed(1)
    batches = [{"color": torch.rand(2, 3, 16, 64, 64, generator=g).pin_memory(), "depth": torch.rand(2, 1, 16, 64, 64, generator=g).pin_memory(), "idx": i} for i in range(5)]
    feed = DevicePrefetcher(iter(batches), DEV)
    lane = torch.cuda.Stream(DEV)
    got = []
    for i, b in enumerate(feed):
        assert b["color"].is_cuda and b["idx"] == i
        assert b["color"].to(DEV) is b["color"]                 # the trainer's .to(self.device) is a no-op (trainer.py:293-297)
        if i % 2:                                               # consumer on another stream (the discriminators' lanes): it must order itself behind the current stream, as the step does
            lane.wait_stream(torch.cuda.current_stream(DEV))
            with torch.cuda.stream(lane):
                s = b["color"].double().sum() + b["depth"].double().sum()
                b["color"].record_stream(lane); b["depth"].record_stream(lane)
            torch.cuda.current_stream(DEV).wait_stream(lane)
        else:
            s = b["color"].double().sum() + b["depth"].double().sum()
        got.append((b["color"].clone(), b["depth"].clone(), s))
    assert len(got) == 5
    torch.cuda.synchronize()
    for (c, d, s), src in zip(got, batches):
        assert torch.equal(c.cpu(), src["color"]) and torch.equal(d.cpu(), src["depth"])
        assert abs(float(s) - float(src["color"].double().sum() + src["depth"].double().sum())) < 1e-6
    # tuples work as well, and an exhausted source ends the iteration
    feed2 = DevicePrefetcher(iter([(batches[0]["color"], batches[0]["depth"])]), DEV)
    (c0, d0), = list(feed2)
    assert torch.equal(c0.cpu(), batches[0]["color"])
