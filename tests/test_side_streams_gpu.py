"""GPU: the discriminators' side streams must not change one bit of a result, in any precision mode and in mixed ones.

Round 4 found (profiles/r04_packed_fp32/SUMMARY.txt) that packed-FP32 VALU code in the small kernels (thin data gradients, BatchNorm, elementwise) gave wrong
upper lanes now and then while ANOTHER stream's bf16-MFMA waves shared their CUs: iterations in the bf16-product and f32x6 modes were not repeatable on the side
streams (6-8 of 12 runs off), and a module left at fp32 beside a bf16 one was hit in most trials.  The library is built without those instructions now; these
tests are the schedule that showed it (the fp32 default was never affected and has had the same test since round 2, tests/test_fullwidth_gpu.py);
tests/test_abi_cpu.py::test_library_has_no_packed_fp32_arithmetic looks at the shipped code object itself."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["bf16", "f32x6"])
def test_iterations_are_bitwise_repeatable_on_side_streams(mode):
    from dcvgan_amd import native, trainer
    from dcvgan_amd.configs import CONFIGS
    from dcvgan_amd.rng import PhiloxRng
    native.lib()
    native.set_precision(mode)
    try:
        dev = torch.device("cuda:0")
        B = 16
        cfg = CONFIGS["surreal-depth1"].scaled(batchsize=B)
        g = torch.Generator().manual_seed(3)
        xc = (torch.rand(B, 3, 16, 64, 64, generator=g) * 2 - 1).to(dev); xg = (torch.rand(B, cfg.channel, 16, 64, 64, generator=g) * 2 - 1).to(dev)

        def run(side_streams):
            torch.manual_seed(11)
            models = trainer.build_models(cfg, dev)
            r = PhiloxRng(5)
            for m in models.values():
                m._rng = r
            runner = trainer.StepRunner(cfg, models, trainer.build_optimizers(cfg, models), trainer.build_loss(cfg), sync_losses=True, side_streams=side_streams)
            losses = [runner.step(xc, xg, 2 + i) for i in range(2)]
            return losses, torch.cat([v.detach().float().reshape(-1) for m in models.values() for v in m.state_dict().values()]).cpu()

        l0, p0 = run(False)
        for i in range(6):
            l, p = run(True)
            assert l == l0, (mode, i, l, l0)
            assert torch.equal(p, p0), (mode, i, int((p != p0).sum()))
    finally:
        native.set_precision("fp32")


def test_a_module_at_another_precision_on_the_next_stream_changes_nothing():
    """vdis at bf16 products, gdis at fp32, forward + backward on two streams: gdis' gradients (and the summed input gradients) against the one-stream pass.
    With packed-FP32 code in thin_quad_kernel this differed in ~85 % of the trials."""
    from dcvgan_amd import native, trainer, util
    from dcvgan_amd.configs import CONFIGS
    from dcvgan_amd.rng import PhiloxRng
    native.lib()
    native.set_precision("fp32")
    dev = torch.device("cuda:0")
    B = 16
    cfg = CONFIGS["surreal-depth1"].scaled(batchsize=B)
    g = torch.Generator().manual_seed(3)
    xc = (torch.rand(B, 3, 16, 64, 64, generator=g) * 2 - 1).to(dev).requires_grad_(True); xg = (torch.rand(B, cfg.channel, 16, 64, 64, generator=g) * 2 - 1).to(dev).requires_grad_(True)
    torch.manual_seed(11)
    models = trainer.build_models(cfg, dev)
    rng = PhiloxRng(5)
    for m in models.values():
        m._rng = rng
        m.train()
    util.set_precision(models["vdis"], "bf16")
    names = ("vdis", "gdis")

    def grads():
        out = {f"{k}.{n}": p.grad.detach().clone() for k in names for n, p in models[k].named_parameters() if p.grad is not None}
        out["xg"] = xg.grad.detach().clone(); out["xc"] = xc.grad.detach().clone()
        for k in names:
            models[k].zero_grad()
        xg.grad = None; xc.grad = None
        return out

    state = (rng._seed_seen, rng._counter)      # every pass draws the same noise
    sum(models[k](xg, xc).float().sum() for k in names).backward()
    ref = grads()
    lanes = [torch.cuda.Stream(dev) for _ in names]
    main = torch.cuda.current_stream()
    for t in range(40):
        rng._seed_seen, rng._counter = state
        ys = []
        for lane, k in zip(lanes, names):
            lane.wait_stream(main)
            with torch.cuda.stream(lane):
                ys.append(models[k](xg, xc))
        for lane, y in zip(lanes, ys):
            main.wait_stream(lane); y.record_stream(main)
        sum(y.float().sum() for y in ys).backward()
        got = grads()
        torch.cuda.synchronize()
        bad = [k for k in ref if not torch.equal(got[k], ref[k])]
        assert not bad, (t, bad[:4])


def test_companion_stream_is_refused_where_autograd_would_touch_its_result():
    """the weight gradient on the companion stream is only safe where AccumulateGrad adopts the tensor without a kernel of its own.  The cases where it
    would launch one (a .grad the in-place sum refuses, create_graph, post-accumulate hooks) stay in-stream; a backward under each still gives the in-stream result."""
    from dcvgan_amd import native, ops
    native.lib()
    dev = torch.device("cuda:0")
    w = torch.nn.Parameter(torch.randn(8, 4, 4, 4, device=dev) * 0.1)
    with torch.no_grad():                                   # a backward node runs with grad mode off ...
        assert ops.wgrad_companion(dev, w) is not None
        w.grad = torch.zeros(4, 8, 4, 4, device=dev).permute(1, 0, 2, 3)      # not contiguous: autograd adds, on the chain's stream
        assert ops.wgrad_companion(dev, w) is None
        w.grad = torch.zeros_like(w)
        assert ops.wgrad_companion(dev, w) is not None
        w.grad = None
        h = w.register_post_accumulate_grad_hook(lambda p: None)
        assert ops.wgrad_companion(dev, w) is None
        h.remove()
        assert ops.wgrad_companion(dev, w) is not None
    assert ops.wgrad_companion(dev, w) is None              # ... unless create_graph=True keeps it on: AccumulateGrad clones then

    g = ops.conv_geom(w, (2, 2), (1, 1), False)
    x = torch.randn(3, 4, 16, 16, device=dev)
    cot = torch.randn(3, 8, 8, 8, device=dev)

    def dw(prepare):
        w.grad = None
        handle = prepare()
        (ops.conv(x, w, g) * cot).sum().backward()
        torch.cuda.synchronize()
        if handle is not None:
            handle.remove()
        out = w.grad.detach().clone()
        w.grad = None
        return out
    seen = []
    base = dw(lambda: None)
    hooked = dw(lambda: w.register_post_accumulate_grad_hook(lambda p: seen.append(p.grad.detach().clone())))
    assert torch.equal(base, hooked) and len(seen) == 1 and torch.equal(seen[0], base)      # the hook read a complete gradient

    def preset():
        w.grad = torch.ones(4, 8, 4, 4, device=dev).permute(1, 0, 2, 3)
    w.grad = None
    preset()
    (ops.conv(x, w, g) * cot).sum().backward()
    torch.cuda.synchronize()
    assert torch.allclose(w.grad, base + 1.0, rtol=0, atol=1e-5 * float(base.abs().max()))
    w.grad = None
