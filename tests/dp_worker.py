"""Child process of tests/test_dp_gpu.py: one data-parallel rank driving the REAL DCVGAN modules (width / 8) through
trainer.StepRunner + optim.DataParallelAdam on cuda:0, gloo collectives (two ranks may share one card; RCCL refuses
that).  Usage: python tests/dp_worker.py RANK WORLD PORT MODE OUT.json
MODE: "distinct" | "same" | "same-cl16" (the bf16 channels-last data path) | "same-overlap" | "distinct-overlap" (GradBucket(overlap=True): chunked collectives launched
from the gradient hooks; three iterations — the arrival order is learned in the first backward — and, for distinct data, a twin reduced the plain way to compare with) """
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world, port, mode, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = port
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dcvgan_amd import ops_cl, optim, trainer
    from dcvgan_amd.configs import CONFIGS
    wdiv = 8
    overlap = mode.endswith("-overlap")
    if overlap:
        mode = mode[:-8]
    if mode.endswith("-cl16"):
        ops_cl.enable(True)
        mode = mode[:-5]
        wdiv = 4          # (that path concatenates 8-channel-aligned slices: the stems' ndf / 2 channels must be a multiple of 8)
    from dcvgan_amd.rng import PhiloxRng
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    cfg = CONFIGS["isogd-depth"].scaled(batchsize=2, width_div=wdiv)
    torch.manual_seed(cfg.seed + 17 * rank)                    # deliberately different replicas ...
    models = trainer.build_models(cfg, dev)
    for m in models.values():
        optim.broadcast_module(m)                              # ... made identical here
    solo = copy.deepcopy(models) if mode == "same" else None   # a world-1 twin (plain Adam) for the exactness check
    plain = copy.deepcopy(models) if (overlap and mode == "distinct") else None      # a twin reduced without overlap
    p_init = torch.cat([p.detach().reshape(-1) for m in models.values() for p in m.parameters()]).cpu()
    opts = trainer.build_optimizers(cfg, models, data_parallel=True, overlap=overlap)
    buckets = {id(o.bucket): o.bucket for o in opts.values()}
    assert len(buckets) == 2, "one bucket per phase"
    seed = cfg.seed + (0 if mode == "same" else rank)
    g = torch.Generator().manual_seed(seed)
    xc = (torch.rand(2, 3, 16, 64, 64, generator=g) * 2 - 1).to(dev)
    xg = (torch.rand(2, 1, 16, 64, 64, generator=g) * 2 - 1).to(dev)

    def with_rng(ms, s):
        r = PhiloxRng(s)
        for m in ms.values():
            m._rng = r

    # capture local gradients right before each reduction, and the reduced ones right after
    captured = []
    for b in ([] if overlap else buckets.values()):      # (with overlap a chunk may already be reduced when reduce() is called: checked through the twins below instead)
        orig = b.reduce

        def wrapped(b=b, orig=orig):
            if b.dirty and b.world > 1:
                local = [None if p.grad is None else p.grad.detach().cpu().clone() for p in b.params]
                orig()
                captured.append((local, [None if p.grad is None else p.grad.detach().cpu().clone() for p in b.params]))
            else:
                orig()
        b.reduce = wrapped

    with_rng(models, 1000 + seed)
    runner = trainer.StepRunner(cfg, models, opts, trainer.build_loss(cfg))
    n_coll = []
    n_it = 3 if overlap else 2
    for it in range(n_it):
        c0 = sum(b.collectives for b in buckets.values())
        runner.step(xc, xg, 3 + it)
        n_coll.append(sum(b.collectives for b in buckets.values()) - c0)
    torch.cuda.synchronize()

    res = {"rank": rank, "collectives_per_iteration": n_coll, "reductions": len(captured)}
    # (1) reduced gradient == sum over ranks of the local gradients (the 1/world factor is Adam's grad_scale)
    worst = 0.0
    for local, reduced in captured:
        gathered = [None] * world
        dist.all_gather_object(gathered, [None if t is None else t.numpy() for t in local])
        for i, r in enumerate(reduced):
            if r is None:
                assert all(gg[i] is None for gg in gathered)
                continue
            want = sum(torch.from_numpy(gg[i]).double() for gg in gathered)
            worst = max(worst, float((r.double() - want).abs().max() / want.abs().max().clamp_min(1e-30)))
    res["grad_sum_relerr"] = worst
    res["grad_scale"] = [o.inner.grad_scale for o in opts.values()]
    # (2) replicas identical after the updates, bit for bit
    params = torch.cat([p.detach().reshape(-1) for m in models.values() for p in m.parameters()]).cpu()
    allp = [None] * world
    dist.all_gather_object(allp, params.numpy())
    res["replicas_identical"] = bool(all((a == allp[0]).all() for a in allp))
    # (3) same data + same draws on every rank: sum of W equal gradients * (1/W) is exact, so the DP run must equal a
    #     plain single-process run bit for bit
    if solo is not None:
        with_rng(solo, 1000 + seed)
        r2 = trainer.StepRunner(cfg, solo, trainer.build_optimizers(cfg, solo), trainer.build_loss(cfg))
        for it in range(n_it):
            r2.step(xc, xg, 3 + it)
        torch.cuda.synchronize()
        ps = torch.cat([p.detach().reshape(-1) for m in solo.values() for p in m.parameters()]).cpu()
        res["equals_single_process"] = bool((ps == params).all())
        res["moved"] = float((ps != p_init).float().mean()) > 0.9   # an optimiser that never stepped would also be "identical"
    if overlap:
        res["early_collectives"] = sum(b.early for b in buckets.values())
    if plain is not None:      # distinct data: the overlapped reduction against the plain one, same collectives on the same ranges -> the same bits
        with_rng(plain, 1000 + seed)
        r3 = trainer.StepRunner(cfg, plain, trainer.build_optimizers(cfg, plain, data_parallel=True, overlap=False), trainer.build_loss(cfg))
        for it in range(n_it):
            r3.step(xc, xg, 3 + it)
        torch.cuda.synchronize()
        pp = torch.cat([p.detach().reshape(-1) for m in plain.values() for p in m.parameters()]).cpu()
        res["equals_plain_reduction"] = bool((pp == params).all())
        res["moved"] = float((pp != p_init).float().mean()) > 0.9
    json.dump(res, open(out, "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
