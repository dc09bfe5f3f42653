"""RCCL on one card, the overlapped reduction: `nccl` backend (= RCCL), world_size 1, three iterations of trainer.StepRunner on reduced-width models with
build_optimizers(data_parallel=True, overlap=...) — per-model chunks whose collectives are launched from the hook of the chunk's last gradient, on the communication
stream, while the backward is still running (optim.GradBucket(overlap=True)).  A world of one runs no collective by itself, so the buckets are forced
(`_force_layout`, `reduce(force=True)`): what is proven is ORDERING — Adam reads a slice only after the collective that was launched during the backward has finished
with it, the next backward writes a slice only after that — and bit-identity with the step-time reduction and with the plain single-process run.
Prints one JSON line.  Fresh process: python3 tools/rccl_overlap_probe.py"""
import json
import os
import socket
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from dcvgan_amd import optim, trainer
    from dcvgan_amd.configs import CONFIGS
    from dcvgan_amd.rng import PhiloxRng
    cfg = CONFIGS["isogd-depth"].scaled(batchsize=4, width_div=4)
    g = torch.Generator().manual_seed(3)
    xc = (torch.rand(4, 3, 16, 64, 64, generator=g) * 2 - 1).to(dev); xg = (torch.rand(4, 1, 16, 64, 64, generator=g) * 2 - 1).to(dev)

    def run(mode):
        torch.manual_seed(11)
        models = trainer.build_models(cfg, dev)
        r = PhiloxRng(5)
        for m in models.values():
            m._rng = r
        opts = trainer.build_optimizers(cfg, models, data_parallel=mode != "plain", overlap=mode == "overlap")
        buckets = []
        if mode != "plain":
            buckets = list({id(o.bucket): o.bucket for o in opts.values()}.values())
            for b in buckets:
                b._force_layout = True
                b.merge_bytes = 128 << 10      # reduced-width models: keep them in chunks of their own (at full width ggen / cgen are 15 / 40 MB)
                b.reduce = (lambda bb: (lambda force=False: optim.GradBucket.reduce(bb, force=True)))(b)      # a world of one reduces nothing by itself
        runner = trainer.StepRunner(cfg, models, opts, trainer.build_loss(cfg), sync_losses=True)
        losses = [runner.step(xc, xg, 2 + i) for i in range(3)]
        torch.cuda.synchronize()
        params = torch.cat([v.detach().float().reshape(-1) for m in models.values() for v in m.state_dict().values()]).cpu()
        return losses, params, buckets

    l0, p0, _ = run("plain")
    l1, p1, b1 = run("sync")
    l2, p2, b2 = run("overlap")
    out = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
           "sync_equals_plain": bool(torch.equal(p0, p1)) and l0 == l1, "overlap_equals_sync": bool(torch.equal(p1, p2)) and l1 == l2,
           "sync": {"collectives": sum(b.collectives for b in b1), "early": sum(b.early for b in b1), "chunks": [len(b._chunks) for b in b1]},
           "overlap": {"collectives": sum(b.collectives for b in b2), "early": sum(b.early for b in b2), "chunks": [len(b._chunks) for b in b2],
                       "chunk_mb": [[round((c.b - c.a) * 4 / 1e6, 2) for c in b._chunks] for b in b2]}}
    print(json.dumps(out))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
