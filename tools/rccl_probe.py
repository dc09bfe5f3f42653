"""RCCL on one card: open the `nccl` backend (= RCCL on ROCm) with world_size 1 on cuda:0 and push the G-phase gradient bucket of the
headline config (ggen + cgen of isogd-depth: 55.1 MB flat fp32) through optim.GradBucket.reduce — flatten, all_reduce(sum),
re-point every .grad at its slice.  With one rank the reduced gradients must equal the local ones bit for bit.  Prints one JSON line
(latency per reduction by HIP events on the collective's stream).  Run as a fresh process: python3 tools/rccl_probe.py [reps]"""
import json
import os
import socket
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from dcvgan_amd import optim, trainer
    from dcvgan_amd.configs import CONFIGS
    cfg = CONFIGS["isogd-depth"]
    torch.manual_seed(cfg.seed)
    models = trainer.build_models(cfg, dev)
    bucket = optim.GradBucket()
    params = list(models["ggen"].parameters()) + list(models["cgen"].parameters())
    bucket.add(params)
    g = torch.Generator(device=dev).manual_seed(1)
    for p in params:
        p.grad = torch.randn(p.shape, device=dev, generator=g)
    local = [p.grad.clone() for p in params]
    nbytes = sum(p.numel() for p in params) * 4
    ms = []
    for i in range(reps + 2):
        bucket.dirty = True                     # what the post-accumulate-grad hooks do after a backward
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        bucket.reduce(force=True)
        e1.record(); e1.synchronize()
        if i >= 2:
            ms.append(e0.elapsed_time(e1))
    same = all(torch.equal(p.grad, l) for p, l in zip(params, local))
    flat_views = len({p.grad.untyped_storage().data_ptr() for p in params})
    opt = optim.DataParallelAdam(optim.Adam(models["ggen"].parameters(), lr=2e-4, betas=(0.5, 0.999), weight_decay=1e-5), bucket)
    before = torch.cat([p.detach().reshape(-1) for p in models["ggen"].parameters()]).clone()
    opt.step()                                  # Adam reads the re-pointed slices of the flat buffer
    torch.cuda.synchronize()
    moved = float((torch.cat([p.detach().reshape(-1) for p in models["ggen"].parameters()]) != before).float().mean())
    print(json.dumps({"backend": dist.get_backend(), "world_size": dist.get_world_size(), "bucket_bytes": nbytes, "tensors": len(params),
                      "collectives": bucket.collectives, "reductions": bucket.reductions, "ms_per_reduction": sum(ms) / len(ms), "ms_min": min(ms),
                      "reduced_equals_local": bool(same), "storages_after_reduce": flat_views, "adam_moved_fraction": moved,
                      "nccl_version": list(torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
