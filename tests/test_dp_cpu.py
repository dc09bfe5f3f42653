"""CPU, world_size 2 over gloo: the data-parallel optimiser wrapper averages gradients across ranks
before the inner step (one collective per backward, also for the trainer's double ggen step), and
broadcast_module makes replicas identical."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _SGD:
    """CPU stand-in for dcvgan_amd.optim.Adam (whose kernel needs the GPU): same duck type."""

    def __init__(self, params, lr):
        self.params, self.lr, self.grad_scale, self.steps = list(params), lr, 1.0, 0

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self):
        self.steps += 1
        for p in self.params:
            if p.grad is not None:
                p.add_(p.grad * self.grad_scale, alpha=-self.lr)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dcvgan_amd import optim
    torch.manual_seed(100 + rank)                       # deliberately different replicas
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 3))
    optim.broadcast_module(net)                         # -> rank 0's parameters and buffers everywhere
    w0 = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    frozen = list(net.parameters())[-1]                 # a parameter that gets no gradient
    opt = optim.DataParallelAdam(_SGD(net.parameters(), 0.1), bucket_bytes=64)   # tiny buckets: several collectives
    x = torch.randn(4, 6, generator=torch.Generator().manual_seed(rank))
    y = net(x)[:, :2].sum() * (rank + 1)
    y.backward()
    frozen.grad = None
    local = [None if p.grad is None else p.grad.clone() for p in net.parameters()]
    opt.step(); opt.step()                              # second step must NOT reduce again
    gathered = [None] * world
    dist.all_gather_object(gathered, [None if g is None else g.numpy() for g in local])
    summed = [p.grad.clone() if p.grad is not None else None for p in net.parameters()]
    w1 = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    allw = [None] * world
    dist.all_gather_object(allw, (w0.numpy(), w1.numpy()))
    ok = True
    for i, s in enumerate(summed):
        if s is None:
            continue
        want = sum(torch.from_numpy(g[i]) for g in gathered)
        ok &= torch.allclose(s, want, atol=1e-6)          # grads hold the SUM; 1/world is the step's grad_scale
    ok &= opt.inner.grad_scale == 1.0 / world and opt.inner.steps == 2
    ok &= all((a[0] == allw[0][0]).all() and (a[1] == allw[0][1]).all() for a in allw)   # replicas stay identical
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_dp_wrapper_world2_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(res) == [(0, True), (1, True)], res
