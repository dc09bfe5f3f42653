"""CPU, world_size 2 over gloo: the data-parallel optimiser wrappers of one phase share a gradient bucket that is
all-reduced once per backward (explicit dirty flag set by autograd hooks — also for the trainer's double ggen
step, for accumulated backwards and for gated-off updates), and broadcast_module makes replicas identical."""
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
    netA = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 3))
    netB = torch.nn.Linear(3, 2)
    for n in (netA, netB):
        optim.broadcast_module(n)                       # -> rank 0's parameters and buffers everywhere
    params = list(netA.parameters()) + list(netB.parameters())
    w0 = torch.cat([p.detach().reshape(-1) for p in params])
    frozen = list(netA.parameters())[-1]                # a parameter that gets no gradient
    # two optimisers stepped after the same backward share ONE bucket (tiny chunks: several collectives per reduction)
    bucket = optim.GradBucket(bucket_bytes=64)
    optA = optim.DataParallelAdam(_SGD(netA.parameters(), 0.1), bucket)
    optB = optim.DataParallelAdam(_SGD(netB.parameters(), 0.1), bucket)
    ok = True

    def run(scale):
        x = torch.randn(4, 6, generator=torch.Generator().manual_seed(rank))
        (netB(netA(x)).sum() * (rank + 1) * scale).backward()
        frozen.grad = None

    def check_sum(local):
        gathered = [None] * world
        dist.all_gather_object(gathered, [None if g is None else g.numpy() for g in local])
        good = True
        for i, p in enumerate(params):
            if p.grad is None:
                good &= local[i] is None
                continue
            want = sum(torch.from_numpy(g[i]) for g in gathered)
            good &= torch.allclose(p.grad, want, atol=1e-6)   # grads hold the SUM; 1/world is the step's grad_scale
        return good

    # iteration 1: one backward, then A.step, B.step, A.step (the trainer's ggen / cgen / ggen pattern)
    run(1.0)
    local = [None if p.grad is None else p.grad.clone() for p in params]
    ok &= bucket.dirty
    optA.step()
    ok &= bucket.reductions == 1 and not bucket.dirty
    optB.step(); optA.step()                            # same backward: must NOT reduce again
    ok &= bucket.reductions == 1 and check_sum(local)
    ok &= optA.inner.grad_scale == 1.0 / world and optA.inner.steps == 2 and optB.inner.steps == 1
    # iteration 2: zero_grad, TWO backwards accumulated (D phase: real + fake), one reduction of the sum
    for n in (netA, netB):
        n.zero_grad()
    run(1.0); run(0.5)
    local = [None if p.grad is None else p.grad.clone() for p in params]
    optB.step()                                         # whichever member steps first reduces the whole bucket
    ok &= bucket.reductions == 2 and check_sum(local)
    optA.step()
    ok &= bucket.reductions == 2
    # iteration 3: no backward at all (update gated off) -> a step must not start a collective
    optA.step()
    ok &= bucket.reductions == 2
    # the slice of a member without a gradient is zeroed before every collective (a stale slice would grow by the world size per reduction)
    ok &= bool(torch.isfinite(bucket._flat).all()) and float(frozen._dcv_grad_slot.abs().max()) == 0.0
    w1 = torch.cat([p.detach().reshape(-1) for p in params])
    allw = [None] * world
    dist.all_gather_object(allw, (w0.numpy(), w1.numpy()))
    ok &= all((a[0] == allw[0][0]).all() and (a[1] == allw[0][1]).all() for a in allw)   # replicas stay identical
    ok &= not (w0 == w1).all()
    # BatchNorm buffers are per-rank state; broadcast_buffers gives every rank rank 0's before a snapshot
    bn = torch.nn.BatchNorm2d(3)
    bn.running_mean.fill_(float(rank + 1)); bn.num_batches_tracked.fill_(rank + 5)
    optim.broadcast_buffers(bn)
    ok &= bool((bn.running_mean == 1.0).all()) and int(bn.num_batches_tracked) == 5
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def _worker_gated(rank, world, port, q):
    """The trainer's gating at world 4 (surreal-depth1: num_gen_update = 2): per iteration a D-phase backward that is only taken — and whose
    three optimisers are only stepped — on even iterations, then a G-phase backward + the ggen / cgen / ggen steps every iteration.  Two buckets
    (D: three members, G: two), one collective per taken backward, none for a gated-off phase; every .grad IS its slice of the bucket's flat
    buffer afterwards; replicas stay bit-identical."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dcvgan_amd import optim
    torch.manual_seed(7 + rank)
    dis = [torch.nn.Linear(4, 3), torch.nn.Linear(4, 2), torch.nn.Linear(4, 1)]
    gen = [torch.nn.Linear(5, 4), torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.Tanh())]
    for n in dis + gen:
        optim.broadcast_module(n)
    bD, bG = optim.GradBucket(), optim.GradBucket()
    oD = [optim.DataParallelAdam(_SGD(n.parameters(), 0.05), bD) for n in dis]
    oG = [optim.DataParallelAdam(_SGD(n.parameters(), 0.05), bG) for n in gen]
    ok = True
    gd = torch.Generator().manual_seed(100 + rank)       # per-rank data
    want_D = want_G = 0
    for it in range(1, 5):
        z = torch.randn(6, 5, generator=gd)
        real = torch.randn(6, 4, generator=gd)
        for n in dis:
            n.zero_grad()
        fake = gen[1](gen[0](z))                          # NOT detached: the D-phase backward also runs through the generators (trainer.py:304-319)
        loss_d = sum((d(real) ** 2).mean() + (d(fake) ** 2).mean() for d in dis)
        if it % 2 == 0:
            loss_d.backward()
            before = bD.collectives
            for o in oD:
                o.step()
            want_D += 1
            ok &= bD.collectives == before + 1            # ONE collective for the three members
            ok &= all(p.grad is not None and p.grad.data_ptr() == p._dcv_grad_slot.data_ptr() for n in dis for p in n.parameters())
        else:
            before = bD.collectives
            for o in oD:
                o.step()                                  # a trainer that stepped anyway must not start a collective: nothing is dirty
            ok &= bD.collectives == before or it > 1      # (iteration 1: nothing ever arrived; later: the G phase's dead D gradients did arrive)
        for n in gen:
            n.zero_grad()
        fake = gen[1](gen[0](z))
        loss_g = sum(-(d(fake)).mean() for d in dis[:2])  # hinge: the third discriminator takes no part in the G loss (loss.py:190-191)
        loss_g.backward()
        before = bG.collectives
        oG[0].step(); oG[1].step(); oG[0].step()          # ggen, cgen, ggen (trainer.py:357-359)
        want_G += 1
        ok &= bG.collectives == before + 1 and bG.reductions == want_G
        ok &= all(p.grad.data_ptr() == p._dcv_grad_slot.data_ptr() for n in gen for p in n.parameters())
    flat = torch.cat([p.detach().reshape(-1) for n in dis + gen for p in n.parameters()])
    allw = [None] * world
    dist.all_gather_object(allw, flat.numpy())
    ok &= all((a == allw[0]).all() for a in allw)
    ok &= bD._flat is not None and bD._flat.numel() >= sum(p.numel() for n in dis for p in n.parameters())
    q.put((rank, bool(ok), bD.reductions, bG.reductions))
    dist.destroy_process_group()


def test_dp_gating_world4_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_gated, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(r[:2] for r in res) == [(r, True) for r in range(4)], res
    assert all(r[3] == 4 for r in res), res              # four G-phase reductions


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


def test_bench_launcher_fails_loudly_without_gpus():
    """bench.py --gpus N (N > 1, no launcher): the parent spawns the ranks itself.  Here there is no GPU, so (a) asking for more ranks
    than visible devices is refused before anything starts, (b) in the rehearsal form every rank dies on its `needs an MI355X`
    assertion and the parent must come back non-zero promptly with no JSON line — never hang, never print a partial result."""
    import subprocess
    import sys
    import time
    import pytest
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible: the ranks would run (tests/test_dp_gpu.py covers that form)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "visible" in p.stderr and p.stdout.strip() == ""
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--all-ranks-on-device0", "--backend", "gloo"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode not in (0, 2) and p.stdout.strip() == "" and "rank" in p.stderr, (p.returncode, p.stderr[-500:])
    assert time.time() - t0 < 120


def _worker_overlap(rank, world, port, q):
    """GradBucket(overlap=True) over gloo: per-model chunks; the first backward is reduced at the step (nothing recorded yet), from the second on every chunk's collective
    is launched from the hook of its last gradient — during the backward — and the step only waits; gradients and parameters equal the non-overlapped bucket's bit for bit;
    a parameter without gradient keeps a zero slice; a changed arrival set falls back to the step-time reduction for that chunk."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dcvgan_amd import optim

    def build(overlap):
        torch.manual_seed(5)
        a = torch.nn.Sequential(torch.nn.Linear(6, 40), torch.nn.Tanh(), torch.nn.Linear(40, 8))       # "ggen"
        b = torch.nn.Sequential(torch.nn.Linear(8, 64), torch.nn.Tanh(), torch.nn.Linear(64, 3))       # "cgen"
        bucket = optim.GradBucket(overlap=overlap, merge_bytes=64)
        oa = optim.DataParallelAdam(_SGD(a.parameters(), 0.05), bucket)
        ob = optim.DataParallelAdam(_SGD(b.parameters(), 0.05), bucket)
        return a, b, bucket, oa, ob

    ok = True
    res = {}
    for overlap in (False, True):
        a, b, bucket, oa, ob = build(overlap)
        gd = torch.Generator().manual_seed(100 + rank)
        early = []
        for it in range(4):
            for n in (a, b):
                n.zero_grad()
            x = torch.randn(5, 6, generator=gd)
            (b(a(x)) ** 2).mean().backward()
            early.append(bucket.early)
            oa.step(); ob.step(); oa.step()
        res[overlap] = torch.cat([p.detach().reshape(-1) for n in (a, b) for p in n.parameters()]).clone()
        if overlap:
            ok &= len(bucket._chunks) == 2                                  # one chunk per model
            ok &= early == [0, 2, 4, 6]                                      # iteration 1: reduced at the step; then both chunks launched DURING each backward
            ok &= bucket.collectives == 2 * 4 and bucket.reductions == 4
        else:
            ok &= len(bucket._chunks) == 1 and bucket.collectives == 4 and bucket.early == 0
    ok &= torch.equal(res[False], res[True])                                 # two ranks: the sum of two numbers is the same in one message or in two
    allw = [None] * world
    dist.all_gather_object(allw, res[True].numpy())
    ok &= all((w == allw[0]).all() for w in allw)                            # replicas identical
    # a backward that skips model b's second layer (another arrival set): no early launch for that chunk, still correct
    a, b, bucket, oa, ob = build(True)
    x = torch.randn(5, 6, generator=torch.Generator().manual_seed(rank))
    for n in (a, b):
        n.zero_grad()
    (b(a(x)) ** 2).mean().backward(); oa.step(); ob.step()
    for n in (a, b):
        n.zero_grad()
    e0 = bucket.early
    (b[0](a(x)) ** 2).mean().backward()                                      # b[2] gets no gradient this time
    ok &= bucket.early == e0 + 1                                             # model a's chunk went early, model b's did not
    oa.step(); ob.step()
    ok &= float(b[2].weight._dcv_grad_slot.abs().max()) == 0.0 and bool(torch.isfinite(bucket._flat).all())
    q.put((rank, bool(ok), early))
    dist.destroy_process_group()


def test_dp_overlap_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ps = [ctx.Process(target=_worker_overlap, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = [q.get(timeout=300) for _ in ps]
    for p in ps:
        p.join(60)
    assert all(g[1] for g in got), got
