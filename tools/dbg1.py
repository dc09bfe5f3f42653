import sys; sys.path.insert(0,'.')
import numpy as np, torch
from oracle import dcvgan_oracle as O
from tests import goldenio as G
from dcvgan_amd import trainer
from dcvgan_amd.rng import InjectedRng
dev = torch.device("cuda:0")
fx = G.load("modules_flow_w4.npz"); cfg = G.cfg_of(fx); B = cfg.batchsize
rel = lambda x,y: float((x.detach().double().cpu()-y.detach().double().cpu()).norm()/y.detach().double().cpu().norm())
for which in ("idis","vdis"):
    st = G.states(fx)
    for m in st: O.require_grad(st[m])
    g = torch.Generator().manual_seed(int(fx["meta/seed_dis_inputs"]))
    xg_c = (torch.rand(B,16,cfg.channel,64,64,generator=g)*2-1); xc_c = (torch.rand(B,16,3,64,64,generator=g)*2-1)
    xg = xg_c.permute(0,2,1,3,4).requires_grad_(True); xc = xc_c.permute(0,2,1,3,4).requires_grad_(True)
    torch.manual_seed(5); rng = O.TorchRng(); t = 5
    if which == "idis": y = O.idis_forward(st["idis"], xg[:,:,t], xc[:,:,t], True, 0.2, rng, True)
    else: y = O.vdis_forward(st["vdis"], xg, xc, True, 0.2, rng, True)
    cot = torch.linspace(-1,1,y.numel()).view(y.shape)
    (y*cot).sum().backward()
    models = trainer.build_models(cfg, dev)
    for n,m in models.items(): m.load_state_dict({k:v.detach().clone() for k,v in G.states(fx)[n].items()}); m.to(dev)
    r = InjectedRng(rng.log)
    for m in models.values(): m._rng = r
    xgd = xg_c.to(dev).permute(0,2,1,3,4).requires_grad_(True); xcd = xc_c.to(dev).permute(0,2,1,3,4).requires_grad_(True)
    yd = models[which](xgd[:,:,t], xcd[:,:,t]) if which=="idis" else models[which](xgd, xcd)
    (yd*cot.to(dev)).sum().backward()
    print(which, "y", rel(yd,y), "gxg", rel(xgd.grad, xg.grad), "gxc", rel(xcd.grad, xc.grad))
    d = (xcd.grad.cpu()-xc.grad).abs()
    print("  max abs diff", d.max().item(), "at", np.unravel_index(d.argmax().item(), d.shape), "ref max", xc.grad.abs().max().item())
    # per-channel / per-position error structure
    e = (xcd.grad.cpu()-xc.grad)
    print("  err by channel", [float(e[:,c].norm()/xc.grad[:,c].norm()) for c in range(3)])
    if which=="vdis":
        print("  err by frame", [round(float(e[:,:,k].norm()/max(xc.grad[:,:,k].norm(),1e-20)),5) for k in range(16)])
    for k,p in models[which].named_parameters():
        print("  ", k, rel(p.grad, st[which][k].grad))
