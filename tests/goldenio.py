"""Helpers shared by the tests: load golden fixtures into oracle states/configs."""
import os
from types import SimpleNamespace

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODELS = ("ggen", "cgen", "idis", "vdis", "gdis")


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def states(fx, prefix="init"):
    """-> {model: {key: tensor}} keeping the fixture's (= state_dict) key order."""
    out = {m: {} for m in MODELS}
    for k, v in fx.items():
        parts = k.split("/")
        if parts[0] == prefix and parts[1] in out:
            t = torch.from_numpy(np.array(v))
            out[parts[1]]["/".join(parts[2:])] = t
    return out


def cfg_of(fx, B=None, loss="adversarial-loss", num_gen_update=1, num_dis_update=1, start_in_eval=False):
    g = lambda k: fx["cfg/" + k]
    names = dict(i="idis", v="vdis", g="gdis")
    return SimpleNamespace(
        geometric_info=str(g("geo")), channel=int(g("Cg")), video_length=16,
        dim_z_content=int(g("dzc")), dim_z_motion=int(g("dzm")), dim_z_color=int(g("dzcol")),
        width=dict(ggen=int(g("ngf_g")), cgen=int(g("ngf_c")), idis=int(g("ndf_i")), vdis=int(g("ndf_v")), gdis=int(g("ndf_g"))),
        use_noise={names[s]: bool(g("noise_" + s)[0]) for s in "ivg"},
        noise_sigma={names[s]: float(g("noise_" + s)[1]) for s in "ivg"},
        batchsize=int(B if B is not None else fx["meta/B"]), loss=loss,
        num_gen_update=num_gen_update, num_dis_update=num_dis_update, start_in_eval=start_in_eval,
        lr={m: float(fx[f"meta/lr/{m}"]) for m in MODELS} if "meta/lr/ggen" in fx else None,
        decay={m: 1e-5 for m in MODELS},
    )


def sub(t, step=37):
    return t.detach().contiguous().view(-1)[::step].cpu().numpy()


def summ(t):
    t = t.detach().double().cpu()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
