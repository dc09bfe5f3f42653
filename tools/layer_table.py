"""Per-layer table of the convolution kernels at a config's real shapes: forward / data gradient / weight gradient
through the C ABI, HIP-event timed -> CSV (layer, op, kernel, GFLOP, ms, TFLOP/s, fraction of the fp32 MFMA peak,
launches of that op per iteration).  This is what DESIGN.md's per-layer statements and bench.py's dominant-kernel
figure are read from; the committed copies live in profiles/r02_layers_<config>.csv.

    python tools/layer_table.py [config] [--csv out.csv] [--filter text] [--plan plan.json --reps N]

`--plan` (for tools/pmc_layers.sh): run each (layer, op) exactly N times, untimed, and write the launch counts in
dispatch order so that a rocprofv3 counter trace of the same process can be attributed layer by layer."""
import argparse
import ctypes as C
import json
import sys

sys.path.insert(0, '.')
import torch
from dcvgan_amd import native as N, ops
from dcvgan_amd.configs import CONFIGS
from dcvgan_amd.native import dims5, ptr, stream_ptr, lib

PEAK = 157.3
ap = argparse.ArgumentParser()
ap.add_argument("config", nargs="?", default="isogd-depth")
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--csv", default="")
ap.add_argument("--filter", default="")
ap.add_argument("--plan", default="")
ap.add_argument("--reps", type=int, default=20)      # timed launches per row, after --warm untimed ones: the SUSTAINED figure (clocks settled), the one
ap.add_argument("--warm", type=int, default=10)      # rocprofv3's per-dispatch durations inside the iteration and bench.py's probe agree with
ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "f32x6", "bf16cl"])
a = ap.parse_args()
cfg = CONFIGS[a.config]
B = a.batch or cfg.batchsize
F = B * 16
Cg = cfg.channel
dev = torch.device("cuda:0")
w_ = cfg.width
gated = 1.0 / cfg.num_gen_update                      # share of iterations whose D phase runs a backward
hinge = cfg.loss != "adversarial-loss"

L = []


def add(name, tr, cin, cout, k, s, p, xshape, nf, nd, nw):
    L.append((name, tr, cin, cout, k, s, p, xshape, nf, nd, nw))


# launches per iteration (forward, data gradient, weight gradient): generators run 2 forwards; their backward runs in the
# D phase (when it is not gated off) and in the G phase.  Discriminators: 3 forwards; backward on real + fake in the D
# phase (no data gradient into real inputs for the first convs) and on fake in the G phase (gdis not under hinge).
gb = gated + 1.0
ngf, ncf = w_["ggen"], w_["cgen"]
add(f"ggen.0 convT 50->{8 * ngf} 1x1->4", True, 50, 8 * ngf, (4, 4), (1, 1), (0, 0), (F, 50, 1, 1), 2, gb, gb)
add(f"ggen.3 convT {8 * ngf}->{4 * ngf} @4", True, 8 * ngf, 4 * ngf, (4, 4), (2, 2), (1, 1), (F, 8 * ngf, 4, 4), 2, gb, gb)
add(f"ggen.6 convT {4 * ngf}->{2 * ngf} @8", True, 4 * ngf, 2 * ngf, (4, 4), (2, 2), (1, 1), (F, 4 * ngf, 8, 8), 2, gb, gb)
add(f"ggen.9 convT {2 * ngf}->{ngf} @16", True, 2 * ngf, ngf, (4, 4), (2, 2), (1, 1), (F, 2 * ngf, 16, 16), 2, gb, gb)
add(f"ggen.12 convT {ngf}->{Cg} @32", True, ngf, Cg, (4, 4), (2, 2), (1, 1), (F, ngf, 32, 32), 2, gb, gb)
add(f"cgen.in conv3 {Cg}->{ncf} @64", False, Cg, ncf, (3, 3), (1, 1), (1, 1), (F, Cg, 64, 64), 2, gb, gb)
for i, (x, y, sp) in enumerate([(1, 1, 64), (1, 2, 32), (2, 4, 16), (4, 4, 8), (4, 4, 4), (4, 4, 2)]):
    add(f"cgen.down{i} conv {x * ncf}->{y * ncf} @{sp}", False, x * ncf, y * ncf, (4, 4), (2, 2), (1, 1), (F, x * ncf, sp, sp), 2, gb, gb)
for i, (x, y, sp) in enumerate([(4 * ncf + 10, 4 * ncf, 1), (8 * ncf, 4 * ncf, 2), (8 * ncf, 4 * ncf, 4), (8 * ncf, 2 * ncf, 8), (4 * ncf, ncf, 16), (2 * ncf, ncf, 32)]):
    add(f"cgen.up{i} convT {x}->{y} @{sp}", True, x, y, (4, 4), (2, 2), (1, 1), (F, x, sp, sp), 2, gb, gb)
add(f"cgen.out convT3 {2 * ncf}->3 @64", True, 2 * ncf, 3, (3, 3), (1, 1), (1, 1), (F, 2 * ncf, 64, 64), 2, gb, gb)
di, dv, dg = w_["idis"], w_["vdis"], w_["gdis"]
dd, dd1, dw = 2 * gated + 1, gated + 1, 2 * gated + 1          # trunk dgrads; first-conv dgrads; wgrads
add(f"idis.g conv {Cg}->{di // 2} @64", False, Cg, di // 2, (4, 4), (2, 2), (1, 1), (B, Cg, 64, 64), 3, dd1, dw)
add(f"idis.c conv 3->{di // 2} @64", False, 3, di // 2, (4, 4), (2, 2), (1, 1), (B, 3, 64, 64), 3, dd1, dw)
add(f"idis.1 conv {di}->{2 * di} @32", False, di, 2 * di, (4, 4), (2, 2), (1, 1), (B, di, 32, 32), 3, dd, dw)
add(f"idis.5 conv {2 * di}->{4 * di} @16", False, 2 * di, 4 * di, (4, 4), (2, 2), (1, 1), (B, 2 * di, 16, 16), 3, dd, dw)
add(f"idis.9 conv {4 * di}->1 @8", False, 4 * di, 1, (4, 4), (2, 2), (1, 1), (B, 4 * di, 8, 8), 3, dd, dw)
S3, P3, K3 = (1, 2, 2), (0, 1, 1), (4, 4, 4)
add(f"vdis.g conv3d {Cg}->{dv // 2}", False, Cg, dv // 2, K3, S3, P3, (B, Cg, 16, 64, 64), 3, dd1, dw)
add(f"vdis.c conv3d 3->{dv // 2}", False, 3, dv // 2, K3, S3, P3, (B, 3, 16, 64, 64), 3, dd1, dw)
add(f"vdis.1 conv3d {dv}->{2 * dv}", False, dv, 2 * dv, K3, S3, P3, (B, dv, 13, 32, 32), 3, dd, dw)
add(f"vdis.5 conv3d {2 * dv}->{4 * dv}", False, 2 * dv, 4 * dv, K3, S3, P3, (B, 2 * dv, 10, 16, 16), 3, dd, dw)
add(f"vdis.9 conv3d {4 * dv}->1", False, 4 * dv, 1, K3, S3, P3, (B, 4 * dv, 7, 8, 8), 3, dd, dw)
gq = 0.0 if hinge else 1.0                                        # gdis takes part in the G-phase backward?
add(f"gdis.1 conv3d {Cg}->{dg}", False, Cg, dg, K3, S3, P3, (B, Cg, 15, 64, 64), 3, gated + gq, 2 * gated + gq)
add(f"gdis.5 conv3d {dg}->{2 * dg}", False, dg, 2 * dg, K3, S3, P3, (B, dg, 12, 32, 32), 3, 2 * gated + gq, 2 * gated + gq)
add(f"gdis.9 conv3d {2 * dg}->{4 * dg}", False, 2 * dg, 4 * dg, K3, S3, P3, (B, 2 * dg, 9, 16, 16), 3, 2 * gated + gq, 2 * gated + gq)
add(f"gdis.13 conv3d {4 * dg}->1", False, 4 * dg, 1, K3, S3, P3, (B, 4 * dg, 6, 8, 8), 3, 2 * gated + gq, 2 * gated + gq)


def timeit(fn, reps):
    for _ in range(max(2, a.warm)):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


Lb = lib()
CL = a.precision == "bf16cl"
if CL:
    from dcvgan_amd import ops_cl
else:
    N.set_precision(a.precision)
if a.precision in ("bf16", "bf16cl"):
    PEAK = 16 * 157.3
if a.precision == "f32x6":
    PEAK = 16 * 157.3 / 6
rows, plan = [], []
tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
totf = 0.0
print("%-34s %9s | %8s %6s | %8s %6s | %8s %6s" % ("layer", "GF", "fwd ms", "TF/s", "dgrad", "TF/s", "wgrad", "TF/s"))
for (name, tr, cin, cout, k, s, p, xs, nf, nd, nw) in L:
    if a.filter and a.filter not in name:
        continue
    x = torch.randn(xs, device=dev)
    w = torch.randn(((cin, cout) if tr else (cout, cin)) + k, device=dev) * 0.05
    g = ops.conv_geom(w, s, p, tr)
    with torch.no_grad():
        if CL:      # bf16 channels-last operands (ops_cl); the packed bf16 weights are made once, as once per optimiser step in the iteration
            x = ops_cl.from_f32(x)
            y = ops_cl.conv(x, w, g)
            dy = ops_cl.from_f32(torch.randn(y.shape, device=dev))
        else:
            y = ops.conv(x, w, g)
            dy = torch.randn_like(y)
    xd, yd = dims5(x), dims5(y)
    taps = k[0] * k[1] * (k[2] if len(k) == 3 else 1)
    macs = (x.numel() // cin if tr else y.numel() // cout) * cin * cout * taps
    gf = 2 * macs / 1e9
    dx = ops_cl.cl_empty(x.shape, dev) if CL else torch.empty_like(x); dw_ = torch.empty_like(w)
    packs = {}
    if CL:
        dxd = dims5(dx)
        pk0 = ops_cl._packed(w, 0, g, xd, yd, tuple(x.shape)); pk1 = ops_cl._packed(w, 1, g, dxd, yd, tuple(x.shape))

    def pack(which):      # the step caches the packed weights between optimiser steps: so does the table
        if which not in packs:
            n = Lb.dcv_conv_packed_bytes(C.byref(g), C.byref(xd), C.byref(yd), which)
            packs[which] = (torch.empty(max(n, 1), dtype=torch.uint8, device=dev), N.WPack(0, n, 0, Lb.dcv_conv_effective_precision(C.byref(g))))
            packs[which][1].buf = packs[which][0].data_ptr()
        pk = packs[which][1]
        ref = C.byref(pk)
        return pk, ref

    def fwd_cl():
        wsp, wsn = ops._ws("clconv", Lb.dcv_cl_conv_workspace_bytes(C.byref(g), C.byref(xd), C.byref(yd), 0), dev)
        N.check(Lb.dcv_cl_conv_forward(C.byref(g), ptr(x), C.byref(xd), ptr(pk0), ptr(y), C.byref(yd), 0, 0.0, wsp, wsn, stream_ptr()), "f")

    def dgrad_cl():
        wsp, wsn = ops._ws("clconv", Lb.dcv_cl_conv_workspace_bytes(C.byref(g), C.byref(dxd), C.byref(yd), 1), dev)
        N.check(Lb.dcv_cl_conv_backward_data(C.byref(g), ptr(dy), C.byref(yd), ptr(pk1), ptr(dx), C.byref(dxd), 0, wsp, wsn, stream_ptr()), "d")

    def wgrad_cl():
        need = Lb.dcv_cl_wgrad_workspace_bytes(C.byref(g), C.byref(xd), C.byref(yd)); wsp, wsn = ops._ws("clconv", need, dev)
        N.check(Lb.dcv_cl_conv_backward_weight(C.byref(g), ptr(x), C.byref(xd), ptr(dy), C.byref(yd), ptr(dw_), wsp, wsn, stream_ptr()), "w")

    def fwd():
        need = Lb.dcv_conv_workspace_bytes(C.byref(g), C.byref(xd), C.byref(yd), 0); wsp, wsn = ops._ws("conv", need, dev)
        pk, ref = pack(0)
        N.check(Lb.dcv_conv_forward(C.byref(g), ptr(x), C.byref(xd), ptr(w), ptr(y), C.byref(yd), 0, 0.0, ref, wsp, wsn, stream_ptr()), "f")
        pk.ready = 1

    def dgrad():
        need = Lb.dcv_conv_workspace_bytes(C.byref(g), C.byref(xd), C.byref(yd), 1); wsp, wsn = ops._ws("conv", need, dev)
        pk, ref = pack(1)
        N.check(Lb.dcv_conv_backward_data(C.byref(g), ptr(dy), C.byref(yd), ptr(w), ptr(dx), C.byref(xd), 0, ref, wsp, wsn, stream_ptr()), "d")
        pk.ready = 1

    def wgrad():
        need = Lb.dcv_conv_workspace_bytes(C.byref(g), C.byref(xd), C.byref(yd), 2); wsp, wsn = ops._ws("conv", need, dev)
        N.check(Lb.dcv_conv_backward_weight(C.byref(g), ptr(x), C.byref(xd), ptr(dy), C.byref(yd), ptr(dw_), wsp, wsn, stream_ptr()), "w")

    line = [name, gf]
    for op, fn, cnt in (("fwd", fwd_cl if CL else fwd, nf), ("dgrad", dgrad_cl if CL else dgrad, nd), ("wgrad", wgrad_cl if CL else wgrad, nw)):
        if a.plan:
            fn(); torch.cuda.synchronize()      # packs / tables built outside the counted launches
            n0 = N.launch_count()
            for _ in range(a.reps):
                fn()
            torch.cuda.synchronize()
            plan.append({"layer": name, "op": op, "launches": N.launch_count() - n0, "reps": a.reps, "gflop": gf,
                         "kernel": Lb.dcv_debug_last_kernel().decode()})
            continue
        ms = timeit(fn, a.reps)
        kn = Lb.dcv_debug_last_kernel().decode()
        rows.append((name, op, kn, gf, ms, gf / ms, gf / ms / PEAK, cnt))
        tot[op] += ms * cnt; totf += gf * cnt
        line += [ms, gf / ms]
    if not a.plan:
        print("%-34s %9.1f | %8.3f %6.1f | %8.3f %6.1f | %8.3f %6.1f" % tuple(line))
if a.plan:
    json.dump({"config": a.config, "batch": B, "entries": plan}, open(a.plan, "w"), indent=1)
    sys.exit(0)
ts = sum(tot.values())
print("per-iteration conv time (ms): fwd %.1f dgrad %.1f wgrad %.1f total %.1f ; %.1f TF/s avg = %.3f of the %s MFMA peak" % (tot["fwd"], tot["dgrad"], tot["wgrad"], ts, totf / ts, totf / ts / PEAK, a.precision))
if a.csv:
    with open(a.csv, "w") as f:
        f.write(f"# {a.config}, per-GPU batch {B}; HIP events, {a.reps} timed launches after {max(2, a.warm)} untimed; launches_per_iteration follows trainer.py:279-363 (gating averaged)\n")
        f.write(f"layer,op,kernel,gflop,ms,tflops,frac_of_{a.precision}_mfma_peak,launches_per_iteration\n")
        for r in rows:
            f.write('"%s",%s,"%s",%.2f,%.4f,%.1f,%.3f,%.2f\n' % r)
        f.write('"TOTAL (conv kernels)",all,,%.1f,%.2f,%.1f,%.3f,\n' % (totf, ts, totf / ts, totf / ts / PEAK))
