"""Which torch-side ops launch the non-dcv kernels of a step (aten copy / add / index...)?"""
import sys; sys.path.insert(0, '.')
import torch
from torch.profiler import profile, ProfilerActivity
from dcvgan_amd import trainer, optim, native
from dcvgan_amd.configs import CONFIGS
native.lib()
if len(sys.argv) > 2 and sys.argv[2] == "bf16cl":
    from dcvgan_amd import ops_cl
    ops_cl.enable(True)
cfg = CONFIGS[sys.argv[3] if len(sys.argv) > 3 else "isogd-depth"].scaled(batchsize=int(sys.argv[1]) if len(sys.argv) > 1 else 16)
dev = torch.device("cuda:0"); B = cfg.batchsize
torch.manual_seed(0)
models = trainer.build_models(cfg, dev)
opts = trainer.build_optimizers(cfg, models, data_parallel=False)
run = trainer.StepRunner(cfg, models, opts, trainer.build_loss(cfg), sync_losses=False)
xc = (torch.rand(B, 3, cfg.video_length, 64, 64) * 2 - 1).to(dev)
xg = (torch.rand(B, cfg.channel, cfg.video_length, 64, 64) * 2 - 1).to(dev)
for _ in range(2): run.step(xc, xg, 3)
run.step(xc, xg, 3) if cfg.num_gen_update > 1 else None
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    run.step(xc, xg, 3)
    torch.cuda.synchronize()
ev = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.self_device_time_total > 0]
ev.sort(key=lambda e: -e.self_device_time_total)
for e in ev[:24]:
    print("%-28s n=%4d  dev us %9.1f  shapes %s" % (e.key, e.count, e.self_device_time_total, str(e.input_shapes)[:120]))
ev = [e for e in prof.key_averages(group_by_stack_n=6) if e.key.startswith("aten::") and e.self_device_time_total > 0]
ev.sort(key=lambda e: -e.self_device_time_total)
for e in ev[:14]:
    print(e.key, e.count, e.self_device_time_total); [print("     ", l) for l in e.stack[:6] if "dcvgan_amd" in l or "tools/" in l]
