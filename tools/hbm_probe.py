"""HBM streaming rates at the RGB head's tensor size (2.35 GB): write-only, read-only, copy — what bounds widen_mfma_kernel (write-only), thin_rows_kernel
and thinj_wgrad_kernel (read-only), and the BatchNorm passes (mixed).  torch's own streaming kernels; HIP events over 20 launches after 5 untimed."""
import torch

dev = torch.device("cuda:0")
n = 1120 * 128 * 64 * 64
a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
a.normal_()


def t(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


gb = n * 4 / 1e9
for name, fn, bytes_ in (("fill (write-only)", lambda: b.fill_(1.5), gb), ("memset (write-only)", lambda: b.zero_(), gb), ("sum (read-only)", lambda: a.sum(), gb),
                         ("copy (read + write)", lambda: b.copy_(a), 2 * gb), ("mul_ in place (read + write)", lambda: a.mul_(1.0001), 2 * gb)):
    ms = t(fn)
    print(f"{name:32s} {ms:7.3f} ms  {bytes_ / ms:6.2f} TB/s")
