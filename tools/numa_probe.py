"""What bench.py's per-rank CPU pinning would do on this box: the NUMA node and CPU share of every visible GPU.  Usage: python3 tools/numa_probe.py"""
import importlib.util
import os
import torch
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
print("affinity", len(os.sched_getaffinity(0)), "devices", torch.cuda.device_count())
for i in range(torch.cuda.device_count()):
    p = torch.cuda.get_device_properties(i)
    print(i, p.name, getattr(p, "pci_domain_id", None), getattr(p, "pci_bus_id", None), getattr(p, "pci_device_id", None))
for r in range(8):
    c = b.gpu_numa_cpus(r, 8)
    print("local rank", r, None if c is None else (len(c), c[:4], c[-2:]))
