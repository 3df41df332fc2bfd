import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lightning_gan_zoo_amd import harness
torch.set_num_threads(8)
dev = torch.device("cuda", 0)
d = torch.distributions.normal.Normal(0, 1)
def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e6
print("set_num_threads 1 then 8:", t(lambda: (torch.set_num_threads(1), torch.set_num_threads(8))))
print("Normal.sample (8 threads):", t(lambda: d.sample((128, 100))))
torch.set_num_threads(1)
print("Normal.sample (1 thread):", t(lambda: d.sample((128, 100))))
print("torch.randn(128,100):", t(lambda: torch.randn(128, 100)))
x = d.sample((128, 100))
print("stager.to_device:", t(lambda: harness._stager.to_device(x, dev)))
torch.set_num_threads(8)
print("draw_on_host total (8 threads outside):", t(lambda: harness.draw_on_host(lambda: d.sample((128, 100)), dev)))
torch.set_num_threads(1)
print("draw_on_host total (1 thread outside):", t(lambda: harness.draw_on_host(lambda: d.sample((128, 100)), dev)))
print("cpus", len(os.sched_getaffinity(0)), os.cpu_count())
