import sys, time, gc
sys.path.insert(0,'.')
import torch, bench
dev = torch.device('cuda',0)
bs = 512
module, trainer = bench.build_trainer('dc_gan', bs, dev, 1)
batch = bench.synthetic_batch(bs, dev, 0)
def run(tag, n=12):
    for _ in range(4): trainer.step(batch)
    torch.cuda.synchronize()
    ts=[]
    for i in range(n):
        t0=time.perf_counter(); trainer.step(batch); torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)*1e3)
    t0=time.perf_counter()
    for i in range(n): trainer.step(batch)
    torch.cuda.synchronize(); tot=(time.perf_counter()-t0)*1e3/n
    print(tag, 'per-step synced:', [round(t,1) for t in ts], '| unsynced avg per step %.2f ms' % tot)
run('baseline(pageable .to)')
# variant A: device RNG
module.sample_noise = lambda n: torch.randn(n, 100, device=dev)
run('device randn')
# variant B: host RNG into pinned buffer, async copy
pin = torch.empty(bs, 100).pin_memory()
def sample_pinned(n):
    torch.normal(0.0, 1.0, (n,100), out=pin[:n])
    return pin[:n].to(dev, non_blocking=True)
module.sample_noise = sample_pinned
run('pinned async')
