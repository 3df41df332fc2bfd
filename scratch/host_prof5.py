import sys, time, gc
sys.path.insert(0,'.')
import torch, bench
from torch.distributions.normal import Normal
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
    print(tag, [round(t,1) for t in ts])
d = Normal(0,1)
# (c) allocate+free 200KB pageable each step, device randn for the actual noise
def c(n):
    junk = d.sample((n,100)); del junk
    return torch.randn(n,100,device=dev)
module.sample_noise = c; run('(c) junk sample + device randn')
def c2(n):
    junk = torch.empty(n,100); junk.fill_(1.0); del junk
    return torch.randn(n,100,device=dev)
module.sample_noise = c2; run('(c2) junk empty+fill + device randn')
def c3(n):
    junk = torch.normal(0.0,1.0,(n,100)); del junk
    return torch.randn(n,100,device=dev)
module.sample_noise = c3; run('(c3) junk torch.normal + device randn')
pin = torch.empty(bs,100).pin_memory()
page = torch.empty(bs,100)
def a(n):
    torch.normal(0.0,1.0,(n,100),out=page); pin.copy_(page); return pin.to(dev, non_blocking=True)
module.sample_noise = a; run('(a) normal into prealloc pageable -> pinned -> async')
torch.set_num_threads(1)
module.sample_noise = c; run('(c) with 1 cpu thread')
