import torch, time
from torch.distributions.normal import Normal
d = Normal(0,1)
print('threads', torch.get_num_threads())
def t(f, n=20):
    f(); t0=time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter()-t0)/n*1e3
print('sample 512x100 ms', t(lambda: d.sample((512,100))))
print('sample 128x100 ms', t(lambda: d.sample((128,100))))
x = d.sample((512,100))
torch.cuda.init(); y = x.cuda(); torch.cuda.synchronize()
print('to cuda ms', t(lambda: x.to('cuda')))
torch.set_num_threads(8)
print('threads 8: sample 512x100 ms', t(lambda: d.sample((512,100))))
torch.set_num_threads(1)
print('threads 1: sample 512x100 ms', t(lambda: d.sample((512,100))))
g = torch.Generator(device='cuda')
print('cuda randn ms', t(lambda: torch.randn(512,100, device='cuda', generator=g)))
