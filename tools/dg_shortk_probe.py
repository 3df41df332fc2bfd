import torch, sys, os
sys.path.insert(0, os.getcwd())
from lightning_gan_zoo_amd import functional as F
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e3
g=F.K4S2P1
for (N,C,H,K) in [(64,64,64,128),(64,64,64,32),(64,64,64,64),(64,64,64,256),(64,128,32,512)]:
    w=torch.randn(K,C,4,4,device='cuda')*0.05; gy=torch.randn(N,K,H//2,H//2,device='cuda')
    fl=2.0*N*(H//2)**2*K*C*16
    t=timeit(lambda: F._conv_dgrad_raw(gy,w,None,g,(H,H),0,0.))
    print("2D Dg N%d C%d H%d K%d chunks/phase %d: %.1f us %.1f TF" % (N,C,H,K,K*4//16,t,fl/t/1e6))
