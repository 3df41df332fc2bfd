import sys, torch
sys.path.insert(0,'.')
from lightning_gan_zoo_amd import functional as F
import torch.nn.functional as TF
N,C,H,K = 128,64,32,128
g=torch.Generator().manual_seed(1)
x=torch.randn(N,C,H,H,generator=g); w=torch.randn(K,C,4,4,generator=g)*0.1
ref=TF.conv2d(x,w,None,2,1)
out=F._conv_fwd_raw(x.cuda(),w.cuda(),None,F.K4S2P1,F.ACT_NONE,0.0).cpu()
d=(out-ref).abs(); bad=d>1e-3
print('max err', d.max().item(), 'bad frac', bad.float().mean().item())
print('by n', bad.float().mean((1,2,3))[:8])
print('by k', bad.float().mean((0,2,3))[:8])
print('by oy', bad.float().mean((0,1,3)))
print('by ox', bad.float().mean((0,1,2)))
# single-tap probes: which taps are wrong?
for (ky,kx) in [(0,0),(0,1),(1,0),(1,1),(2,2),(3,3),(0,3),(3,0)]:
    w1=torch.zeros_like(w); w1[:,:,ky,kx]=w[:,:,ky,kx]
    r=TF.conv2d(x,w1,None,2,1); o=F._conv_fwd_raw(x.cuda(),w1.cuda(),None,F.K4S2P1,F.ACT_NONE,0.0).cpu()
    dd=(o-r).abs()>1e-3
    print('tap',ky,kx,'bad frac',dd.float().mean().item(), 'by ox', [round(v,2) for v in dd.float().mean((0,1,2)).tolist()][:16], 'by oy', [round(v,2) for v in dd.float().mean((0,1,3)).tolist()][:16])
