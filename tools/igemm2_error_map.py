"""Where does a transposed-convolution result differ from torch's?  Error fractions by phase / image / channel /
row / column and the reference channel a wrong channel actually holds -- how the two igemm2 bugs of round 3 were found
(a VALU -> asm-MFMA hazard hit accumulator blocks (i in {0, 2}, j = 0); a bit_cast of a vector element stored element 0).

    python tools/igemm2_error_map.py        # on the GPU box
"""
import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0,'.')
from lightning_gan_zoo_amd import functional as F
import torch.nn.functional as TF
N,K,OH,C = 64,64,32,128
g=torch.Generator().manual_seed(1)
gy=torch.randn(N,K,OH,OH,generator=g); w=torch.randn(K,C,4,4,generator=g)*0.1
ref=TF.conv_transpose2d(gy,w,None,2,1)
out=F._conv_dgrad_raw(gy.cuda(),w.cuda(),None,F.K4S2P1,(2*OH,2*OH),F.ACT_NONE,0.0).cpu()
d=(out-ref).abs()
print('max err', d.max().item(), 'ref max', ref.abs().max().item())
bad=d>1e-3
print('bad frac', bad.float().mean().item())
print('by py,px', [[bad[:,:,py::2,px::2].float().mean().item() for px in (0,1)] for py in (0,1)])
print('by n (first 8)', bad.float().mean((1,2,3))[:8])
print('by c (first 16)', bad.float().mean((0,2,3))[:16], bad.float().mean((0,2,3))[60:70])
print('by row', bad.float().mean((0,1,3))[:16])
print('by col', bad.float().mean((0,1,2))[:16])
out2=F._conv_dgrad_raw(gy.cuda(),w.cuda(),None,F.K4S2P1,(2*OH,2*OH),F.ACT_NONE,0.0).cpu()
print('deterministic', torch.equal(out,out2), (out-out2).abs().max().item())
b=torch.zeros(C)
out3=F._conv_dgrad_raw(gy.cuda(),w.cuda(),b.cuda(),F.K4S2P1,(2*OH,2*OH),F.ACT_NONE,0.0).cpu()
print('slow path (zero bias): max err', (out3-ref).abs().max().item())
d=(out-ref).abs(); bad=d>1e-3
# where do wrong-channel values come from?  compare out[:,c] with ref[:,c'] for a few c
for c in (1,2,3,5,8,9):
    best=min(range(C), key=lambda cc: (out[0,c]-ref[0,cc]).abs().max().item())
    print('out channel', c, 'matches ref channel', best, (out[0,c]-ref[0,best]).abs().max().item())
