import torch, sys, torch.nn.functional as TF
sys.path.insert(0,'.')
from lightning_gan_zoo_amd import functional as F
g=F.K4S2P1
for (N,C,H,K) in [(2,3,16,5),(4,8,16,16),(8,64,16,128),(64,16,32,256)]:
    x=torch.randn(N,C,H,H); w=torch.randn(K,C,4,4)*0.1
    print('case',(N,C,H,K), flush=True)
    y=F._conv_fwd_raw(x.cuda(),w.cuda(),None,g,0,0.); torch.cuda.synchronize()
    ref=TF.conv2d(x,w,None,2,1)
    print('  rel', float((y.cpu()-ref).abs().max()/ref.abs().max()), flush=True)
