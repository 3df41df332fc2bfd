import torch, torch.nn.functional as TF, sys
sys.path.insert(0, '.')
from lightning_gan_zoo_amd import functional as F
def rel(a,b):
    a=a.double().cpu(); b=b.double().cpu(); return float((a-b).abs().max()/b.abs().max())
g=F.K4S2P1
torch.manual_seed(0)
for (N,C,H,K) in [(8,3,64,64),(8,64,32,128),(8,128,16,256),(8,256,8,512),(8,512,4,1024),(8,3,64,128)]:
    x=torch.randn(N,C,H,H); w=torch.randn(K,C,4,4)*0.05
    y=TF.conv2d(x,w,None,2,1); gy=torch.randn_like(y)
    dx=TF.conv_transpose2d(gy,w,None,2,1); dw=torch.nn.grad.conv2d_weight(x,w.shape,gy,stride=2,padding=1)
    xd,wd,gd=x.cuda(),w.cuda(),gy.cuda()
    print((N,C,H,K), 'F',rel(F._conv_fwd_raw(xd,wd,None,g,0,0.),y),'Dg',rel(F._conv_dgrad_raw(gd,wd,None,g,(H,H),0,0.),dx),'Wg',rel(F._conv_wgrad_raw(xd,gd,g),dw))
