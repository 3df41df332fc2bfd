import torch, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from lightning_gan_zoo_amd import functional as F
g=F.K4S2P1
bs=512
layers={'D.b1':(64,32,128),'G.b3':(256,16,512),'G.b4':(128,32,256),'G.b2':(512,8,1024),'G.out':(3,64,128)}
C,H,K=layers[sys.argv[1]]
x=torch.randn(bs,C,H,H,device='cuda'); w=torch.randn(K,C,4,4,device='cuda')*0.05; gy=torch.randn(bs,K,H//2,H//2,device='cuda')
for _ in range(3):
    F._conv_fwd_raw(x,w,None,g,0,0.); F._conv_dgrad_raw(gy,w,None,g,(H,H),0,0.); F._conv_wgrad_raw(x,gy,g)
torch.cuda.synchronize()
