import torch, sys, os
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from lightning_gan_zoo_amd import functional as F
g=F.K4S2P1
bs=int(sys.argv[1]) if len(sys.argv)>1 else 512
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n
layers=[('D.conv_in',3,64,64),('D.b1',64,32,128),('D.b2',128,16,256),('D.b3',256,8,512),('G.b2',512,8,1024),('G.b3',256,16,512),('G.b4',128,32,256),('G.out',3,64,128)]
print('bs',bs,'tile env',os.environ.get('GZ_TILE'))
for name,C,H,K in layers:
    x=torch.randn(bs,C,H,H,device='cuda'); w=torch.randn(K,C,4,4,device='cuda')*0.05; gy=torch.randn(bs,K,H//2,H//2,device='cuda')
    fl=2.0*bs*(H//2)**2*K*C*16
    tf=timeit(lambda: F._conv_fwd_raw(x,w,None,g,0,0.)); td=timeit(lambda: F._conv_dgrad_raw(gy,w,None,g,(H,H),0,0.)); tw=timeit(lambda: F._conv_wgrad_raw(x,gy,g))
    print('%-10s C%4d H%3d K%5d  GF %7.1f | F %7.3f ms %6.1f TF | Dg %7.3f ms %6.1f TF | Wg %7.3f ms %6.1f TF' % (name,C,H,K,fl/1e9,tf,fl/tf/1e9,td,fl/td/1e9,tw,fl/tw/1e9))
