import torch, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from lightning_gan_zoo_amd.core.models.standard_networks import Generator, Discriminator
from oracle import reference_cpu as O
from helpers import fill_closed_form
def rel(a,b):
    a=a.double().cpu(); b=b.double().cpu(); return float((a-b).abs().max()/b.abs().max())
torch.manual_seed(0)
for feats, bs, nz in [(8,4,16),(64,8,100)]:
    g1, g2 = Generator(nz,3,feats), O.Generator(nz,3,feats)
    fill_closed_form(g1,1); fill_closed_form(g2,1)
    g1.cuda()
    z = torch.randn(bs,nz); wv = torch.randn(bs,3,64,64)
    o1 = g1(z.cuda()); o2 = g2(z)
    print('fwd', rel(o1,o2))
    (o1*wv.cuda()).sum().backward(); (o2*wv).sum().backward()
    for (n,p),(_,q) in zip(g1.named_parameters(), g2.named_parameters()):
        print(feats, n, rel(p.grad,q.grad))
    d1, d2 = Discriminator(3,feats,final_sigmoid=False), O.Discriminator(3,feats,final_sigmoid=False)
    fill_closed_form(d1,2); fill_closed_form(d2,2); d1.cuda()
    x = torch.randn(bs,3,64,64)
    x1 = x.cuda().requires_grad_(); x2 = x.clone().requires_grad_()
    o1 = d1(x1); o2 = d2(x2); print('D fwd', rel(o1,o2))
    o1.sum().backward(); o2.sum().backward()
    print('D dx', rel(x1.grad, x2.grad))
    for (n,p),(_,q) in zip(d1.named_parameters(), d2.named_parameters()):
        print(feats, n, rel(p.grad,q.grad))
