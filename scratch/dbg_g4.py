import torch, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from lightning_gan_zoo_amd.core.models.standard_networks import Generator, Discriminator
from oracle import reference_cpu as O
from helpers import fill_closed_form
def rel(a,b):
    a=a.detach().double().cpu(); b=b.detach().double().cpu(); return float((a-b).abs().max()/b.abs().max())
torch.manual_seed(0)
for shift in (0.0, 8.0):
  for feats, bs, nz in [(64,8,100)]:
    g1, g2 = Generator(nz,3,feats), O.Generator(nz,3,feats)
    fill_closed_form(g1,1); fill_closed_form(g2,1)
    with torch.no_grad():
        for g in (g1,g2):
            for n,p in g.named_parameters():
                if n.endswith('batch_norm.bias'): p.add_(shift)
    g1.cuda()
    z = torch.randn(bs,nz); wv = torch.randn(bs,3,64,64)
    o1 = g1(z.cuda()); o2 = g2(z)
    print('shift', shift, 'fwd', rel(o1,o2))
    (o1*wv.cuda()).sum().backward(); (o2*wv).sum().backward()
    for (n,p),(_,q) in zip(g1.named_parameters(), g2.named_parameters()):
        print(feats, n, rel(p.grad,q.grad))
