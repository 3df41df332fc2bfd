import torch, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from lightning_gan_zoo_amd.core.models.standard_networks import Generator
from lightning_gan_zoo_amd import functional as F
from oracle import reference_cpu as O
from helpers import fill_closed_form
def rel(a,b):
    a=a.detach().double().cpu(); b=b.detach().double().cpu(); return float((a-b).abs().max()/b.abs().max())
torch.manual_seed(0)
feats, bs, nz = 64, 8, 100
g1, g2 = Generator(nz,3,feats), O.Generator(nz,3,feats)
fill_closed_form(g1,1); fill_closed_form(g2,1)
g1.cuda()
z = torch.randn(bs,nz); wv = torch.randn(bs,3,64,64)
acts1, acts2 = [], []
x1 = z.cuda().unsqueeze(-1).unsqueeze(-1); x2 = z[:, :, None, None]
for name, m in g1.net.named_children():
    if name.startswith('block'):
        x1 = m(x1); x1.retain_grad(); acts1.append((name, x1))
o1 = F.conv_transpose2d(x1, g1.net.transpose_conv_out.weight, None, F.K4S2P1, F.ACT_TANH, 0.0)
for name, m in g2.net.named_children():
    if name.startswith('block'):
        x2 = m(x2); x2.retain_grad(); acts2.append((name, x2))
o2 = torch.tanh(g2.net.transpose_conv_out(x2))
(o1*wv.cuda()).sum().backward(); (o2*wv).sum().backward()
for (n,a),(_,b) in zip(acts1, acts2):
    print(n, 'act', rel(a,b), 'grad', rel(a.grad, b.grad))
# direct BN test at block4 shape
x = torch.randn(8,128,32,32)*0.1; go = torch.randn(8,128,32,32)
bn = torch.nn.BatchNorm2d(128)
xr = x.clone().requires_grad_(); ref = torch.relu(bn(xr)); ref.backward(go)
xd = x.cuda().requires_grad_(); gd = torch.ones(128).cuda().requires_grad_(); bd = torch.zeros(128).cuda().requires_grad_()
out = F.batch_norm_act(xd, gd, bd, torch.zeros(128).cuda(), torch.ones(128).cuda(), torch.zeros((),dtype=torch.int64).cuda(), True, 0.1, 1e-5, F.ACT_RELU, 0.0)
out.backward(go.cuda())
print('bn', rel(out,ref), rel(xd.grad,xr.grad), rel(gd.grad,bn.weight.grad), rel(bd.grad,bn.bias.grad))
