import torch, sys, copy
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import reference_cpu as O
from helpers import fill_closed_form
def rel(a,b):
    a=a.detach().double(); b=b.detach().double(); return float((a-b).abs().max()/b.abs().max())
torch.manual_seed(0)
feats, bs, nz = 64, 8, 100
g32 = O.Generator(nz,3,feats); fill_closed_form(g32,1)
g64 = copy.deepcopy(g32).double()
z = torch.randn(bs,nz); wv = torch.randn(bs,3,64,64)
def run(g, z, wv):
    acts=[]; x = z[:, :, None, None]
    for name, m in g.net.named_children():
        if name.startswith('block'):
            x = m(x); x.retain_grad(); acts.append((name,x))
    o = torch.tanh(g.net.transpose_conv_out(x))
    (o*wv).sum().backward()
    return acts
a32 = run(g32, z, wv); a64 = run(g64, z.double(), wv.double())
for (n,a),(_,b) in zip(a32,a64):
    print(n, 'act', rel(a,b), 'grad', rel(a.grad,b.grad))
for (n,p),(_,q) in zip(g32.named_parameters(), g64.named_parameters()):
    print(n, rel(p.grad,q.grad))
# how close to zero are pre-activations? 
x = z[:, :, None, None]
with torch.no_grad():
    for name, m in g32.net.named_children():
        if name.startswith('block'):
            y = m.batch_norm(m.transpose_conv(x)); print(name, 'frac |z|<1e-5:', float((y.abs()<1e-5).float().mean()), 'numel', y.numel(), 'mean/std of conv out', float((m.transpose_conv(x).mean((0,2,3)).abs()/m.transpose_conv(x).std((0,2,3))).max()))
            x = torch.relu(y)
