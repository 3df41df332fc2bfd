"""Per-launch times of HoloGAN's two ConvTranspose3d layers (k3 s2 p1 op1) at bs 64: forward (= Dg kernel), input
gradient (= F kernel), weight gradient.  python tools/conv3d_bench.py [bs]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightning_gan_zoo_amd import functional as F      # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


print("bs", bs, {k: v for k, v in os.environ.items() if k.startswith("GZ_")})
for name, cin, cout, d in (("block1", 512, 128, 4), ("block2", 128, 64, 8)):
    x = torch.randn(bs, cin, d, d, d, device="cuda", requires_grad=True)
    w = (torch.randn(cin, cout, 3, 3, 3, device="cuda") * 0.05).requires_grad_()
    b = torch.zeros(cout, device="cuda", requires_grad=True)
    flop = 2.0 * bs * (2 * d) ** 3 * cin * cout * 27 / 8
    tf = timeit(lambda: F.conv_transpose3d(x.detach(), w.detach(), b.detach()))
    y = F.conv_transpose3d(x, w, b)
    gy = torch.randn_like(y)
    tall = timeit(lambda: torch.autograd.grad(y, (x, w), gy, retain_graph=True))
    tx = timeit(lambda: torch.autograd.grad(y, (x,), gy, retain_graph=True))
    tw = timeit(lambda: torch.autograd.grad(y, (w,), gy, retain_graph=True))
    print("%s %d->%d %d^3  GF %.1f | fwd %.3f ms %.1f TF | dx %.3f ms %.1f TF | dw %.3f ms %.1f TF | dx+dw %.3f" % (
        name, cin, cout, d, flop / 1e9, tf, flop / tf / 1e9, tx, flop / tx / 1e9, tw, flop / tw / 1e9, tall))
