"""AdaIN / InstanceNorm + activation, forward and backward, at HoloGAN's bs-64 shapes.  python tools/norm_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightning_gan_zoo_amd import functional as F      # noqa: E402


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


print({k: v for k, v in os.environ.items() if k.startswith("GZ_")})
for shape in ((64, 128, 8, 8, 8), (64, 64, 16, 16, 16), (64, 256, 32, 32), (64, 64, 64, 64), (64, 128, 16, 16), (64, 512, 4, 4)):
    x = torch.randn(*shape, device="cuda", requires_grad=True)
    sb = torch.rand(shape[0], 2 * shape[1], device="cuda", requires_grad=True)
    go = torch.randn(*shape, device="cuda")
    mb = x.numel() * 4 / 1e6
    tf = timeit(lambda: F.adain_act_packed(x.detach(), sb.detach(), 1e-8, F.ACT_RELU))
    y = F.adain_act_packed(x, sb, 1e-8, F.ACT_RELU)
    tb = timeit(lambda: torch.autograd.grad(y, (x, sb), go, retain_graph=True))
    print("%-22s %6.1f MB | fwd %.3f ms %.2f TB/s (r+w) | bwd %.3f ms %.2f TB/s (2r+w)" % (
        shape, mb, tf, 2 * mb / tf / 1e6, tb, 3 * mb / tb / 1e6))
