"""Per-launch times of HoloGAN's discriminator convolutions (k5 s2 p2) at bs 64.  python tools/conv5_bench.py [bs]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightning_gan_zoo_amd import functional as F      # noqa: E402

g = F.Geom(5, 5, 2, 2)
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
for name, C, H, K in (("conv_in", 3, 64, 64), ("block0", 64, 32, 128), ("block1", 128, 16, 256), ("block2", 256, 8, 512)):
    x = torch.randn(bs, C, H, H, device="cuda")
    w = torch.randn(K, C, 5, 5, device="cuda") * 0.05
    gy = torch.randn(bs, K, H // 2, H // 2, device="cuda")
    fl = 2.0 * bs * (H // 2) ** 2 * K * C * 25
    tf = timeit(lambda: F._conv_fwd_raw(x, w, None, g, 0, 0.))
    td = timeit(lambda: F._conv_dgrad_raw(gy, w, None, g, (H, H), 0, 0.))
    tw = timeit(lambda: F._conv_wgrad_raw(x, gy, g))
    print("%-8s C%4d H%3d K%4d GF %6.2f | F %6.3f ms %6.1f TF | Dg %6.3f ms %6.1f TF | Wg %6.3f ms %6.1f TF" % (
        name, C, H, K, fl / 1e9, tf, fl / tf / 1e9, td, fl / td / 1e9, tw, fl / tw / 1e9))
