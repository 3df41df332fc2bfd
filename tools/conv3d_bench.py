"""F / Dg / Wg timing of HoloGAN's two ConvTranspose3d layers (bs 64): python tools/conv3d_bench.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightning_gan_zoo_amd import functional as F  # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


# ConvTranspose3d(Cin -> Cout), input D^3 -> (2D)^3; in conv terms K = Cin (feature side), C = Cout (image side)
for name, cin, cout, d in [("512->128 4^3->8^3", 512, 128, 4), ("128->64 8^3->16^3", 128, 64, 8)]:
    g = torch.randn(bs, cin, d, d, d, device="cuda")
    w = torch.randn(cin, cout, 3, 3, 3, device="cuda") * 0.05
    x = torch.randn(bs, cout, 2 * d, 2 * d, 2 * d, device="cuda")
    fl = 2.0 * bs * d ** 3 * cin * cout * 27
    td = timeit(lambda: F._conv3d_dgrad_raw(g, w, None, 0, 0.))
    tf = timeit(lambda: F._conv3d_fwd_raw(x, w, None, 0, 0.))
    tw = timeit(lambda: F._conv3d_wgrad_raw(x, g, 3))
    print("%-20s GF %5.1f | Dg %6.1f us %5.1f TF | F %6.1f us %5.1f TF | Wg %6.1f us %5.1f TF"
          % (name, fl / 1e9, td, fl / td / 1e6, tf, fl / tf / 1e6, tw, fl / tw / 1e6))
