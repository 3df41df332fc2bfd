"""Per-layer timing of the three implicit-GEMM ops on the R1 ResNet layer shapes (conf/expt/gan_stability_r1.yaml:
nfilter 16, 128x128, bs 64), with the HBM floor (inputs + outputs once) next to each time.

    python tools/resnet_bench.py [batch]
"""
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
    return s.elapsed_time(e) / n


# (name, C, K, H, k)
layers = [("img3->16", 3, 16, 128, 3), ("16->16@128", 16, 16, 128, 3), ("16->32@64", 16, 32, 64, 3),
          ("32->32@64", 32, 32, 64, 3), ("32->64@32", 32, 64, 32, 3), ("64->64@32", 64, 64, 32, 3),
          ("64->128@16", 64, 128, 16, 3), ("128->128@16", 128, 128, 16, 3), ("128->256@8", 128, 256, 8, 3),
          ("256->256@8", 256, 256, 8, 3), ("256->512@4", 256, 512, 4, 3), ("512->512@4", 512, 512, 4, 3),
          ("s16->32@64", 16, 32, 64, 1), ("s64->128@16", 64, 128, 16, 1), ("s256->512@4", 256, 512, 4, 1),
          ("16->3@128", 16, 3, 128, 3)]
print("bs", bs)
for name, C, K, H, k in layers:
    g = F.K3S1P1 if k == 3 else F.K1S1P0
    x = torch.randn(bs, C, H, H, device="cuda")
    w = torch.randn(K, C, k, k, device="cuda") * 0.05
    gy = torch.randn(bs, K, H, H, device="cuda")
    fl = 2.0 * bs * H * H * K * C * k * k
    floor_us = (x.numel() + gy.numel()) * 4 / 8e12 * 1e6
    tf = timeit(lambda: F._conv_fwd_raw(x, w, None, g, 0, 0.))
    td = timeit(lambda: F._conv_dgrad_raw(gy, w, None, g, (H, H), 0, 0.))
    tw = timeit(lambda: F._conv_wgrad_raw(x, gy, g))
    print("%-13s GF %6.2f  hbm floor %6.1f us | F %7.1f us %5.1f TF | Dg %7.1f us %5.1f TF | Wg %7.1f us %5.1f TF"
          % (name, fl / 1e9, floor_us, tf * 1e3, fl / tf / 1e9, td * 1e3, fl / td / 1e9, tw * 1e3, fl / tw / 1e9))
