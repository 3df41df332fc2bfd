"""Forward timing of the tap-major layers (HoloGAN's 5x5 s2 p2 critic blocks, a 3x3 s1 p1 residual-block conv) with a
warm clock:   python tools/tap_bench.py [bs]        (GZ_NO_IGEMM2_TAP=1: the round-2 kernels)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightning_gan_zoo_amd import functional as F      # noqa: E402
from lightning_gan_zoo_amd._lib import lib      # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64


def timeit(fn, n=30, warm_s=0.3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < warm_s:
        for _ in range(10):
            fn()
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


layers = [("holo128 D.b1", 64, 64, 128, 5, 2, 2), ("holo128 D.b2", 128, 32, 256, 5, 2, 2), ("holo128 D.b3", 256, 16, 512, 5, 2, 2),
          ("holo64 D.b1", 64, 32, 128, 5, 2, 2), ("holo64 D.b2", 128, 16, 256, 5, 2, 2), ("holo64 D.b3", 256, 8, 512, 5, 2, 2),
          ("res 3x3 256@32", 256, 32, 256, 3, 1, 1), ("res 3x3 128@64", 128, 64, 128, 3, 1, 1),
          ("holo 1x1 1024@16", 1024, 16, 1024, 1, 1, 0)]
print("bs", bs, "GZ_NO_IGEMM2_TAP", os.environ.get("GZ_NO_IGEMM2_TAP"))
for name, C, H, K, k, st, pd in layers:
    geom = F.Geom(k, k, st, pd)
    OH = (H + 2 * pd - k) // st + 1
    x = torch.randn(bs, C, H, H, device="cuda")
    w = torch.randn(K, C, k, k, device="cuda") * 0.05
    fl = 2.0 * bs * OH * OH * K * C * k * k
    t = timeit(lambda: F._conv_fwd_raw(x, w, None, geom, 0, 0.))
    lab = F._TILES[lib.gz_conv2d_tile(0, bs, C, H, H, K, OH, OH, k, k, st)]
    line = "%-16s C%4d H%3d K%4d  %6.1f GF  F %-8s %7.3f ms %6.1f TF" % (name, C, H, K, fl / 1e9, lab, t, fl / t / 1e9)
    if True:
        gy = torch.randn(bs, K, OH, OH, device="cuda")
        t = timeit(lambda: F._conv_dgrad_raw(gy, w, None, geom, (H, H), 0, 0.))
        lab = F._TILES[lib.gz_conv2d_tile(1, bs, C, H, H, K, OH, OH, k, k, st)]
        line += " | Dg %-8s %7.3f ms %6.1f TF" % (lab, t, fl / t / 1e9)
    if k != 1:
        gy = torch.randn(bs, K, OH, OH, device="cuda")
        t = timeit(lambda: F._conv_wgrad_raw(x, gy, geom))
        lab = F._TILES[lib.gz_conv2d_tile(2, bs, C, H, H, K, OH, OH, k, k, st)]
        line += " | Wg %-8s %7.3f ms %6.1f TF" % (lab, t, fl / t / 1e9)
    print(line)
