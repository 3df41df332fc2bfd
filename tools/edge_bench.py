"""Timing of the DCGAN's small-channel edge layers (3 <-> 64 channels, k4 s2 p1) with a warm clock, beside the HBM
bound of each (bytes that must move / 8 TB/s):

    python tools/edge_bench.py [bs]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightning_gan_zoo_amd import functional as F      # noqa: E402

g = F.K4S2P1
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 512


def timeit(fn, n=50, warm_s=0.3):
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


C, K, H = 3, 64, 64
x = torch.randn(bs, C, H, H, device="cuda")
w = torch.randn(K, C, 4, 4, device="cuda") * 0.05
gy = torch.randn(bs, K, H // 2, H // 2, device="cuda")
gy2 = torch.randn(bs, 2 * K, H // 2, H // 2, device="cuda")
w2 = torch.randn(2 * K, C, 4, 4, device="cuda") * 0.05
mb = (x.numel() + gy.numel()) * 4 / 1e6
print("bs %d: image side %d x %d x %d, feature side %d x %d x %d, %.0f MB one pass -> %.1f us at 8 TB/s" %
      (bs, C, H, H, K, H // 2, H // 2, mb, mb / 8.0))
for name, fn in [("F   3 -> 64 (D.block1 forward)", lambda: F._conv_fwd_raw(x, w, None, g, 0, 0.)),
                 ("Dg 64 -> 3  (G.block5 forward, D.block1 backward-data)", lambda: F._conv_dgrad_raw(gy, w, None, g, (H, H), 0, 0.)),
                 ("Dg 64 -> 3 + tanh (G.block5 forward)", lambda: F._conv_dgrad_raw(gy, w, None, g, (H, H), F.ACT_TANH, 0.)),
                 ("Dg 128 -> 3 + tanh (G.block5 forward at features_g 64: 293 MB)",
                  lambda: F._conv_dgrad_raw(gy2, w2, None, g, (H, H), F.ACT_TANH, 0.)),
                 ("Wg 128 x 48 (G.block5)", lambda: F._conv_wgrad_raw(x, gy2, g)),
                 ("Wg 64 x 48  (both edge layers)", lambda: F._conv_wgrad_raw(x, gy, g))]:
    t = timeit(fn)
    print("%-58s %7.1f us  %5.2f TB/s" % (name, t * 1e3, mb / t / 1e3))

# the same transposed convolution right behind a kernel that has just WRITTEN its input (as inside the step, where
# the producer is the normalisation pass): pair time minus producer time
buf = torch.empty_like(gy)
src = torch.randn_like(gy)
ta = timeit(lambda: torch.clamp(src, min=0, out=buf))
tb = timeit(lambda: (torch.clamp(src, min=0, out=buf), F._conv_dgrad_raw(buf, w, None, g, (H, H), F.ACT_TANH, 0.)))
print("producer %.1f us, producer + Dg %.1f us -> Dg behind its producer %.1f us" % (ta * 1e3, tb * 1e3, (tb - ta) * 1e3))
tc = timeit(lambda: (torch.neg(src, out=buf), F._conv_dgrad_raw(buf, w, None, g, (H, H), F.ACT_TANH, 0.)))
ta2 = timeit(lambda: torch.neg(src, out=buf))
print("dense producer %.1f us, + Dg %.1f us -> %.1f us" % (ta2 * 1e3, tc * 1e3, (tc - ta2) * 1e3))
