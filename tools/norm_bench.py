"""HBM throughput of the BatchNorm+ReLU passes on the DCGAN generator shapes (bs 512): forward = statistics pass
(1 read) + apply pass (1 read, 1 write); backward = row-sums pass (2 reads) + apply pass (2 reads, 1 write).

    python tools/norm_bench.py [batch]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightning_gan_zoo_amd import functional as F  # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 512


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
    return s.elapsed_time(e) / n * 1e3


for name, C, H in [("G.block2 512@8x8", 512, 8), ("G.block3 256@16x16", 256, 16), ("G.block4 128@32x32", 128, 32),
                   ("D.block1 128@16x16", 128, 16)]:
    x = torch.randn(bs, C, H, H, device="cuda", requires_grad=True)
    g, b = torch.ones(C, device="cuda", requires_grad=True), torch.zeros(C, device="cuda", requires_grad=True)
    rm, rv, nbt = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), torch.zeros((), dtype=torch.int64, device="cuda")
    go = torch.randn(bs, C, H, H, device="cuda")
    mb = x.numel() * 4 / 1e6

    def fwd():
        return F.batch_norm_act(x, g, b, rm, rv, nbt, True, 0.1, 1e-5, F.ACT_RELU, 0.0)

    y = fwd()
    tf = timeit(lambda: fwd())
    tfb = timeit(lambda: torch.autograd.grad(fwd(), (x, g, b), go))
    tb = tfb - tf
    print("%-20s %6.1f MB | fwd %6.1f us %5.2f TB/s (3 passes) | bwd %6.1f us %5.2f TB/s (5 passes)"
          % (name, mb, tf, 3 * mb / tf, tb, 5 * mb / tb))
