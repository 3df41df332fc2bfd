"""G.block1 of the DCGAN generator (ConvTranspose2d k4 s1 p0 on a 1x1 input = one GEMM [bs,100] x [100,16384]) and its
two gradient GEMMs, against torch.mm.   python tools/gb1_bench.py [bs]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightning_gan_zoo_amd import functional as F      # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 512


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


z = torch.randn(bs, 100, device="cuda")
w = torch.randn(100, 16384, device="cuda")
gy = torch.randn(bs, 16384, device="cuda")
print("bs", bs)
print("fwd  z @ w        : gz %.1f us   torch %.1f us" % (timeit(lambda: F.gemm(z, w)), timeit(lambda: z @ w)))
print("dz   gy @ w^T     : gz %.1f us   torch %.1f us" % (timeit(lambda: F.gemm(gy, w, trans_b=True)), timeit(lambda: gy @ w.t())))
print("dw   z^T @ gy     : gz %.1f us   torch %.1f us" % (timeit(lambda: F.gemm(z, gy, trans_a=True)), timeit(lambda: z.t() @ gy)))
