"""Timing of gz_gemm on skinny shapes (the nn.Linear heads): python tools/gemm_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightning_gan_zoo_amd import functional as F  # noqa: E402


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


for M, N, K, tb in [(64, 128, 8192, True), (64, 1, 8192, True), (64, 128, 32768, True), (64, 8192, 128, False),
                    (128, 8192, 64, False), (64, 256, 16384, True), (512, 16384, 100, False)]:
    a = torch.randn(M, K, device="cuda")
    b = torch.randn(N, K, device="cuda") if tb else torch.randn(K, N, device="cuda")
    t = timeit(lambda: F.gemm(a, b, trans_b=tb))
    print("M%-4d N%-5d K%-6d %s  %7.1f us  %6.2f TF  %6.1f GB/s" % (M, N, K, "NT" if tb else "NN", t, 2.0 * M * N * K / t / 1e6,
                                                                  (M * K + N * K + M * N) * 4 / t / 1e3))
