"""What the vendor fp32 GEMM (torch.matmul -> hipBLASLt / rocBLAS) reaches on this chip for GEMMs of the same size as
the DCGAN layers' implicit GEMMs: a yardstick for the hand-written igemm kernels (which also gather / scatter)."""
import torch

torch.backends.cuda.matmul.allow_tf32 = False


def timeit(fn, n=10):
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


for M, N, K in ((32768, 512, 4096), (131072, 256, 2048), (8192, 1024, 8192), (131072, 128, 1024), (32768, 256, 2048),
                (8192, 8192, 8192), (16384, 4096, 4096)):
    a = torch.randn(M, K, device="cuda")
    b = torch.randn(K, N, device="cuda")
    bt = torch.randn(N, K, device="cuda")
    t1 = timeit(lambda: a @ b)
    t2 = timeit(lambda: a @ bt.t())
    fl = 2.0 * M * N * K
    print("M%7d N%5d K%5d  NN %.3f ms %6.1f TF/s | NT %.3f ms %6.1f TF/s" % (M, N, K, t1, fl / t1 / 1e9, t2, fl / t2 / 1e9))
