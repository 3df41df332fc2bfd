"""One weight-gradient shape, 30 calls (run under rocprofv3 --kernel-trace --stats to see its kernels):
   python tools/wg_one.py N C H K k stride pad"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightning_gan_zoo_amd import functional as F      # noqa: E402
N, C, H, K, k, st, pd = (int(a) for a in sys.argv[1:8])
geom = F.Geom(k, k, st, pd)
OH = (H + 2 * pd - k) // st + 1
x = torch.randn(N, C, H, H, device="cuda")
gy = torch.randn(N, K, OH, OH, device="cuda")
for _ in range(30):
    F._conv_wgrad_raw(x, gy, geom)
torch.cuda.synchronize()
