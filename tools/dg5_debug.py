import sys, ctypes, torch
import torch.nn.functional as TF
sys.path.insert(0, "/root/repo")
from lightning_gan_zoo_amd import functional as F
from lightning_gan_zoo_amd._lib import lib
def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
for case in [(64,128,64,16), (37,208,64,8), (3,32,192,64), (16,256,128,8), (16,512,256,4), (8,128,64,32), (48,128,64,32), (1,1104,64,16), (40,32,64,16)]:
    N,K,C,OH = case; H = 2*OH
    g = torch.Generator().manual_seed(1)
    gy = torch.randn(N,K,OH,OH, generator=g); w = torch.randn(K,C,5,5, generator=g)*0.05
    ref = TF.conv_transpose2d(gy, w, None, 2, 2, output_padding=1)
    text = ctypes.create_string_buffer(256)
    lib.gz_conv2d_plan(1, N, C, H, H, K, OH, OH, 5, 5, 2, 2, text, 256)
    out = F._conv_dgrad_raw(gy.cuda(), w.cuda(), None, F.Geom(5,5,2,2), (H,H), F.ACT_NONE, 0.0).cpu()
    ph = {(py,px): rel(out[:,:,py::2,px::2], ref[:,:,py::2,px::2]) for py in (0,1) for px in (0,1)}
    chs = [rel(out[:,c0:c0+32], ref[:,c0:c0+32]) for c0 in range(0, C, 32)]
    ns = [rel(out[n0:n0+max(1,N//4)], ref[n0:n0+max(1,N//4)]) for n0 in range(0, N, max(1,N//4))]
    print(case, text.value.decode()[:70], "err %.1e" % rel(out, ref), {k: "%.0e" % v for k,v in ph.items()}, ["%.0e" % c for c in chs][:8], ["%.0e" % c for c in ns][:5])
