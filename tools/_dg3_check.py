import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightning_gan_zoo_amd import functional as F
torch.manual_seed(0)
for bs, cin, cout, d in ((8, 512, 128, 4), (8, 128, 64, 8), (64, 512, 128, 4), (3, 512, 128, 4), (8, 256, 96, 4)):
    x = torch.randn(bs, cin, d, d, d, device="cuda")
    w = torch.randn(cin, cout, 3, 3, 3, device="cuda") * 0.05
    b = torch.randn(cout, device="cuda")
    y = F.conv_transpose3d(x, w, b)
    ref = torch.nn.functional.conv_transpose3d(x.double(), w.double(), b.double(), stride=2, padding=1, output_padding=1)
    err = ((y.double() - ref).norm() / ref.norm()).item()
    print(os.environ.get("GZ_DG3_EVEN_SPLIT"), bs, cin, cout, d, "rel err %.2e" % err, "max abs %.2e" % (y.double() - ref).abs().max().item())
