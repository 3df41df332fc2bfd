"""R1-regularised ResNet generator / discriminator on the MI355X HIP kernels.

Drop-in for reference core/submodules/gan_stability/models/resnet.py:9-133: same constructor signatures, same
``state_dict`` keys (``fc.*``, ``resnet.<2i>.conv_{0,1,s}.*``, ``conv_img.*``), same parameter creation order
(so equal seeds give equal default initialisations).  The nn.Conv2d / nn.Linear children hold parameters only;
``forward`` runs on libgz_hip.so:

    conv_0 / conv_1 / conv_img : 3x3 s1 p1 MFMA implicit GEMM, bias (+ LeakyReLU / tanh) in the epilogue
    conv_s                     : 1x1 implicit GEMM
    actvn, x_s + 0.1*dx        : one HBM pass each (gz_act_fwd, gz_axpby)
    AvgPool2d(3,2,1), Upsample : gz_avgpool3s2_*, gz_upsample2_*

Every op is differentiable to second order (compute_grad2 differentiates D's input gradient again).
"""
import math

import torch
from torch import nn

from ..... import functional as F

SLOPE = 2e-1


def actvn(x):
    """resnet.py:131-133"""
    return F.activation(x, F.ACT_LRELU, SLOPE)


class ResnetBlock(nn.Module):
    """resnet.py:96-129"""

    def __init__(self, fin, fout, fhidden=None, is_bias=True):
        super().__init__()
        self.is_bias = is_bias
        self.learned_shortcut = (fin != fout)
        self.fin = fin
        self.fout = fout
        self.fhidden = min(fin, fout) if fhidden is None else fhidden
        self.conv_0 = nn.Conv2d(self.fin, self.fhidden, 3, stride=1, padding=1)
        self.conv_1 = nn.Conv2d(self.fhidden, self.fout, 3, stride=1, padding=1, bias=is_bias)
        if self.learned_shortcut:
            self.conv_s = nn.Conv2d(self.fin, self.fout, 1, stride=1, padding=0, bias=False)

    def forward(self, x):
        x_s = self._shortcut(x)
        # conv_0's output is only ever seen through actvn, so the LeakyReLU rides in its epilogue
        h = F.conv2d(actvn(x), self.conv_0.weight, self.conv_0.bias, F.K3S1P1, F.ACT_LRELU, SLOPE)
        dx = F.conv2d(h, self.conv_1.weight, self.conv_1.bias, F.K3S1P1)
        return F.add_scaled(x_s, dx, 0.1)

    def _shortcut(self, x):
        if self.learned_shortcut:
            return F.conv2d(x, self.conv_s.weight, None, F.K1S1P0)
        return x


class Upsample2(nn.Upsample):
    """nn.Upsample(scale_factor=2) (nearest) served by gz_upsample2_fwd / _bwd."""

    def forward(self, x):
        if self.mode != "nearest" or float(self.scale_factor) != 2.0:
            raise RuntimeError("the HIP path implements nearest-neighbour 2x upsampling only")
        return F.upsample2(x)


class AvgPool3s2(nn.AvgPool2d):
    """nn.AvgPool2d(3, stride=2, padding=1) served by gz_avgpool3s2_fwd / _bwd."""

    def forward(self, x):
        if (self.kernel_size, self.stride, self.padding) != (3, 2, 1) or not self.count_include_pad:
            raise RuntimeError("the HIP path implements AvgPool2d(3, stride=2, padding=1) only")
        return F.avg_pool3s2(x)


class Generator(nn.Module):
    """resnet.py:9-51"""

    def __init__(self, z_dim, nlabels, size, embed_size=256, nfilter=64, nfilter_max=512, **kwargs):
        super().__init__()
        s0 = self.s0 = 4
        nf = self.nf = nfilter
        nf_max = self.nf_max = nfilter_max
        self.z_dim = z_dim
        nlayers = int(math.log2(size / s0))
        self.nf0 = min(nf_max, nf * 2 ** nlayers)
        self.fc = nn.Linear(z_dim, self.nf0 * s0 * s0)
        blocks = []
        for i in range(nlayers):
            nf0 = min(nf * 2 ** (nlayers - i), nf_max)
            nf1 = min(nf * 2 ** (nlayers - i - 1), nf_max)
            blocks += [ResnetBlock(nf0, nf1), Upsample2(scale_factor=2)]
        blocks += [ResnetBlock(nf, nf)]
        self.resnet = nn.Sequential(*blocks)
        self.conv_img = nn.Conv2d(nf, 3, 3, padding=1)

    def forward(self, z):
        z = z.squeeze(-1).squeeze(-1)
        batch_size = z.size(0)
        out = F.linear_act(z, self.fc.weight, self.fc.bias)
        out = out.view(batch_size, self.nf0, self.s0, self.s0)
        out = self.resnet(out)
        # tanh(conv_img(actvn(out))): bias and tanh in the conv epilogue
        return F.conv2d(actvn(out), self.conv_img.weight, self.conv_img.bias, F.K3S1P1, F.ACT_TANH, 0.0)


class Discriminator(nn.Module):
    """resnet.py:54-93"""

    def __init__(self, z_dim, nlabels, size, embed_size=256, nfilter=64, nfilter_max=1024, **root_config_keys):
        # ``img_size`` / ``final_sigmoid`` arrive from the reference's root config (conf/config.yaml:35-37); its own class
        # (resnet.py:55) takes no **kwargs and cannot be constructed from the shipped tree -- ignored here, as the
        # reference's Generator (:10) already does
        super().__init__()
        self.embed_size = embed_size
        s0 = self.s0 = 4
        nf = self.nf = nfilter
        nf_max = self.nf_max = nfilter_max
        nlayers = int(math.log2(size / s0))
        self.nf0 = min(nf_max, nf * 2 ** nlayers)
        blocks = [ResnetBlock(nf, nf)]
        for i in range(nlayers):
            nf0 = min(nf * 2 ** i, nf_max)
            nf1 = min(nf * 2 ** (i + 1), nf_max)
            blocks += [AvgPool3s2(3, stride=2, padding=1), ResnetBlock(nf0, nf1)]
        self.conv_img = nn.Conv2d(3, 1 * nf, 3, padding=1)
        self.resnet = nn.Sequential(*blocks)
        self.fc = nn.Linear(self.nf0 * s0 * s0, nlabels)
        self.final_sigmoid = nn.Sigmoid()

    def forward(self, x):
        batch_size = x.size(0)
        out = F.conv2d(x, self.conv_img.weight, self.conv_img.bias, F.K3S1P1)
        out = self.resnet(out)
        out = out.view(batch_size, self.nf0 * self.s0 * self.s0)
        out = F.linear(actvn(out), self.fc.weight, self.fc.bias)
        out = out.squeeze(1)
        return torch.sigmoid(out)
