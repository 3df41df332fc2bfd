"""DCGAN / WGAN / WGAN-GP generator and discriminator on the MI355X HIP kernels.

Drop-in for the reference's core/models/standard_networks.py:9-93: same constructor
signatures, same ``state_dict`` keys and shapes (``net.block1.transpose_conv.weight``,
``disc.block2.batch_norm.running_var`` ...), same default initialisation and the same
host-RNG consumption order, so Lightning checkpoints interchange.  The child modules
(nn.ConvTranspose2d, nn.Conv2d, nn.BatchNorm2d, nn.InstanceNorm2d) are parameter holders
only: ``forward`` never calls them, it calls the fused HIP ops in ``functional``:

    generator block     : conv_transpose2d (MFMA dgrad-form implicit GEMM) -> BatchNorm+ReLU
    discriminator block : conv2d (MFMA implicit GEMM) -> BatchNorm|InstanceNorm + LeakyReLU(0.2)
    first G layer       : 1x1 -> 4x4 transposed conv == plain GEMM [N,nz] x [nz, 16*C]
    last  D layer       : 4x4 valid conv == one dot product per sample
"""
import math
from collections import OrderedDict

import torch
from torch import nn

from ... import functional as F


class _GBlock(nn.Module):
    """transpose_conv -> batch_norm -> relu   (reference standard_networks.py:78-89)"""

    def __init__(self, cin, cout, stride, padding):
        super().__init__()
        self.transpose_conv = nn.ConvTranspose2d(cin, cout, 4, stride, padding, bias=False)
        self.batch_norm = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU()
        self.stride, self.padding = stride, padding

    def forward(self, x):
        w = self.transpose_conv.weight
        bn = self.batch_norm
        F.ready(w, bn.weight, bn.bias)      # data parallel: this layer's gradient bucket + optimizer step have landed
        training = bn.training or bn.running_mean is None
        stats = None
        if self.stride == 1 and self.padding == 0 and x.shape[2] == 1 and x.shape[3] == 1:
            n = x.shape[0]
            y = F.matmul(x.reshape(n, -1), w.view(w.shape[0], -1), w).view(n, w.shape[1], 4, 4)
        elif self.stride == 2 and self.padding == 1:
            if training:        # the transposed convolution's epilogue also emits the BatchNorm partial sums
                y, stats = F.conv_transpose2d_with_stats(x, w, F.K4S2P1)
            else:
                y = F.conv_transpose2d(x, w, None, F.K4S2P1)
        else:
            raise RuntimeError("unsupported transposed-convolution geometry on the HIP path")
        return F.batch_norm_act(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                training, bn.momentum, bn.eps, F.ACT_RELU, 0.0, stats)


class _DBlock(nn.Module):
    """conv -> batch_norm | instance_norm2d | identity -> leaky_relu   (standard_networks.py:34-50)"""

    def __init__(self, cin, cout, norm):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 4, 2, 1, bias=False)
        if norm == "batch_norm":
            self.batch_norm = nn.BatchNorm2d(cout)
        elif norm == "instance_norm2d":
            self.instance_norm2d = nn.InstanceNorm2d(cout, affine=True)
        else:
            self.identity = nn.Identity()
        self.leaky_relu = nn.LeakyReLU(0.2)
        self.norm = norm

    def forward(self, x, groups=1):
        slope = self.leaky_relu.negative_slope
        F.ready(*self.parameters())
        if self.norm == "batch_norm":
            bn = self.batch_norm
            stats = None
            if bn.training:     # statistics from the convolution's own epilogue
                y, stats = F.conv2d_with_stats(x, self.conv.weight, F.K4S2P1)
            else:
                y = F.conv2d(x, self.conv.weight, None, F.K4S2P1)
            return F.batch_norm_act(y, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                    bn.num_batches_tracked, bn.training, bn.momentum, bn.eps, F.ACT_LRELU, slope, stats,
                                    groups)
        if self.norm == "instance_norm2d":
            y = F.conv2d(x, self.conv.weight, None, F.K4S2P1)
            inn = self.instance_norm2d
            return F.instance_norm_act(y, inn.weight, inn.bias, inn.eps, F.ACT_LRELU, slope)
        return F.conv2d(x, self.conv.weight, None, F.K4S2P1, F.ACT_LRELU, slope)


class _DiscStack(nn.Sequential):
    def forward(self, x, groups=1):
        F.ready(self.conv_in.weight)
        x = F.conv2d(x, self.conv_in.weight, None, F.K4S2P1, F.ACT_LRELU, self.leaky_relu.negative_slope)
        for name, m in self.named_children():
            if name.startswith("block"):
                x = m(x, groups)
        w = self.conv_out.weight
        F.ready(w)
        if tuple(x.shape[2:]) != tuple(w.shape[2:]):
            raise RuntimeError("conv_out expects a %dx%d map, got %s" % (w.shape[2], w.shape[3], tuple(x.shape)))
        x = F.full_dot_conv(x, w)
        if hasattr(self, "sigmoid"):
            x = torch.sigmoid(x)
        return x


class Discriminator(nn.Module):
    supports_stacked_batches = True      # forward(x, groups=G): G batches stacked along n, own BatchNorm statistics each
    gates_parameters = True              # every layer announces its parameters with F.ready before reading them

    def __init__(self, channels_img, features_d, norm="batch_norm", img_size=64, final_sigmoid=True):
        super().__init__()
        self.norm = norm
        n_blocks = int(math.log2(img_size // 8))
        # creation order as in the reference (:15-30): strided blocks, then conv_in, then conv_out
        blocks = [(f"block{i}", _DBlock(features_d * 2 ** (i - 1), features_d * 2 ** i, norm))
                  for i in range(1, n_blocks + 1)]
        conv_in = nn.Conv2d(channels_img, features_d, kernel_size=4, stride=2, padding=1, bias=False)
        conv_out = nn.Conv2d(features_d * 2 ** n_blocks, 1, kernel_size=4, stride=2, padding=0, bias=False)
        tail = ("sigmoid", nn.Sigmoid()) if final_sigmoid else ("identity", nn.Identity())
        self.disc = _DiscStack(OrderedDict(
            [("conv_in", conv_in), ("leaky_relu", nn.LeakyReLU(0.2))] + blocks + [("conv_out", conv_out), tail]))

    def forward(self, x, groups=1):
        """``groups`` > 1: x holds that many batches stacked along n, each normalised with its OWN BatchNorm statistics
        (and the running buffers updated batch after batch) -- D(cat(real, fake), groups=2) is D(real), D(fake) of the
        reference (core/lightning_module.py:112-119) in one pass.  Instance / no normalisation: per-sample anyway."""
        if groups > 1 and self.norm == "batch_norm" and not self.training:
            raise RuntimeError("stacked batches with their own statistics exist in training mode only")
        return self.disc(x, groups)


class _GenStack(nn.Sequential):
    def forward(self, x):
        for name, m in self.named_children():
            if name.startswith("block"):
                x = m(x)
        F.ready(self.transpose_conv_out.weight)
        return F.conv_transpose2d(x, self.transpose_conv_out.weight, None, F.K4S2P1, F.ACT_TANH, 0.0)


class Generator(nn.Module):
    gates_parameters = True              # see Discriminator

    def __init__(self, channels_noise, channels_img, features_g, img_size=64):
        super().__init__()
        n_blocks = int(math.log2(img_size / 4))
        widths = [channels_noise] + [features_g * 2 ** (n_blocks - i) for i in range(n_blocks)]
        layers = [(f"block{i + 1}", _GBlock(widths[i], widths[i + 1], 1 if i == 0 else 2, 0 if i == 0 else 1))
                  for i in range(n_blocks)]
        layers.append(("transpose_conv_out", nn.ConvTranspose2d(features_g * 2, channels_img, kernel_size=4,
                                                                stride=2, padding=1, bias=False)))
        layers.append(("tanh", nn.Tanh()))
        self.net = _GenStack(OrderedDict(layers))

    def forward(self, x):
        return self.net(x.unsqueeze(-1).unsqueeze(-1))
