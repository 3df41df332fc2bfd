"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the R1-regularised ResNet path.

Plain ``torch.nn`` fp32 restatement of
  ResNet Generator / Discriminator / ResnetBlock ... core/submodules/gan_stability/models/resnet.py:9-133
  GANStabilityR1.training_step .................... core/lightning_module.py:130-156
  compute_grad2 .................................... core/utils/utils.py:60-69
pinned by tests/golden/gan_stability_r1_*.npz, which tests/golden/make_golden.py generates by running the
unmodified reference classes.  Same import rule as oracle/reference_cpu.py: tests, smoke and the bench's
cpu_baseline leg only.
"""
import math

import torch
from torch import nn
from torch.nn import functional as TF

SLOPE = 0.2          # resnet.py:131-133
RESIDUAL_GAIN = 0.1  # resnet.py:122


class ResnetBlock(nn.Module):
    """resnet.py:96-129: out = shortcut(x) + 0.1 * conv_1(lrelu(conv_0(lrelu(x)))); the shortcut is a bias-free
    1x1 conv when the channel count changes, the identity otherwise.  Sub-module creation order conv_0, conv_1,
    conv_s fixes the default-init RNG stream."""

    def __init__(self, fin, fout, fhidden=None, is_bias=True):
        super().__init__()
        self.fin, self.fout = fin, fout
        self.fhidden = min(fin, fout) if fhidden is None else fhidden
        self.learned_shortcut = fin != fout
        self.conv_0 = nn.Conv2d(fin, self.fhidden, 3, 1, 1)
        self.conv_1 = nn.Conv2d(self.fhidden, fout, 3, 1, 1, bias=is_bias)
        if self.learned_shortcut:
            self.conv_s = nn.Conv2d(fin, fout, 1, 1, 0, bias=False)

    def forward(self, x):
        skip = self.conv_s(x) if self.learned_shortcut else x
        h = self.conv_0(TF.leaky_relu(x, SLOPE))
        h = self.conv_1(TF.leaky_relu(h, SLOPE))
        return skip + RESIDUAL_GAIN * h


def _widths(size, nf, nf_max, s0=4):
    levels = int(math.log2(size / s0))
    return levels, [min(nf * 2 ** i, nf_max) for i in range(levels + 1)]


class Generator(nn.Module):
    """resnet.py:9-51: fc -> [nf0,4,4] -> (block, nearest x2) * levels -> block(nf,nf) -> lrelu -> conv3x3 -> tanh."""

    def __init__(self, z_dim, nlabels, size, embed_size=256, nfilter=64, nfilter_max=512, **kwargs):
        super().__init__()
        self.z_dim, self.s0, self.nf = z_dim, 4, nfilter
        levels, w = _widths(size, nfilter, nfilter_max)
        self.nf0 = w[levels]
        self.fc = nn.Linear(z_dim, self.nf0 * 16)
        stack = []
        for i in range(levels, 0, -1):
            stack += [ResnetBlock(w[i], w[i - 1]), nn.Upsample(scale_factor=2)]
        stack.append(ResnetBlock(nfilter, nfilter))
        self.resnet = nn.Sequential(*stack)
        self.conv_img = nn.Conv2d(nfilter, 3, 3, padding=1)

    def forward(self, z):
        z = z.squeeze(-1).squeeze(-1)
        x = self.fc(z).view(z.size(0), self.nf0, 4, 4)
        x = self.resnet(x)
        return torch.tanh(self.conv_img(TF.leaky_relu(x, SLOPE)))


class Discriminator(nn.Module):
    """resnet.py:54-93: conv3x3 -> block(nf,nf) -> (avgpool(3,2,1), block) * levels -> lrelu -> fc -> sigmoid.
    The stack is created before conv_img and fc (:69-82)."""

    def __init__(self, z_dim, nlabels, size, embed_size=256, nfilter=64, nfilter_max=1024):
        super().__init__()
        self.s0, self.nf = 4, nfilter
        levels, w = _widths(size, nfilter, nfilter_max)
        self.nf0 = w[levels]
        stack = [ResnetBlock(nfilter, nfilter)]
        for i in range(levels):
            stack += [nn.AvgPool2d(3, stride=2, padding=1), ResnetBlock(w[i], w[i + 1])]
        self.conv_img = nn.Conv2d(3, nfilter, 3, padding=1)
        self.resnet = nn.Sequential(*stack)
        self.fc = nn.Linear(self.nf0 * 16, nlabels)
        self.final_sigmoid = nn.Sigmoid()

    def forward(self, x):
        h = self.resnet(self.conv_img(x))
        h = h.view(x.size(0), self.nf0 * 16)
        return self.final_sigmoid(self.fc(TF.leaky_relu(h, SLOPE)).squeeze(1))


def compute_grad2(d_out, x_in):
    """utils.py:60-69: per-sample squared L2 norm of d(sum d_out)/d x_in, kept differentiable."""
    (g,) = torch.autograd.grad(outputs=d_out.sum(), inputs=x_in, create_graph=True, retain_graph=True,
                               only_inputs=True)
    assert g.size() == x_in.size()
    return g.pow(2).view(x_in.size(0), -1).sum(1)


def r1_training_step(step, batch, batch_idx, optimizer_idx):
    """lightning_module.py:130-156."""
    real, _ = batch
    fake = step.generator(step._noise(len(real)))
    if optimizer_idx == 0:
        real.requires_grad_()                                           # :139
        out_r = step.discriminator(real).reshape(-1)
        out_f = step.discriminator(fake.detach()).reshape(-1)
        loss_r = step.criterion(out_r, torch.ones_like(out_r))
        loss_f = step.criterion(out_f, torch.zeros_like(out_f))
        reg = step.cfg["loss_weight"]["reg"] * compute_grad2(out_r, real).mean()
        loss = reg + (loss_r + loss_f)                                  # :147 (no /2 here)
        step.log("train/d_loss", loss)
        return loss
    if optimizer_idx == 1:
        out = step.discriminator(fake).reshape(-1)
        loss = step.criterion(out, torch.ones_like(out))
        step.log("train/g_loss", loss)
        return loss
