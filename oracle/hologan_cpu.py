"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the HoloGAN leg of the hot path.

Plain torch / numpy fp32 restatement of
  core/models/hologan_generator.py:7-345     (ZMapping, BasicBlock, Generator incl. the 3-D rigid
                                              transform + trilinear resampling, AdaIn)
  core/models/hologan_discriminator.py:7-78  (spectral-norm conv blocks, heads, truncated normal)
  core/lightning_module.py:209-237           (HOLOGAN.training_step)
  core/utils/hologan.py:3-9                  (LR schedule)
Pinned like oracle/reference_cpu.py: against fixtures generated from the unmodified reference
(tests/golden/make_golden.py hologan) by tests/test_oracle_golden.py.  Constructor call order
matches the reference so that equal seeds give equal initial parameters and spectral-norm vectors.
"""
import math

import numpy as np
import torch
from torch import nn
from torch.optim.lr_scheduler import LambdaLR


# --------------------------------------------------------------------------
# generator
# --------------------------------------------------------------------------
def adain(feat, scale, bias):
    """hologan_generator.py:333-345: per-(n,c) statistics over all spatial dims, UNBIASED variance,
    eps 1e-8, then per-(n,c) scale and bias."""
    n, c = feat.shape[:2]
    bshape = (n, c) + (1,) * (feat.dim() - 2)
    flat = feat.reshape(n, c, -1)
    mu = flat.mean(2).reshape(bshape)
    inv = torch.rsqrt(flat.var(2).reshape(bshape) + 1e-8)
    return scale.reshape(bshape) * ((feat - mu) * inv) + bias.reshape(bshape)


class ZMapping(nn.Module):
    def __init__(self, z_dimension, output_channel):
        super().__init__()
        self.output_channel = output_channel
        self.linear1 = nn.Linear(z_dimension, output_channel * 2)
        nn.init.normal_(self.linear1.weight, std=0.02)
        nn.init.constant_(self.linear1.bias, val=0.0)

    def forward(self, z):
        h = torch.relu(self.linear1(z))
        return h[:, :self.output_channel], h[:, self.output_channel:]


class UpBlock(nn.Module):
    """hologan_generator.py:20-42: transposed conv (3-D k3 s2 p1 op1, or 2-D k4 s2 p1) -> AdaIN(z) -> ReLU"""

    def __init__(self, z_planes, in_planes, out_planes, transpose_dim):
        super().__init__()
        if transpose_dim == 2:
            self.convTranspose = nn.ConvTranspose2d(in_planes, out_planes, kernel_size=4, stride=2, padding=1)
        else:
            self.convTranspose = nn.ConvTranspose3d(in_planes, out_planes, kernel_size=3, stride=2,
                                                    output_padding=1, padding=1)
        nn.init.normal_(self.convTranspose.weight, std=0.02)
        nn.init.constant_(self.convTranspose.bias, val=0.0)
        self.zMapping = ZMapping(z_planes, out_planes)

    def forward(self, h, z):
        s, b = self.zMapping(z)
        return torch.relu(adain(self.convTranspose(h), s, b))


def sample_view(args, batch_size):
    """hologan_generator.py:80-114 -- numpy global RNG, call order preserved."""
    theta = np.random.randint(args["azimuth_low"], args["azimuth_high"], (batch_size)).astype(float)
    theta = theta * math.pi / 180.0
    if args["elevation_low"] < args["elevation_high"]:
        gamma = np.random.randint(args["elevation_low"], args["elevation_high"], (batch_size)).astype(float)
        gamma = gamma * math.pi / 180.0
    else:
        gamma = np.zeros(batch_size).astype(float)
    scale = float(np.random.uniform(args["scale_low"], args["scale_high"]))
    shifts = []
    for ax in ("X", "Y", "Z"):
        lo, hi = args["trans%s_low" % ax], args["trans%s_high" % ax]
        shifts.append(lo + np.random.random(batch_size) * (hi - lo))
    view = np.zeros((batch_size, 6))
    view[:, 0], view[:, 1], view[:, 2] = theta, gamma, scale
    view[:, 3], view[:, 4], view[:, 5] = shifts
    return view


def _mat(rows):
    return torch.cat([torch.cat(r, dim=2) for r in rows], dim=1)


def view_matrices(view, size=16, new_size=16):
    """Inverse of  C_new . (T . S . (Rz . Ry)) . C_old  per sample, fp32, same association order as
    hologan_generator.py:145-214."""
    view = torch.as_tensor(view)
    # always float32, as the reference hard-codes it (:148-186): the clamped-corner interpolation is DISCONTINUOUS
    # at the volume faces (x = -0 gives weights ~0, x = +0 gives the voxel), which right-angle views hit exactly,
    # so the coordinates must be the reference's fp32 ones even in make_golden.py's float64 conditioning run
    dt = torch.float32
    col = lambda i: view[:, i].reshape(-1, 1, 1).to(dt)       # noqa: E731
    th, ga, sc, tx, ty, tz = (col(i) for i in range(6))
    one, zero = torch.ones_like(th), torch.zeros_like(th)
    rot_z = _mat([[th.cos(), th.sin(), zero, zero], [-th.sin(), th.cos(), zero, zero],
                  [zero, zero, one, zero], [zero, zero, zero, one]])
    rot_y = _mat([[ga.cos(), zero, ga.sin(), zero], [zero, one, zero, zero],
                  [-ga.sin(), zero, ga.cos(), zero], [zero, zero, zero, one]])
    rot = torch.matmul(rot_z, rot_y)
    scl = _mat([[sc, zero, zero, zero], [zero, sc, zero, zero], [zero, zero, sc, zero], [zero, zero, zero, one]])
    trn = _mat([[one, zero, zero, tx], [zero, one, zero, ty], [zero, zero, one, tz], [zero, zero, zero, one]])
    m = torch.matmul(torch.matmul(trn, scl), rot)
    n = view.shape[0]

    def centre(v):
        c = torch.tensor([[1, 0, 0, v], [0, 1, 0, v], [0, 0, 1, v], [0, 0, 0, 1.0]], dtype=dt)
        return c.reshape(1, 4, 4).repeat(n, 1, 1)

    full = torch.matmul(torch.matmul(centre(new_size * 0.5), m), centre(-size * 0.5))
    return full.inverse()


def resample_coords(inv, new_size=16):
    """source coordinates (x, y, z), each [N * S^3], of every output voxel (flat order z, y, x)."""
    z, y, x = torch.meshgrid(torch.arange(new_size), torch.arange(new_size), torch.arange(new_size), indexing="ij")
    dt = inv.dtype
    grid = torch.cat([x.reshape(1, -1).to(dt), y.reshape(1, -1).to(dt), z.reshape(1, -1).to(dt),
                      torch.ones(1, new_size ** 3, dtype=dt)], dim=0)
    pts = torch.matmul(inv, grid.reshape(1, 4, -1).repeat(inv.shape[0], 1, 1))
    return pts[:, 0, :].reshape(-1), pts[:, 1, :].reshape(-1), pts[:, 2, :].reshape(-1)


def trilinear_indices(vox_shape, x, y, z):
    """The eight clamped corner indices (into the [N*D*H*W, C] flattening) and weights, in the
    reference's a..h order (hologan_generator.py:245-311).  Weights use the CLAMPED corners, so
    they can be negative / not sum to one outside the volume -- preserved."""
    n, c, d0, d1, d2 = vox_shape          # dims 2,3,4 are indexed by z, y, x
    lo = [torch.floor(v).long() for v in (x, y, z)]
    x0, y0, z0 = (torch.clamp(v, 0, lim - 1) for v, lim in zip(lo, (d2, d1, d0)))
    x1, y1, z1 = (torch.clamp(v + 1, 0, lim - 1) for v, lim in zip(lo, (d2, d1, d0)))
    base = (torch.arange(n) * (d0 * d1 * d2)).reshape(-1, 1).repeat(1, x.numel() // n).reshape(-1)
    fx0, fx1, fy0, fy1, fz0, fz1 = (t.to(x.dtype) for t in (x0, x1, y0, y1, z0, z1))
    wx = (fx1 - x, x - fx0)
    wy = (fy1 - y, y - fy0)
    wz = (fz1 - z, z - fz0)
    xs, ys, zs = (x0, x1), (y0, y1), (z0, z1)
    idx, wts = [], []
    for kz in (0, 1):
        for kx in (0, 1):
            for ky in (0, 1):
                idx.append(base + zs[kz] * (d1 * d2) + ys[ky] * d2 + xs[kx])
                wts.append(wx[kx] * wy[ky] * wz[kz])
    return idx, wts


def rigid_resample(vox, view, size=16, new_size=16):
    """transformation3d + apply_transformation + interpolation: [N,C,16,16,16] -> [N,C,16,16,16]."""
    n, c = vox.shape[:2]
    x, y, z = resample_coords(view_matrices(view, size, new_size), new_size)     # float32 coordinates
    idx, wts = trilinear_indices(vox.shape, x, y, z)
    wts = [w.to(vox.dtype) for w in wts]
    flat = vox.permute(0, 2, 3, 4, 1).reshape(-1, c)
    out = None
    for i, w in zip(idx, wts):
        term = w.unsqueeze(1) * flat[i]
        out = term if out is None else out + term
    return out.reshape(n, new_size, new_size, new_size, c).permute(0, 4, 1, 2, 3)


class Generator(nn.Module):
    def __init__(self, in_planes, out_planes, z_planes, view_args, img_size, view_planes=6, gpu=True,
                 ext128=False):
        super().__init__()
        self.x = nn.Parameter((torch.randn(1, in_planes * 8, 4, 4, 4) - 0.5) / 0.5)
        self.view_args = view_args
        self.zMapping = ZMapping(z_planes, in_planes * 8)
        self.block1 = UpBlock(z_planes, in_planes * 8, in_planes * 2, 3)
        self.block2 = UpBlock(z_planes, in_planes * 2, in_planes, 3)
        self.convTranspose2d1 = nn.ConvTranspose2d(in_planes * 16, in_planes * 16, kernel_size=1)
        nn.init.normal_(self.convTranspose2d1.weight, std=0.02)
        nn.init.constant_(self.convTranspose2d1.bias, val=0.0)
        self.block3 = UpBlock(z_planes, in_planes * 16, in_planes * 4, 2)
        self.block4 = UpBlock(z_planes, in_planes * 4, in_planes, 2)
        if img_size == 64:
            self.final_layer = nn.Conv2d(in_planes, out_planes, kernel_size=3, padding=1)
        elif img_size == 128:
            # reference: stride 1 (65x65, unusable); ext128: the stride-2 extension the product offers
            self.final_layer = nn.ConvTranspose2d(in_planes, out_planes, kernel_size=4, padding=1,
                                                  stride=2 if ext128 else 1)
        nn.init.normal_(self.final_layer.weight, std=0.02)
        nn.init.constant_(self.final_layer.bias, val=0.0)

    def sample_view(self, batch_size):
        return sample_view(self.view_args, batch_size)

    def forward(self, z, view_in=None):
        n = z.shape[0]
        if view_in is None:
            view_in = self.sample_view(n)
        s0, b0 = self.zMapping(z)
        h = torch.relu(adain(self.x.repeat(n, 1, 1, 1, 1), s0, b0))
        h = self.block2(self.block1(h, z), z)
        h = rigid_resample(h, view_in)
        # swap the two middle spatial axes, mirror the new first one, fold it into channels (:130-133)
        h = h.permute(0, 1, 3, 2, 4).flip(2).reshape(n, -1, 16, 16)
        h = torch.relu(self.convTranspose2d1(h))
        h = self.block4(self.block3(h, z), z)
        return torch.tanh(self.final_layer(h))


# --------------------------------------------------------------------------
# discriminator
# --------------------------------------------------------------------------
def truncated_normal_(weight, mean=0, std=0.02):
    """hologan_discriminator.py:72-78: first of 4 normal draws inside (-2, 2), scaled."""
    draws = weight.new_empty(tuple(weight.shape) + (4,)).normal_()
    ok = (draws < 2) & (draws > -2)
    first = ok.max(-1, keepdim=True)[1]
    weight.data.copy_(draws.gather(-1, first).squeeze(-1))
    weight.data.mul_(std).add_(mean)


class SNBlock(nn.Module):
    """spectral-norm conv k5 s2 p2 -> InstanceNorm2d (no affine) -> LeakyReLU(0.2)"""

    def __init__(self, in_planes, out_planes):
        super().__init__()
        self.conv2d = nn.Conv2d(in_planes, out_planes, kernel_size=5, stride=2, padding=2)
        truncated_normal_(self.conv2d.weight)
        nn.init.constant_(self.conv2d.bias, val=0.0)
        self.conv2d_spec_norm = nn.utils.spectral_norm(self.conv2d)     # same module, second name
        self.instance_norm = nn.InstanceNorm2d(out_planes)
        self.lrelu = nn.LeakyReLU(0.2)

    def forward(self, x):
        return self.lrelu(self.instance_norm(self.conv2d_spec_norm(x)))


class Discriminator(nn.Module):
    def __init__(self, in_planes, out_planes, z_planes, img_size=64):
        super().__init__()
        self.conv2d = nn.Conv2d(in_planes, out_planes, kernel_size=5, stride=2, padding=2)
        truncated_normal_(self.conv2d.weight)
        nn.init.constant_(self.conv2d.bias, val=0.0)
        self.lrelu = nn.LeakyReLU(0.2)
        self.blocks = nn.Sequential(SNBlock(out_planes, out_planes * 2), SNBlock(out_planes * 2, out_planes * 4),
                                    SNBlock(out_planes * 4, out_planes * 8))
        feat = out_planes * 8 * (img_size // 16) ** 2
        self.linear1 = nn.Linear(feat, 1)
        truncated_normal_(self.linear1.weight)
        nn.init.constant_(self.linear1.bias, val=0.0)
        self.linear2 = nn.Linear(feat, 128)
        truncated_normal_(self.linear1.weight)      # reference quirk (:46): linear2's weight keeps its default init
        nn.init.constant_(self.linear2.bias, val=0.0)
        self.linear3 = nn.Linear(128, z_planes)
        truncated_normal_(self.linear1.weight)      # same quirk (:50)
        nn.init.constant_(self.linear3.bias, val=0.0)

    def forward(self, x):
        h = self.blocks(self.lrelu(self.conv2d(x))).reshape(x.size(0), -1)
        logit = self.linear1(h)
        z_pred = torch.tanh(self.linear3(self.lrelu(self.linear2(h))))
        return logit, z_pred


# --------------------------------------------------------------------------
# step + scheduler
# --------------------------------------------------------------------------
def create_hologan_lr_scheduler(total_epochs, optimizer):
    half = total_epochs / 2
    return LambdaLR(optimizer, lambda epoch: 1 if epoch <= half else 1 - ((epoch - half) / half))


def hologan_training_step(self, batch, batch_idx, optimizer_idx):
    """core/lightning_module.py:210-237"""
    real, _ = batch
    z = self._noise(len(real))
    fake = self.generator(z)
    if optimizer_idx == 0:
        out_r, _ = self.discriminator(real)
        out_f, z_pred = self.discriminator(fake.detach())
        loss = (self.criterion(out_r, torch.ones_like(out_r)) + self.criterion(out_f, torch.zeros_like(out_f))) / 2
        q = torch.mean((z_pred - z) ** 2)
        self.log("train/d_loss", loss)
        self.log("train/q_loss", q)
        return loss + q
    if optimizer_idx == 1:
        out, z_pred = self.discriminator(fake)
        loss = self.criterion(out, torch.ones_like(out))
        q = torch.mean((z_pred - z) ** 2)
        self.log("train/g_loss", loss)
        self.log("train/q_loss", q)
        return loss + q
