"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the G+D training-step hot path.

A plain ``torch.nn`` / fp32 restatement of the reference algorithm, written so
that it can travel to the GPU box (the reference's own files cannot).  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module; the product package (``lightning_gan_zoo_amd``) never
does and fails loudly when its HIP library is missing.

Pinning: the reference holds no tests or golden vectors, so
this oracle is pinned against outputs of the reference itself, imported
unmodified in the build container by ``tests/golden/make_golden.py`` (through
``oracle/ref_import.py``); the resulting vectors live in ``tests/golden/*.npz``
and ``tests/test_oracle_golden.py`` checks this file against them on every run.

Reference citations (relative to /root/reference):
  Generator / Discriminator ...... core/models/standard_networks.py:9-93
  DCGAN / WGAN / WGANGP steps .... core/lightning_module.py:104-128,158-207
  configure_optimizers ........... core/lightning_module.py:75-87
  gradient_penalty ............... core/utils/utils.py:39-58
  GANStabilityR1 + ResNet G/D .... oracle/resnet_cpu.py
"""
import math
from collections import OrderedDict

import torch
from torch import nn


# --------------------------------------------------------------------------
# networks (core/models/standard_networks.py)
# --------------------------------------------------------------------------
def _d_stage(cin, cout, norm):
    # standard_networks.py:34-50 -- conv(k4,s2,p1,no bias) -> norm -> LeakyReLU(0.2)
    if norm == "batch_norm":
        mid = ("batch_norm", nn.BatchNorm2d(cout))
    elif norm == "instance_norm2d":
        mid = ("instance_norm2d", nn.InstanceNorm2d(cout, affine=True))
    else:
        mid = ("identity", nn.Identity())
    return nn.Sequential(OrderedDict([
        ("conv", nn.Conv2d(cin, cout, 4, 2, 1, bias=False)),
        mid,
        ("leaky_relu", nn.LeakyReLU(0.2)),
    ]))


class Discriminator(nn.Module):
    """standard_networks.py:9-53.  Parameter creation order follows the
    reference (the strided blocks are built before ``conv_in``/``conv_out``,
    :15-30) so that equal seeds give equal default initialisations."""

    def __init__(self, channels_img, features_d, norm="batch_norm", img_size=64,
                 final_sigmoid=True):
        super().__init__()
        self.norm = norm
        depth = int(math.log2(img_size // 8))
        stages = [(f"block{i}", _d_stage(features_d * 2 ** (i - 1), features_d * 2 ** i, norm))
                  for i in range(1, depth + 1)]
        conv_in = nn.Conv2d(channels_img, features_d, 4, 2, 1, bias=False)
        conv_out = nn.Conv2d(features_d * 2 ** depth, 1, 4, 2, 0, bias=False)
        last = ("sigmoid", nn.Sigmoid()) if final_sigmoid else ("identity", nn.Identity())
        self.disc = nn.Sequential(OrderedDict(
            [("conv_in", conv_in), ("leaky_relu", nn.LeakyReLU(0.2))] + stages +
            [("conv_out", conv_out), last]))

    def forward(self, x):
        return self.disc(x)


def _g_stage(cin, cout, stride, pad):
    # standard_networks.py:78-89 -- convT(k4,no bias) -> BatchNorm -> ReLU
    return nn.Sequential(OrderedDict([
        ("transpose_conv", nn.ConvTranspose2d(cin, cout, 4, stride, pad, bias=False)),
        ("batch_norm", nn.BatchNorm2d(cout)),
        ("relu", nn.ReLU()),
    ]))


class Generator(nn.Module):
    """standard_networks.py:55-93: z[N,C] -> 1x1 map -> 4x4 -> ... -> img."""

    def __init__(self, channels_noise, channels_img, features_g, img_size=64):
        super().__init__()
        depth = int(math.log2(img_size / 4))
        widths = [channels_noise] + [features_g * 2 ** (depth - i) for i in range(depth)]
        layers = []
        for i in range(depth):
            layers.append((f"block{i + 1}", _g_stage(widths[i], widths[i + 1],
                                                      1 if i == 0 else 2, 0 if i == 0 else 1)))
        layers.append(("transpose_conv_out",
                       nn.ConvTranspose2d(features_g * 2, channels_img, 4, 2, 1, bias=False)))
        layers.append(("tanh", nn.Tanh()))
        self.net = nn.Sequential(OrderedDict(layers))

    def forward(self, x):
        return self.net(x[:, :, None, None])


# --------------------------------------------------------------------------
# loss utilities (core/utils/utils.py)
# --------------------------------------------------------------------------
def gradient_penalty(critic, real, fake, device="cpu", alpha=None):
    """utils.py:39-58.  ``alpha`` may be injected ([N,1,1,1]) so that tests do not
    depend on the host RNG stream; by default it is drawn exactly like the
    reference does (host generator, one value per sample)."""
    n, c, h, w = real.shape
    if alpha is None:
        alpha = torch.rand((n, 1, 1, 1))
    alpha = alpha.repeat(1, c, h, w).to(device)
    mix = real * alpha + fake * (1 - alpha)
    mix.requires_grad_()
    score = critic(mix)
    (grad,) = torch.autograd.grad(outputs=score, inputs=mix,
                                  grad_outputs=torch.ones_like(score),
                                  create_graph=True, retain_graph=True)
    norm = grad.view(n, -1).norm(2, dim=1)
    return torch.mean((norm - 1) ** 2)


# --------------------------------------------------------------------------
# step logic (core/lightning_module.py)
# --------------------------------------------------------------------------
def _locate(path):
    import importlib
    parts = path.split(".")
    for n in range(len(parts) - 1, 0, -1):
        try:
            obj = importlib.import_module(".".join(parts[:n]))
        except ModuleNotFoundError:
            continue
        for name in parts[n:]:
            obj = getattr(obj, name)
        return obj
    raise ImportError(path)


def _build(node, *args, **kwargs):
    node = dict(node)
    fn = _locate(node.pop("_target_"))
    return fn(*args, **{**node, **kwargs})


class _StepBase(nn.Module):
    """BaseGAN (lightning_module.py:35-102) minus data loading / figures.
    Construction order D, G, criterion, noise distribution, fixed noise (:38-50)
    is kept because it fixes the host RNG stream."""

    def __init__(self, cfg, logging_dir=None):
        super().__init__()
        self.cfg = cfg
        self.logging_dir = logging_dir
        self.discriminator = _build(cfg["discriminator"])
        self.generator = _build(cfg["generator"])
        self.criterion = _build(cfg["train"]["criterion"])
        self.noise_distn = _build(cfg["model"]["noise_distn"])
        self.fixed_noise = self.noise_distn.sample((8, cfg["model"]["noise_dim"]))
        self.logged = {}

    @property
    def device(self):
        return next(self.parameters()).device

    def log(self, key, value, *a, **k):
        self.logged[key] = value.detach().clone()

    def _noise(self, n):
        return self.noise_distn.sample((n, self.cfg["model"]["noise_dim"])).to(self.device)

    def configure_optimizers(self):
        # lightning_module.py:75-87 -- discriminator entry first
        opt_d = _build(self.cfg["disc_optimiser"], self.discriminator.parameters())
        opt_g = _build(self.cfg["gen_optimiser"], self.generator.parameters())
        sch = self.cfg["optimisation"]["lr_scheduler"]
        return ({"optimizer": opt_d, "lr_scheduler": _build(sch, optimizer=opt_d),
                 "frequency": self.cfg["optimisation"]["disc_freq"]},
                {"optimizer": opt_g, "lr_scheduler": _build(sch, optimizer=opt_g),
                 "frequency": self.cfg["optimisation"]["gen_freq"]})


class DCGAN(_StepBase):
    def training_step(self, batch, batch_idx, optimizer_idx):
        real, _ = batch                                   # :106
        fake = self.generator(self._noise(len(real)))     # :107-109, both branches
        if optimizer_idx == 0:                            # :112-121
            out_r = self.discriminator(real).reshape(-1)
            out_f = self.discriminator(fake.detach()).reshape(-1)
            loss = (self.criterion(out_r, torch.ones_like(out_r)) +
                    self.criterion(out_f, torch.zeros_like(out_f))) / 2
            self.log("train/d_loss", loss)
            return loss
        if optimizer_idx == 1:                            # :124-128
            out = self.discriminator(fake).reshape(-1)
            loss = self.criterion(out, torch.ones_like(out))
            self.log("train/g_loss", loss)
            return loss


class WGAN(_StepBase):
    def training_step(self, batch, batch_idx, optimizer_idx):
        clip = self.cfg["train"]["weight_clip"]           # :160-162, every call
        for p in self.discriminator.parameters():
            p.data.clamp_(-clip, clip)
        real, _ = batch
        fake = self.generator(self._noise(len(real)))
        if optimizer_idx == 0:                            # :170-175
            out_r = self.discriminator(real).reshape(-1)
            out_f = self.discriminator(fake.detach()).reshape(-1)
            loss = -(torch.mean(out_r) - torch.mean(out_f))
            self.log("train/d_loss", loss)
            return loss
        if optimizer_idx == 1:                            # :178-182
            loss = -torch.mean(self.discriminator(fake).reshape(-1))
            self.log("train/g_loss", loss)
            return loss


class WGANGP(_StepBase):
    gp_alpha = None   # test hook: inject alpha [N,1,1,1]

    def training_step(self, batch, batch_idx, optimizer_idx):
        real, _ = batch
        fake = self.generator(self._noise(len(real)))
        if optimizer_idx == 0:                            # :192-201; fake NOT detached in GP
            out_r = self.discriminator(real).reshape(-1)
            out_f = self.discriminator(fake.detach()).reshape(-1)
            gp = gradient_penalty(self.discriminator, real, fake, device=self.device,
                                  alpha=self.gp_alpha)
            loss = self.cfg["loss_weight"]["lambda_gp"] * gp - (torch.mean(out_r) - torch.mean(out_f))
            self.log("train/d_loss", loss)
            return loss
        if optimizer_idx == 1:                            # :204-207
            loss = -torch.mean(self.discriminator(fake).reshape(-1))
            self.log("train/g_loss", loss)
            return loss


class GANStabilityR1(_StepBase):
    """core/lightning_module.py:130-156 (restated in oracle/resnet_cpu.py)"""

    def training_step(self, batch, batch_idx, optimizer_idx):
        from .resnet_cpu import r1_training_step
        return r1_training_step(self, batch, batch_idx, optimizer_idx)


class HOLOGAN(_StepBase):
    """core/lightning_module.py:209-237 (restated in oracle/hologan_cpu.py)"""

    def training_step(self, batch, batch_idx, optimizer_idx):
        from .hologan_cpu import hologan_training_step
        return hologan_training_step(self, batch, batch_idx, optimizer_idx)


# --------------------------------------------------------------------------
# harness semantics (Lightning's per-batch optimizer alternation + toggle)
# --------------------------------------------------------------------------
def toggle(step, optimizer_idx):
    """Lightning's toggle_optimizer: only the active network's parameters
    require grad during a training_step."""
    for p in step.discriminator.parameters():
        p.requires_grad_(optimizer_idx == 0)
    for p in step.generator.parameters():
        p.requires_grad_(optimizer_idx == 1)


def run_step(step, optimizers, batch, batch_idx, optimizer_idx, apply_update=True):
    toggle(step, optimizer_idx)
    loss = step.training_step(batch, batch_idx, optimizer_idx)
    loss.backward()
    if apply_update:
        opt = optimizers[optimizer_idx]["optimizer"]
        opt.step()
        opt.zero_grad()
    return loss.detach()


# --------------------------------------------------------------------------
# ``_target_`` namespaces so that a config composed with
# module_root="oracle.reference_cpu" resolves to the classes above
# --------------------------------------------------------------------------
class models:                      # noqa: N801
    class standard_networks:       # noqa: N801
        pass


class lightning_module:            # noqa: N801
    pass


class utils:                       # noqa: N801
    class hologan:                 # noqa: N801
        pass


def _wire_hologan():
    from . import hologan_cpu as H

    class hologan_generator:       # noqa: N801
        Generator = H.Generator

    class hologan_discriminator:   # noqa: N801
        Discriminator = H.Discriminator

    models.hologan_generator = hologan_generator
    models.hologan_discriminator = hologan_discriminator
    utils.hologan.create_hologan_lr_scheduler = staticmethod(H.create_hologan_lr_scheduler)


def _wire_resnet():
    from . import resnet_cpu as R

    class resnet:                  # noqa: N801
        Generator = R.Generator
        Discriminator = R.Discriminator

    class _models:                 # noqa: N801
        pass

    class gan_stability:           # noqa: N801
        models = _models

    _models.resnet = resnet
    submodules.gan_stability = gan_stability


class submodules:                  # noqa: N801
    pass


_wire_hologan()
_wire_resnet()
lightning_module.GANStabilityR1 = GANStabilityR1
lightning_module.HOLOGAN = HOLOGAN
models.standard_networks.Generator = Generator
models.standard_networks.Discriminator = Discriminator
lightning_module.DCGAN = DCGAN
lightning_module.WGAN = WGAN
lightning_module.WGANGP = WGANGP
