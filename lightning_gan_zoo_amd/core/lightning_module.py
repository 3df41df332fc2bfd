"""Step classes of the hot path -- DCGAN, WGAN, WGANGP (HOLOGAN: see hologan section) -- with the
reference's LightningModule surface (core/lightning_module.py:35-237):

    __init__(cfg, logging_dir)
    training_step(batch, batch_idx, optimizer_idx) -> loss tensor carrying an autograd graph
    configure_optimizers() -> (dict_D, dict_G) with keys optimizer / lr_scheduler / frequency

The generator / discriminator built from ``cfg`` are the HIP-kernel modules of
``lightning_gan_zoo_amd.core.models``; the arithmetic of every line below that touches an
image-sized tensor runs in libgz_hip.so.
"""
from abc import abstractmethod

import torch

from .. import functional as F
from ..config import instantiate
from ..harness import LightningModule, draw_on_host
from .utils.utils import VerboseShapeExecution, compute_grad2, gradient_penalty


class _HostImageTransform:
    """Resize((S, S)) -> ToTensor -> Normalize(mean, std) for installations without torchvision: PIL bilinear resize,
    HWC uint8 -> CHW float in [0, 1], then (x - mean) / std.  Tensors pass through (synthetic / tensor datasets)."""

    def __init__(self, size, mean, std, channels):
        self.size, self.mean, self.std, self.channels = size, float(mean), float(std), channels

    def __call__(self, img):
        if torch.is_tensor(img):
            return img
        import numpy as np
        from PIL import Image
        if not isinstance(img, Image.Image):
            img = Image.fromarray(np.asarray(img))
        img = img.resize((self.size, self.size), Image.BILINEAR)
        a = np.asarray(img, dtype=np.float32)
        a = a[:, :, None] if a.ndim == 2 else a
        return torch.from_numpy(a).permute(2, 0, 1).div(255.0).sub(self.mean).div(self.std)


def _build_transform(cfg):
    """Resize -> ToTensor -> Normalize (reference :42-47); torchvision's when it is installed."""
    t = cfg.train
    try:
        from torchvision import transforms
    except Exception:  # noqa: BLE001
        return _HostImageTransform(t.img_size, t.data_mean, t.data_std, t.channels_img)
    return transforms.Compose([
        transforms.Resize((t.img_size, t.img_size)),
        transforms.ToTensor(),
        transforms.Normalize(mean=[t.data_mean for _ in range(t.channels_img)],
                             std=[t.data_std for _ in range(t.channels_img)])])


class BaseGAN(LightningModule):
    real_first = True        # discriminator steps run D(real) before G(z) (see DCGAN.training_step)
    # Round 4: a discriminator step applies D ONCE to the stacked batch [real; fake] with two BatchNorm statistics
    # groups (standard_networks.Discriminator.forward(x, groups=2)) instead of twice -- the same arithmetic as the
    # reference's two calls (per-call batch statistics, running buffers updated real then fake), half the launches and
    # twice the rows per launch; the weight gradients of the two passes come out of one launch already summed.  Tests
    # that replay the reference's mask decisions in ITS call order turn it off.
    stack_d_passes = True

    def __init__(self, cfg, logging_dir=None):
        super().__init__()
        # construction order fixes the host RNG stream: D, then G, then fixed_noise (reference :38-50)
        self.discriminator = instantiate(cfg.discriminator)
        self.generator = instantiate(cfg.generator)
        self.cfg = cfg
        self.logging_dir = logging_dir
        self.transform = _build_transform(cfg)
        self.criterion = instantiate(cfg.train.criterion)
        self.noise_distn = instantiate(cfg.model.noise_distn)
        self.fixed_noise = self.noise_distn.sample((8, cfg.model.noise_dim))
        if cfg.debug.verbose_shape:                      # reference :53-54
            self.apply(VerboseShapeExecution)

    @abstractmethod
    def training_step(self, batch, batch_idx, optimizer_idx):
        pass

    def criterion_const(self, logits, value):
        """``self.criterion(logits, ones_like(logits))`` / ``zeros_like`` (reference :114-119,126): the shipped
        criterion (BCEWithLogitsLoss, mean reduction, conf/config.yaml:19-20) against a constant target is one fused
        launch forward and one backward; any other criterion object is called as the reference calls it."""
        c = self.criterion
        if (isinstance(c, torch.nn.BCEWithLogitsLoss) and c.reduction == "mean" and c.weight is None
                and c.pos_weight is None and logits.is_cuda):
            return F.bce_logits_mean(logits, value)
        return c(logits, torch.full_like(logits, value))

    def stacked_discriminator(self, real, fake):
        """D(cat(real, fake.detach()), groups=2).reshape(-1) -- [logits(real); logits(fake)] -- when the discriminator
        supports stacked batches, else None (the caller takes the reference's two calls)."""
        d = self.discriminator
        if not (self.stack_d_passes and getattr(d, "supports_stacked_batches", False) and real.is_cuda and d.training
                and real.shape == fake.shape):
            return None
        return d(torch.cat([real, fake.detach()]), groups=2).reshape(-1)

    def sample_noise(self, n):
        # drawn on the host generator, then copied to the device (reference :107-108); single host
        # thread + pinned staging, see harness.few_host_threads / HostStager
        return draw_on_host(lambda: self.noise_distn.sample((n, self.cfg.model.noise_dim)), self.device)

    def validation_step(self, batch, batch_idx):
        real, _ = batch
        return {"real": real}

    def validation_epoch_end(self, outputs):
        """reference :64-73: the first validation batch's reals and the generator's images for the 8 ``fixed_noise``
        latents drawn at construction, each as a normalised grid, logged as 'Real' / 'Fake' when a TensorBoard-style
        logger is attached (``self.logger.experiment.add_image``); the two grids are also returned."""
        from ..eval import make_grid
        real = outputs[0]["real"][:len(self.fixed_noise)]
        noise = self.fixed_noise.to(self.device)
        with torch.no_grad():
            fake = self.generator(noise)
        img_grid_real = make_grid(real, normalize=True)
        img_grid_fake = make_grid(fake, normalize=True)
        experiment = getattr(getattr(self, "logger", None), "experiment", None)
        if experiment is not None:
            experiment.add_image("Real", img_grid_real, self.current_epoch)
            experiment.add_image("Fake", img_grid_fake, self.current_epoch)
        return img_grid_real, img_grid_fake

    def configure_optimizers(self):
        # the reference's nodes target torch.optim.Adam / RMSprop (conf/expt/*.yaml); on the GPU they are
        # served by the fused HIP implementations with the same arguments and state layout
        # (lightning_gan_zoo_amd.optim); `fused_optimizer: false` in the config keeps torch's own
        d_params, g_params = list(self.discriminator.parameters()), list(self.generator.parameters())
        d_node, g_node = self.cfg.disc_optimiser, self.cfg.gen_optimiser
        if self.cfg.get("fused_optimizer", True):
            from ..optim import fused_node
            d_node, g_node = fused_node(d_node, d_params), fused_node(g_node, g_params)
        opt_disc = instantiate(d_node, d_params)
        opt_gen = instantiate(g_node, g_params)
        scheduler_disc = instantiate(self.cfg.optimisation.lr_scheduler, optimizer=opt_disc)
        scheduler_gen = instantiate(self.cfg.optimisation.lr_scheduler, optimizer=opt_gen)
        return ({"optimizer": opt_disc, "lr_scheduler": scheduler_disc,
                 "frequency": self.cfg.optimisation.disc_freq},
                {"optimizer": opt_gen, "lr_scheduler": scheduler_gen,
                 "frequency": self.cfg.optimisation.gen_freq})


    # -- data hooks Lightning's ``trainer.fit(model)`` calls (reference :89-102): the dataset node of the config
    #    (``torchvision.datasets.ImageFolder`` in conf/dataset/celeb_a.yaml) with this module's transform, the
    #    configured batch size and worker count, NO shuffling, incomplete last batch kept.  The batches are host
    #    tensors; Lightning moves them to ``self.device`` before ``training_step``.
    def _loader(self, split):
        from torch.utils.data import DataLoader
        dataset = instantiate(self.cfg.dataset[split], transform=self.transform)
        return DataLoader(dataset, num_workers=self.cfg.train.num_workers, batch_size=self.cfg.train.batch_size)

    def train_dataloader(self):
        return self._loader("train")

    def val_dataloader(self):
        return self._loader("val")

    def test_dataloader(self):
        return self._loader("test")


class DCGAN(BaseGAN):
    def training_step(self, batch, batch_idx, optimizer_idx):
        real, _ = batch
        noise = self.sample_noise(len(real))         # host RNG order of the reference: the noise is drawn first

        if optimizer_idx == 0:      # discriminator (reference :112-121)
            # D(real) is launched BEFORE G(z) (`real_first`): it does not read the generator, so under data
            # parallelism the generator's last gradient bucket + optimizer step (ddp.GradSync finalizes them in the
            # generator's forward-pre hook) hide behind it.  Same kernels on the same operands, D's norm buffers
            # still see real then fake: bit-identical to the reference's order (tests: test_real_first_order_...)
            c = self.criterion
            if (self.stack_d_passes and isinstance(c, torch.nn.BCEWithLogitsLoss) and c.reduction == "mean"
                    and c.weight is None and c.pos_weight is None):
                fake = self.generator(noise)
                logits = self.stacked_discriminator(real, fake)
                if logits is not None:
                    loss_disc = F.bce_logits_pair_mean(logits, 1.0, 0.0)      # (BCE(real, 1) + BCE(fake, 0)) / 2
                    self.log("train/d_loss", loss_disc)
                    return loss_disc
                disc_real = self.discriminator(real).reshape(-1)
            elif self.real_first:
                disc_real = self.discriminator(real).reshape(-1)
                fake = self.generator(noise)
            else:
                fake = self.generator(noise)
                disc_real = self.discriminator(real).reshape(-1)
            loss_disc_real = self.criterion_const(disc_real, 1.0)
            disc_fake = self.discriminator(fake.detach()).reshape(-1)
            loss_disc_fake = self.criterion_const(disc_fake, 0.0)
            loss_disc = (loss_disc_real + loss_disc_fake) / 2
            self.log("train/d_loss", loss_disc)
            return loss_disc

        if optimizer_idx == 1:      # generator (reference :124-128)
            fake = self.generator(noise)
            output = self.discriminator(fake).reshape(-1)
            loss_gen = self.criterion_const(output, 1.0)
            self.log("train/g_loss", loss_gen)
            return loss_gen


class GANStabilityR1(BaseGAN):
    """reference core/lightning_module.py:130-156"""

    def training_step(self, batch, batch_idx, optimizer_idx):
        real, _ = batch
        fake = self.generator(self.sample_noise(len(real)))

        if optimizer_idx == 0:
            real.requires_grad_()
            disc_real = self.discriminator(real).reshape(-1)
            loss_disc_real = self.criterion_const(disc_real, 1.0)
            disc_fake = self.discriminator(fake.detach()).reshape(-1)
            loss_disc_fake = self.criterion_const(disc_fake, 0.0)
            r1_reg = self.cfg.loss_weight.reg * compute_grad2(disc_real, real).mean()
            loss_disc = r1_reg + (loss_disc_real + loss_disc_fake)
            self.log("train/d_loss", loss_disc)
            return loss_disc

        if optimizer_idx == 1:
            output = self.discriminator(fake).reshape(-1)
            loss_gen = self.criterion_const(output, 1.0)
            self.log("train/g_loss", loss_gen)
            return loss_gen


class WGAN(BaseGAN):
    mutates_discriminator_before_forward = True     # tells ddp.GradSync not to defer D's step past the clamp

    def training_step(self, batch, batch_idx, optimizer_idx):
        clip = self.cfg.train.weight_clip
        for p in self.discriminator.parameters():       # every call, both branches (reference :160-162)
            F.clamp_(p, -clip, clip)

        real, _ = batch
        noise = self.sample_noise(len(real))

        if optimizer_idx == 0:
            if self.stack_d_passes:
                fake = self.generator(noise)
                logits = self.stacked_discriminator(real, fake)
                if logits is not None:
                    loss_disc = F.weighted_half_means(logits, -1.0, 1.0)      # -(mean(D(real)) - mean(D(fake)))
                    self.log("train/d_loss", loss_disc)
                    return loss_disc
                disc_real = self.discriminator(real).reshape(-1)
            elif self.real_first:
                disc_real = self.discriminator(real).reshape(-1)
                fake = self.generator(noise)
            else:
                fake = self.generator(noise)
                disc_real = self.discriminator(real).reshape(-1)
            disc_fake = self.discriminator(fake.detach()).reshape(-1)
            loss_disc = -(torch.mean(disc_real) - torch.mean(disc_fake))
            self.log("train/d_loss", loss_disc)
            return loss_disc

        if optimizer_idx == 1:
            fake = self.generator(noise)
            gen_fake = self.discriminator(fake).reshape(-1)
            loss_gen = -torch.mean(gen_fake)
            self.log("train/g_loss", loss_gen)
            return loss_gen


class WGANGP(BaseGAN):
    gp_alpha = None     # test hook: pin the interpolation coefficients ([N,1,1,1])

    def training_step(self, batch, batch_idx, optimizer_idx):
        real, _ = batch
        noise = self.sample_noise(len(real))

        if optimizer_idx == 0:
            logits = None
            if self.stack_d_passes:
                # D(real) and D(fake.detach()) in one pass over the stacked batch (InstanceNorm: per-sample statistics,
                # nothing couples the two halves); the penalty's D(x_hat) below stays a pass of its own
                fake = self.generator(noise)
                logits = self.stacked_discriminator(real, fake)
                if logits is None:
                    disc_real = self.discriminator(real).reshape(-1)
            elif self.real_first:
                disc_real = self.discriminator(real).reshape(-1)
                fake = self.generator(noise)
            else:
                fake = self.generator(noise)
                disc_real = self.discriminator(real).reshape(-1)
            if logits is None:
                disc_fake = self.discriminator(fake.detach()).reshape(-1)
            # `fake` is NOT detached here, as in the reference (:195-196)
            gp = gradient_penalty(self.discriminator, real, fake, device=self.device, alpha=self.gp_alpha)
            if logits is not None:
                loss_disc = (self.cfg.loss_weight.lambda_gp * gp) + F.weighted_half_means(logits, -1.0, 1.0)
            else:
                loss_disc = (self.cfg.loss_weight.lambda_gp * gp) - (torch.mean(disc_real) - torch.mean(disc_fake))
            self.log("train/d_loss", loss_disc)
            return loss_disc

        if optimizer_idx == 1:
            fake = self.generator(noise)
            gen_fake = self.discriminator(fake).reshape(-1)
            loss_gen = -torch.mean(gen_fake)
            self.log("train/g_loss", loss_gen)
            return loss_gen


class HOLOGAN(BaseGAN):
    """reference core/lightning_module.py:209-237"""

    def training_step(self, batch, batch_idx, optimizer_idx):
        real, _ = batch
        z = self.sample_noise(len(real))

        if optimizer_idx == 0:
            d, c = self.discriminator, self.criterion
            fake = None
            if (self.stack_d_passes and getattr(d, "supports_stacked_batches", False) and real.is_cuda and d.training
                    and isinstance(c, torch.nn.BCEWithLogitsLoss) and c.reduction == "mean" and c.weight is None
                    and c.pos_weight is None):
                # ONE pass over [real; fake]: InstanceNorm is per sample, and each half gets the sigma of its own power
                # iteration (real's first, as in the reference's call order) -- hologan_discriminator.Discriminator
                fake = self.generator(z)
                if real.shape == fake.shape:
                    logits, d_z_pred = d(torch.cat([real, fake.detach()]), groups=2)
                    loss_disc = F.bce_logits_pair_mean(logits, 1.0, 0.0)       # (BCE(real, 1) + BCE(fake, 0)) / 2
                    q_loss = F.mse_mean(d_z_pred[len(real):], z)
                    self.log("train/d_loss", loss_disc)
                    self.log("train/q_loss", q_loss)
                    return loss_disc + q_loss
            # (the generator draws its views from numpy's generator inside forward(): no other numpy draw happens in
            # this step, so running D(real) first leaves that stream untouched)
            if fake is not None:
                disc_real, _ = self.discriminator(real)
            elif self.real_first:
                disc_real, _ = self.discriminator(real)
                fake = self.generator(z)
            else:
                fake = self.generator(z)
                disc_real, _ = self.discriminator(real)
            loss_disc_real = self.criterion_const(disc_real, 1.0)
            disc_fake, d_z_pred = self.discriminator(fake.detach())
            loss_disc_fake = self.criterion_const(disc_fake, 0.0)
            loss_disc = (loss_disc_real + loss_disc_fake) / 2
            q_loss = F.mse_mean(d_z_pred, z)
            self.log("train/d_loss", loss_disc)
            self.log("train/q_loss", q_loss)
            return loss_disc + q_loss

        if optimizer_idx == 1:
            fake = self.generator(z)
            output, d_z_pred = self.discriminator(fake)
            loss_gen = self.criterion_const(output, 1.0)
            q_loss = F.mse_mean(d_z_pred, z)
            self.log("train/g_loss", loss_gen)
            self.log("train/q_loss", q_loss)
            return loss_gen + q_loss
