"""Loss utilities of the hot path on the HIP kernels; mirrors reference core/utils/utils.py."""
import torch
from torch import nn

from ... import functional as F
from ...harness import draw_on_host


_INIT_RULES = (
    # (layer types, attribute, mean, std) -- the DCGAN-paper initialisation the reference defines at utils.py:5-11
    # and never applies (its two call sites are commented out, lightning_module.py:51-52)
    ((nn.Conv2d, nn.ConvTranspose2d), "weight", 0.0, 0.02),
    ((nn.BatchNorm2d,), "weight", 1.0, 0.02),
    ((nn.BatchNorm2d,), "bias", 0.0, 0.0),
)


def init_weights(m):
    """``net.apply(init_weights)``: N(0, 0.02) conv weights, N(1, 0.02) BatchNorm scales, zero BatchNorm shifts."""
    for types, attr, mean, std in _INIT_RULES:
        t = getattr(m, attr, None) if isinstance(m, types) else None
        if t is None:
            continue
        with torch.no_grad():
            t.normal_(mean, std) if std > 0 else t.fill_(mean)


class VerboseShapeExecution(nn.Module):
    """``cfg.debug.verbose_shape`` (reference utils.py:13-27, used as ``self.apply(VerboseShapeExecution)`` at
    lightning_module.py:53-54): constructing it on a module makes every direct child print
    ``<child name>: <input shape> --> <output shape>`` after each forward.  ``Module.apply`` constructs one per
    module of the tree (and throws the wrapper away), so every layer of both networks reports once per call."""

    def __init__(self, model):
        super().__init__()
        self.model = model
        for child_name, child in model.named_children():
            child.__name__ = child_name
            child.register_forward_hook(_print_shapes)

    def forward(self, x):
        return self.model(x)


def _print_shapes(layer, inputs, output):
    print(f"{layer.__name__}: {inputs[0].shape} --> {output.shape}")


def interpolate_sphere(z1, z2, t):
    """Spherical interpolation between latent rows (reference utils.py:29-37; the figure callbacks import it from
    this module).  Latent-sized host / device arithmetic, not part of the image path."""
    cos_omega = (z1 * z2).sum(dim=1, keepdim=True)
    cos_omega = cos_omega / z1.pow(2).sum(dim=1, keepdim=True).sqrt()
    cos_omega = cos_omega / z2.pow(2).sum(dim=1, keepdim=True).sqrt()
    omega = torch.acos(cos_omega)
    so = torch.sin(omega)
    return torch.sin((1 - t) * omega) / so * z1 + torch.sin(t * omega) / so * z2


def gradient_penalty(critic, real, fake, device="cpu", alpha=None):
    """WGAN-GP penalty, reference utils.py:39-58.

    alpha ~ U[0,1) is drawn per sample on the HOST generator and moved to the device, exactly like
    the reference (:41); ``alpha`` may be passed in ([N,1,1,1]) to pin it.  The interpolation, the
    critic, its input gradient (create_graph=True -> differentiable HIP backward ops), the
    per-sample squared norm and their second-order backward all run on the HIP kernels.
    """
    bs = real.shape[0]
    given = alpha
    if given is not None and given.is_cuda:          # already staged (harness.GraphedTrainer's static buffer)
        alpha = given.reshape(bs)
    else:
        alpha = draw_on_host(lambda: (torch.rand((bs, 1, 1, 1)) if given is None else given).reshape(bs).float(),
                             real.device)
    interpolated_images = F.lerp_rows(real, fake, alpha)
    interpolated_images.requires_grad_()
    mixed_scores = critic(interpolated_images)
    gradient = torch.autograd.grad(
        inputs=interpolated_images,
        outputs=mixed_scores,
        grad_outputs=F.ones_like_const(mixed_scores),
        create_graph=True,
        retain_graph=True,
    )[0]
    sumsq = F.row_sumsq(gradient.reshape(bs, -1))
    # mean((||g_n|| - 1)^2) and its gradient in one launch each (gz_gp_penalty); torch.norm's subgradient at an
    # exactly-zero gradient is 0 (reference :55), kept by the kernel -- a bare sqrt would give 0.5 / 0 there
    return F.gp_penalty(sumsq)


def compute_grad2(d_out, x_in):
    """R1 regulariser, reference utils.py:60-69 (used by GANStabilityR1; 'next' row of SURVEY 8-f)."""
    batch_size = x_in.size(0)
    grad_dout = torch.autograd.grad(outputs=d_out.sum(), inputs=x_in, create_graph=True, retain_graph=True,
                                    only_inputs=True)[0]
    assert grad_dout.size() == x_in.size()
    return F.row_sumsq(grad_dout.reshape(batch_size, -1))
