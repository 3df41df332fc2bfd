"""Loss utilities of the hot path on the HIP kernels; mirrors reference core/utils/utils.py."""
import torch
from torch import nn

from ... import functional as F
from ...harness import draw_on_host


@torch.no_grad()
def init_weights(m):
    # reference utils.py:5-11 (its call sites are commented out at lightning_module.py:51-52)
    if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
        nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif isinstance(m, (nn.BatchNorm2d)):
        nn.init.normal_(m.weight.data, 1.0, 0.02)
        nn.init.constant_(m.bias.data, 0)


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
        grad_outputs=torch.ones_like(mixed_scores),
        create_graph=True,
        retain_graph=True,
    )[0]
    sumsq = F.row_sumsq(gradient.reshape(bs, -1))
    # torch.norm's subgradient at an exactly-zero gradient is 0 (reference :55); a bare sqrt would give 0.5 / 0 = inf
    # there and NaN after the chain rule
    nonzero = sumsq > 0
    gradient_norm = torch.where(nonzero, sumsq, torch.ones_like(sumsq)).sqrt() * nonzero
    return torch.mean((gradient_norm - 1) ** 2)


def compute_grad2(d_out, x_in):
    """R1 regulariser, reference utils.py:60-69 (used by GANStabilityR1; 'next' row of SURVEY 8-f)."""
    batch_size = x_in.size(0)
    grad_dout = torch.autograd.grad(outputs=d_out.sum(), inputs=x_in, create_graph=True, retain_graph=True,
                                    only_inputs=True)[0]
    assert grad_dout.size() == x_in.size()
    return F.row_sumsq(grad_dout.reshape(batch_size, -1))
