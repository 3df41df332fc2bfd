"""torch.autograd.Function wrappers over the C-ABI HIP kernels (include/gz_ops.h).

PyTorch is used for device memory, streams and the autograd tape only; every
arithmetic step of the hot path is a call into libgz_hip.so.  The convolution
family is closed under differentiation:

    F(x, w)  = conv2d            dF/dx^T g  = Dg(g, w)   dF/dw^T g  = Wg(x, g)
    Dg(g, w) = conv_transpose2d  dDg/dg^T v = F(v, w)    dDg/dw^T v = Wg(v, g)
    Wg(x, g) = weight gradient   dWg/dx^T v = Dg(g, v)   dWg/dg^T v = F(x, v)

so `torch.autograd.grad(..., create_graph=True)` (the WGAN-GP gradient penalty,
reference core/utils/utils.py:48-54) works through these ops to any order.

Round 6: one flat namespace, five source files by operator family (the single file had 2.4 k lines) --
``_base`` (call helpers, timer, packed-weight caches, raw launchers, gradient sinks, parameter gate), ``_conv`` (F / Dg /
Wg, dense, row-dot), ``_norm``, ``_misc`` (penalty tail, loss heads, ResNet helpers, input step), ``_hologan``.  Users
keep ``from lightning_gan_zoo_amd import functional as F``; tests that replace ``F.<op>`` patch THIS namespace, which
is what the model modules call through.
"""
from ._base import *      # noqa: F401,F403
from ._conv import *      # noqa: F401,F403
from ._norm import *      # noqa: F401,F403
from ._misc import *      # noqa: F401,F403
from ._hologan import *      # noqa: F401,F403
