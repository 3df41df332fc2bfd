"""lightning_gan_zoo_amd.functional, part 4: the WGAN-GP penalty tail, loss heads, the R1 ResNet helpers and the input step."""
import ctypes
import os
import weakref
from collections import namedtuple

import torch

from .._lib import check, lib
from ._base import *      # noqa: F401,F403
from ._conv import *      # noqa: F401,F403
from ._norm import *      # noqa: F401,F403

# ---------------------------------------------------------------------------
# gradient-penalty tail (reference core/utils/utils.py:41-42, 55-57)
# ---------------------------------------------------------------------------
class _Lerp(torch.autograd.Function):
    """out[n] = alpha[n]*a[n] + (1-alpha[n])*b[n] for a, b [N, L], alpha [N]."""

    @staticmethod
    def forward(ctx, a, b, alpha):
        a, b, alpha = _req(a, "a"), _req(b, "b"), _req(alpha, "alpha")
        R, L = a.shape[0], a.numel() // a.shape[0]
        out = torch.empty_like(a)
        check(lib.gz_rowscale(_p(a), _p(alpha), _p(b), None, _p(out), R, L, 0, 1, _stream()), "rowscale(lerp)")
        ctx.save_for_backward(alpha)
        return out

    @staticmethod
    def backward(ctx, g):
        (alpha,) = ctx.saved_tensors
        ga = gb = None
        if ctx.needs_input_grad[0]:
            ga = row_scale(g, alpha)
        if ctx.needs_input_grad[1]:
            gb = row_scale(g, 1.0 - alpha)
        return ga, gb, None


def lerp_rows(a, b, alpha):
    return _Lerp.apply(a, b, alpha.reshape(-1))


class _RowScale(torch.autograd.Function):
    """out[n] = s[n] * x[n]"""

    @staticmethod
    def forward(ctx, x, s):
        x, s = _req(x, "x"), _req(s, "s")
        R, L = x.shape[0], x.numel() // x.shape[0]
        out = torch.empty_like(x)
        check(lib.gz_rowscale(_p(x), _p(s), None, None, _p(out), R, L, 0, 0, _stream()), "rowscale")
        ctx.save_for_backward(x, s)
        return out

    @staticmethod
    def backward(ctx, g):
        x, s = ctx.saved_tensors
        gx = _RowScale.apply(g, s) if ctx.needs_input_grad[0] else None
        gs = _RowDot.apply(g, x) if ctx.needs_input_grad[1] else None
        return gx, gs


class _RowDot(torch.autograd.Function):
    """y[n] = <a[n], b[n]>"""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _req(a, "a"), _req(b, "b")
        ctx.save_for_backward(a, b)
        R = a.shape[0]
        return _rowdot_raw(a.reshape(R, -1), b.reshape(R, -1), False)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga = _RowScale.apply(b, g) if ctx.needs_input_grad[0] else None
        gb = _RowScale.apply(a, g) if ctx.needs_input_grad[1] else None
        return ga, gb


def row_scale(x, s):
    return _RowScale.apply(x, s.reshape(-1))


def row_dot(a, b):
    return _RowDot.apply(a, b)


def row_sumsq(x):
    """sum of squares per sample, [N, ...] -> [N]"""
    return _RowDot.apply(x, x)


class _GPPenalty(torch.autograd.Function):
    """mean((sqrt(sumsq) - 1)^2): the tail of the gradient penalty in one launch each way (gz_gp_penalty)."""

    @staticmethod
    def forward(ctx, sumsq):
        sumsq = _req(sumsq, "sumsq")
        out = torch.empty((), device=sumsq.device, dtype=torch.float32)
        check(lib.gz_gp_penalty(_p(sumsq), _p(out), sumsq.numel(), _stream()), "gp_penalty")
        ctx.save_for_backward(sumsq)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (sumsq,) = ctx.saved_tensors
        ds = torch.empty_like(sumsq)
        check(lib.gz_gp_penalty_bwd(_p(sumsq), _p(_req(g)), _p(ds), sumsq.numel(), _stream()), "gp_penalty_bwd")
        return ds


def gp_penalty(sumsq):
    """``torch.mean((norm - 1) ** 2)`` with ``norm = sqrt(sumsq)`` per sample (reference core/utils/utils.py:55-57);
    the subgradient at an exactly-zero gradient is 0, as torch.norm's."""
    return _GPPenalty.apply(sumsq.reshape(-1))


_ones = {}


def ones_like_const(t):
    """A read-only tensor of ones shaped like ``t`` (``grad_outputs=torch.ones_like(scores)``, reference utils.py:51):
    cached per shape and device instead of a fill launch per step.  Never written by anyone."""
    key = (tuple(t.shape), t.device, t.dtype)
    o = _ones.get(key)
    if o is None:
        o = _ones[key] = torch.ones(t.shape, device=t.device, dtype=t.dtype)
    return o


@torch.no_grad()
def clamp_(t, lo, hi):
    """in-place clamp of a parameter tensor (WGAN weight clipping, lightning_module.py:160-162)."""
    if not t.is_cuda or not t.is_contiguous() or t.dtype != torch.float32:
        raise RuntimeError("clamp_: expected a contiguous float32 GPU tensor")
    check(lib.gz_clamp_(_p(t), t.numel(), float(lo), float(hi), _stream()), "clamp_")
    invalidate(t)
    return t



# ---------------------------------------------------------------------------
# loss heads (csrc/gz_loss.hip)
# ---------------------------------------------------------------------------
class _BCELogitsMean(torch.autograd.Function):
    """mean BCE-with-logits against a constant target: criterion(x, ones_like(x)) / zeros_like(x)."""

    @staticmethod
    def forward(ctx, x, target):
        x = _req(x, "logits").reshape(-1)
        loss = torch.empty(1, device=x.device, dtype=torch.float32)
        check(lib.gz_bce_logits_mean(_p(x), _p(loss), x.numel(), float(target), _stream()), "bce_logits_mean")
        ctx.save_for_backward(x)
        ctx.target = float(target)
        return loss.reshape(())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = _req(g).reshape(1)
        dx = torch.empty_like(x)
        check(lib.gz_bce_logits_mean_bwd(_p(x), _p(g), _p(dx), x.numel(), ctx.target, _stream()), "bce_logits_mean_bwd")
        return dx, None


def bce_logits_mean(logits, target):
    return _BCELogitsMean.apply(logits.reshape(-1), target)


class _PairLoss(torch.autograd.Function):
    """Loss head over the stacked logits [first half; second half] (gz_pair_loss): mode 0 = the mean of two BCE means
    against constants t0 / t1, mode 1 = t0 * mean(first) + t1 * mean(second)."""

    @staticmethod
    def forward(ctx, x, t0, t1, mode):
        x = _req(x, "logits").reshape(-1)
        if x.numel() % 2:
            raise RuntimeError("pair loss: an even number of logits (two stacked batches) is required")
        loss = torch.empty(1, device=x.device, dtype=torch.float32)
        check(lib.gz_pair_loss(_p(x), _p(loss), x.numel() // 2, float(t0), float(t1), int(mode), _stream()), "pair_loss")
        ctx.save_for_backward(x)
        ctx.cfg = (float(t0), float(t1), int(mode))
        return loss.reshape(())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        t0, t1, mode = ctx.cfg
        g = _req(g).reshape(1)
        dx = torch.empty_like(x)
        check(lib.gz_pair_loss_bwd(_p(x), _p(g), _p(dx), x.numel() // 2, t0, t1, mode, _stream()), "pair_loss_bwd")
        return dx, None, None, None


def bce_logits_pair_mean(logits, t_first, t_second):
    """(BCE(logits[:n], t_first).mean() + BCE(logits[n:], t_second).mean()) / 2 in one launch."""
    return _PairLoss.apply(logits.reshape(-1), t_first, t_second, 0)


def weighted_half_means(logits, w_first, w_second):
    """w_first * logits[:n].mean() + w_second * logits[n:].mean() in one launch."""
    return _PairLoss.apply(logits.reshape(-1), w_first, w_second, 1)


class _MSEMean(torch.autograd.Function):
    """mean((a - b)^2) with b a constant (HoloGAN's q_loss: b is the latent the generator was fed)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _req(a, "a"), _req(b, "b")
        loss = torch.empty(1, device=a.device, dtype=torch.float32)
        check(lib.gz_mse_mean(_p(a), _p(b), _p(loss), a.numel(), _stream()), "mse_mean")
        ctx.save_for_backward(a, b)
        return loss.reshape(())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = _req(g).reshape(1)
        da = torch.empty_like(a)
        check(lib.gz_mse_mean_bwd(_p(a), _p(b), _p(g), _p(da), a.numel(), _stream()), "mse_mean_bwd")
        return da, (-da if ctx.needs_input_grad[1] else None)


def mse_mean(a, b):
    return _MSEMean.apply(a, b)



# ---------------------------------------------------------------------------
# R1-regularised ResNet path (reference core/submodules/gan_stability/models/resnet.py).
# Every op here is linear or piecewise linear, so forward/adjoint pairs close under differentiation and
# compute_grad2's create_graph=True (core/utils/utils.py:60-69) works to any order.
# ---------------------------------------------------------------------------
K3S1P1 = Geom(3, 3, 1, 1)
K1S1P0 = Geom(1, 1, 1, 0)


class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act, slope):
        x = _req(x, "x")
        y = torch.empty_like(x)
        check(lib.gz_act_fwd(_p(x), _p(y), x.numel(), act, slope, _stream()), "act_fwd")
        ctx.act, ctx.slope = act, slope
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return _ActBwd.apply(g, y, ctx.act, ctx.slope), None, None


def activation(x, act=ACT_LRELU, slope=0.2):
    return _Act.apply(x, act, slope)


def _axpby_raw(a, alpha, b, beta, act_out=None, act=ACT_NONE, slope=0.0):
    out = torch.empty_like(a)
    check(lib.gz_axpby(_p(a), alpha, _p(b), beta, _p(out), _p(act_out), a.numel(), act, slope, _stream()), "axpby")
    return out


class _Scale(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, alpha):
        ctx.alpha = alpha
        return _axpby_raw(_req(a, "a"), alpha, None, 0.0)

    @staticmethod
    def backward(ctx, g):
        return _Scale.apply(g, ctx.alpha), None


class _AddScaled(torch.autograd.Function):
    """out = a + beta*b (the residual tail, resnet.py:121-122)."""

    @staticmethod
    def forward(ctx, a, b, beta):
        ctx.beta = beta
        return _axpby_raw(_req(a, "a"), 1.0, _req(b, "b"), beta)

    @staticmethod
    def backward(ctx, g):
        ga = g if ctx.needs_input_grad[0] else None
        gb = _Scale.apply(g, ctx.beta) if ctx.needs_input_grad[1] else None
        return ga, gb, None


class _AddScaledAct(torch.autograd.Function):
    """(out, act(out)) with out = a + beta*b: the residual tail and the next block's pre-activation in one pass."""

    @staticmethod
    def forward(ctx, a, b, beta, act, slope):
        a, b = _req(a, "a"), _req(b, "b")
        act_out = torch.empty_like(a)
        out = _axpby_raw(a, 1.0, b, beta, act_out, act, slope)
        ctx.beta, ctx.act, ctx.slope = beta, act, slope
        ctx.save_for_backward(act_out)
        return out, act_out

    @staticmethod
    def backward(ctx, g_out, g_act):
        (act_out,) = ctx.saved_tensors
        g = None
        if g_act is not None:
            g = _ActBwd.apply(g_act, act_out, ctx.act, ctx.slope)
        if g_out is not None:
            g = g_out if g is None else _AddScaled.apply(g_out, g, 1.0)
        ga = g if ctx.needs_input_grad[0] else None
        gb = _Scale.apply(g, ctx.beta) if ctx.needs_input_grad[1] else None
        return ga, gb, None, None, None


def scale(a, alpha):
    return _Scale.apply(a, alpha)


def add_scaled(a, b, beta):
    return _AddScaled.apply(a, b, beta)


def add_scaled_act(a, b, beta, act=ACT_LRELU, slope=0.2):
    return _AddScaledAct.apply(a, b, beta, act, slope)


def _planes(x):
    if x.dim() != 4:
        raise RuntimeError("lightning_gan_zoo_amd: expected an NCHW tensor, got shape %s" % (tuple(x.shape),))
    return x.shape[0] * x.shape[1]


def _avgpool_fwd_raw(x):
    N, C, H, W = x.shape
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((N, C, OH, OW), device=x.device, dtype=torch.float32)
    check(lib.gz_avgpool3s2_fwd(_p(x), _p(y), _planes(x), H, W, OH, OW, _stream()), "avgpool3s2_fwd")
    return y


def _avgpool_bwd_raw(gy, hw):
    N, C, OH, OW = gy.shape
    H, W = hw
    gx = torch.empty((N, C, H, W), device=gy.device, dtype=torch.float32)
    check(lib.gz_avgpool3s2_bwd(_p(gy), _p(gx), _planes(gy), H, W, OH, OW, _stream()), "avgpool3s2_bwd")
    return gx


class _AvgPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.hw = tuple(x.shape[2:])
        return _avgpool_fwd_raw(_req(x, "x"))

    @staticmethod
    def backward(ctx, g):
        return _AvgPoolT.apply(g, ctx.hw)


class _AvgPoolT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, hw):
        return _avgpool_bwd_raw(_req(g, "g"), hw)

    @staticmethod
    def backward(ctx, v):
        return _AvgPool.apply(v), None


def avg_pool3s2(x):
    """nn.AvgPool2d(3, stride=2, padding=1) (resnet.py:72)."""
    return _AvgPool.apply(x)


def _upsample_fwd_raw(x):
    N, C, H, W = x.shape
    y = torch.empty((N, C, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
    check(lib.gz_upsample2_fwd(_p(x), _p(y), _planes(x), H, W, _stream()), "upsample2_fwd")
    return y


def _upsample_bwd_raw(gy):
    N, C, H2, W2 = gy.shape
    gx = torch.empty((N, C, H2 // 2, W2 // 2), device=gy.device, dtype=torch.float32)
    check(lib.gz_upsample2_bwd(_p(gy), _p(gx), _planes(gy), H2 // 2, W2 // 2, _stream()), "upsample2_bwd")
    return gx


class _Upsample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return _upsample_fwd_raw(_req(x, "x"))

    @staticmethod
    def backward(ctx, g):
        return _UpsampleT.apply(g)


class _UpsampleT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g):
        return _upsample_bwd_raw(_req(g, "g"))

    @staticmethod
    def backward(ctx, v):
        return _Upsample.apply(v)


def upsample2(x):
    """nn.Upsample(scale_factor=2), nearest (resnet.py:31)."""
    return _Upsample.apply(x)



# ---------------------------------------------------------------------------
# input step
# ---------------------------------------------------------------------------
def normalize_u8_images(u8, mean, std):
    """Decoded uint8 images [N,H,W,C] on the GPU -> float [N,C,H,W] = (x / 255 - mean) / std: ToTensor() +
    Normalize(mean, std) of reference core/lightning_module.py:42-47, one pass on the device."""
    if not u8.is_cuda or u8.dtype != torch.uint8 or u8.dim() != 4:
        raise RuntimeError("lightning_gan_zoo_amd: expected a uint8 NHWC tensor on the GPU")
    u8 = u8 if u8.is_contiguous() else u8.contiguous()
    N, H, W, C = u8.shape
    out = torch.empty((N, C, H, W), device=u8.device, dtype=torch.float32)
    check(lib.gz_u8hwc_to_nchw(_p(u8), _p(out), N, H, W, C, float(mean), float(std), _stream()), "u8hwc_to_nchw")
    return out


__all__ = [n for n in list(globals()) if not n.startswith("__")]     # the flat namespace of the package (private helpers included)
