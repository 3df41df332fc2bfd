"""lightning_gan_zoo_amd.functional, part 2: the differentiable convolution family F / Dg / Wg, dense layers on the GEMM core and
the critics' last layer (row-dot family)."""
import ctypes
import os
import weakref
from collections import namedtuple

import torch

from .._lib import check, lib
from ._base import *      # noqa: F401,F403

# ---------------------------------------------------------------------------
# activation backward as a differentiable op (linear in g; the mask is piecewise constant)
# ---------------------------------------------------------------------------
class _ActBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, out, act, slope):
        ctx.act, ctx.slope = act, slope
        ctx.save_for_backward(g, out)
        return _act_bwd_raw(_req(g), out, act, slope)

    @staticmethod
    def backward(ctx, v):
        g, out = ctx.saved_tensors
        gg = _ActBwd.apply(v, out, ctx.act, ctx.slope) if ctx.needs_input_grad[0] else None
        go = None
        if ctx.needs_input_grad[1] and ctx.act == ACT_TANH:
            # d/d(out) of g*(1-out^2); only tanh has a non-constant mask
            if out.numel() % 4 == 0:
                v, g, o = _req(v), _req(g), _req(out)
                go = torch.empty_like(o)
                check(lib.gz_tanh_bwd2(_p(v), _p(g), _p(o), _p(go), o.numel(), _stream()), "tanh_bwd2")
            else:
                go = v * g * (-2.0 * out)
        return gg, go, None, None


# ---------------------------------------------------------------------------
# F / Dg / Wg
# ---------------------------------------------------------------------------
# In a first-order backward the weight gradient and the input gradient of a layer are independent and could share
# the GPU from two streams (one kernel's drain covered by the other).  Measured and left OFF (threshold 0): two large
# MFMA-bound kernels running together lose more to cache / LDS contention than their drains cost (dc_gan +8 %,
# hologan +9 % step time), and for the small R1 layers the two extra stream waits per layer make the already
# launch-heavy step host-bound (15.4 -> 19.0 ms).  GZ_WG_SIDE_STREAM_FLOPS=<flops> enables it for layers below
# that size.  When enabled, the main stream waits for the side stream before anything else is enqueued, so every
# consumer sees both results and no record_stream is needed.
_side_streams = {}
_WG_SIDE_FLOPS = float(os.environ.get("GZ_WG_SIDE_STREAM_FLOPS", "0"))


def _conv_flops(x, gy, w):
    return 2.0 * gy.shape[0] * gy.shape[2] * gy.shape[3] * w.shape[0] * w.shape[1] * w.shape[2] * w.shape[3]


def _pair_wgrad_dgrad(wgrad, dgrad, flops):
    """Run wgrad() on the side stream and dgrad() on the current one; returns (dx, dw)."""
    if flops > _WG_SIDE_FLOPS or torch.is_grad_enabled():     # double backward: one stream (autograd tracks nothing across)
        dx = dgrad()
        return dx, wgrad()
    main = torch.cuda.current_stream()
    key = (main.device_index, main.cuda_stream)
    side = _side_streams.get(key)
    if side is None:
        side = _side_streams[key] = torch.cuda.Stream(device=main.device)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        dw = wgrad()
    dx = dgrad()
    main.wait_stream(side)
    return dx, dw


class _ConvF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, geom, act, slope, want_stats=False):
        x, w = _req(x, "x"), _req(w, "w")
        ctx.geom, ctx.act, ctx.slope = geom, act, slope
        ctx.has_bias = bias is not None
        ctx.bias_ref = bias if isinstance(bias, torch.nn.Parameter) else None
        if want_stats:          # BatchNorm follows: no bias, no activation; second output = partial statistics
            y, stats = _conv_fwd_stats_raw(x, w, geom)
            ctx.save_for_backward(x, w, None)
            if stats is None:
                stats = torch.empty(0, device=x.device)
            ctx.mark_non_differentiable(stats)
            ctx.set_materialize_grads(False)      # else every backward launches a zero fill for the statistics' "gradient"
            return y, stats
        y = _conv_fwd_raw(x, w, bias, geom, act, slope)
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, gy, *_stats_grad):
        if gy is None:
            return (None,) * 7
        x, w, y = ctx.saved_tensors
        geom = ctx.geom
        if (ctx.act != ACT_NONE and not torch.is_grad_enabled() and ctx.needs_input_grad[1] and not ctx.needs_input_grad[0]
                and (not ctx.has_bias or not ctx.needs_input_grad[2] or ctx.bias_ref is not None)):
            # weights (and bias) only: activation backward, weight gradient and bias gradient in one launch where it exists
            b = ctx.bias_ref if ctx.has_bias and ctx.needs_input_grad[2] else None
            if _sink_conv_wgrad_act(w, b, x, gy, y, geom, ctx.act, ctx.slope):
                return (None,) * 7
        if (ctx.act in (ACT_RELU, ACT_LRELU) and not torch.is_grad_enabled() and ctx.needs_input_grad[0]
                and not ctx.needs_input_grad[1] and not (ctx.has_bias and ctx.needs_input_grad[2])):
            # input only (a generator step through the frozen critic's first layer): the mask is formed on load
            dx = _conv_dgrad_act_raw(gy, y, ctx.act, ctx.slope, w, geom, tuple(x.shape[2:]))
            if dx is not None:
                return (dx,) + (None,) * 6
        if ctx.act != ACT_NONE:
            gy = _ActBwd.apply(gy, y, ctx.act, ctx.slope)
        if not torch.is_grad_enabled() and _WG_SIDE_FLOPS <= 0:
            return _ConvF._first_order(ctx, gy, x, w, geom)
        dx = dw = db = None
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            dx, dw = _pair_wgrad_dgrad(lambda: _ConvWg.apply(x, gy, geom),
                                       lambda: _ConvDg.apply(gy, w, None, geom, tuple(x.shape[2:]), ACT_NONE, 0.0),
                                       _conv_flops(x, gy, w))
        elif ctx.needs_input_grad[0]:
            dx = _ConvDg.apply(gy, w, None, geom, tuple(x.shape[2:]), ACT_NONE, 0.0)
        elif ctx.needs_input_grad[1]:
            dw = _ConvWg.apply(x, gy, geom)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _ChannelSum.apply(gy)
        return dx, dw, db, None, None, None, None

    @staticmethod
    def _first_order(ctx, gy, x, w, geom):
        """No graph is being recorded: raw launches, weight and bias gradient from one kernel where it can."""
        gy = _req(gy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _conv_dgrad_raw(gy, w, None, geom, tuple(x.shape[2:]), ACT_NONE, 0.0)
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            if want_b:
                dw, db = _conv_wgrad_raw(x, gy, geom, with_bias=True)
                dw = _sink_or_return(w, dw)       # (complete gradient: one-slab source, no AccumulateGrad `add_`)
            elif not _sink_conv_wgrad(w, x, gy, geom):
                dw = _conv_wgrad_raw(x, gy, geom)
        elif want_b:
            db = _channel_sum_raw(gy)
        if want_b and ctx.bias_ref is not None:
            db = _sink_or_return(ctx.bias_ref, db)
        return dx, dw, db, None, None, None, None


class _ConvDg(torch.autograd.Function):
    """x = act(conv_transpose2d(g, w) + bias): ConvTranspose2d forward and Conv2d input gradient."""

    @staticmethod
    def forward(ctx, g, w, bias, geom, hw, act, slope, want_stats=False, bias_cancels=False):
        g, w = _req(g, "g"), _req(w, "w")
        ctx.geom, ctx.act, ctx.slope = geom, act, slope
        ctx.has_bias = bias is not None
        ctx.bias_cancels = bias_cancels      # the caller normalises the output per (sample, channel): d/d bias == 0
        ctx.bias_ref = bias if isinstance(bias, torch.nn.Parameter) else None
        if want_stats:
            x, stats = _conv_dgrad_stats_raw(g, w, geom, hw)
            ctx.save_for_backward(g, w, None)
            if stats is None:
                stats = torch.empty(0, device=g.device)
            ctx.mark_non_differentiable(stats)
            ctx.set_materialize_grads(False)
            return x, stats
        x = _conv_dgrad_raw(g, w, bias, geom, hw, act, slope)
        ctx.save_for_backward(g, w, x if act != ACT_NONE else None)
        return x

    @staticmethod
    def backward(ctx, v, *_stats_grad):
        if v is None:
            return (None,) * 9
        g, w, x = ctx.saved_tensors
        geom = ctx.geom
        if ctx.act != ACT_NONE:
            v = _ActBwd.apply(v, x, ctx.act, ctx.slope)
        if not torch.is_grad_enabled() and _WG_SIDE_FLOPS <= 0:      # no graph is being recorded: raw launches
            v = _req(v)
            dg = _conv_fwd_raw(v, w, None, geom, ACT_NONE, 0.0) if ctx.needs_input_grad[0] else None
            dw = None
            if ctx.needs_input_grad[1] and not _sink_conv_wgrad(w, v, g, geom):
                dw = _conv_wgrad_raw(v, g, geom)
            db = None
            if ctx.has_bias and ctx.needs_input_grad[2]:
                if ctx.bias_cancels:
                    db = _sink_zero(ctx.bias_ref, (v.shape[1],), v.device)
                else:
                    db = _channel_sum_raw(v)
                    if ctx.bias_ref is not None:
                        db = _sink_or_return(ctx.bias_ref, db)
            return dg, dw, db, None, None, None, None, None, None
        dg = dw = db = None
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            dg, dw = _pair_wgrad_dgrad(lambda: _ConvWg.apply(v, g, geom),
                                       lambda: _ConvF.apply(v, w, None, geom, ACT_NONE, 0.0), _conv_flops(v, g, w))
        elif ctx.needs_input_grad[0]:
            dg = _ConvF.apply(v, w, None, geom, ACT_NONE, 0.0)
        elif ctx.needs_input_grad[1]:
            dw = _ConvWg.apply(v, g, geom)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _ChannelSum.apply(v)
        return dg, dw, db, None, None, None, None, None, None


class _ConvWg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, geom):
        x, g = _req(x, "x"), _req(g, "g")
        ctx.geom = geom
        ctx.save_for_backward(x, g)
        return _conv_wgrad_raw(x, g, geom)

    @staticmethod
    def backward(ctx, v):
        x, g = ctx.saved_tensors
        geom = ctx.geom
        v = _req(v)
        dx = dg = None
        if ctx.needs_input_grad[0]:
            dx = _ConvDg.apply(g, v, None, geom, tuple(x.shape[2:]), ACT_NONE, 0.0)
        if ctx.needs_input_grad[1]:
            dg = _ConvF.apply(x, v, None, geom, ACT_NONE, 0.0)
        return dx, dg, None


def conv2d(x, w, bias=None, geom=K4S2P1, act=ACT_NONE, slope=0.0):
    return _ConvF.apply(x, w, bias, geom, act, slope)


def conv_transpose2d(x, w, bias=None, geom=K4S2P1, act=ACT_NONE, slope=0.0, bias_cancels=False):
    """w is the ConvTranspose2d weight [Cin, Cout, KH, KW]; output size (H-1)*S - 2P + KH.
    bias_cancels: the caller feeds the output straight into a normalisation over each (sample, channel) plane (AdaIN,
    InstanceNorm), which removes a per-channel constant -- the bias gradient is exactly zero and is returned as such
    (a first-order backward skips the two reduction launches; the reference's value is rounding noise)."""
    H, W = x.shape[2:]
    oh = (H - 1) * geom.stride - 2 * geom.pad + geom.kh
    ow = (W - 1) * geom.stride - 2 * geom.pad + geom.kw
    return _ConvDg.apply(x, w, bias, geom, (oh, ow), act, slope, False, bias_cancels)


def conv2d_with_stats(x, w, geom=K4S2P1):
    """(conv2d(x, w), stats) for a convolution that feeds a training-mode BatchNorm: the launch also produces the
    per-tile (sum, sum of squares) of its output; pass ``stats`` to batch_norm_act.  stats is an empty tensor when
    the launch could not carry them (batch_norm_act then reads the feature map itself)."""
    return _ConvF.apply(x, w, None, geom, ACT_NONE, 0.0, True)


def conv_transpose2d_with_stats(x, w, geom=K4S2P1):
    H, W = x.shape[2:]
    oh = (H - 1) * geom.stride - 2 * geom.pad + geom.kh
    ow = (W - 1) * geom.stride - 2 * geom.pad + geom.kw
    return _ConvDg.apply(x, w, None, geom, (oh, ow), ACT_NONE, 0.0, True)


# ---------------------------------------------------------------------------
# dense layers on the same GEMM core: the generator's 1x1 -> 4x4 ConvTranspose2d
# (reference standard_networks.py:60) and nn.Linear
# ---------------------------------------------------------------------------
class _MatMul(torch.autograd.Function):
    """c = a @ b for row-major a [M,K], b [K,N].  ``param``: the Parameter ``b`` is a view of (the DCGAN generator's
    first layer): a first-order backward hands its gradient to that parameter's sink instead of autograd."""

    @staticmethod
    def forward(ctx, a, b, param=None):
        a, b = _req(a, "a"), _req(b, "b")
        ctx.save_for_backward(a, b)
        ctx.param = param
        return gemm(a, b)

    @staticmethod
    def backward(ctx, gc):
        a, b = ctx.saved_tensors
        gc = _req(gc)
        ga = gb = None
        if ctx.needs_input_grad[0]:
            ga = _MatMulNT.apply(gc, b)            # gc @ b^T
        if ctx.needs_input_grad[1]:
            if ctx.param is not None and not torch.is_grad_enabled() and _sinks.enabled:
                gb = gemm(a, gc, trans_a=True)
                if _sink_grad(ctx.param, gb.view_as(ctx.param)):
                    gb = None
            else:
                gb = _MatMulTN.apply(a, gc)        # a^T @ gc
        return ga, gb, None


class _MatMulNT(torch.autograd.Function):
    """c = a @ b^T for a [M,K], b [N,K]."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _req(a, "a"), _req(b, "b")
        ctx.save_for_backward(a, b)
        return gemm(a, b, trans_b=True)

    @staticmethod
    def backward(ctx, gc):
        a, b = ctx.saved_tensors
        gc = _req(gc)
        ga = _MatMul.apply(gc, b) if ctx.needs_input_grad[0] else None       # gc @ b
        gb = _MatMulTN.apply(gc, a) if ctx.needs_input_grad[1] else None     # gc^T @ a
        return ga, gb


class _MatMulTN(torch.autograd.Function):
    """c = a^T @ b for a [K,M], b [K,N]."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _req(a, "a"), _req(b, "b")
        ctx.save_for_backward(a, b)
        return gemm(a, b, trans_a=True)

    @staticmethod
    def backward(ctx, gc):
        a, b = ctx.saved_tensors
        gc = _req(gc)
        ga = _MatMulNT.apply(b, gc) if ctx.needs_input_grad[0] else None     # b @ gc^T
        gb = _MatMul.apply(a, gc) if ctx.needs_input_grad[1] else None       # a @ gc
        return ga, gb


def matmul(a, b, param=None):
    return _MatMul.apply(a, b, param)


def matmul_nt(a, b):
    return _MatMulNT.apply(a, b)


def linear(x, weight, bias=None):
    y = _MatMulNT.apply(x, weight)
    return y if bias is None else y + bias


# ---------------------------------------------------------------------------
# last discriminator layer: Conv2d(C, 1, k4, s2, p0) on a 4x4 map == per-sample dot
# ---------------------------------------------------------------------------
def _rowdot_raw(a, b, bcast):
    R, L = a.shape
    y = torch.empty(R, device=a.device, dtype=torch.float32)
    check(lib.gz_rowdot(_p(a), _p(b), _p(y), R, L, int(bcast), _stream()), "rowdot")
    return y


def _rowscale_raw(x, s, R, L, bcast):
    out = torch.empty((R, L), device=s.device, dtype=torch.float32)
    check(lib.gz_rowscale(_p(x), _p(s), None, None, _p(out), R, L, int(bcast), 0, _stream()), "rowscale")
    return out


def _coldot_raw(g, x):
    R, L = x.shape
    out = torch.empty(L, device=x.device, dtype=torch.float32)
    nbytes = lib.gz_coldot_workspace_bytes(R, L)
    ws = _ws(max(nbytes // 4, 1), x.device)
    check(lib.gz_coldot(_p(g), _p(x), _p(out), _p(ws), nbytes, R, L, _stream()), "coldot")
    return out


def _sink_coldot(param, g, x):
    """``param``'s gradient sum_r g[r] * x[r, :] into its sink as UNREDUCED row slices (gz_coldot_partial): the slab sum
    that would follow is done by the launch that sums everything else.  False = not taken."""
    if not _sinks.enabled or not isinstance(param, torch.nn.Parameter) or (param.numel() & 3):
        return False
    if param.grad is not None and (param.grad.data_ptr() & 15 or not param.grad.is_contiguous()
                                   or param.grad.dtype != torch.float32):
        return False
    R, L = x.shape
    if L != param.numel():
        return False
    nbytes = lib.gz_coldot_workspace_bytes(R, L)
    ws = _ws(max(nbytes // 4, 1), x.device)
    out = torch.empty(L, device=x.device, dtype=torch.float32)
    nz = ctypes.c_int(0)
    check(lib.gz_coldot_partial(_p(g), _p(x), _p(out), _p(ws), nbytes, R, L, ctypes.byref(nz), _stream()), "coldot_partial")
    src = (ws, nz.value, L) if nz.value > 1 else (out, 1, L)
    if src[0].data_ptr() & 15:
        _sink_fail("coldot slices")
    _sinks.pending.setdefault(id(param), [param, []])[1].append(src)
    return True


class _DotF(torch.autograd.Function):
    """y[r] = <x[r,:], w>"""

    @staticmethod
    def forward(ctx, x, w, param=None):
        x, w = _req(x, "x"), _req(w, "w")
        ctx.save_for_backward(x, w)
        ctx.param = param          # the Parameter ``w`` is a view of: a first-order backward feeds its sink
        return _rowdot_raw(x, w, True)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        dx = _DotDg.apply(g, w, ctx.param) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            if ctx.param is not None and not torch.is_grad_enabled() and _sinks.enabled:
                if not _sink_coldot(ctx.param, _req(g), x):
                    dw = _coldot_raw(_req(g), x)
                    if _sink_grad(ctx.param, dw.view_as(ctx.param)):
                        dw = None
            else:
                dw = _DotWg.apply(x, g)
        return dx, dw, None


class _DotDg(torch.autograd.Function):
    """x[r,:] = g[r] * w"""

    @staticmethod
    def forward(ctx, g, w, param=None):
        g, w = _req(g, "g"), _req(w, "w")
        ctx.save_for_backward(g, w)
        ctx.param = param
        return _rowscale_raw(w, g, g.numel(), w.numel(), True)

    @staticmethod
    def backward(ctx, v):
        g, w = ctx.saved_tensors
        dg = _DotF.apply(v, w, ctx.param) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            if ctx.param is not None and not torch.is_grad_enabled() and _sinks.enabled:
                if not _sink_coldot(ctx.param, _req(g), _req(v)):     # == _DotWg(v, g)
                    dw = _coldot_raw(_req(g), _req(v))
                    if _sink_grad(ctx.param, dw.view_as(ctx.param)):
                        dw = None
            else:
                dw = _DotWg.apply(v, g)
        return dg, dw, None


class _DotWg(torch.autograd.Function):
    """dw = sum_r g[r] * x[r,:]"""

    @staticmethod
    def forward(ctx, x, g):
        x, g = _req(x, "x"), _req(g, "g")
        ctx.save_for_backward(x, g)
        return _coldot_raw(g, x)

    @staticmethod
    def backward(ctx, v):
        x, g = ctx.saved_tensors
        dx = _DotDg.apply(g, v) if ctx.needs_input_grad[0] else None
        dg = _DotF.apply(x, v) if ctx.needs_input_grad[1] else None
        return dx, dg


def full_dot_conv(x, w):
    """Conv2d whose kernel covers the whole (unpadded) input: [N,C,H,W] x [1,C,H,W] -> [N,1,1,1]."""
    n = x.shape[0]
    y = _DotF.apply(x.reshape(n, -1), w.reshape(-1), w if isinstance(w, torch.nn.Parameter) else None)
    return y.reshape(n, 1, 1, 1)


__all__ = [n for n in list(globals()) if not n.startswith("__")]     # the flat namespace of the package (private helpers included)
