"""lightning_gan_zoo_amd.functional, part 5: HoloGAN -- ConvTranspose3d, AdaIN, fused linear layers, rigid-body resampling,
spectral normalisation (reference core/models/hologan_generator.py, hologan_discriminator.py)."""
import ctypes
import os
import weakref
from collections import namedtuple

import torch

from .._lib import check, lib
from ._base import *      # noqa: F401,F403
from ._conv import *      # noqa: F401,F403
from ._norm import *      # noqa: F401,F403
from ._misc import *      # noqa: F401,F403

# ---------------------------------------------------------------------------
# HoloGAN: 3-D transposed convolution family, AdaIN, dense layers with fused epilogue,
# rigid-body resampling, spectral normalisation
# ---------------------------------------------------------------------------

def _conv3d_fwd_raw(x, w, bias, act, slope):
    N, C, D, H, W = x.shape
    K, KS = w.shape[0], w.shape[2]
    OD, OH, OW = D // 2, H // 2, W // 2
    y = torch.empty((N, K, OD, OH, OW), device=x.device, dtype=torch.float32)
    wp = _packed3(w, "f")
    ws, nbytes = _scratch(lib.gz_conv3d_fwd_workspace_bytes(N, C, K, OD, OH, OW, KS), x.device)
    _timed_detail(lambda: "igemm3d<F> N%d C%d D%d K%d" % (N, C, D, K), 2.0 * N * OD * OH * OW * K * C * KS ** 3,
                  lambda: check(lib.gz_conv3d_fwd(_p(x), _p(wp), _p(bias), _p(y), _p(ws), nbytes, N, C, D, H, W, K, OD,
                                                  OH, OW, KS, 2, 1, act, slope, _stream()), "conv3d_fwd"))
    return y


def _conv3d_dgrad_raw(g, w, bias, act, slope):
    N, K, OD, OH, OW = g.shape
    C, KS = w.shape[1], w.shape[2]
    D, H, W = 2 * OD, 2 * OH, 2 * OW
    x = torch.empty((N, C, D, H, W), device=g.device, dtype=torch.float32)
    wp = _packed3(w, "d")
    ws, nbytes = _scratch(lib.gz_conv3d_dgrad_workspace_bytes(N, C, K, OD, OH, OW, KS), g.device)
    _timed_detail(lambda: "igemm3d<Dg> N%d C%d D%d K%d" % (N, C, D, K), 2.0 * N * OD * OH * OW * K * C * KS ** 3,
                  lambda: check(lib.gz_conv3d_dgrad(_p(g), _p(wp), _p(bias), _p(x), _p(ws), nbytes, N, C, D, H, W, K,
                                                    OD, OH, OW, KS, 2, 1, act, slope, _stream()), "conv3d_dgrad"))
    return x


def _conv3d_wgrad_raw(x, g, ks):
    N, C, D, H, W = x.shape
    _, K, OD, OH, OW = g.shape
    dw = torch.empty((K, C, ks, ks, ks), device=x.device, dtype=torch.float32)
    nbytes = lib.gz_conv3d_wgrad_workspace_bytes(N, C, K, OD, OH, OW, ks)
    ws = _ws(max(nbytes // 4, 1), x.device)
    _timed_detail(lambda: "igemm3d<Wg> N%d C%d D%d K%d" % (N, C, D, K), 2.0 * N * OD * OH * OW * K * C * ks ** 3,
                  lambda: check(lib.gz_conv3d_wgrad(_p(x), _p(g), _p(dw), _p(ws), nbytes, N, C, D, H, W, K, OD, OH, OW,
                                                    ks, 2, 1, _stream()), "conv3d_wgrad"))
    return dw


class _Conv3DDg(torch.autograd.Function):
    """x = conv_transpose3d(g, w) + bias   (k3, s2, p1, output_padding 1)"""

    @staticmethod
    def forward(ctx, g, w, bias, bias_cancels=False):
        g, w = _req(g, "g"), _req(w, "w")
        ctx.save_for_backward(g, w)
        ctx.w_ref = w
        ctx.has_bias = bias is not None
        ctx.bias_cancels = bias_cancels
        ctx.bias_ref = bias if isinstance(bias, torch.nn.Parameter) else None
        return _conv3d_dgrad_raw(g, w, bias, ACT_NONE, 0.0)

    @staticmethod
    def backward(ctx, v):
        g, w = ctx.saved_tensors
        v = _req(v)
        if torch.is_grad_enabled():
            dg = _Conv3DF.apply(v, w) if ctx.needs_input_grad[0] else None
            dw = _Conv3DWg.apply(v, g, w.shape[2]) if ctx.needs_input_grad[1] else None
        else:                                # no graph is being recorded: raw launches, the weight gradient into its sink
            dg = _conv3d_fwd_raw(v, w, None, ACT_NONE, 0.0) if ctx.needs_input_grad[0] else None
            dw = _sink_or_return(ctx.w_ref, _conv3d_wgrad_raw(v, g, w.shape[2])) if ctx.needs_input_grad[1] else None
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            if ctx.bias_cancels and not torch.is_grad_enabled():
                db = _sink_zero(ctx.bias_ref, (v.shape[1],), v.device)
            else:
                db = _ChannelSum.apply(v) if torch.is_grad_enabled() else _channel_sum_raw(v)
        return dg, dw, db, None


class _Conv3DF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        x, w = _req(x, "x"), _req(w, "w")
        ctx.save_for_backward(x, w)
        return _conv3d_fwd_raw(x, w, None, ACT_NONE, 0.0)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = _req(gy)
        dx = _Conv3DDg.apply(gy, w, None, False) if ctx.needs_input_grad[0] else None
        dw = _Conv3DWg.apply(x, gy, w.shape[2]) if ctx.needs_input_grad[1] else None
        return dx, dw


class _Conv3DWg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, ks):
        x, g = _req(x, "x"), _req(g, "g")
        ctx.save_for_backward(x, g)
        return _conv3d_wgrad_raw(x, g, ks)

    @staticmethod
    def backward(ctx, v):
        x, g = ctx.saved_tensors
        v = _req(v)
        dx = _Conv3DDg.apply(g, v, None, False) if ctx.needs_input_grad[0] else None
        dg = _Conv3DF.apply(x, v) if ctx.needs_input_grad[1] else None
        return dx, dg, None


def conv_transpose3d(x, w, bias=None, bias_cancels=False):
    """nn.ConvTranspose3d(kernel 3, stride 2, padding 1, output_padding 1); w [Cin, Cout, 3, 3, 3]; bias_cancels as in
    conv_transpose2d."""
    return _Conv3DDg.apply(x, w, bias, bias_cancels)


class _AdaINAct(torch.autograd.Function):
    """act(scale[n,c] * (x - mean) * rsqrt(var_unbiased + eps) + bias[n,c]); reference AdaIn
    (core/models/hologan_generator.py:333-345) followed by ReLU."""

    @staticmethod
    def forward(ctx, x, scale, bias, eps, act, slope):
        x, scale, bias = _req(x, "x"), _req(scale, "scale"), _req(bias, "bias")
        N, C = x.shape[:2]
        inner = x.numel() // (N * C)
        coef = torch.empty(4 * N * C, device=x.device, dtype=torch.float32)
        out = torch.empty_like(x)
        check(lib.gz_rownorm_act_fwd(_p(x), _p(scale), _p(bias), _p(coef), _p(out), N, C, inner, eps, 1, 1, act, slope,
                                     _stream()), "rownorm_act_fwd(adain)")
        ctx.save_for_backward(x, coef)
        ctx.cfg = (N, C, inner, act, slope)
        ctx.x_ref = x
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        x, coef = ctx.saved_tensors
        N, C, inner, act, slope = ctx.cfg
        gout = _req(gout)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        ds = torch.empty((N, C), device=x.device, dtype=torch.float32)
        db = torch.empty((N, C), device=x.device, dtype=torch.float32)
        kbuf = torch.empty(2 * N * C, device=x.device, dtype=torch.float32)
        ws = _norm_ws(x, N, C)
        check(lib.gz_norm_act_bwd(_p(gout), _p(x), _p(coef), _p(dx), _p(ds), _p(db), _p(ws), _p(kbuf), N, C, inner, 0,
                                  1, 1, act, slope, _stream()), "norm_act_bwd(adain)")
        return dx, ds, db, None, None, None


def adain_act(x, scale, bias, eps=1e-8, act=ACT_RELU, slope=0.0):
    return _AdaINAct.apply(x, scale, bias, eps, act, slope)


class _AdaINActPacked(torch.autograd.Function):
    """adain_act with scale and shift given as the two halves of ONE [N, 2C] tensor (the ZMapping output,
    hologan_generator.py:15-18): no slicing / copying on the way in, and the gradient comes back as one [N, 2C]
    tensor written by the kernel (the framework spelling cost 2 copies forward and a zeros + 2 slice copies + add
    backward per block)."""

    @staticmethod
    def forward(ctx, x, sb, eps, act, slope):
        x, sb = _req(x, "x"), _req(sb, "scale|shift")
        N, C = x.shape[:2]
        if sb.shape != (N, 2 * C):
            raise RuntimeError("adain_act_packed: expected scale|shift of shape [N, 2C]")
        inner = x.numel() // (N * C)
        coef = torch.empty(4 * N * C, device=x.device, dtype=torch.float32)
        sbp = sb.data_ptr()
        out = torch.empty_like(x)
        check(lib.gz_rownorm_act_fwd(_p(x), ctypes.c_void_p(sbp), ctypes.c_void_p(sbp + 4 * C), _p(coef), _p(out), N, C,
                                     inner, eps, 2, 1, act, slope, _stream()), "rownorm_act_fwd(adain, packed)")
        ctx.save_for_backward(x, coef)
        ctx.cfg = (N, C, inner, act, slope)
        ctx.x_ref = x
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        x, coef = ctx.saved_tensors
        N, C, inner, act, slope = ctx.cfg
        gout = _req(gout)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dsb = torch.empty((N, 2 * C), device=x.device, dtype=torch.float32)
        kbuf = torch.empty(2 * N * C, device=x.device, dtype=torch.float32)
        ws = _norm_ws(x, N, C)
        p0 = dsb.data_ptr()
        check(lib.gz_norm_act_bwd(_p(gout), _p(x), _p(coef), _p(dx), ctypes.c_void_p(p0), ctypes.c_void_p(p0 + 4 * C),
                                  _p(ws), _p(kbuf), N, C, inner, 0, 2, 1, act, slope, _stream()),
              "norm_act_bwd(adain, packed)")
        return dx, dsb, None, None, None


def adain_act_packed(x, sb, eps=1e-8, act=ACT_RELU, slope=0.0):
    return _AdaINActPacked.apply(x, sb, eps, act, slope)


class _AdaINConst(torch.autograd.Function):
    """adain_act_packed(x.repeat(N, ...), sb) for a constant x of shape [1, C, ...] without materialising the
    repeat or the per-sample input gradient (reference hologan_generator.py:141-142)."""

    @staticmethod
    def forward(ctx, x, sb, eps, act, slope):
        x, sb = _req(x, "x"), _req(sb, "scale|shift")
        N, C = sb.shape[0], x.shape[1]
        inner = x.numel() // C
        if x.shape[0] != 1 or sb.shape[1] != 2 * C:
            raise RuntimeError("adain_const: expected x [1, C, ...] and scale|shift [N, 2C]")
        coef = torch.empty(4 * N * C, device=x.device, dtype=torch.float32)
        out = torch.empty((N,) + tuple(x.shape[1:]), device=x.device, dtype=torch.float32)
        check(lib.gz_adain_const_fwd(_p(x), _p(sb), _p(coef), _p(out), N, C, inner, eps, act, slope, _stream()),
              "adain_const_fwd")
        ctx.save_for_backward(x, coef)
        ctx.cfg = (N, C, inner, act, slope)
        ctx.x_ref = x
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        x, coef = ctx.saved_tensors
        N, C, inner, act, slope = ctx.cfg
        gout = _req(gout)
        dx = torch.empty_like(x)
        dsb = torch.empty((N, 2 * C), device=x.device, dtype=torch.float32)
        check(lib.gz_adain_const_bwd(_p(gout), _p(x), _p(coef), _p(dx), _p(dsb), N, C, inner, act, slope, _stream()),
              "adain_const_bwd")
        return (_sink_or_return(ctx.x_ref, dx) if ctx.needs_input_grad[0] else None), dsb, None, None, None


def adain_const_act(x, sb, eps=1e-8, act=ACT_RELU, slope=0.0):
    if x.numel() // x.shape[1] > 1024:          # rows longer than the constant-input kernel keeps in registers
        return adain_act_packed(x.repeat(sb.shape[0], *([1] * (x.dim() - 1))), sb, eps, act, slope)
    return _AdaINConst.apply(x, sb, eps, act, slope)


class _LinearAct(torch.autograd.Function):
    """act(x @ W^T + b) with bias and activation fused in the GEMM epilogue."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, slope):
        x, weight = _req(x, "x"), _req(weight, "weight")
        out = gemm(x, weight, bias, trans_b=True, act=act, slope=slope)
        ctx.save_for_backward(x, weight, out if act != ACT_NONE else None)
        ctx.act, ctx.slope, ctx.has_bias = act, slope, bias is not None
        ctx.bias_ref = bias if isinstance(bias, torch.nn.Parameter) else None
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x, weight, out = ctx.saved_tensors
        g = _req(g)
        if ctx.act != ACT_NONE:
            if g.numel() % 4 == 0:
                g = _act_bwd_raw(g, out, ctx.act, ctx.slope)
            else:   # odd tiny shapes
                d = {ACT_RELU: (out > 0).float(), ACT_TANH: 1 - out * out}.get(ctx.act)
                g = g * (d if d is not None else torch.where(out > 0, 1.0, ctx.slope))
        dx = gemm(g, weight) if ctx.needs_input_grad[0] else None
        dw = gemm(g, x, trans_a=True) if ctx.needs_input_grad[1] else None
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty(g.shape[1], device=g.device, dtype=torch.float32)
            check(lib.gz_colsum(_p(g), _p(db), g.shape[0], g.shape[1], _stream()), "colsum")
            db = _sink_or_return(ctx.bias_ref, db) if ctx.bias_ref is not None else db
        if dw is not None:
            dw = _sink_or_return(weight, dw)
        return dx, dw, db, None, None


def linear_act(x, weight, bias=None, act=ACT_NONE, slope=0.0):
    return _LinearAct.apply(x, weight, bias, act, slope)


class _LinearActMulti(torch.autograd.Function):
    """Several act(x @ W_j^T + b_j) over ONE x in one launch; one launch for every dW_j / db_j backward."""

    @staticmethod
    def forward(ctx, x, act, slope, *wb):
        x = _req(x, "x")
        ws = [_req(w, "weight") for w in wb[0::2]]
        bs = [None if b is None else _req(b, "bias") for b in wb[1::2]]
        N, K = x.shape
        if any(w.dim() != 2 or w.shape[1] != K for w in ws) or len(ws) > lib.gz_linear_multi_max_jobs():
            raise ValueError("linear_act_multi: weights must be [J, %d], at most %d of them"
                             % (K, lib.gz_linear_multi_max_jobs()))
        outs = tuple(torch.empty((N, w.shape[0]), device=x.device, dtype=torch.float32) for w in ws)
        table = (ctypes.c_char * lib.gz_linear_multi_table_bytes())()
        for w, b, o in zip(ws, bs, outs):
            check(lib.gz_linear_multi_add(table, _p(w), _p(b), _p(o), None, None, None, w.shape[0]), "linear_multi_add")
        check(lib.gz_linear_multi_fwd(table, _p(x), N, K, act, slope, _stream()), "linear_multi_fwd")
        ctx.save_for_backward(x, *ws, *outs)
        ctx.act, ctx.slope, ctx.has_bias = act, slope, [b is not None for b in bs]
        ctx.params = tuple(wb)               # the Parameters themselves: their gradients join the sinks
        return outs

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gs):
        saved = ctx.saved_tensors
        n = len(ctx.has_bias)
        x, ws, outs = saved[0], saved[1:1 + n], saved[1 + n:]
        N, K = x.shape
        gs = [torch.zeros_like(o) if g is None else _req(g) for g, o in zip(gs, outs)]
        grads = []
        table = (ctypes.c_char * lib.gz_linear_multi_table_bytes())()
        for j, (w, o, g) in enumerate(zip(ws, outs, gs)):
            dw = torch.empty_like(w)
            db = torch.empty(w.shape[0], device=w.device, dtype=torch.float32) if ctx.has_bias[j] else None
            check(lib.gz_linear_multi_add(table, None, None, _p(o), _p(g), _p(dw), _p(db), w.shape[0]), "linear_multi_add")
            grads += [dw, db]
        check(lib.gz_linear_multi_bwd(table, _p(x), N, K, ctx.act, ctx.slope, _stream()), "linear_multi_bwd")
        dx = None
        if ctx.needs_input_grad[0]:          # (z is noise in the shipped models: not on their path)
            for w, o, g in zip(ws, outs, gs):
                gm = g if ctx.act == ACT_NONE else g * _act_derivative(o, ctx.act, ctx.slope)
                part = gemm(gm.contiguous(), w)
                dx = part if dx is None else dx + part
        # complete gradients: one-slab sources of the sink flush (under data parallelism p.grad is a view of the flat
        # exchange buffer and autograd would spend an `add_` launch per parameter: 10 per generator step)
        grads = [None if not ctx.needs_input_grad[3 + j] else _sink_or_return(p_, g_)
                 for j, (p_, g_) in enumerate(zip(ctx.params, grads))]
        return (dx, None, None, *grads)


def _act_derivative(out, act, slope):
    if act == ACT_RELU:
        return (out > 0).float()
    if act == ACT_TANH:
        return 1 - out * out
    return torch.where(out > 0, 1.0, slope)


def linear_act_multi(x, layers, act=ACT_NONE, slope=0.0):
    """``[linear_act(x, w, b, act) for (w, b) in layers]`` in ONE launch (and one launch for all the weight and bias
    gradients): HoloGAN's five ZMapping layers over the same z (reference core/models/hologan_generator.py:7-19)."""
    flat = []
    for w, b in layers:
        flat += [w, b]
    return _LinearActMulti.apply(x, act, slope, *flat)


class _RigidResample(torch.autograd.Function):
    """[N,C,S,S,S] voxels + [N,16] inverse view matrices -> [N, C*S, S, S] projected feature map."""

    @staticmethod
    def forward(ctx, vox, minv):
        vox, minv = _req(vox, "vox"), _req(minv, "minv")
        N, C, S = vox.shape[0], vox.shape[1], vox.shape[2]
        out = torch.empty((N, C * S, S, S), device=vox.device, dtype=torch.float32)
        check(lib.gz_rigid_resample_fwd(_p(vox), _p(minv), _p(out), None, N, C, S, _stream()), "rigid_resample_fwd")
        ctx.save_for_backward(minv)
        ctx.shape = (N, C, S)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (minv,) = ctx.saved_tensors
        N, C, S = ctx.shape
        g = _req(g)
        gv = torch.empty((N, C, S, S, S), device=g.device, dtype=torch.float32)
        ws, nbytes = _scratch((lib.gz_rigid_resample_bwd_workspace_bytes(N, S) + 3) // 4 * 4, g.device)
        check(lib.gz_rigid_resample_bwd(_p(g), _p(minv), _p(gv), _p(ws), nbytes, N, C, S, _stream()),
              "rigid_resample_bwd")
        return gv, None


def rigid_resample(vox, minv):
    return _RigidResample.apply(vox, minv)


def rigid_resample_indices(vox, minv):
    """Debug / test hook: the int64 corner indices idx_a..idx_h the kernel uses, [8, N*S^3]."""
    N, C, S = vox.shape[0], vox.shape[1], vox.shape[2]
    out = torch.empty((N, C * S, S, S), device=vox.device, dtype=torch.float32)
    idx = torch.empty((8, N * S ** 3), device=vox.device, dtype=torch.int64)
    check(lib.gz_rigid_resample_fwd(_p(_req(vox)), _p(_req(minv)), _p(out), _p(idx), N, C, S, _stream()),
          "rigid_resample_fwd")
    return out, idx


class _SpectralNormWeight(torch.autograd.Function):
    """weight_orig / sigma with sigma = u^T W v after one power iteration (training: the module's u / v buffers are
    updated in place, like torch.nn.utils.spectral_norm); u, v enter sigma as constants.  Six launches forward, two
    backward (the framework spelling took ~16 + ~10)."""

    @staticmethod
    def forward(ctx, weight_orig, u, v, training, eps):
        W = _req(weight_orig, "weight_orig")
        R = W.shape[0]
        L = W.numel() // R
        st = _stream()
        dev = W.device
        us = torch.empty(R, device=dev, dtype=torch.float32)       # the copies this node keeps (the buffers move on)
        vs = torch.empty(L, device=dev, dtype=torch.float32)
        sigma = torch.empty(1, device=dev, dtype=torch.float32)
        Wm = W.view(R, L)
        if training:
            v_raw = _coldot_raw(u, Wm)                              # W^T u
            check(lib.gz_vec_normalize(_p(v_raw), _p(v), _p(vs), None, L, eps, st), "vec_normalize(v)")
            wv = _rowdot_raw(Wm, vs, True)                          # W v
            check(lib.gz_vec_normalize_dot(_p(wv), _p(u), _p(us), _p(sigma), R, eps, st), "vec_normalize(u), sigma")
        else:
            us.copy_(u)
            vs.copy_(v)
            wv = _rowdot_raw(Wm, vs, True)
            check(lib.gz_vec_dot(_p(us), _p(wv), _p(sigma), R, st), "vec_dot(sigma)")
        w = torch.empty_like(W)
        check(lib.gz_div_scalar(_p(W), _p(sigma), _p(w), W.numel(), st), "div_scalar")
        ctx.save_for_backward(w, us, vs, sigma)
        return w

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        w, us, vs, sigma = ctx.saved_tensors
        g = _req(g)
        R = w.shape[0]
        L = w.numel() // R
        rowdots = _rowdot_raw(g.view(R, L), w.view(R, L), False)
        out = torch.empty_like(w)
        check(lib.gz_spectral_norm_bwd(_p(g), _p(rowdots), _p(us), _p(vs), _p(sigma), _p(out), R, L, _stream()),
              "spectral_norm_bwd")
        return out, None, None, None, None


def spectral_normalize(weight_orig, u, v, training, eps=1e-12):
    """torch.nn.utils.spectral_norm's weight: one power iteration (training: u, v updated in place),
    then weight_orig / sigma with sigma = u^T W v differentiable w.r.t. weight_orig.  The two
    matrix-vector products per iteration run on gz_coldot / gz_rowdot, the rest on csrc/gz_loss.hip."""
    L = weight_orig.numel() // weight_orig.shape[0]
    if L % 4 == 0:
        return _SpectralNormWeight.apply(weight_orig, u, v, bool(training), float(eps))
    w_mat = weight_orig.reshape(weight_orig.shape[0], -1)           # odd row lengths: framework arithmetic
    if training:
        with torch.no_grad():
            wd = _req(w_mat.detach())
            v_new = _coldot_raw(u, wd)                       # W^T u
            v_new = v_new / v_new.norm().clamp_min(eps)
            u_new = _rowdot_raw(wd, v_new, True)             # W v
            u_new = u_new / u_new.norm().clamp_min(eps)
            v.copy_(v_new)
            u.copy_(u_new)
    uc, vc = u.clone(), v.clone()
    sigma = torch.dot(uc, _DotF.apply(w_mat, vc))
    return weight_orig / sigma


class _SpectralNormMulti(torch.autograd.Function):
    """spectral_normalize for SEVERAL conv weights of one discriminator call: the power iterations in four launches
    (gz_sn_power_iteration), w = weight_orig / sigma together with both packed images of every layer in a fifth
    (gz_conv2d_pack_table_launch) -- 8 launches per layer before.  The packed images ride on the returned tensors
    (``_gz_packs``) and are picked up by ``_packed``."""

    @staticmethod
    def forward(ctx, eps, geom, *wuv):
        ctx.params = wuv[0::3]
        Ws = [_req(w, "weight_orig") for w in wuv[0::3]]
        us_buf, vs_buf = wuv[1::3], wuv[2::3]
        dev, st = Ws[0].device, _stream()
        if len(Ws) > lib.gz_sn_max_jobs() or 3 * len(Ws) > lib.gz_conv2d_pack_table_max_jobs():
            raise ValueError("spectral_normalize_multi: at most %d layers" % lib.gz_sn_max_jobs())
        table = (ctypes.c_char * lib.gz_sn_table_bytes())()
        packs = (ctypes.c_char * lib.gz_conv2d_pack_table_bytes())()
        keep, saved, outs = [], [], []
        for W, u, v in zip(Ws, us_buf, vs_buf):
            K, C, KH, KW = W.shape
            R, L = K, C * KH * KW
            us = torch.empty(R, device=dev, dtype=torch.float32)
            vs = torch.empty(L, device=dev, dtype=torch.float32)
            sigma = torch.empty(1, device=dev, dtype=torch.float32)
            ws = _ws(lib.gz_sn_workspace_floats(R, L), dev)
            check(lib.gz_sn_add(table, _p(W), _p(u), _p(v), _p(us), _p(vs), _p(sigma), _p(ws), R, L), "sn_add")
            w = torch.empty_like(W)
            wf = torch.empty(lib.gz_conv2d_pack_fwd_elems(K, C, KH, KW), device=dev, dtype=torch.float32)
            wd = torch.empty(lib.gz_conv2d_pack_dgrad_elems(K, C, KH, KW, geom.stride), device=dev, dtype=torch.float32)
            for what, dst in ((2, w), (0, wf), (1, wd)):
                check(lib.gz_conv2d_pack_table_add(packs, _p(W), _p(dst), _p(sigma), what, K, C, KH, KW, geom.stride,
                                                   geom.pad), "pack_table_add")
            w._gz_packs = {"f": wf, "d": wd, "geom": geom, "version": w._version}
            keep.append(ws)
            saved += [w, us, vs, sigma]
            outs.append(w)
        check(lib.gz_sn_power_iteration(table, eps, st), "sn_power_iteration")
        check(lib.gz_conv2d_pack_table_launch(packs, st), "pack_table_launch")
        del keep            # (stream-ordered allocator: the workspaces may be reused by later launches of this stream)
        ctx.save_for_backward(*saved)
        return tuple(outs)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gs):
        saved = ctx.saved_tensors
        grads = []
        for k, g in enumerate(gs):
            w, us, vs, sigma = saved[4 * k: 4 * k + 4]
            if g is None:
                grads += [None, None, None]
                continue
            g = _req(g)
            R = w.shape[0]
            L = w.numel() // R
            rowdots = _rowdot_raw(g.view(R, L), w.view(R, L), False)
            out = torch.empty_like(w)
            check(lib.gz_spectral_norm_bwd(_p(g), _p(rowdots), _p(us), _p(vs), _p(sigma), _p(out), R, L, _stream()),
                  "spectral_norm_bwd")
            grads += [_sink_or_return(ctx.params[k], out), None, None]
        return (None, None, *grads)


def spectral_normalize_multi(layers, training, geom, eps=1e-12):
    """``[spectral_normalize(W, u, v, training) for (W, u, v) in layers]`` for the conv weights of one discriminator call
    (all of geometry ``geom``); training mode on 4-D weights with C*KH*KW % 4 == 0 takes the five-launch path."""
    ok = training and all(W.dim() == 4 and (W.numel() // W.shape[0]) % 4 == 0 and W.is_cuda and W.is_contiguous()
                          and W.data_ptr() % 16 == 0 and v.data_ptr() % 16 == 0 for W, _, v in layers)
    if not ok or len(layers) > lib.gz_sn_max_jobs():
        return [spectral_normalize(W, u, v, training, eps) for W, u, v in layers]
    flat = []
    for W, u, v in layers:
        flat += [W, u, v]
    return list(_SpectralNormMulti.apply(float(eps), geom, *flat))


def spectral_power_iterations(layers, calls=1, eps=1e-12):
    """``calls`` consecutive power iterations of torch.nn.utils.spectral_norm for each (weight_orig, u, v) -- what
    ``calls`` discriminator forward passes would run one after the other (the module buffers end up where the last call
    leaves them).  Returns per layer (sigma [calls], u [calls, R], v [calls, L]): the values each call works with.
    No autograd: sigma's dependence on the weight is handled by sn_conv_in_act's backward (u, v constants, as in torch)."""
    st = _stream()
    out = []
    with torch.no_grad():
        for W, _, _ in layers:
            R, L = W.shape[0], W.numel() // W.shape[0]
            out.append((torch.empty(calls, device=W.device, dtype=torch.float32),
                        torch.empty((calls, R), device=W.device, dtype=torch.float32),
                        torch.empty((calls, L), device=W.device, dtype=torch.float32)))
        keep = []
        for g in range(calls):
            table = (ctypes.c_char * lib.gz_sn_table_bytes())()
            for (W, u, v), (sig, us, vs) in zip(layers, out):
                W = _req(W.detach(), "weight_orig")
                R, L = W.shape[0], W.numel() // W.shape[0]
                ws = _ws(lib.gz_sn_workspace_floats(R, L), W.device)
                keep.append(ws)
                check(lib.gz_sn_add(table, _p(W), _p(u), _p(v), _p(us[g]), _p(vs[g]), _p(sig[g:g + 1]), _p(ws), R, L),
                      "sn_add")
            check(lib.gz_sn_power_iteration(table, eps, st), "sn_power_iteration")
        del keep
    return out


class _SNConvINAct(torch.autograd.Function):
    """act(InstanceNorm(conv2d(x, weight_orig / sigma) + bias)) for the spectral-normalised blocks of HoloGAN's critic
    (reference core/models/hologan_discriminator.py:28-38), computed as act(IN_{eps sigma^2}(conv2d(x, weight_orig))):
    the InstanceNorm removes the bias and any scale except through eps, so the convolution runs on weight_orig's own
    packed images (cached for the whole optimizer step instead of re-packed at every call) and ``groups`` discriminator
    calls with different sigma can share one pass over the stacked batch.  Backward: weight_orig's gradient is the
    convolution's weight gradient plus dL/dsigma_g u_g v_g^T (gz_sn_sigma_term: exactly torch's -(sum g w) u v^T / sigma);
    the bias gradient is exactly zero (the reference's is rounding noise)."""

    @staticmethod
    def forward(ctx, x, weight_orig, bias, sigma, us, vs, geom, in_eps, act, slope):
        x, W = _req(x, "x"), _req(weight_orig, "weight_orig")
        groups = sigma.numel()
        N = x.shape[0]
        y = _conv_fwd_raw(x, W, None, geom, ACT_NONE, 0.0)
        C, inner = y.shape[1], y.shape[2] * y.shape[3]
        coef = torch.empty(4 * N * C, device=x.device, dtype=torch.float32)
        out = torch.empty_like(y)
        check(lib.gz_rownorm_act_fwd_sigma(_p(y), _p(sigma), groups, _p(coef), _p(out), N, C, inner, in_eps, act, slope,
                                           _stream()), "rownorm_act_fwd_sigma")
        ctx.save_for_backward(x, W, y, coef, sigma, us, vs)
        ctx.cfg = (geom, in_eps, act, slope, groups, bias is not None)
        ctx.bias_shape = None if bias is None else tuple(bias.shape)
        ctx.bias_ref = bias if isinstance(bias, torch.nn.Parameter) else None
        ctx.param = weight_orig
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        x, W, y, coef, sigma, us, vs = ctx.saved_tensors
        geom, in_eps, act, slope, groups, has_bias = ctx.cfg
        gout = _req(gout)
        N, C = y.shape[0], y.shape[1]
        inner = y.shape[2] * y.shape[3]
        st = _stream()
        g_raw = torch.empty_like(y)
        rowsums = torch.empty(2 * N * C, device=y.device, dtype=torch.float32)
        check(lib.gz_rownorm_act_bwd_rows(_p(gout), _p(y), _p(coef), _p(g_raw), _p(rowsums), N, C, inner, act, slope, st),
              "rownorm_act_bwd_rows")
        dx = _conv_dgrad_raw(g_raw, W, None, geom, tuple(x.shape[2:]), ACT_NONE, 0.0) if ctx.needs_input_grad[0] else None
        dW = None
        if ctx.needs_input_grad[1]:
            R, L = W.shape[0], W.numel() // W.shape[0]
            coefs = torch.empty(lib.gz_sn_sigma_coef_floats(groups), device=y.device, dtype=torch.float32)
            term = torch.empty_like(W)
            check(lib.gz_sn_sigma_term(_p(rowsums), _p(coef[3 * N * C:]), _p(sigma), _p(us), _p(vs), _p(coefs), _p(term),
                                       N * C, groups, R, L, in_eps, st), "sn_sigma_term")
            p = ctx.param
            # weight_orig is a leaf: the convolution's weight gradient and the sigma term join its sink as two sources
            sunk = isinstance(p, torch.nn.Parameter) and not (W.numel() & 3) and _sink_conv_wgrad(p, x, g_raw, geom)
            if sunk:     # (setdefault: a weight whose launch GradSync postponed has no entry yet)
                _sinks.pending.setdefault(id(p), [p, []])[1].append((term, 1, W.numel()))
            if not sunk:
                dW = _conv_wgrad_raw(x, g_raw, geom)
                dW.add_(term)
        db = _sink_zero(ctx.bias_ref, ctx.bias_shape, y.device) if (has_bias and ctx.needs_input_grad[2]) else None
        return dx, dW, db, None, None, None, None, None, None, None


def sn_conv_in_act(x, weight_orig, bias, sigma, us, vs, geom, in_eps=1e-5, act=ACT_NONE, slope=0.0):
    """One spectral-normalised conv + InstanceNorm2d(affine=False) + activation block; sigma [groups], us [groups, R],
    vs [groups, L] from spectral_power_iterations (groups = discriminator calls stacked along the batch)."""
    return _SNConvINAct.apply(x, weight_orig, bias, sigma, us, vs, geom, float(in_eps), act, slope)


__all__ = [n for n in list(globals()) if not n.startswith("__")]     # the flat namespace of the package (private helpers included)
