"""The FID / KID feature extractor on the HIP kernels.

``InceptionV3([3])`` of the reference (core/callback_inception_metrics.py:204-222,
core/submodules/gan_stability/metrics/inception.py:16-311): torchvision's ``inception_v3(num_classes=1008,
aux_logits=False)`` with the FID patches, bilinear resize to 299 x 299, [0, 1] -> [-1, 1], 2048 pool features.
Same module / parameter names as torchvision, so the weight file the reference downloads
(``pt_inception-2015-12-05-6726825d.pth``, inception.py:13) loads with ``load_fid_weights(path)`` -- there is no
network here, the file has to be supplied by the user.

Forward only (evaluation): every convolution is ``gz_conv2d_fwd_any`` (MFMA implicit GEMM with run-time geometry:
3x3 s2, 5x5 p2, 1x7 / 7x1, 1x3 / 3x1 ...) with the eval-mode BatchNorm folded into its weights and bias and the ReLU
in its epilogue; pools and the resize are ``gz_pool2d`` / ``gz_resize_bilinear``.  The nn.Conv2d / nn.BatchNorm2d
children are parameter holders.
"""
import ctypes

import numpy as np
import torch
from torch import nn

from . import functional as F
from ._lib import check, lib

_p, _stream = F._p, F._stream


def _pool(x, ks, stride, pad, mode):
    N, C, H, W = x.shape
    OH, OW = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    y = torch.empty((N, C, OH, OW), device=x.device, dtype=torch.float32)
    check(lib.gz_pool2d(_p(x), _p(y), N * C, H, W, OH, OW, ks, stride, pad, mode, _stream()), "pool2d")
    return y


MAX, AVG_NOPAD, AVG = 0, 1, 2
FLOPS = None          # bench.py: a one-element list that accumulates the convolutions' algorithmic FLOP (2 x MAC)


class BasicConv2d(nn.Module):
    """conv (no bias) -> BatchNorm(eps 1e-3, running statistics) -> ReLU as ONE launch: w' = w * gamma / sqrt(var + eps),
    b' = beta - mean * gamma / sqrt(var + eps).  The folded, packed weights are cached until a parameter changes."""

    def __init__(self, cin, cout, kernel_size, stride=1, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size, stride, padding, bias=False)
        self.bn = nn.BatchNorm2d(cout, eps=0.001)
        self._folded = None

    def _fold(self):
        c, bn = self.conv, self.bn
        key = tuple((t.data_ptr(), t._version) for t in (c.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var))
        if self._folded is not None and self._folded[0] == key:
            return self._folded[1:]
        with torch.no_grad():
            scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
            w = (c.weight * scale.view(-1, 1, 1, 1)).contiguous()
            b = (bn.bias - bn.running_mean * scale).contiguous()
            K, C, KH, KW = w.shape
            wp = torch.empty(lib.gz_conv2d_pack_fwd_any_elems(K, C, KH, KW), device=w.device, dtype=torch.float32)
            check(lib.gz_conv2d_pack_fwd_any(_p(w), _p(wp), K, C, KH, KW, _stream()), "conv2d_pack_fwd_any")
        self._folded = (key, wp, b)
        return wp, b

    def out_shape(self, x):
        c = self.conv
        (KH, KW), (SH, SW), (PH, PW) = c.kernel_size, c.stride, c.padding
        return (x.shape[0], c.out_channels, (x.shape[2] + 2 * PH - KH) // SH + 1, (x.shape[3] + 2 * PW - KW) // SW + 1)

    def forward(self, x, out=None):
        """``out``: a channel slice [:, a:b] of a contiguous (N, channels, OH, OW) tensor -- the block's concatenation --
        to write into instead of a fresh tensor (round 6: the eleven torch.cat of a pass were 3 % of it)."""
        x = F._req(x, "x")
        wp, b = self._fold()
        c = self.conv
        N, C, H, W = x.shape
        K = c.out_channels
        (KH, KW), (SH, SW), (PH, PW) = c.kernel_size, c.stride, c.padding
        OH, OW = (H + 2 * PH - KH) // SH + 1, (W + 2 * PW - KW) // SW + 1
        if FLOPS is not None:
            FLOPS[0] += 2.0 * N * OH * OW * K * C * KH * KW
        if out is not None:
            assert out.shape == (N, K, OH, OW) and out.stride()[1:] == (OH * OW, OW, 1) and out.stride(0) % (OH * OW) == 0
            rc = lib.gz_conv2d_fwd_any_into(_p(x), _p(wp), _p(b), ctypes.c_void_p(out.data_ptr()),
                                            out.stride(0) // (OH * OW), N, C, H, W, K, OH, OW, KH, KW, SH, SW, PH, PW,
                                            F.ACT_RELU, 0.0, _stream())
            if rc == 0:
                return out
            if rc != -2:                     # (-2: this shape does not run on the launch with a destination stride)
                check(rc, "conv2d_fwd_any_into")
        y = torch.empty((N, K, OH, OW), device=x.device, dtype=torch.float32)
        ws, nbytes = F._scratch(lib.gz_conv2d_fwd_any_workspace_bytes(N, C, H, W, K, OH, OW, KH, KW, SH, SW, PH, PW),
                                x.device)
        check(lib.gz_conv2d_fwd_any(_p(x), _p(wp), _p(b), _p(y), _p(ws), nbytes, N, C, H, W, K, OH, OW, KH, KW, SH, SW,
                                    PH, PW, F.ACT_RELU, 0.0, _stream()), "conv2d_fwd_any")
        if out is not None:
            out.copy_(y)
            return out
        return y


def _concat(x, OH, OW, parts):
    """torch.cat(..., 1) of a block's branches without the copy: every branch's last layer writes its channel slice of
    the result.  parts: [(last BasicConv2d | channel count, fn(out_slice))] in the reference's concatenation order."""
    chans = [p.conv.out_channels if isinstance(p, BasicConv2d) else int(p) for p, _ in parts]
    y = torch.empty((x.shape[0], sum(chans), OH, OW), device=x.device, dtype=torch.float32)
    a = 0
    for c, (_, fn) in zip(chans, parts):
        fn(y[:, a:a + c])
        a += c
    return y


class InceptionA(nn.Module):
    def __init__(self, cin, pool_features):
        super().__init__()
        self.branch1x1 = BasicConv2d(cin, 64, 1)
        self.branch5x5_1 = BasicConv2d(cin, 48, 1)
        self.branch5x5_2 = BasicConv2d(48, 64, 5, padding=2)
        self.branch3x3dbl_1 = BasicConv2d(cin, 64, 1)
        self.branch3x3dbl_2 = BasicConv2d(64, 96, 3, padding=1)
        self.branch3x3dbl_3 = BasicConv2d(96, 96, 3, padding=1)
        self.branch_pool = BasicConv2d(cin, pool_features, 1)

    def forward(self, x):
        return _concat(x, x.shape[2], x.shape[3], [
            (self.branch1x1, lambda o: self.branch1x1(x, o)),
            (self.branch5x5_2, lambda o: self.branch5x5_2(self.branch5x5_1(x), o)),
            (self.branch3x3dbl_3, lambda o: self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x)), o)),
            (self.branch_pool, lambda o: self.branch_pool(_pool(x, 3, 1, 1, AVG_NOPAD), o))])


class InceptionB(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.branch3x3 = BasicConv2d(cin, 384, 3, stride=2)
        self.branch3x3dbl_1 = BasicConv2d(cin, 64, 1)
        self.branch3x3dbl_2 = BasicConv2d(64, 96, 3, padding=1)
        self.branch3x3dbl_3 = BasicConv2d(96, 96, 3, stride=2)

    def forward(self, x):
        return _concat(x, (x.shape[2] - 3) // 2 + 1, (x.shape[3] - 3) // 2 + 1, [
            (self.branch3x3, lambda o: self.branch3x3(x, o)),
            (self.branch3x3dbl_3, lambda o: self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x)), o)),
            (x.shape[1], lambda o: o.copy_(_pool(x, 3, 2, 0, MAX)))])


class InceptionC(nn.Module):
    def __init__(self, cin, c7):
        super().__init__()
        self.branch1x1 = BasicConv2d(cin, 192, 1)
        self.branch7x7_1 = BasicConv2d(cin, c7, 1)
        self.branch7x7_2 = BasicConv2d(c7, c7, (1, 7), padding=(0, 3))
        self.branch7x7_3 = BasicConv2d(c7, 192, (7, 1), padding=(3, 0))
        self.branch7x7dbl_1 = BasicConv2d(cin, c7, 1)
        self.branch7x7dbl_2 = BasicConv2d(c7, c7, (7, 1), padding=(3, 0))
        self.branch7x7dbl_3 = BasicConv2d(c7, c7, (1, 7), padding=(0, 3))
        self.branch7x7dbl_4 = BasicConv2d(c7, c7, (7, 1), padding=(3, 0))
        self.branch7x7dbl_5 = BasicConv2d(c7, 192, (1, 7), padding=(0, 3))
        self.branch_pool = BasicConv2d(cin, 192, 1)

    def forward(self, x):
        return _concat(x, x.shape[2], x.shape[3], [
            (self.branch1x1, lambda o: self.branch1x1(x, o)),
            (self.branch7x7_3, lambda o: self.branch7x7_3(self.branch7x7_2(self.branch7x7_1(x)), o)),
            (self.branch7x7dbl_5, lambda o: self.branch7x7dbl_5(self.branch7x7dbl_4(self.branch7x7dbl_3(
                self.branch7x7dbl_2(self.branch7x7dbl_1(x)))), o)),
            (self.branch_pool, lambda o: self.branch_pool(_pool(x, 3, 1, 1, AVG_NOPAD), o))])


class InceptionD(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.branch3x3_1 = BasicConv2d(cin, 192, 1)
        self.branch3x3_2 = BasicConv2d(192, 320, 3, stride=2)
        self.branch7x7x3_1 = BasicConv2d(cin, 192, 1)
        self.branch7x7x3_2 = BasicConv2d(192, 192, (1, 7), padding=(0, 3))
        self.branch7x7x3_3 = BasicConv2d(192, 192, (7, 1), padding=(3, 0))
        self.branch7x7x3_4 = BasicConv2d(192, 192, 3, stride=2)

    def forward(self, x):
        return _concat(x, (x.shape[2] - 3) // 2 + 1, (x.shape[3] - 3) // 2 + 1, [
            (self.branch3x3_2, lambda o: self.branch3x3_2(self.branch3x3_1(x), o)),
            (self.branch7x7x3_4, lambda o: self.branch7x7x3_4(self.branch7x7x3_3(self.branch7x7x3_2(
                self.branch7x7x3_1(x))), o)),
            (x.shape[1], lambda o: o.copy_(_pool(x, 3, 2, 0, MAX)))])


class InceptionE(nn.Module):
    def __init__(self, cin, pool_mode):
        super().__init__()
        self.pool_mode = pool_mode     # AVG_NOPAD in Mixed_7b, MAX in Mixed_7c (inception.py:258-306)
        self.branch1x1 = BasicConv2d(cin, 320, 1)
        self.branch3x3_1 = BasicConv2d(cin, 384, 1)
        self.branch3x3_2a = BasicConv2d(384, 384, (1, 3), padding=(0, 1))
        self.branch3x3_2b = BasicConv2d(384, 384, (3, 1), padding=(1, 0))
        self.branch3x3dbl_1 = BasicConv2d(cin, 448, 1)
        self.branch3x3dbl_2 = BasicConv2d(448, 384, 3, padding=1)
        self.branch3x3dbl_3a = BasicConv2d(384, 384, (1, 3), padding=(0, 1))
        self.branch3x3dbl_3b = BasicConv2d(384, 384, (3, 1), padding=(1, 0))
        self.branch_pool = BasicConv2d(cin, 192, 1)

    def forward(self, x):
        b3 = self.branch3x3_1(x)
        d = self.branch3x3dbl_2(self.branch3x3dbl_1(x))
        return _concat(x, x.shape[2], x.shape[3], [
            (self.branch1x1, lambda o: self.branch1x1(x, o)),
            (self.branch3x3_2a, lambda o: self.branch3x3_2a(b3, o)), (self.branch3x3_2b, lambda o: self.branch3x3_2b(b3, o)),
            (self.branch3x3dbl_3a, lambda o: self.branch3x3dbl_3a(d, o)),
            (self.branch3x3dbl_3b, lambda o: self.branch3x3dbl_3b(d, o)),
            (self.branch_pool, lambda o: self.branch_pool(_pool(x, 3, 1, 1, self.pool_mode), o))])


class FIDInceptionV3(nn.Module):
    def __init__(self):
        super().__init__()
        self.Conv2d_1a_3x3 = BasicConv2d(3, 32, 3, stride=2)
        self.Conv2d_2a_3x3 = BasicConv2d(32, 32, 3)
        self.Conv2d_2b_3x3 = BasicConv2d(32, 64, 3, padding=1)
        self.Conv2d_3b_1x1 = BasicConv2d(64, 80, 1)
        self.Conv2d_4a_3x3 = BasicConv2d(80, 192, 3)
        self.Mixed_5b = InceptionA(192, 32)
        self.Mixed_5c = InceptionA(256, 64)
        self.Mixed_5d = InceptionA(288, 64)
        self.Mixed_6a = InceptionB(288)
        self.Mixed_6b = InceptionC(768, 128)
        self.Mixed_6c = InceptionC(768, 160)
        self.Mixed_6d = InceptionC(768, 160)
        self.Mixed_6e = InceptionC(768, 192)
        self.Mixed_7a = InceptionD(768)
        self.Mixed_7b = InceptionE(1280, AVG_NOPAD)
        self.Mixed_7c = InceptionE(2048, MAX)
        self.fc = nn.Linear(2048, 1008)      # part of the weight file; the features are taken in front of it
        self.eval()

    @torch.no_grad()
    def forward(self, x, resize_input=True, normalize_input=True):
        """x [N, 3, H, W] float in [0, 1] on the GPU -> [N, 2048] pool features."""
        x = F._req(x, "x")
        N, C, H, W = x.shape
        if resize_input or normalize_input:
            OH, OW = (299, 299) if resize_input else (H, W)
            y = torch.empty((N, C, OH, OW), device=x.device, dtype=torch.float32)
            mul, add = (2.0, -1.0) if normalize_input else (1.0, 0.0)
            check(lib.gz_resize_bilinear(_p(x), _p(y), N * C, H, W, OH, OW, mul, add, _stream()), "resize_bilinear")
            x = y
        x = self.Conv2d_2b_3x3(self.Conv2d_2a_3x3(self.Conv2d_1a_3x3(x)))
        x = _pool(x, 3, 2, 0, MAX)
        x = self.Conv2d_4a_3x3(self.Conv2d_3b_1x1(x))
        x = _pool(x, 3, 2, 0, MAX)
        for name in ("Mixed_5b", "Mixed_5c", "Mixed_5d", "Mixed_6a", "Mixed_6b", "Mixed_6c", "Mixed_6d", "Mixed_6e",
                     "Mixed_7a", "Mixed_7b", "Mixed_7c"):
            x = getattr(self, name)(x)
        return _pool(x, x.shape[2], 1, 0, AVG).reshape(N, -1)


FID_WEIGHTS_SHA256_PREFIX = "6726825d"       # pt_inception-2015-12-05-<sha256[:8]>.pth (reference inception.py:13)


def fid_weights_state(path, check_hash=True):
    """The state dict of the reference's weight file.  ``check_hash``: the file must be the published one -- torch.hub
    names its files ``<name>-<first 8 hex digits of the sha256>.pth`` and ``load_state_dict_from_url(check_hash=...)``
    verifies exactly that prefix; a different file gives FID numbers that compare with nobody's."""
    if check_hash:
        import hashlib
        h = hashlib.sha256()
        with open(path, "rb") as f:
            for block in iter(lambda: f.read(1 << 20), b""):
                h.update(block)
        if not h.hexdigest().startswith(FID_WEIGHTS_SHA256_PREFIX):
            raise RuntimeError("lightning_gan_zoo_amd: %s is not pytorch-fid's pt_inception-2015-12-05 weight file "
                               "(sha256 %s..., expected %s...); pass check_hash=False to load it anyway"
                               % (path, h.hexdigest()[:8], FID_WEIGHTS_SHA256_PREFIX))
    state = torch.load(path, map_location="cpu")
    return state.get("state_dict", state) if isinstance(state, dict) else state


def load_fid_weights(path, device="cuda", check_hash=True):
    """The reference's weight file (pytorch-fid's ``pt_inception-2015-12-05-6726825d.pth``: a torchvision-keyed
    state_dict) -> a ready feature extractor."""
    net = FIDInceptionV3()
    net.load_state_dict(fid_weights_state(path, check_hash))
    return net.to(device)


class InceptionFeatures:
    """``feature_fn`` of eval.evaluate: uint8 images [n, H, W, 3] (what the callback writes to PNG and reads back
    through ToTensor: x / 255) -> float64 activations [n, 2048], in batches of 16 like
    ``compute_activations_of_path(batch_size=16)`` (callback_inception_metrics.py:211-221)."""

    def __init__(self, net, batch_size=16):
        self.net, self.batch_size = net, batch_size
        self.device = next(net.parameters()).device

    def __call__(self, images_u8):
        out = []
        for i in range(0, len(images_u8), self.batch_size):
            u8 = torch.from_numpy(np.ascontiguousarray(images_u8[i:i + self.batch_size])).to(self.device)
            x = F.normalize_u8_images(u8, 0.0, 1.0)            # ToTensor: [n, 3, H, W] in [0, 1]
            out.append(self.net(x).double().cpu().numpy())
        return np.concatenate(out, axis=0)
