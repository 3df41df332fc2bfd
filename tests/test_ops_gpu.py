"""Per-operator parity of the HIP kernels (through the C ABI / autograd wrappers) against the
plain PyTorch fp32 CPU implementation of the same operator.  Tolerance: 1e-3 relative to the
largest reference magnitude (the north-star bar); observed errors are ~1e-6."""
import numpy as np
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu

TOL = 1e-3


def _F():
    from lightning_gan_zoo_amd import functional as F
    return F


def rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + int(np.prod(shape)) % 1000)
    return torch.randn(*shape, generator=g) * scale


GEOMS = {
    "k4s2p1": (4, 2, 1),
    "k5s2p2": (5, 2, 2),
    "k3s1p1": (3, 1, 1),
    "k1s1p0": (1, 1, 0),
}

# (N, C, H, K) -- chosen to hit every tile configuration, ragged M/N tails, K tails
CONV_CASES = [
    (2, 3, 16, 5),        # tiny everything (128x32 tile, masked)
    (4, 8, 16, 16),       # small
    (3, 20, 8, 40),       # ragged channels, N<=64 tiles
    (8, 64, 16, 128),     # 128-wide N
    (16, 32, 32, 96),     # larger M, ragged N tile (96)
    (64, 16, 32, 256),    # >= 256 tiles of 128x128
]


@pytest.mark.parametrize("geom", list(GEOMS))
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_family(geom, case):
    F = _F()
    k, s, p = GEOMS[geom]
    N, C, H, K = case
    if geom != "k4s2p1" and N * H > 300:
        pytest.skip("large cases only for the headline geometry")
    g = F.Geom(k, k, s, p)
    x = rnd(N, C, H, H, seed=1)
    w = rnd(K, C, k, k, seed=2, scale=0.1)
    y_ref = TF.conv2d(x, w, None, s, p)
    gy = rnd(*y_ref.shape, seed=3)
    dx_ref = TF.conv_transpose2d(gy, w, None, s, p, output_padding=H - ((y_ref.shape[2] - 1) * s - 2 * p + k))
    dw_ref = torch.nn.grad.conv2d_weight(x, w.shape, gy, stride=s, padding=p)

    xd, wd, gyd = x.cuda(), w.cuda(), gy.cuda()
    y = F._conv_fwd_raw(xd, wd, None, g, F.ACT_NONE, 0.0)
    assert rel(y, y_ref) < TOL
    dx = F._conv_dgrad_raw(gyd, wd, None, g, (H, H), F.ACT_NONE, 0.0)
    assert rel(dx, dx_ref) < TOL
    dw = F._conv_wgrad_raw(xd, gyd, g)
    assert rel(dw, dw_ref) < TOL


def test_conv_bias_act_epilogues():
    F = _F()
    g = F.K4S2P1
    x, w, b = rnd(4, 8, 16, 16, seed=5), rnd(12, 8, 4, 4, seed=6, scale=0.2), rnd(12, seed=7)
    y = F._conv_fwd_raw(x.cuda(), w.cuda(), b.cuda(), g, F.ACT_LRELU, 0.2)
    assert rel(y, TF.leaky_relu(TF.conv2d(x, w, b, 2, 1), 0.2)) < TOL
    wt = rnd(8, 12, 4, 4, seed=8, scale=0.2)   # ConvTranspose2d weight [Cin, Cout, k, k]
    yt = F.conv_transpose2d(x.cuda(), wt.cuda(), b.cuda(), g, F.ACT_TANH, 0.0)
    assert rel(yt, torch.tanh(TF.conv_transpose2d(x, wt, b, 2, 1))) < TOL


@pytest.mark.parametrize("shape", [(5, 7, 9), (128, 100, 256), (100, 512, 300), (512, 16384, 100), (33, 1, 64)])
def test_gemm_all_transposes(shape):
    F = _F()
    M, N, K = shape
    a, b = rnd(M, K, seed=11), rnd(K, N, seed=12)
    ref = a @ b
    assert rel(F.gemm(a.cuda(), b.cuda()), ref) < TOL
    assert rel(F.gemm(a.t().contiguous().cuda(), b.cuda(), trans_a=True), ref) < TOL
    assert rel(F.gemm(a.cuda(), b.t().contiguous().cuda(), trans_b=True), ref) < TOL
    assert rel(F.gemm(a.t().contiguous().cuda(), b.t().contiguous().cuda(), trans_a=True, trans_b=True), ref) < TOL


def test_autograd_conv_first_and_second_order():
    """F/Dg/Wg closure: double backward through conv -> LeakyReLU -> conv vs torch autograd on CPU."""
    F = _F()
    x = rnd(3, 3, 16, 16, seed=21)
    w1, w2 = rnd(8, 3, 4, 4, seed=22, scale=0.3), rnd(6, 8, 4, 4, seed=23, scale=0.3)

    def run(dev, hip):
        xx = x.to(dev).requires_grad_()
        a, b = w1.to(dev).requires_grad_(), w2.to(dev).requires_grad_()
        if hip:
            h = F.conv2d(xx, a, None, F.K4S2P1, F.ACT_LRELU, 0.2)
            out = F.conv2d(h, b, None, F.K4S2P1)
        else:
            h = TF.leaky_relu(TF.conv2d(xx, a, None, 2, 1), 0.2)
            out = TF.conv2d(h, b, None, 2, 1)
        (gx,) = torch.autograd.grad(out.sum(), xx, create_graph=True)
        pen = ((gx.reshape(3, -1).pow(2).sum(1) + 1e-12).sqrt() - 1).pow(2).mean() + out.pow(2).mean()
        ga, gb = torch.autograd.grad(pen, (a, b))
        return gx.detach(), ga, gb

    r = run("cpu", False)
    h = run("cuda", True)
    for got, ref in zip(h, r):
        assert rel(got, ref) < TOL


@pytest.mark.parametrize("shape", [(4, 8, 4, 4), (16, 32, 16, 16), (6, 5, 8, 8), (2, 3, 64, 64)])
@pytest.mark.parametrize("act", ["relu", "lrelu"])
def test_batchnorm_act_fwd_bwd_and_buffers(shape, act):
    F = _F()
    N, C, H, W = shape
    x = rnd(*shape, seed=31) * 2 + 0.5
    gamma, beta = rnd(C, seed=32) * 0.1 + 1, rnd(C, seed=33) * 0.1
    go = rnd(*shape, seed=34)
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    xr = x.clone().requires_grad_()
    fn = torch.relu if act == "relu" else (lambda t: TF.leaky_relu(t, 0.2))
    ref = fn(bn(xr))
    ref.backward(go)

    xd = x.cuda().requires_grad_()
    gd, bd = gamma.cuda().requires_grad_(), beta.cuda().requires_grad_()
    rm, rv, nbt = torch.zeros(C).cuda(), torch.ones(C).cuda(), torch.zeros((), dtype=torch.int64).cuda()
    out = F.batch_norm_act(xd, gd, bd, rm, rv, nbt, True, 0.1, 1e-5,
                           F.ACT_RELU if act == "relu" else F.ACT_LRELU, 0.2)
    out.backward(go.cuda())
    assert rel(out, ref) < TOL
    assert rel(xd.grad, xr.grad) < TOL
    assert rel(gd.grad, bn.weight.grad) < TOL
    assert rel(bd.grad, bn.bias.grad) < TOL
    assert rel(rm, bn.running_mean) < TOL and rel(rv, bn.running_var) < TOL
    assert int(nbt.item()) == int(bn.num_batches_tracked.item()) == 1      # bit-exact counter
    # eval mode uses the running statistics
    bn.eval()
    out_e = F.batch_norm_act(xd.detach(), gd.detach(), bd.detach(), rm, rv, nbt, False, 0.1, 1e-5, F.ACT_NONE, 0.0)
    assert rel(out_e, bn(x)) < TOL


@pytest.mark.parametrize("shape", [(4, 8, 4, 4), (8, 16, 16, 16), (3, 5, 8, 8)])
def test_instancenorm_lrelu_first_and_second_order(shape):
    """The InstanceNorm leg of the gradient-penalty double backward against torch autograd."""
    F = _F()
    N, C, H, W = shape
    x = rnd(*shape, seed=41) * 1.5 + 0.3
    gamma, beta = rnd(C, seed=42) * 0.2 + 1, rnd(C, seed=43) * 0.1
    wv = rnd(*shape, seed=44)

    def run(dev, hip):
        xx = x.to(dev).requires_grad_()
        g, b = gamma.to(dev).requires_grad_(), beta.to(dev).requires_grad_()
        if hip:
            out = F.instance_norm_act(xx, g, b, 1e-5, F.ACT_LRELU, 0.2)
        else:
            out = TF.leaky_relu(TF.instance_norm(xx, weight=g, bias=b, eps=1e-5), 0.2)
        (gx,) = torch.autograd.grad((out * wv.to(dev)).sum(), xx, create_graph=True)
        pen = gx.pow(2).sum() + (out * out).sum() * 0.1
        g2 = torch.autograd.grad(pen, (xx, g, b))
        return (out.detach(), gx.detach()) + g2

    r = run("cpu", False)
    h = run("cuda", True)
    for i, (got, ref) in enumerate(zip(h, r)):
        assert rel(got, ref) < TOL, i


def test_full_dot_conv_closure():
    F = _F()
    x, w = rnd(6, 32, 4, 4, seed=51), rnd(1, 32, 4, 4, seed=52, scale=0.2)

    def run(dev, hip):
        xx, ww = x.to(dev).requires_grad_(), w.to(dev).requires_grad_()
        out = F.full_dot_conv(xx, ww) if hip else TF.conv2d(xx, ww, None, 2, 0)
        (gx,) = torch.autograd.grad(out.sum(), xx, create_graph=True)
        pen = (gx.reshape(6, -1).pow(2).sum(1).sqrt() - 1).pow(2).mean() + out.pow(2).sum()
        return (out.detach(),) + torch.autograd.grad(pen, (xx, ww))

    for got, ref in zip(run("cuda", True), run("cpu", False)):
        assert rel(got, ref) < TOL


def test_gp_tail_ops():
    F = _F()
    a, b, al = rnd(5, 3, 8, 8, seed=61), rnd(5, 3, 8, 8, seed=62), torch.rand(5)
    ad, bd = a.cuda().requires_grad_(), b.cuda().requires_grad_()
    out = F.lerp_rows(ad, bd, al.cuda())
    ref = a * al.view(5, 1, 1, 1) + b * (1 - al.view(5, 1, 1, 1))
    assert rel(out, ref) < TOL
    ss = F.row_sumsq(out.reshape(5, -1))
    assert rel(ss, ref.reshape(5, -1).pow(2).sum(1)) < TOL
    ss.sum().backward()
    assert rel(ad.grad, 2 * ref * al.view(5, 1, 1, 1)) < TOL
    assert rel(bd.grad, 2 * ref * (1 - al.view(5, 1, 1, 1))) < TOL
    p = torch.nn.Parameter(rnd(1000, seed=63).cuda())
    F.clamp_(p, -0.5, 0.5)
    assert torch.equal(p.detach().cpu(), rnd(1000, seed=63).clamp(-0.5, 0.5))   # clamp is exact
