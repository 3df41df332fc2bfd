"""Per-operator parity of the HIP kernels (through the C ABI / autograd wrappers) against the
plain PyTorch fp32 CPU implementation of the same operator.  Tolerance: 1e-3 relative to the
largest reference magnitude (the north-star bar); observed errors are ~1e-6."""
import os

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
    # round 5 (tools/kernel_reach.sh: no test reached the generic weight-gradient loaders -- every case above has feature
    # rows of 4, 8, 16 or 32 pixels, which the row-aligned loaders take): feature rows that neither divide nor are
    # multiples of 16
    (3, 6, 12, 10),       # rows of 6 (s2) / 12 (s1) pixels
    (2, 5, 20, 7),        # rows of 10 / 20
    (2, 4, 40, 72),       # rows of 20 / 40, 72 output channels
]


@pytest.mark.parametrize("geom", list(GEOMS))
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_family(geom, case):
    F = _F()
    k, s, p = GEOMS[geom]
    N, C, H, K = case
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


DG5_CASES = [
    # (N, K feature channels, C image-side channels, OH = feature rows): M = N * OH * OH pixels per output phase
    (40, 32, 64, 16),      # shortest reduction (K = 32: 12 chunks, the mode switch after 8), 40 x 2 tiles
    (37, 208, 64, 8),      # M = 2368: the last tile is ragged; K = 208 (ty rows of 208: no chunk crosses a ty boundary)
    (3, 64, 128, 32),      # four column tiles, rows of 32 pixels
    (3, 32, 96, 64),       # three column tiles (C = 96), rows of 64 pixels (four image rows per tile)
    (64, 128, 64, 16),     # HoloGAN 64x64, D.block1's input gradient at bs 64 (split)
    (16, 256, 128, 8),     # ... D.block2's shape, 8 x 8 features: a tile spans four samples (split)
    (16, 512, 256, 4),     # ... D.block3's, 4 x 4 features: sixteen samples per tile, long reduction, many slabs
    (8, 128, 64, 32),      # EXT-128 D.block1's shape
    (1, 1104, 64, 16),     # ONE pixel tile, 69 channel blocks
    (64, 256, 128, 16),    # EXT-128 D.block2 at bs 64: default dispatch, unsplit (256 tiles)
    (64, 512, 256, 8),     # EXT-128 D.block3 at bs 64: default dispatch, split
]


@pytest.mark.parametrize("case", DG5_CASES)
def test_transposed_conv_5x5_row_shared_four_phases(case):
    """Round 6: the 5x5 s2 p2 input gradient on row-shared LDS rows with all four output phases per workgroup (csrc/
    gz_igemm2.h ConvDg5A2 / DgQuadB2, EpiPhaseQuadB; HoloGAN's critic, reference core/models/hologan_discriminator.py:7-23)
    against torch's conv_transpose2d: every row / column phase, image-edge zero columns (shift -1 at b = 0, +1 at b = AW
    - 1), top / bottom rows, ragged last tile, one to eight column tiles, both chunk modes, unsplit and split launches
    (the finish pass's single-block store path); bit-for-bit repeatable; the packed image is the tap-major one the gather
    loader reads, so an activation (which this kernel does not take) must give the gather loader's result on the same
    pack.  By default only launches with >= 80 chunk-tiles per CU take this kernel (it loses to the gather loader below
    that): the small cases run in test_..._small_shapes' child process (GZ_DG5_MIN_UNITS=0)."""
    import ctypes
    F = _F()
    from lightning_gan_zoo_amd._lib import lib
    N, K, C, OH = case
    H = 2 * OH
    text = ctypes.create_string_buffer(256)
    lib.gz_conv2d_plan(1, N, C, H, H, K, OH, OH, 5, 5, 2, 2, text, 256)
    if b"ConvDg5A2" not in text.value:
        assert not os.environ.get("GZ_RUN_DG5_SMALL_CASES"), text.value
        pytest.skip("little work per CU: the gather loader keeps it; covered by the GZ_DG5_MIN_UNITS=0 child process")
    g = F.Geom(5, 5, 2, 2)
    gy = rnd(N, K, OH, OH, seed=61)
    w = rnd(K, C, 5, 5, seed=62, scale=0.05)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = TF.conv_transpose2d(gy, w, None, 2, 2, output_padding=1)
    gyd, wd = gy.cuda(), w.cuda()
    out = F._conv_dgrad_raw(gyd, wd, None, g, (H, H), F.ACT_NONE, 0.0)
    assert out.shape == ref.shape
    err = rel(out, ref)
    # per phase, so that a wrong tap in one of them is not averaged away
    worst = max(rel(out[:, :, py::2, px::2], ref[:, :, py::2, px::2]) for py in (0, 1) for px in (0, 1))
    edge = max(rel(out[:, :, :, :2], ref[:, :, :, :2]), rel(out[:, :, :, -2:], ref[:, :, :, -2:]),
               rel(out[:, :, :2], ref[:, :, :2]), rel(out[:, :, -2:], ref[:, :, -2:]))
    print(case, text.value.decode()[:90], "err %.1e worst phase %.1e edges %.1e" % (err, worst, edge))
    assert err < TOL and worst < TOL and edge < TOL
    again = F._conv_dgrad_raw(gyd, wd, None, g, (H, H), F.ACT_NONE, 0.0)
    assert torch.equal(out, again)
    # with an activation the launch takes the gather loader (same packed weights): act(x) of the above
    relu = F._conv_dgrad_raw(gyd, wd, None, g, (H, H), F.ACT_RELU, 0.0)
    assert rel(relu, torch.relu(ref)) < TOL


def test_transposed_conv_5x5_row_shared_small_shapes():
    """The same cases with GZ_DG5_MIN_UNITS=0 (an experiment switch, read once per process: hence a child process): every
    case takes the row-shared kernel, whatever its size."""
    import subprocess
    import sys
    env = dict(os.environ, GZ_EXPERIMENTS="1", GZ_DG5_MIN_UNITS="0", GZ_RUN_DG5_SMALL_CASES="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-k",
                        "test_transposed_conv_5x5_row_shared_four_phases"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout and "skipped" not in r.stdout, r.stdout[-3000:]


@pytest.mark.parametrize("case", [(3, 1, 8, 6), (2, 2, 16, 9), (4, 3, 64, 16), (2, 4, 32, 7), (5, 3, 128, 4),
                                  (2, 3, 24, 5), (1, 3, 8, 128)])
def test_small_channel_transposed_conv(case):
    """ConvTranspose2d k4 s2 p1 onto <= 4 image channels (G's output layer, the critic's input gradient): the
    4-positions-per-lane direct kernel (feature rows of 4 .. 64 columns; 24 columns -> 6 lanes per row do not divide
    a wavefront, which takes the one-position kernel) with bias + tanh, against torch."""
    F = _F()
    N, C, H, K = case                      # image side [N, C, H, H], feature side [N, K, H/2, H/2]
    gy = rnd(N, K, H // 2, H // 2, seed=7)
    w = rnd(K, C, 4, 4, seed=8, scale=0.2)
    b = rnd(C, seed=9)
    ref = torch.tanh(TF.conv_transpose2d(gy, w, b, 2, 1))
    out = F._conv_dgrad_raw(gy.cuda(), w.cuda(), b.cuda(), F.K4S2P1, (H, H), F.ACT_TANH, 0.0)
    assert out.shape == ref.shape and rel(out, ref) < TOL
    assert rel(F._conv_dgrad_raw(gy.cuda(), w.cuda(), None, F.K4S2P1, (H, H), F.ACT_NONE, 0.0),
               TF.conv_transpose2d(gy, w, None, 2, 1)) < TOL


@pytest.mark.parametrize("case", [(4, 3, 64, 16), (128, 3, 64, 64), (256, 3, 64, 64), (1100, 3, 64, 16), (3, 4, 32, 40)])
@pytest.mark.parametrize("act", ["relu", "lrelu"])
def test_input_gradient_through_a_fused_activation(case, act):
    """d/dx of ``act(conv(x, w) + b)`` for the critics' first layer with frozen weights (the generator step): the direct
    few-channel kernel masks the incoming gradient with the saved output on load (gz_conv2d_dgrad_act) -- every
    wavefront split of the channel loop (1 / 4 / 8), against act_bwd followed by the plain transposed convolution."""
    F = _F()
    N, C, H, K = case
    A, slope = (F.ACT_RELU, 0.0) if act == "relu" else (F.ACT_LRELU, 0.2)
    x = rnd(N, C, H, H, seed=47).cuda().requires_grad_()
    w, b = rnd(K, C, 4, 4, seed=48, scale=0.2).cuda(), rnd(K, seed=49).cuda()
    gy = rnd(N, K, H // 2, H // 2, seed=50).cuda()
    from lightning_gan_zoo_amd._lib import lib
    assert lib.gz_conv2d_dgrad_act_fuses(N, C, H, H, K, H // 2, H // 2, 4, 4, 2, 1, A)
    y = F.conv2d(x, w, b, F.K4S2P1, A, slope)
    y.backward(gy)
    g_pre = F._act_bwd_raw(gy, y.detach(), A, slope)
    ref = F._conv_dgrad_raw(g_pre, w, None, F.K4S2P1, (H, H), F.ACT_NONE, 0.0)
    assert x.grad.shape == ref.shape and rel(x.grad, ref) < 1e-5
    t = TF.conv_transpose2d(g_pre.cpu(), w.cpu(), None, 2, 1)
    assert rel(x.grad, t) < TOL


@pytest.mark.parametrize("case", [(512, 3, 64, 7), (128, 3, 128, 2), (512, 2, 64, 65), (520, 4, 64, 16), (512, 1, 64, 193),
                                  (128, 3, 64, 128), (64, 3, 64, 40), (256, 3, 64, 64), (130, 4, 64, 33), (1100, 3, 64, 16)])
def test_small_channel_transposed_conv_full_chip(case):
    """The same layer at sizes that give every SIMD >= 2 wavefronts of lane positions (the unsplit channel loop: 256
    lane positions per workgroup), odd channel counts and a batch that is not a multiple of the workgroup; round 5:
    the benchmarked sizes, where eight (bs <= 128) or four (bs <= 512) wavefronts share the channel loop of 64 lane
    positions and meet in an LDS tree -- channel counts that are not multiples of the wavefront count included."""
    F = _F()
    N, C, H, K = case
    gy = rnd(N, K, H // 2, H // 2, seed=17)
    w = rnd(K, C, 4, 4, seed=18, scale=0.2)
    b = rnd(C, seed=19)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = torch.tanh(TF.conv_transpose2d(gy, w, b, 2, 1))
    out = F._conv_dgrad_raw(gy.cuda(), w.cuda(), b.cuda(), F.K4S2P1, (H, H), F.ACT_TANH, 0.0)
    assert out.shape == ref.shape and rel(out, ref) < TOL


@pytest.mark.parametrize("case", [(16, 3, 64, 64), (20, 4, 64, 128), (16, 1, 64, 16), (33, 2, 64, 32), (128, 3, 64, 128),
                                  (16, 3, 128, 64), (128, 3, 64, 64), (24, 4, 64, 32), (8, 3, 64, 64)])
def test_few_channel_k4s2p1_weight_gradient(case):
    """Weight gradient of the k4 s2 p1 layers with <= 4 image channels (D.conv_in, and G's last layer as its adjoint):
    the direct 16x16x4-MFMA kernel (k = 4 pixels, one slab per workgroup), reduced by the library and -- the training
    path -- left as slabs for gz_reduce_multi, against torch."""
    import ctypes
    from lightning_gan_zoo_amd._lib import lib
    F = _F()
    N, C, H, K = case
    x = rnd(N, C, H, H, seed=37)
    gy = rnd(N, K, H // 2, H // 2, seed=38)
    text = ctypes.create_string_buffer(256)
    lib.gz_conv2d_plan(2, N, C, H, H, K, H // 2, H // 2, 4, 4, 2, 1, text, 256)
    # (K = 128, G's last layer as the adjoint, stays on the tile path: no faster in the step; fewer than 1024 row segments
    # -- 8 samples -- too)
    direct = K <= 64 and N * (H // 2) * (H // 32) >= 1024
    assert text.value.decode().startswith("Wg direct wgrad_k4s2p1_fewc") == direct, text.value
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = torch.nn.grad.conv2d_weight(x, (K, C, 4, 4), gy, stride=2, padding=1)
    dw = F._conv_wgrad_raw(x.cuda(), gy.cuda(), F.K4S2P1)
    assert dw.shape == ref.shape and rel(dw, ref) < TOL
    # the sink route: slabs stay in the workspace, one gz_reduce_multi launch writes the gradient
    w = torch.nn.Parameter(torch.zeros(K, C, 4, 4, device="cuda"))
    prev = F.set_grad_sinks(True)
    try:
        xin = x.cuda()
        out = F.conv2d(xin, w, None, F.K4S2P1)
        out.backward(gy.cuda())
        F.flush_grad_sinks()
    finally:
        F.set_grad_sinks(*prev)
    assert w.grad is not None and rel(w.grad, ref) < TOL
    # ... and the layer as the critics use it, LeakyReLU(conv(x) + bias) on an input that needs no gradient: activation
    # backward, weight gradient and bias gradient in one launch (gz_conv2d_wgrad_act_partial), twice into the same sinks
    # (a critic applied to a real and a fake batch)
    # (the reference takes the LeakyReLU decisions of the HIP forward: a pre-activation within rounding of zero may fall on
    # either side in two implementations, and one flipped element moves a weight gradient by more than 1e-3)
    w0, b0 = rnd(K, C, 4, 4, seed=39, scale=0.2), rnd(K, seed=40)
    x2 = rnd(N, C, H, H, seed=41)
    w, b = torch.nn.Parameter(w0.cuda()), torch.nn.Parameter(b0.cuda())
    assert bool(lib.gz_conv2d_wgrad_act_fuses(N, C, H, H, K, H // 2, H // 2, 4, 4, 2, 1, F.ACT_LRELU)) == direct
    dw_ref, db_ref = torch.zeros_like(w0), torch.zeros_like(b0)
    prev = F.set_grad_sinks(True)
    try:
        for xh in (x, x2):
            out = F.conv2d(xh.cuda(), w, b, F.K4S2P1, F.ACT_LRELU, 0.2)
            assert rel(out, TF.leaky_relu(TF.conv2d(xh, w0, b0, 2, 1), 0.2)) < TOL
            out.backward(gy.cuda())
            g_pre = gy * torch.where(out.detach().cpu() > 0, 1.0, 0.2)
            dw_ref += torch.nn.grad.conv2d_weight(xh, (K, C, 4, 4), g_pre, stride=2, padding=1)
            db_ref += g_pre.sum((0, 2, 3))
        assert not direct or (w.grad is None and b.grad is None)   # nothing went through autograd's accumulation
        F.flush_grad_sinks()
    finally:
        F.set_grad_sinks(*prev)
    assert rel(w.grad, dw_ref) < TOL and rel(b.grad, db_ref) < TOL


@pytest.mark.parametrize("case", [(2, 3, 16, 5), (4, 8, 16, 16), (3, 20, 8, 40), (8, 64, 16, 128), (16, 32, 32, 96),
                                  (64, 16, 32, 256), (512, 64, 32, 128), (37, 24, 16, 72)])
def test_conv_with_fused_batchnorm_statistics(case):
    """conv / transposed conv whose epilogue also emits the BatchNorm partial sums (gz_conv2d_*_stats): the output is
    bit-identical to the plain launch, the partial sums add up to the per-channel sum / sum of squares of the output
    (every tile shape, ragged pixel and channel tails), and batch_norm_act fed with them gives the same result,
    gradients and running buffers as when it reads the feature map itself."""
    F = _F()
    N, C, H, K = case
    g = F.K4S2P1
    x, w = rnd(N, C, H, H, seed=301).cuda(), rnd(K, C, 4, 4, seed=302, scale=0.1).cuda()
    for name in ("conv", "conv_transpose"):
        if name == "conv":
            y_plain = F.conv2d(x, w, None, g)
            y, stats = F.conv2d_with_stats(x, w, g)
            ch = K
        else:
            xt = rnd(N, K, H // 2, H // 2, seed=303).cuda()
            y_plain = F.conv_transpose2d(xt, w, None, g)          # w [Cin = K, Cout = C, 4, 4]
            y, stats = F.conv_transpose2d_with_stats(xt, w, g)
            ch = C
        assert torch.equal(y, y_plain)
        if stats.numel() == 0:
            continue            # a split-K launch (few output tiles): not fused, by design
        assert stats.shape[1:] == (ch, 2)
        tot = stats.double().sum(0).cpu()
        yd = y.double().cpu()
        assert rel(tot[:, 0], yd.sum((0, 2, 3))) < 1e-5 and rel(tot[:, 1], (yd * yd).sum((0, 2, 3))) < 1e-5
        gamma, beta = (rnd(ch, seed=304) * 0.1 + 1).cuda(), (rnd(ch, seed=305) * 0.1).cuda()
        go = rnd(*y.shape, seed=306).cuda()
        res = []
        for st in (stats, None):
            ya = y.detach().clone().requires_grad_()
            ga, ba = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
            rm, rv, nbt = torch.zeros(ch).cuda(), torch.ones(ch).cuda(), torch.zeros((), dtype=torch.int64).cuda()
            out = F.batch_norm_act(ya, ga, ba, rm, rv, nbt, True, 0.1, 1e-5, F.ACT_LRELU, 0.2, st)
            out.backward(go)
            res.append((out.detach(), ya.grad, ga.grad, ba.grad, rm, rv, nbt))
        for a, b in zip(res[0][:6], res[1][:6]):
            assert rel(a, b) < 1e-5
        assert int(res[0][6]) == int(res[1][6]) == 1


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


# ---------------------------------------------------------------------------
# HoloGAN operators
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("case", [(2, 8, 4, 4), (3, 20, 4, 12), (4, 64, 8, 16), (2, 128, 4, 130),
                                  (8, 512, 4, 128),       # few output tiles -> split-K forward and dgrad
                                  (8, 128, 8, 64),        # HoloGAN block2 at bs 8: phases split with 8-chunk slabs
                                  (64, 512, 4, 128),      # HoloGAN block1 at the benched bs 64: 27 slabs over 8 phases
                                  (3, 64, 4, 64),         # igemm2 gathers with fewer rows than one 256-row tile
                                  (2, 192, 4, 192),       # ... 192 columns: three 64-wide tiles, ragged 16-channel blocks
                                  (5, 72, 8, 64)])        # ... 72 input channels: the last channel block is half empty
def test_conv3d_family(case):
    """ConvTranspose3d(k3,s2,p1,op1) forward (Dg), its input gradient (F) and weight gradient (Wg)."""
    F = _F()
    N, Cin, D, Cout = case
    x = rnd(N, Cin, D, D, D, seed=71)
    w = rnd(Cin, Cout, 3, 3, 3, seed=72, scale=0.1)     # ConvTranspose3d weight [Cin, Cout, 3,3,3]
    b = rnd(Cout, seed=73)
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    ref = TF.conv_transpose3d(xr, wr, b, stride=2, padding=1, output_padding=1)
    go = rnd(*ref.shape, seed=74)
    ref.backward(go)
    xd, wd, bd = x.cuda().requires_grad_(), w.cuda().requires_grad_(), b.cuda().requires_grad_()
    out = F.conv_transpose3d(xd, wd, bd)
    out.backward(go.cuda())
    assert rel(out, ref) < TOL
    assert rel(xd.grad, xr.grad) < TOL
    assert rel(wd.grad, wr.grad) < TOL
    assert rel(bd.grad, go.sum((0, 2, 3, 4))) < TOL


@pytest.mark.parametrize("shape", [(4, 8, 4, 4, 4), (3, 6, 8, 8), (2, 16, 16, 16, 16)])
def test_adain_relu_fwd_bwd(shape):
    F = _F()
    from oracle.hologan_cpu import adain
    N, C = shape[:2]
    x = rnd(*shape, seed=81) * 1.3 + 0.2
    s, b = rnd(N, C, seed=82).abs() + 0.5, rnd(N, C, seed=83) * 0.3
    go = rnd(*shape, seed=84)
    xr, sr, br = x.clone().requires_grad_(), s.clone().requires_grad_(), b.clone().requires_grad_()
    ref = torch.relu(adain(xr, sr, br))
    ref.backward(go)
    xd, sd, bd = x.cuda().requires_grad_(), s.cuda().requires_grad_(), b.cuda().requires_grad_()
    out = F.adain_act(xd, sd, bd, 1e-8, F.ACT_RELU)
    out.backward(go.cuda())
    for got, want in ((out, ref), (xd.grad, xr.grad), (sd.grad, sr.grad), (bd.grad, br.grad)):
        assert rel(got, want) < TOL
    # scale | shift packed in one [N, 2C] tensor (the ZMapping output as it is)
    xp = x.cuda().requires_grad_()
    sbp = torch.cat([s, b], 1).cuda().requires_grad_()
    outp = F.adain_act_packed(xp, sbp, 1e-8, F.ACT_RELU)
    outp.backward(go.cuda())
    assert torch.equal(outp, out) and rel(xp.grad, xr.grad) < TOL
    assert rel(sbp.grad[:, :C], sr.grad) < TOL and rel(sbp.grad[:, C:], br.grad) < TOL


def test_adain_of_a_shared_constant_equals_adain_of_its_repeat():
    """HoloGAN's first layer (reference hologan_generator.py:141-142): AdaIN(self.x.repeat(N, ...)) without the repeat;
    the constant's gradient is the sum over the samples."""
    F = _F()
    N, C = 6, 24
    x = (rnd(1, C, 4, 4, 4, seed=111) + 0.3).cuda()
    sb = (rnd(N, 2 * C, seed=112).abs() + 0.2).cuda()
    go = rnd(N, C, 4, 4, 4, seed=113).cuda()
    xa, sa = x.clone().requires_grad_(), sb.clone().requires_grad_()
    ya = F.adain_act_packed(xa.repeat(N, 1, 1, 1, 1), sa, 1e-8, F.ACT_RELU)
    ya.backward(go)
    xb, sbb = x.clone().requires_grad_(), sb.clone().requires_grad_()
    yb = F.adain_const_act(xb, sbb, 1e-8, F.ACT_RELU)
    yb.backward(go)
    assert torch.equal(ya, yb)
    assert rel(xb.grad, xa.grad) < 1e-5 and rel(sbb.grad, sa.grad) < 1e-5


@pytest.mark.parametrize("inner_shape", [(8, 8, 8), (32, 32)])
def test_row_norms_keep_their_digits_when_the_mean_dwarfs_the_spread(inner_shape):
    """AdaIN / InstanceNorm rows with |mean| = 300 sigma (the output of a convolution over an all-positive, nearly
    constant AdaIN+ReLU map looks like this): the statistics are two-pass, so the result stays within 1e-3 of fp64
    -- a one-pass E[x^2] - mean^2 in fp32 is off by 1e-7 * 300^2 = 1 % of the variance here."""
    F = _F()
    N, C = 4, 8
    x = (rnd(N, C, *inner_shape, seed=91) * 0.1 + 30.0)
    s, b = rnd(N, C, seed=92).abs() + 0.5, rnd(N, C, seed=93) * 0.3
    go = rnd(N, C, *inner_shape, seed=94)
    x64 = x.double().requires_grad_()
    flat = x64.reshape(N, C, -1)
    xh = (flat - flat.mean(2, keepdim=True)) * torch.rsqrt(flat.var(2, keepdim=True) + 1e-8)
    ref = torch.relu(s.double()[..., None] * xh + b.double()[..., None]).reshape(x.shape)
    ref.backward(go.double())
    xd = x.cuda().requires_grad_()
    out = F.adain_act(xd, s.cuda(), b.cuda(), 1e-8, F.ACT_RELU)
    out.backward(go.cuda())
    assert rel(out, ref.float()) < TOL and rel(xd.grad, x64.grad.float()) < TOL
    if len(inner_shape) == 2:
        x64 = x.double().requires_grad_()
        ref = torch.nn.functional.leaky_relu(torch.nn.functional.instance_norm(x64), 0.2)
        ref.backward(go.double())
        xd = x.cuda().requires_grad_()
        out = F.instance_norm_act(xd, None, None, 1e-5, F.ACT_LRELU, 0.2)
        out.backward(go.cuda())
        assert rel(out, ref.float()) < TOL and rel(xd.grad, x64.grad.float()) < TOL


def test_rigid_resample_matches_oracle_and_indices_are_bit_exact():
    """int64 voxel indices (reference idx_a..idx_h) bit-exact; resampled features and their adjoint
    within 1e-3 (observed 1e-6)."""
    F = _F()
    import numpy as np
    from oracle import hologan_cpu as H
    from lightning_gan_zoo_amd.core.models.hologan_generator import view_inverse_matrices
    rng = np.random.RandomState(5)
    n, c = 6, 5
    view = np.zeros((n, 6))
    view[:, 0] = rng.randint(220, 320, n) * np.pi / 180.0
    view[:, 1] = rng.randint(70, 110, n) * np.pi / 180.0
    view[:, 2] = 1.0
    view[0] = (0.0, 0.0, 1.0, 0.0, 0.0, 0.0)         # identity view: every coordinate lands on an integer
    view[1, 3:] = (0.7, -1.2, 0.4)
    view[4, 2] = 2.0         # zoom in: each source voxel is hit by ~64 output voxels (the adjoint's hit lists overflow
    view[5, 2] = 0.6         # and the launch falls back to the direct gather); zoom out: fewer hits than usual
    vox = rnd(n, c, 16, 16, 16, seed=91)
    minv = view_inverse_matrices(view)
    assert torch.equal(minv, H.view_matrices(view))                       # host matrices: bit-identical
    x, y, z = H.resample_coords(minv)
    idx_ref, _ = H.trilinear_indices(vox.shape, x, y, z)
    voxr = vox.clone().requires_grad_()
    ref = H.rigid_resample(voxr, view).permute(0, 1, 3, 2, 4).flip(2).reshape(n, -1, 16, 16)
    go = rnd(*ref.shape, seed=92)
    ref.backward(go)

    md = minv.reshape(n, 16).cuda()
    out, idx = F.rigid_resample_indices(vox.cuda(), md)
    mism = sum(int((idx[k].cpu() != idx_ref[k]).sum()) for k in range(8))
    assert mism == 0, "%d of %d voxel indices differ from the reference's" % (mism, 8 * idx_ref[0].numel())
    assert rel(out, ref) < TOL
    vd = vox.cuda().requires_grad_()
    F.rigid_resample(vd, md).backward(go.cuda())
    assert rel(vd.grad, voxr.grad) < TOL


def test_spectral_normalize_matches_torch():
    F = _F()
    torch.manual_seed(3)
    conv = torch.nn.utils.spectral_norm(torch.nn.Conv2d(8, 16, 5, 2, 2))
    u0, v0 = conv.weight_u.clone(), conv.weight_v.clone()
    x = rnd(2, 8, 16, 16, seed=95)
    conv.train()
    ref = conv(x)
    ref.sum().backward()
    w_orig = conv.weight_orig.detach().clone().cuda().requires_grad_()
    u, v = u0.cuda(), v0.cuda()
    w = F.spectral_normalize(w_orig, u, v, True)
    out = F.conv2d(x.cuda(), w, conv.bias.detach().cuda(), F.Geom(5, 5, 2, 2))
    out.sum().backward()
    assert rel(out, ref) < TOL
    assert rel(u, conv.weight_u) < TOL and rel(v, conv.weight_v) < TOL
    assert rel(w_orig.grad, conv.weight_orig.grad) < TOL


@pytest.mark.parametrize("n", [1, 7, 64, 513])
def test_fused_loss_heads(n):
    """BCE-with-logits against a constant target (mean) and the mean squared error of HoloGAN's q_loss: value and
    gradient against torch, with a non-trivial upstream factor (the steps combine them as (a + b) / 2 + q)."""
    F = _F()
    x = rnd(n, seed=201) * 3
    for target in (0.0, 1.0):
        xr = x.clone().requires_grad_()
        ref = TF.binary_cross_entropy_with_logits(xr, torch.full_like(xr, target))
        (ref * 0.37).backward()
        xd = x.cuda().requires_grad_()
        out = F.bce_logits_mean(xd.reshape(n, 1), target)
        (out * 0.37).backward()
        assert out.dim() == 0 and rel(out, ref) < 1e-5 and rel(xd.grad, xr.grad) < 1e-5
    a, b = rnd(n, 12, seed=202), rnd(n, 12, seed=203)
    ar = a.clone().requires_grad_()
    ref = torch.mean((ar - b) ** 2)
    (ref * 1.7).backward()
    ad = a.cuda().requires_grad_()
    out = F.mse_mean(ad, b.cuda())
    (out * 1.7).backward()
    assert rel(out, ref) < 1e-5 and rel(ad.grad, ar.grad) < 1e-5


def test_spectral_normalize_eval_mode_and_two_uses():
    """Eval mode leaves u / v alone; two forward uses before one backward (the D step runs D(real), D(fake)) keep the
    vectors each use saw."""
    F = _F()
    torch.manual_seed(5)
    conv = torch.nn.utils.spectral_norm(torch.nn.Conv2d(8, 12, 5, 2, 2))
    x1, x2 = rnd(2, 8, 16, 16, seed=211), rnd(2, 8, 16, 16, seed=212)
    u0, v0 = conv.weight_u.clone(), conv.weight_v.clone()
    conv.train()
    ref = conv(x1).sum() + 2 * conv(x2).pow(2).sum()
    ref.backward()
    w_orig = conv.weight_orig.detach().clone().cuda().requires_grad_()
    u, v = u0.cuda(), v0.cuda()
    g = F.Geom(5, 5, 2, 2)
    b = conv.bias.detach().cuda()
    out = F.conv2d(x1.cuda(), F.spectral_normalize(w_orig, u, v, True), b, g).sum() + \
        2 * F.conv2d(x2.cuda(), F.spectral_normalize(w_orig, u, v, True), b, g).pow(2).sum()
    out.backward()
    assert rel(out, ref) < TOL and rel(u, conv.weight_u) < TOL and rel(v, conv.weight_v) < TOL
    assert rel(w_orig.grad, conv.weight_orig.grad) < TOL
    conv.eval()
    ue, ve = u.clone(), v.clone()
    w_eval = F.spectral_normalize(w_orig.detach(), u, v, False)
    assert torch.equal(u, ue) and torch.equal(v, ve)
    assert rel(F.conv2d(x1.cuda(), w_eval, b, g), conv(x1)) < TOL


def test_linear_act_fwd_bwd():
    F = _F()
    x, w, b = rnd(6, 40, seed=96), rnd(24, 40, seed=97, scale=0.2), rnd(24, seed=98)
    for act, fn in ((F.ACT_NONE, lambda t: t), (F.ACT_RELU, torch.relu), (F.ACT_TANH, torch.tanh),
                    (F.ACT_LRELU, lambda t: TF.leaky_relu(t, 0.2))):
        xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
        ref = fn(TF.linear(xr, wr, br))
        ref.pow(2).sum().backward()
        xd, wd, bd = x.cuda().requires_grad_(), w.cuda().requires_grad_(), b.cuda().requires_grad_()
        out = F.linear_act(xd, wd, bd, act, 0.2)
        out.pow(2).sum().backward()
        for got, want in ((out, ref), (xd.grad, xr.grad), (wd.grad, wr.grad), (bd.grad, br.grad)):
            assert rel(got, want) < TOL
    # the logit head's shapes: one output column, and a batch longer than the 16 row groups of colsum_kernel
    x, w, b = rnd(70, 64, seed=99), rnd(1, 64, seed=100, scale=0.2), rnd(1, seed=101)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    TF.linear(xr, wr, br).pow(2).sum().backward()
    xd, wd, bd = x.cuda().requires_grad_(), w.cuda().requires_grad_(), b.cuda().requires_grad_()
    F.linear_act(xd, wd, bd).pow(2).sum().backward()
    assert rel(bd.grad, br.grad) < TOL and rel(wd.grad, wr.grad) < TOL


@pytest.mark.parametrize("shape", [(64, 128, (1024, 256, 128, 512, 128)), (5, 7, (3, 70)), (130, 33, (65,))])
def test_linear_act_multi_matches_separate_layers(shape):
    """HoloGAN's five ZMapping layers in one launch (gz_linear_multi_fwd / _bwd): values, weight, bias and input
    gradients against torch's nn.functional.linear per layer; ragged N, K, J exercise the tile edges."""
    F = _F()
    N, K, Js = shape
    x = rnd(N, K, seed=300)
    ws = [rnd(J, K, seed=301 + i, scale=0.2) for i, J in enumerate(Js)]
    bs = [rnd(J, seed=331 + i) for i, J in enumerate(Js)]
    for act, fn in ((F.ACT_RELU, torch.relu), (F.ACT_NONE, lambda t: t), (F.ACT_TANH, torch.tanh),
                    (F.ACT_LRELU, lambda t: TF.leaky_relu(t, 0.2))):
        xr = x.clone().requires_grad_()
        wr, br = [w.clone().requires_grad_() for w in ws], [b.clone().requires_grad_() for b in bs]
        refs = [fn(TF.linear(xr, w, b)) for w, b in zip(wr, br)]
        sum((i + 1.0) * r.pow(2).sum() for i, r in enumerate(refs)).backward()
        xd = x.cuda().requires_grad_()
        wd, bd = [w.cuda().requires_grad_() for w in ws], [b.cuda().requires_grad_() for b in bs]
        outs = F.linear_act_multi(xd, list(zip(wd, bd)), act, 0.2)
        sum((i + 1.0) * o.pow(2).sum() for i, o in enumerate(outs)).backward()
        assert rel(xd.grad, xr.grad) < TOL
        for o, r, w1, w0, b1, b0 in zip(outs, refs, wd, wr, bd, br):
            assert o.shape == r.shape and rel(o, r) < TOL
            assert rel(w1.grad, w0.grad) < TOL and rel(b1.grad, b0.grad) < TOL
    # an output nobody uses gets a zero gradient; a layer without bias
    wd = [w.cuda().requires_grad_() for w in ws]
    outs = F.linear_act_multi(x.cuda(), [(w, None) for w in wd], F.ACT_RELU)
    outs[0].sum().backward()
    assert rel(outs[0], torch.relu(TF.linear(x, ws[0]))) < TOL
    assert all(float(w.grad.abs().max()) == 0.0 for w in wd[1:]) and float(wd[0].grad.abs().max()) > 0


def test_spectral_normalize_multi_matches_per_layer_path():
    """The five-launch spectral normalisation of all blocks of a HoloGAN discriminator call (power iterations as jobs of
    four launches, w and both packed images in one table launch) against the per-layer path and against
    torch.nn.utils.spectral_norm's arithmetic on the CPU: u / v buffers, sigma-scaled weights, the convolution run
    from the attached images (forward, input gradient) and the weight_orig gradients."""
    F = _F()
    geom = F.Geom(5, 5, 2, 2)
    shapes = [(128, 64), (256, 128), (512, 256), (24, 12)]            # the reference's three blocks + a ragged one
    g = torch.Generator().manual_seed(77)
    Ws = [torch.randn(k, c, 5, 5, generator=g) * 0.05 for k, c in shapes]
    us = [torch.nn.functional.normalize(torch.randn(k, generator=g), dim=0) for k, _ in shapes]
    vs = [torch.nn.functional.normalize(torch.randn(c * 25, generator=g), dim=0) for _, c in shapes]
    xs = [torch.randn(2, c, 16, 16, generator=g) for _, c in shapes]
    for rounds in (1, 2):            # twice: the buffers carry over
        res = {}
        for mode in ("multi", "single", "cpu"):
            dev = "cpu" if mode == "cpu" else "cuda"
            W = [w.clone().to(dev).requires_grad_() for w in Ws]
            u, v = [t.clone().to(dev) for t in us], [t.clone().to(dev) for t in vs]
            x = [t.clone().to(dev).requires_grad_() for t in xs]
            for _ in range(rounds):
                if mode == "multi":
                    ws = F.spectral_normalize_multi(list(zip(W, u, v)), True, geom)
                    assert all(hasattr(w, "_gz_packs") for w in ws)
                elif mode == "single":
                    ws = [F.spectral_normalize(a, b, c, True) for a, b, c in zip(W, u, v)]
                else:
                    ws = []
                    for a, b, c in zip(W, u, v):
                        m = a.detach().reshape(a.shape[0], -1)
                        c.copy_(torch.nn.functional.normalize(m.t() @ b, dim=0, eps=1e-12))
                        b.copy_(torch.nn.functional.normalize(m @ c, dim=0, eps=1e-12))
                        ws.append(a / torch.dot(b, a.reshape(a.shape[0], -1) @ c))
            ys = [(F.conv2d(xi, w, None, geom) if mode != "cpu" else TF.conv2d(xi, w, None, 2, 2)) for xi, w in zip(x, ws)]
            sum(y.pow(2).sum() for y in ys).backward()
            res[mode] = [t.detach().cpu() for t in ws + u + v + ys + [a.grad for a in W] + [xi.grad for xi in x]]
        for k, (a, b, c) in enumerate(zip(res["multi"], res["single"], res["cpu"])):
            assert rel(a, b) < 2e-6, (rounds, k, rel(a, b))
            assert rel(a, c) < TOL, (rounds, k, rel(a, c))


@pytest.mark.parametrize("groups", [1, 2])
def test_sn_conv_in_act_matches_spectral_norm_conv_instance_norm(groups):
    """functional.sn_conv_in_act: act(IN_{eps sigma^2}(conv(x, weight_orig))) on weight_orig's own packed images against
    the reference's composition conv2d(x, weight_orig / sigma) + bias -> InstanceNorm2d -> LeakyReLU in torch on the CPU
    (sigma = u^T W v after the power iteration, u / v constants): values, input gradient, weight_orig gradient including
    the sigma term, u / v buffers; groups = 2: two consecutive power iterations, each half of the batch with its own
    sigma -- what two discriminator calls see."""
    F = _F()
    geom = F.Geom(5, 5, 2, 2)
    g = torch.Generator().manual_seed(5)
    for (K, C, N, H) in ((32, 16, 4 * groups, 16), (128, 64, 2 * groups, 8)):
        W = torch.randn(K, C, 5, 5, generator=g) * 0.05
        b = torch.randn(K, generator=g) * 0.1
        u0 = torch.nn.functional.normalize(torch.randn(K, generator=g), dim=0)
        v0 = torch.nn.functional.normalize(torch.randn(C * 25, generator=g), dim=0)
        x = torch.randn(N, C, H, H, generator=g)
        go = torch.randn(N, K, H // 2, H // 2, generator=g)
        # torch, CPU
        Wr, xr = W.clone().requires_grad_(), x.clone().requires_grad_()
        u, v = u0.clone(), v0.clone()
        outs = []
        for h in range(groups):
            with torch.no_grad():
                m = Wr.detach().reshape(K, -1)
                v = torch.nn.functional.normalize(m.t() @ u, dim=0, eps=1e-12)
                u = torch.nn.functional.normalize(m @ v, dim=0, eps=1e-12)
            sigma = torch.dot(u, Wr.reshape(K, -1) @ v)
            xs = xr[h * (N // groups):(h + 1) * (N // groups)]
            y = TF.conv2d(xs, Wr / sigma, b, 2, 2)
            outs.append(TF.leaky_relu(TF.instance_norm(y, eps=1e-5), 0.2))
        ref = torch.cat(outs)
        ref.backward(go)
        # product
        Wd = torch.nn.Parameter(W.cuda())
        bd = torch.nn.Parameter(b.cuda())
        xd = x.cuda().requires_grad_()
        ud, vd = u0.cuda(), v0.cuda()
        (sig, us, vs), = F.spectral_power_iterations([(Wd, ud, vd)], calls=groups)
        out = F.sn_conv_in_act(xd, Wd, bd, sig, us, vs, geom, 1e-5, F.ACT_LRELU, 0.2)
        out.backward(go.cuda())
        assert rel(ud, u) < 1e-5 and rel(vd, v) < 1e-5
        assert rel(out, ref) < TOL, rel(out, ref)
        assert rel(xd.grad, xr.grad) < TOL, rel(xd.grad, xr.grad)
        assert rel(Wd.grad, Wr.grad) < TOL, rel(Wd.grad, Wr.grad)
        assert float(bd.grad.abs().max()) == 0.0          # exactly zero (the reference's is rounding noise)


@pytest.mark.parametrize("width", [(8, 16), (64, 128)])      # (in_planes, z): small, and the reference's default width
def test_hologan_ext128_matches_oracle_extension(width):
    """EXT-128: the reference cannot run at 128x128; product and oracle implement the
    same stride-2 extension, so this checks HIP-vs-CPU consistency only (no reference parity)."""
    feat, zdim = width
    import numpy as np
    from helpers import fill_closed_form
    from lightning_gan_zoo_amd.config import make_cfg
    from lightning_gan_zoo_amd.core.models import hologan_discriminator as PD, hologan_generator as PG
    from oracle import hologan_cpu as H
    va = make_cfg("hologan", features=feat, batch_size=2, noise_dim=zdim).generator.view_args
    torch.manual_seed(0)
    gp, go = PG.Generator(feat, 3, zdim, va, 128, ext128=True), H.Generator(feat, 3, zdim, va, 128, ext128=True)
    torch.manual_seed(0)
    dp, do = PD.Discriminator(3, feat, zdim, img_size=128), H.Discriminator(3, feat, zdim, img_size=128)
    for a, b in ((gp, go), (dp, do)):
        fill_closed_form(a, 3)
        fill_closed_form(b, 3)
        b.load_state_dict(a.state_dict())
    gp.cuda(), dp.cuda()
    z = rnd(2, zdim, seed=99)
    np.random.seed(1)
    view = go.sample_view(2)
    img_p, img_o = gp(z.cuda(), view), go(z, view)
    assert img_p.shape == (2, 3, 128, 128) and rel(img_p, img_o) < TOL
    lp, zp = dp(img_p)
    lo, zo = do(img_o)
    assert rel(lp, lo) < TOL and rel(zp, zo) < TOL


@pytest.mark.parametrize("kind", ["adam", "adam_b0", "rmsprop"])
def test_fused_optimizers_match_torch(kind):
    """Fused multi-tensor Adam / RMSprop vs torch.optim on the same GPU tensors over 5 steps; state_dicts
    interchange."""
    from lightning_gan_zoo_amd import optim as O
    torch.manual_seed(0)
    shapes = [(64, 3, 4, 4), (128,), (100, 1024, 4, 4), (7,), (1, 512, 4, 4)]
    pa = [torch.nn.Parameter(torch.randn(s, device="cuda") * 0.05) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    if kind == "rmsprop":
        oa, ob = O.RMSprop(pa, lr=5e-5), torch.optim.RMSprop(pb, lr=5e-5)
    else:
        betas = (0.0, 0.9) if kind == "adam_b0" else (0.5, 0.999)
        oa, ob = O.Adam(pa, lr=2e-4, betas=betas), torch.optim.Adam(pb, lr=2e-4, betas=betas)
    for step in range(5):
        for a, b in zip(pa, pb):
            g = torch.randn_like(a) * (0.1 + step)
            a.grad, b.grad = g.clone(), g.clone()
        oa.step()
        ob.step()
    for a, b in zip(pa, pb):
        assert rel(a, b) < 1e-6
    # a fresh fused optimizer continues from TORCH's state_dict (same keys / shapes) and keeps matching torch
    pc = [torch.nn.Parameter(b.detach().clone()) for b in pb]
    oc = type(oa)(pc, **{k: v for k, v in ob.param_groups[0].items() if k in ("lr", "betas", "alpha", "eps")})
    import copy
    oc.load_state_dict(copy.deepcopy(ob.state_dict()))    # (load_state_dict would alias torch's CPU `step` tensors)
    assert set(oa.state_dict()["state"][0]) == set(ob.state_dict()["state"][0])
    for c, b in zip(pc, pb):
        g = torch.randn_like(b)
        c.grad, b.grad = g.clone(), g.clone()
    oc.step()
    ob.step()
    for c, b in zip(pc, pb):
        assert rel(c, b) < 1e-6


@pytest.mark.parametrize("kind", ["adam", "rmsprop"])
def test_fused_optimizers_step_parameter_subsets_and_zero_the_gradients(kind):
    """Round 5 (ddp.GradSync lands its gradient buckets one at a time): ``step(params=subset, zero_grads=True,
    grad_scale=s)`` updates exactly the subset -- bit-identical to what a whole-optimizer step does to those parameters
    -- leaves the others and their gradients alone, and overwrites the subset's gradients with zeros; two subset steps
    equal one full step.  Unaligned tensors (a view that starts 4 bytes into a buffer) take the scalar path."""
    from lightning_gan_zoo_amd import optim as O
    torch.manual_seed(1)
    shapes = [(64, 3, 4, 4), (128,), (33, 17), (7,), (1, 512, 4, 4)]

    def make():
        torch.manual_seed(2)
        ps = [torch.nn.Parameter(torch.randn(s, device="cuda") * 0.05) for s in shapes]
        odd = torch.randn(1001, device="cuda")
        ps.append(torch.nn.Parameter(odd[1:]))                  # data_ptr % 16 == 4
        return ps

    pa, pb = make(), make()
    mk = (lambda ps: O.RMSprop(ps, lr=5e-5)) if kind == "rmsprop" else (lambda ps: O.Adam(ps, lr=2e-4, betas=(0.5, 0.999)))
    oa, ob = mk(pa), mk(pb)
    for step in range(3):
        gs = [torch.randn_like(p) * (0.1 + step) for p in pa]
        for a, b, g in zip(pa, pb, gs):
            a.grad, b.grad = g.clone(), g.clone()
        oa.step(grad_scale=0.5)
        first, second = pb[:2] + pb[5:], pb[2:5]
        before = [p.detach().clone() for p in second]
        ob.step(grad_scale=0.5, params=first, zero_grads=True)
        assert all(torch.equal(p.detach(), q) for p, q in zip(second, before))          # untouched
        assert all(float(p.grad.abs().max()) == 0.0 for p in first)                     # zeroed behind the read
        assert all(torch.equal(p.grad, g) for p, g in zip(second, gs[2:5]))             # still there
        ob.step(grad_scale=0.5, params=second, zero_grads=True)
        assert all(float(p.grad.abs().max()) == 0.0 for p in pb)
        for a, b in zip(pa, pb):
            assert torch.equal(a.detach(), b.detach())
    sa, sb = oa.state_dict()["state"], ob.state_dict()["state"]
    assert all(float(sa[k]["step"]) == float(sb[k]["step"]) == 3.0 for k in sa)


def test_gp_penalty_tail_matches_torch():
    """functional.gp_penalty(sumsq) == mean((sqrt(sumsq) - 1)^2) of the reference (core/utils/utils.py:55-57:
    ``gradient.norm(2, dim=1)``, ``torch.mean((gradient_norm - 1) ** 2)``) and its gradient, including the zero
    subgradient of torch.norm at an exactly-zero row."""
    F = _F()
    g = rnd(37, 500, seed=5) * 0.05
    g[3] = 0.0                                   # a sample whose input gradient vanishes
    g[7] *= 40.0
    gr = g.clone().requires_grad_()
    ref = torch.mean((gr.norm(2, dim=1) - 1) ** 2)
    ref.backward()
    gd = g.cuda().requires_grad_()
    out = F.gp_penalty(F.row_sumsq(gd))
    out.backward()
    assert rel(out, ref) < 1e-6 and rel(gd.grad, gr.grad) < 1e-5
    assert float(gd.grad[3].abs().max()) == 0.0 and torch.isfinite(gd.grad).all()
    ones = F.ones_like_const(out.reshape(1, 1))
    assert ones is F.ones_like_const(out.reshape(1, 1)) and float(ones) == 1.0


def test_exact_zero_bias_gradients_join_the_sink_flush():
    """A bias in front of a per-plane normalisation (``bias_cancels``): with sinks on no zero tensor is created -- the
    flush writes zeros into a fresh ``p.grad`` (a job without sources) or leaves an existing gradient alone."""
    F = _F()
    x = rnd(2, 8, 4, 4, seed=1).cuda().requires_grad_()
    w = torch.nn.Parameter(rnd(8, 12, 4, 4, seed=2).cuda() * 0.1)
    b = torch.nn.Parameter(rnd(12, seed=3).cuda())
    prev = F.set_grad_sinks(True)
    try:
        y = F.conv_transpose2d(x, w, b, F.K4S2P1, bias_cancels=True)
        y.sum().backward()
        assert b.grad is None and id(b) in F._sinks.pending       # nothing launched for it yet
        F.flush_grad_sinks()
        assert b.grad is not None and float(b.grad.abs().max()) == 0.0 and w.grad is not None
        b.grad.fill_(2.0)                                           # an existing gradient is left alone
        y = F.conv_transpose2d(x, w, b, F.K4S2P1, bias_cancels=True)
        y.sum().backward()
        F.flush_grad_sinks()
        assert float((b.grad - 2.0).abs().max()) == 0.0
    finally:
        F.set_grad_sinks(*prev)


def test_host_draws_reach_the_device_through_the_pinned_ring():
    """harness.HostStager: the per-step host draws (latent noise, GP alpha, HoloGAN's view matrices; reference
    core/lightning_module.py:107-108 ``sample(...).to(device)``) are read by the device straight out of pinned host
    buffers (gz_copy_words) -- same values as ``.to(device)``, for more calls than the ring has slots, and for sizes
    that are no multiple of four bytes (plain asynchronous copy)."""
    from lightning_gan_zoo_amd.harness import HostStager
    st = HostStager(depth=3)
    g = torch.Generator().manual_seed(5)
    outs, refs = [], []
    for k in range(10):
        t = torch.randn(128, 100, generator=g)
        outs.append(st.to_device(t, "cuda"))
        refs.append(t.clone())
    a = torch.rand(37, 1, 1, 1, generator=g)
    b = torch.arange(5, dtype=torch.uint8)
    c = torch.arange(7, dtype=torch.int64)
    da, db, dc = st.to_device(a, "cuda"), st.to_device(b, "cuda"), st.to_device(c, "cuda")
    torch.cuda.synchronize()
    assert all(o.is_cuda and torch.equal(o.cpu(), r) for o, r in zip(outs, refs))
    assert torch.equal(da.cpu(), a) and torch.equal(db.cpu(), b) and torch.equal(dc.cpu(), c)
    assert da.shape == a.shape and dc.dtype == torch.int64


# ---------------------------------------------------------------------------
# R1 / ResNet path
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(2, 3, 8, 8), (3, 5, 16, 12), (4, 16, 128, 128), (1, 7, 4, 4), (2, 2, 6, 10)])
def test_avgpool_and_upsample_pairs(shape):
    """AvgPool2d(3,2,1) / nearest 2x upsample: forward, adjoint, and adjoint-of-adjoint through autograd."""
    F = _F()
    x = rnd(*shape, seed=11)
    for op, ref_op in ((F.avg_pool3s2, lambda t: TF.avg_pool2d(t, 3, 2, 1)),
                       (F.upsample2, lambda t: TF.interpolate(t, scale_factor=2))):
        xr = x.clone().requires_grad_()
        yr = ref_op(xr)
        gy = rnd(*yr.shape, seed=12)
        (gxr,) = torch.autograd.grad(yr, xr, gy, create_graph=True)
        xd = x.cuda().requires_grad_()
        gyd = gy.cuda().requires_grad_()
        yd = op(xd)
        assert yd.shape == yr.shape and rel(yd, yr) < TOL
        (gxd,) = torch.autograd.grad(yd, xd, gyd, create_graph=True)
        assert rel(gxd, gxr) < TOL
        # the adjoint is linear in gy: differentiating it w.r.t. gy must give the forward map back
        v = rnd(*x.shape, seed=13)
        (ggy,) = torch.autograd.grad(gxd, gyd, v.cuda())
        assert rel(ggy, ref_op(v)) < TOL


def test_activation_and_residual_tail():
    F = _F()
    a, b = rnd(3, 8, 16, 16, seed=21), rnd(3, 8, 16, 16, seed=22)
    ar, br = a.clone().requires_grad_(), b.clone().requires_grad_()
    ad, bd = a.cuda().requires_grad_(), b.cuda().requires_grad_()
    out_r = ar + 0.1 * br
    act_r = TF.leaky_relu(out_r, 0.2)
    out_d, act_d = F.add_scaled_act(ad, bd, 0.1)
    assert rel(out_d, out_r) < 1e-6 and rel(act_d, act_r) < 1e-6
    g1, g2 = rnd(*a.shape, seed=23), rnd(*a.shape, seed=24)
    (out_r * g1 + act_r * g2).sum().backward()
    (out_d * g1.cuda() + act_d * g2.cuda()).sum().backward()
    assert rel(ad.grad, ar.grad) < 1e-5 and rel(bd.grad, br.grad) < 1e-5
    y = F.add_scaled(ad, F.activation(bd), 0.1)
    assert rel(y, a + 0.1 * TF.leaky_relu(b, 0.2)) < 1e-6
    assert rel(F.scale(ad, -2.5), -2.5 * a) < 1e-6


def _r1_pair(size=32, nf=4, nf_max=16, bs=3, zdim=8, seed=3):
    from helpers import fill_closed_form
    from lightning_gan_zoo_amd.core.submodules.gan_stability.models import resnet as P
    from oracle import resnet_cpu as O
    torch.manual_seed(seed)
    nets = []
    for m in (P, O):
        torch.manual_seed(seed)
        nets.append((m.Generator(zdim, 1, size, nfilter=nf, nfilter_max=nf_max),
                     m.Discriminator(zdim, 1, size, nfilter=nf, nfilter_max=nf_max)))
    return nets


def test_resnet_default_init_and_state_dict_match_oracle():
    """Same seed -> same default initialisation (parameter creation order) and state_dict layout."""
    (gp, dp), (go, do) = _r1_pair()
    for a, b in ((gp, go), (dp, do)):
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa) == list(sb)
        for k in sa:
            assert torch.equal(sa[k], sb[k]), k


@pytest.mark.parametrize("cfg", [dict(size=32, nf=4, nf_max=16, bs=3), dict(size=64, nf=8, nf_max=32, bs=2),
                                 dict(size=16, nf=6, nf_max=6, bs=5)])
def test_resnet_r1_regulariser_matches_oracle(cfg):
    """D(real), the R1 regulariser compute_grad2(D(real), real) and the parameter gradients of
    reg * mean(R1) + sum(D) -- the double backward through conv3x3 / conv1x1 / LeakyReLU / AvgPool / fc /
    sigmoid -- and G's forward + first-order gradients, against the CPU oracle."""
    from helpers import fill_closed_form
    from lightning_gan_zoo_amd.core.utils.utils import compute_grad2 as cg_p
    from oracle.resnet_cpu import compute_grad2 as cg_o
    (gp, dp), (go, do) = _r1_pair(cfg["size"], cfg["nf"], cfg["nf_max"], cfg["bs"])
    for a, b in ((gp, go), (dp, do)):
        fill_closed_form(a, 4)
        with torch.no_grad():
            for p in a.parameters():
                if p.ndim >= 2:
                    p.mul_(1.0 / (p[0].numel() ** 0.5 * 0.02 * 2 ** 0.5))
        b.load_state_dict(a.state_dict())
    gp.cuda(), dp.cuda()
    x = rnd(cfg["bs"], 3, cfg["size"], cfg["size"], seed=31)
    z = rnd(cfg["bs"], 8, seed=32)

    def l2(a, b):
        return float((a.double().cpu() - b.double()).norm() / b.double().norm().clamp_min(1e-30))

    res = []
    for D, cg, xx in ((dp, cg_p, x.cuda().requires_grad_()), (do, cg_o, x.clone().requires_grad_())):
        out = D(xx)
        reg = cg(out, xx)
        (300.0 * reg.mean() + out.sum()).backward()
        res.append((out.detach(), reg.detach(), xx.grad, {n: p.grad for n, p in D.named_parameters()}))
    (op_, rp, gxp, pgp), (oo, ro, gxo, pgo) = res
    assert rel(op_, oo) < TOL and l2(rp, ro) < TOL and l2(gxp, gxo) < TOL
    worst = max((l2(pgp[n], pgo[n]), n) for n in pgo if float(pgo[n].norm()) > 0)
    assert worst[0] < TOL, worst

    res = []
    for G, D, zz in ((gp, dp, z.cuda().requires_grad_()), (go, do, z.clone().requires_grad_())):
        G.zero_grad()
        fake = G(zz)
        (D(fake).sum() + 0.01 * fake.pow(2).sum()).backward()
        res.append((fake.detach(), zz.grad, {n: p.grad for n, p in G.named_parameters()}))
    (fp, gzp, pgp), (fo, gzo, pgo) = res
    assert fp.shape == (cfg["bs"], 3, cfg["size"], cfg["size"]) and rel(fp, fo) < TOL and l2(gzp, gzo) < TOL
    worst = max((l2(pgp[n], pgo[n]), n) for n in pgo)
    assert worst[0] < TOL, worst


# ---------------------------------------------------------------------------
# split-K of F / Dg / GEMM (few output tiles, long reduction)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("geom,case", [
    ("k3s1p1", (64, 512, 4, 512)),     # R1 ResNet 512->512 @ 4x4, bs 64
    ("k3s1p1", (16, 96, 8, 200)),      # ragged channels
    ("k5s2p2", (64, 256, 8, 512)),     # HoloGAN discriminator block 3, bs 64
    ("k4s2p1", (32, 256, 8, 512)),     # DCGAN D.block3 at a small batch
    ("k1s1p0", (8, 512, 4, 512)),
])
def test_conv_split_k(geom, case):
    F = _F()
    from lightning_gan_zoo_amd._lib import lib
    k, s, p = GEOMS[geom]
    N, C, H, K = case
    g = F.Geom(k, k, s, p)
    OH = (H + 2 * p - k) // s + 1
    assert lib.gz_conv2d_fwd_workspace_bytes(N, C, H, H, K, OH, OH, k, k, s, p) > 0, "case no longer splits"
    assert lib.gz_conv2d_dgrad_workspace_bytes(N, C, H, H, K, OH, OH, k, k, s, p) > 0
    x = rnd(N, C, H, H, seed=41)
    w = rnd(K, C, k, k, seed=42, scale=0.05)
    b = rnd(K, seed=43)
    y_ref = TF.leaky_relu(TF.conv2d(x, w, b, s, p), 0.2)
    gy = rnd(*y_ref.shape, seed=44)
    bc = rnd(C, seed=45)
    dx_ref = torch.tanh(TF.conv_transpose2d(gy, w, bc, s, p, output_padding=H - ((OH - 1) * s - 2 * p + k)))
    y = F._conv_fwd_raw(x.cuda(), w.cuda(), b.cuda(), g, F.ACT_LRELU, 0.2)
    assert rel(y, y_ref) < TOL
    dx = F._conv_dgrad_raw(gy.cuda(), w.cuda(), bc.cuda(), g, (H, H), F.ACT_TANH, 0.0)
    assert rel(dx, dx_ref) < TOL
    # same launch twice: the slab sum order is fixed, so the result is bit-reproducible
    assert torch.equal(y, F._conv_fwd_raw(x.cuda(), w.cuda(), b.cuda(), g, F.ACT_LRELU, 0.2))


@pytest.mark.parametrize("shape", [(64, 8192, 128), (64, 8192, 1), (3, 4100, 70), (200, 2048, 130)])
@pytest.mark.parametrize("tb", [False, True])
def test_gemm_split_k(shape, tb):
    F = _F()
    from lightning_gan_zoo_amd._lib import lib
    M, K, N = shape
    assert lib.gz_gemm_workspace_bytes(M, N, K) > 0, "case no longer splits"
    a, b, bias = rnd(M, K, seed=51), rnd(K, N, seed=52, scale=0.05), rnd(N, seed=53)
    ref = torch.relu(a.double() @ b.double() + bias.double())
    bd = b.t().contiguous().cuda() if tb else b.cuda()
    c = F.gemm(a.cuda(), bd, bias.cuda(), trans_b=tb, act=F.ACT_RELU)
    assert rel(c, ref) < TOL
    c2 = F.gemm(a.t().contiguous().cuda(), bd, bias.cuda(), trans_a=True, trans_b=tb, act=F.ACT_RELU)
    assert rel(c2, ref) < TOL


@pytest.mark.parametrize("case", [(16, 16, 64, 16), (4, 3, 128, 16), (4, 16, 128, 3), (16, 32, 64, 32),
                                  (16, 16, 64, 32), (32, 20, 48, 24), (16, 64, 64, 3), (16, 5, 64, 50),
                                  (16, 7, 64, 4), (17, 10, 64, 1), (64, 64, 64, 2)])      # <= 4 output channels: FMA kernel
def test_wgrad_small_channel_3x3(case):
    """The 16x16x4-MFMA weight-gradient kernel of 3x3 s1 p1 layers with <= 32 channels (R1 ResNet high-resolution
    stages) and the plain-FMA kernel of layers with <= 4 output channels on 64-wide maps (HoloGAN's last layer):
    against torch (GZ_NO_SMALLCH_WG is read once per process, so the comparison is with the CPU reference only)."""
    F = _F()
    from lightning_gan_zoo_amd._lib import lib
    N, C, H, K = case
    x, gy = rnd(N, C, H, H, seed=61), rnd(N, K, H, H, seed=62)
    count = K * C * 9
    assert lib.gz_conv2d_wgrad_workspace_bytes(N, C, H, H, K, H, H, 3, 3) >= 4 * count
    dw_ref = torch.nn.grad.conv2d_weight(x.double(), (K, C, 3, 3), gy.double(), stride=1, padding=1)
    dw = F._conv_wgrad_raw(x.cuda(), gy.cuda(), F.K3S1P1)
    assert rel(dw, dw_ref) < TOL
    assert torch.equal(dw, F._conv_wgrad_raw(x.cuda(), gy.cuda(), F.K3S1P1))      # fixed summation order
    # the same launch also yields the bias gradient
    assert lib.gz_conv2d_wgrad_fuses_bias(N, C, H, H, K, H, H, 3, 3, 1, 1) == 1
    dw2, db = F._conv_wgrad_raw(x.cuda(), gy.cuda(), F.K3S1P1, with_bias=True)
    assert torch.equal(dw2, dw) and rel(db, gy.double().sum((0, 2, 3))) < TOL


@pytest.mark.parametrize("case", [(16, 16, 64, 16), (4, 3, 128, 16), (4, 16, 128, 3), (16, 32, 64, 32),
                                  (16, 16, 64, 32), (32, 20, 48, 24), (8, 5, 96, 30), (16, 64, 64, 3),
                                  (16, 9, 64, 50), (16, 9, 64, 4), (17, 33, 64, 1), (64, 64, 64, 2)])   # <= 4 outputs: FMA kernel
def test_conv3x3_small_channel_fwd_dgrad(case):
    """The direct 16x16x4-MFMA forward / input-gradient kernel of 3x3 s1 p1 layers with <= 32 channels, with the
    bias + activation epilogues, for both weight image formats (tap-major when the input side has >= 16
    channels) and ragged channel counts."""
    F = _F()
    N, C, H, K = case
    g = F.K3S1P1
    x, w, b = rnd(N, C, H, H, seed=81), rnd(K, C, 3, 3, seed=82, scale=0.2), rnd(K, seed=83)
    y_ref = TF.leaky_relu(TF.conv2d(x, w, b, 1, 1), 0.2)
    gy, bc = rnd(N, K, H, H, seed=84), rnd(C, seed=85)
    dx_ref = torch.tanh(TF.conv_transpose2d(gy, w, bc, 1, 1))
    y = F._conv_fwd_raw(x.cuda(), w.cuda(), b.cuda(), g, F.ACT_LRELU, 0.2)
    assert rel(y, y_ref) < TOL
    dx = F._conv_dgrad_raw(gy.cuda(), w.cuda(), bc.cuda(), g, (H, H), F.ACT_TANH, 0.0)
    assert rel(dx, dx_ref) < TOL
    assert rel(F._conv_dgrad_raw(gy.cuda(), w.cuda(), None, g, (H, H), F.ACT_NONE, 0.0),
               TF.conv_transpose2d(gy, w, None, 1, 1)) < TOL


@pytest.mark.parametrize("shape", [(8, 16, 32, 32), (3, 5, 4, 4), (64, 64, 16, 16, 16), (2, 7, 6, 10)])
def test_channel_sum(shape):
    F = _F()
    g = rnd(*shape, seed=91)
    ref = g.double().sum([d for d in range(g.dim()) if d != 1])
    assert rel(F._channel_sum_raw(g.cuda()), ref) < 1e-5


@pytest.mark.parametrize("act", ["relu", "lrelu"])
def test_batchnorm_eval_mode_backward(act):
    """Eval-mode BatchNorm + activation (running statistics as constants): dx, dgamma, dbeta against torch."""
    F = _F()
    N, C, H = 6, 12, 8
    x, g, b = rnd(N, C, H, H, seed=101), rnd(C, seed=102).abs() + 0.5, rnd(C, seed=103)
    rm, rv, go = rnd(C, seed=104), rnd(C, seed=105).abs() + 0.3, rnd(N, C, H, H, seed=106)
    xr, gr, br = (t.clone().requires_grad_() for t in (x, g, b))
    y = TF.batch_norm(xr, rm.clone(), rv.clone(), gr, br, False, 0.1, 1e-5)
    y = TF.relu(y) if act == "relu" else TF.leaky_relu(y, 0.2)
    y.backward(go)
    xd, gd, bd = (t.cuda().requires_grad_() for t in (x, g, b))
    nbt = torch.zeros((), dtype=torch.int64, device="cuda")
    yd = F.batch_norm_act(xd, gd, bd, rm.cuda(), rv.cuda(), nbt, False, 0.1, 1e-5,
                          F.ACT_RELU if act == "relu" else F.ACT_LRELU, 0.2)
    yd.backward(go.cuda())
    assert rel(yd, y) < TOL and rel(xd.grad, xr.grad) < TOL and rel(gd.grad, gr.grad) < TOL and rel(bd.grad, br.grad) < TOL
    assert int(nbt) == 0


def test_second_order_through_fused_tanh_and_bias():
    """create_graph=True through conv_transpose2d(+bias, tanh epilogue): the double backward of the activation
    (gz_tanh_bwd2) and the differentiable bias gradient (_ChannelSum) against torch's own double backward."""
    F = _F()
    x, w, b = rnd(4, 8, 8, 8, seed=201), rnd(8, 4, 4, 4, seed=202, scale=0.2), rnd(4, seed=203)
    outs = []
    for dev in ("cpu", "cuda"):
        xd, wd, bd = (t.detach().clone().to(dev).requires_grad_() for t in (x, w, b))
        if dev == "cpu":
            y = torch.tanh(TF.conv_transpose2d(xd, wd, bd, 2, 1))
        else:
            y = F.conv_transpose2d(xd, wd, bd, F.K4S2P1, F.ACT_TANH)
        gx, gb = torch.autograd.grad(y.pow(2).sum(), (xd, bd), create_graph=True)
        (gx.pow(2).sum() + gb.pow(2).sum()).backward()
        outs.append([t.detach().cpu() for t in (y, gx, gb, xd.grad, wd.grad, bd.grad)])
    for got, want in zip(outs[1], outs[0]):
        assert rel(got, want) < TOL


def test_group_repack_equals_single_packs():
    """gz_conv2d_pack_multi (one launch for every packed image of an optimizer's weights) writes exactly what the
    per-image pack launches write: all four layouts (forward / transposed, reduction in (c, tap) or tap-major order)."""
    F = _F()
    F.set_pack_cache(True)       # (a GraphedTrainer test that ran earlier in the process leaves the cache off)
    geoms = [F.K4S2P1, F.Geom(5, 5, 2, 2), F.Geom(3, 3, 1, 1), F.Geom(5, 5, 2, 2), F.K4S2P1, F.K4S2P1]
    # (the k4 s2 p1 transposed image has a group-launch body of its own, 64 channels per LDS pass: 70 -> two passes and a
    # padded tail, 3 -> one partial quad)
    shapes = [(24, 12, 4, 4), (40, 32, 5, 5), (16, 20, 3, 3), (6, 3, 5, 5), (130, 70, 4, 4), (64, 3, 4, 4)]
    ws = [torch.nn.Parameter(rnd(*s, seed=300 + i).cuda()) for i, s in enumerate(shapes)]
    F.register_pack_group(ws)
    first = [(F._packed(w, "f", g), F._packed(w, "d", g)) for w, g in zip(ws, geoms)]       # single launches
    with torch.no_grad():
        for i, w in enumerate(ws):
            w.mul_(1.5).add_(0.01 * i)                    # version bump: every image is stale
    again = [(F._packed(w, "f", g), F._packed(w, "d", g)) for w, g in zip(ws, geoms)]       # one group launch
    for (f0, d0), (f1, d1), w, g in zip(first, again, ws, geoms):
        assert f0.data_ptr() == f1.data_ptr() and d0.data_ptr() == d1.data_ptr()            # buffers kept
        for kind, got in (("f", f1), ("d", d1)):
            # the tap-major transposed layout has per-phase tails that no pack writes and no kernel reads: start the
            # reference from a copy, so that only what a pack writes is compared (a group launch that had written
            # nothing would leave the OLD weights' image in `got`)
            ref = got.clone()
            F._pack_one(w, ref, kind, g)
            assert torch.equal(got.view(torch.int32), ref.view(torch.int32)), (tuple(w.shape), kind)      # bitwise
    F.invalidate(ws[0])                                   # raw in-place rewrite: stale without a version bump
    with torch.no_grad():
        ws[0].data.zero_()
    assert float(F._packed(ws[0], "f", geoms[0]).abs().sum()) == 0.0
    F.clear_pack_cache()


# ---------------------------------------------------------------------------------------------------------------
# round 3: the one-wavefront-per-SIMD implicit GEMM (csrc/gz_igemm.h igemm2, 256x256 / 256x128 workgroup tiles)
# ---------------------------------------------------------------------------------------------------------------
IGEMM2_DG_CASES = [
    # N, K (feature channels), OH (feature side), C (output channels): label expected from gz_conv2d_tile
    (64, 64, 32, 128, "256x128"),        # G.block4-like: 32x32 phase grid, 1024 tiles of 256x128
    (64, 96, 16, 256, "256x128"),        # K = 96: 24 chunks, the ring wraps 8 times; 512 tiles
    (128, 64, 8, 512, "256x128"),        # 8x8 feature rows (32 rows of a tile), 512 tiles
    (128, 72, 8, 384, "256x256"),        # 10 chunks; 256 tiles of 256x256 (384 tiles of 256x128 would not fill two per CU)
    (512, 72, 4, 512, "256x128"),        # 4x4 feature maps: 64 image rows per tile; 18 chunks
    (52, 64, 16, 384, "256x128"),        # ragged pixel tail (13312 = 52 * 256) and 3 N tiles
    (67, 68, 16, 160, "256x128"),
    (64, 256, 8, 256, "256x64"),         # round 4: 128 tiles of 256x128 -> 256 tiles of 256x64, unsplit
    (32, 256, 8, 256, "256x64"),         # ... and with half the batch: 128 tiles of 256x64, the reduction split in two
    (64, 256, 8, 160, "256x128"),        # 160 channels (not a multiple of 64): 256x128, reduction split in two
    (8, 64, 64, 128, "256x128"),         # 64-pixel feature rows: four image rows per tile
    (64, 64, 32, 64, "512x64"),          # D.block1-size: 64 output channels, 512-pixel tiles (two 256-pixel pieces per row)
    (272, 68, 16, 48, "512x64"),         # 48 channels: the general epilogue; 17 chunks        # pixel tail inside a tile (67 * 256 = 17152 = 67 tiles), channel tail 160 = 128 + 32
]


@pytest.mark.parametrize("case", IGEMM2_DG_CASES)
def test_igemm2_transposed_conv_matches_torch(case):
    """ConvTranspose2d k4 s2 p1 (= the data gradient of the strided conv) on the igemm2 skeleton against torch's CPU
    operator: plain, with bias + activation epilogue, and with the BatchNorm partial sums (bit-identical output,
    sums = per-channel sums of the output)."""
    F = _F()
    from lightning_gan_zoo_amd._lib import lib
    N, K, OH, C, label = case
    H = 2 * OH
    tile = lib.gz_conv2d_tile(1, N, C, H, H, K, OH, OH, 4, 4, 2)
    assert F._TILES[tile] == label, (F._TILES[tile], label)
    gy = rnd(N, K, OH, OH, seed=11)
    w = rnd(K, C, 4, 4, seed=12, scale=0.1)
    b = rnd(C, seed=13)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = TF.conv_transpose2d(gy, w, None, 2, 1)
    gyd, wd = gy.cuda(), w.cuda()
    out = F._conv_dgrad_raw(gyd, wd, None, F.K4S2P1, (H, H), F.ACT_NONE, 0.0)
    assert out.shape == ref.shape
    err = rel(out, ref)
    assert err < TOL, err
    out_b = F._conv_dgrad_raw(gyd, wd, b.cuda(), F.K4S2P1, (H, H), F.ACT_LRELU, 0.2)
    assert rel(out_b, TF.leaky_relu(ref + b.view(1, -1, 1, 1), 0.2)) < TOL
    y, stats = F.conv_transpose2d_with_stats(gyd, wd, F.K4S2P1)
    assert torch.equal(y, out)
    if stats.numel():               # (a split reduction is not fused with the statistics, by design)
        st = stats.double().sum(0)
        o64 = out.double()
        assert rel(st[:, 0], o64.sum((0, 2, 3))) < 1e-5 and rel(st[:, 1], (o64 * o64).sum((0, 2, 3))) < 1e-5


IGEMM2_F_CASES = [
    # N, C (input channels), H (input side), K (output channels), expected label
    # round 4: launches whose 256x128 tiles would not give every CU a workgroup take 256x64 tiles (three per CU) ...
    (32, 64, 64, 128, "256x64"),         # OW = 32: 8 output rows per tile
    (128, 64, 32, 128, "256x64"),        # D.block1-like: OW = 16
    (256, 72, 16, 256, "256x64"),        # OW = 8, 72 chunks
    (512, 64, 8, 512, "256x64"),         # OW = 4 (64 output rows per tile)
    (50, 68, 32, 384, "256x64"),         # ragged pixel tail (12800 = 50 tiles), 6 column tiles
    (512, 256, 8, 512, "256x64"),        # 32 row tiles x 8 column tiles = 256 tiles, unsplit
    (128, 256, 8, 512, "256x64"),        # 64 tiles of 256x64: reduction split in four (slabs + finish kernel)
    # ... and the same geometries with >= 256 tiles of 256x128 keep those
    (64, 64, 64, 128, "256x128"),        # OW = 32, 256 tiles
    (512, 64, 32, 128, "256x128"),       # OW = 16
    (512, 72, 16, 256, "256x128"),       # OW = 8
    (2048, 64, 8, 512, "256x128"),       # OW = 4
    (100, 68, 32, 384, "256x128"),       # ragged pixel tail (25600 = 100 tiles), 3 column tiles
    (32, 64, 128, 128, "256x128"),       # OW = 64: four output rows per tile
    (128, 256, 8, 416, "256x128"),       # 416 channels (not a multiple of 64): 32 tiles, reduction split in eight
]


@pytest.mark.parametrize("case", IGEMM2_F_CASES)
def test_igemm2_forward_conv_matches_torch(case):
    """Conv2d k4 s2 p1 on the igemm2 skeleton (raw input rows staged by LDS-DMA, taps on the fragment read, borders
    read from a zeroed LDS region) against torch's CPU operator: plain, bias + activation, BatchNorm partial sums,
    and -- last case -- with the reduction split over two workgroups (slab + finish kernel)."""
    F = _F()
    from lightning_gan_zoo_amd._lib import lib
    N, C, H, K, label = case
    OH = H // 2
    tile = lib.gz_conv2d_tile(0, N, C, H, H, K, OH, OH, 4, 4, 2)
    assert F._TILES[tile] == label, (F._TILES[tile], label)
    x = rnd(N, C, H, H, seed=21)
    w = rnd(K, C, 4, 4, seed=22, scale=0.1)
    b = rnd(K, seed=23)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = TF.conv2d(x, w, None, 2, 1)
    xd, wd = x.cuda(), w.cuda()
    out = F._conv_fwd_raw(xd, wd, None, F.K4S2P1, F.ACT_NONE, 0.0)
    assert out.shape == ref.shape
    err = rel(out, ref)
    assert err < TOL, err
    out_b = F._conv_fwd_raw(xd, wd, b.cuda(), F.K4S2P1, F.ACT_LRELU, 0.2)
    assert rel(out_b, TF.leaky_relu(ref + b.view(1, -1, 1, 1), 0.2)) < TOL
    y, stats = F.conv2d_with_stats(xd, wd, F.K4S2P1)
    assert torch.equal(y, out)
    if stats.numel():               # (a split reduction is not fused with the statistics, by design)
        st = stats.double().sum(0)
        o64 = out.double()
        assert rel(st[:, 0], o64.sum((0, 2, 3))) < 1e-5 and rel(st[:, 1], (o64 * o64).sum((0, 2, 3))) < 1e-5


IGEMM2_WG_CASES = [
    # N, C (image-side channels), H (image side), K (feature-side channels)
    (512, 32, 32, 256),         # OW = 16: a chunk is one output row; 1 x 4 tiles, 8192 chunks -> 128-way split
    (512, 64, 16, 512),         # OW = 8: two rows per chunk; 2 x 8 tiles, 32-way split
    (1024, 128, 8, 256),        # OW = 4: four rows per chunk (a whole 4x4 map); 1 x 16 tiles, 16-way split
    (400, 24, 32, 288),         # ragged: 288 = 256 + 32 output channels, 384 columns, 6400 chunks
    (512, 32, 32, 128),         # 128 output channels: the 128 x 256 tile
    (512, 64, 16, 160),         # 160 = 128 + 32 output channels on that tile
]


@pytest.mark.parametrize("case", IGEMM2_WG_CASES)
def test_igemm2_weight_gradient_matches_torch(case):
    """Weight gradient of the k4 s2 p1 convolution on the igemm2 skeleton (register-staged transposing loaders, two
    LDS stages, reduction split over workgroups + slab reduction) against torch's CPU operator."""
    F = _F()
    from lightning_gan_zoo_amd._lib import lib
    N, C, H, K = case
    OH = H // 2
    tile = lib.gz_conv2d_tile(2, N, C, H, H, K, OH, OH, 4, 4, 2)
    assert F._TILES[tile] == ("256x128" if K >= 256 else "128x256"), F._TILES[tile]
    x = rnd(N, C, H, H, seed=31)
    gy = rnd(N, K, OH, OH, seed=32)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = torch.nn.grad.conv2d_weight(x, (K, C, 4, 4), gy, stride=2, padding=1)
    dw = F._conv_wgrad_raw(x.cuda(), gy.cuda(), F.K4S2P1)
    assert dw.shape == ref.shape
    err = rel(dw, ref)
    assert err < TOL, err
    assert torch.equal(dw, F._conv_wgrad_raw(x.cuda(), gy.cuda(), F.K4S2P1))      # fixed summation order


IGEMM2W_CASES = [
    # N, C, H, W (image side), K: the shapes that reach igemm2w_kernel's less-travelled paths
    (64, 32, 64, 64, 256),       # rows of 32 pixels: two chunks per row, left / right halo quads differ per chunk
    (32, 16, 128, 128, 256),     # rows of 64 pixels: interior chunks with both halo quads inside the image
    (2048, 20, 16, 16, 256),     # 20 input channels: the last column tile holds 4 channels (the others out of range)
    (2048, 32, 16, 8, 256),      # 8 x 4 feature map: two 4 x 4 chunks per image (top / bottom halo rows differ)
    (1024, 32, 8, 32, 256),      # 4 x 16 feature map: one-row chunks, rows 0 and 3 lose their top / bottom halo row
    (2048, 40, 16, 16, 128),     # the 128 x 256 tile with a ragged column tile (640 columns = 2.5 tiles)
]


@pytest.mark.parametrize("case", IGEMM2W_CASES)
def test_igemm2w_weight_gradient_edge_shapes(case):
    """igemm2w_kernel (weight gradient of k4 s2 p1, both operands by LDS-DMA): halo quads / rows at every kind of
    chunk position, ragged channel tiles, non-square maps -- against torch's CPU operator, and against the
    register-staged kernel (GZ_NO_IGEMM2W is read once per process, so that comparison is indirect: same reference)."""
    F = _F()
    from lightning_gan_zoo_amd._lib import lib
    N, C, H, W, K = case
    OH, OW = H // 2, W // 2
    tile = lib.gz_conv2d_tile(2, N, C, H, W, K, OH, OW, 4, 4, 2)
    assert F._TILES[tile] == ("256x128" if K >= 256 else "128x256"), F._TILES[tile]
    x = rnd(N, C, H, W, seed=41)
    gy = rnd(N, K, OH, OW, seed=42)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = torch.nn.grad.conv2d_weight(x, (K, C, 4, 4), gy, stride=2, padding=1)
    dw = F._conv_wgrad_raw(x.cuda(), gy.cuda(), F.K4S2P1)
    err = rel(dw, ref)
    assert err < TOL, err
    # an unaligned view takes the register-staged kernel: same numbers to rounding
    xs = torch.empty(x.numel() + 1, device="cuda")[1:].view_as(x).copy_(x)
    dw2 = F._conv_wgrad_raw(xs, gy.cuda(), F.K4S2P1)
    assert rel(dw2, ref) < TOL
    assert rel(dw2, dw) < 1e-5


IGEMM2_TAP_CASES = [
    # N, C, H, K, k, stride, pad
    (64, 64, 64, 128, 5, 2, 2),       # HoloGAN EXT-128 D.block1: 256 tiles, 100 chunks, no split
    (64, 128, 32, 256, 5, 2, 2),      # D.block2: 128 tiles, 200 chunks -> 2 splits
    (64, 256, 16, 512, 5, 2, 2),      # D.block3: 64 tiles, 400 chunks -> 4 splits
    (64, 72, 32, 160, 3, 1, 1),       # 3x3 s1 p1, 72 channels (last chunk of a tap: 8 live rows), 160 = 128 + 32 columns
    (67, 72, 32, 160, 3, 1, 1),       # the same with a ragged last pixel tile (68608 pixels = 268 tiles)
    (64, 1024, 16, 1024, 1, 1, 0),    # HoloGAN's 1x1 projection (its input gradient): one tap, 64 chunks
    (70, 200, 32, 160, 1, 1, 0),      # 1x1 with 200 channels: 13 chunks, the last one 8 live rows; weight image without padding rows
    (70, 64, 64, 128, 5, 2, 2),       # 71680 pixels = 280 tiles: more than one workgroup per CU for some
    (33, 128, 32, 256, 5, 2, 2),      # 8448 pixels = 33 x 2 tiles, 200 chunks -> 3 splits of 67, ragged batch
]


@pytest.mark.parametrize("case", IGEMM2_TAP_CASES)
def test_igemm2_tap_major_forward_matches_torch(case):
    """Forward convolutions whose reduction is tap-major (5x5 s2 p2, 3x3 ...) on the igemm2 skeleton: gather loader
    with 4-byte LDS-DMA pieces (three per k-step), padding taps as out-of-range lanes; bias + LeakyReLU epilogue."""
    F = _F()
    from lightning_gan_zoo_amd._lib import lib
    N, C, H, K, k, st, pd = case
    geom = F.Geom(k, k, st, pd)
    OH = (H + 2 * pd - k) // st + 1
    tile = lib.gz_conv2d_tile(0, N, C, H, H, K, OH, OH, k, k, st)
    x = rnd(N, C, H, H, seed=51)
    w = rnd(K, C, k, k, seed=52, scale=0.05)
    b = rnd(K, seed=53)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = TF.leaky_relu(TF.conv2d(x, w, b, st, pd), 0.2)
    out = F._conv_fwd_raw(x.cuda(), w.cuda(), b.cuda(), geom, F.ACT_LRELU, 0.2)
    assert out.shape == ref.shape
    err = rel(out, ref)
    assert err < TOL, (err, F._TILES[tile])
    if case[:4] in ((64, 64, 64, 128), (64, 128, 32, 256), (64, 256, 16, 512), (70, 64, 64, 128), (64, 1024, 16, 1024),
                    (70, 200, 32, 160)):
        assert F._TILES[tile] == "256x128", F._TILES[tile]


IGEMM2_TAP_DG_CASES = [
    # N, C (image side), H (image side), K (feature side): dx[N, C, H, H] from dy[N, K, H/2, H/2], 5x5 s2 p2
    (64, 128, 32, 256),       # HoloGAN EXT-128 D.block2 backward-data: 64 tiles x 4 phases, 144 / 96 / 96 / 64 chunks, 3 splits
    (64, 256, 16, 512),       # D.block3: 16 x 2 tiles, 288 / 192 / 192 / 128 chunks, 6 splits
    (33, 160, 32, 200),       # ragged: 8448 pixels per phase, 160 = 128 + 32 columns, 200 feature channels (208 padded)
    (256, 128, 32, 128),      # many tiles, unsplit: round 6 -- the row-shared kernel (ConvDg5A2, column-phase pairs)
]


@pytest.mark.parametrize("case", IGEMM2_TAP_DG_CASES)
def test_igemm2_tap_major_transposed_conv_matches_torch(case):
    """Backward-data of the 5x5 s2 p2 convolution (phases of 9 / 6 / 6 / 4 taps) on the igemm2 skeleton: gather loader,
    per-phase chunk counts, reduction cut into equal pieces across the phases (slabs + finish kernel)."""
    F = _F()
    from lightning_gan_zoo_amd._lib import lib
    N, C, H, K = case
    geom = F.Geom(5, 5, 2, 2)
    tile = lib.gz_conv2d_tile(1, N, C, H, H, K, H // 2, H // 2, 5, 5, 2)
    assert F._TILES[tile] == ("256x(4x32)" if case in ((256, 128, 32, 128), (64, 128, 32, 256), (64, 256, 16, 512)) else "256x128"), F._TILES[tile]
    gy = rnd(N, K, H // 2, H // 2, seed=61)
    w = rnd(K, C, 5, 5, seed=62, scale=0.05)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = torch.nn.grad.conv2d_input((N, C, H, H), w, gy, stride=2, padding=2)
    out = F._conv_dgrad_raw(gy.cuda(), w.cuda(), None, geom, (H, H), F.ACT_NONE, 0.0)
    assert out.shape == ref.shape
    err = rel(out, ref)
    assert err < TOL, err
    assert torch.equal(out, F._conv_dgrad_raw(gy.cuda(), w.cuda(), None, geom, (H, H), F.ACT_NONE, 0.0))


IGEMM2_TAP_DG_S1_CASES = [
    # N, C (image side), H, K (feature side), k, pad: stride-1 input gradients (one phase)
    (64, 256, 32, 256, 3, 1),         # residual-block 3x3: 256 x 2 tiles, 144 chunks, unsplit
    (16, 128, 64, 136, 3, 1),         # 136 feature channels (144 padded), 256 tiles
    (8, 256, 32, 512, 3, 1),          # 32 x 2 tiles, 288 chunks -> 4 splits
    (64, 1024, 16, 1024, 1, 0),       # HoloGAN's 1x1 transposed convolution (forward)
    (70, 160, 32, 200, 1, 0),         # 1x1, ragged everything
]


@pytest.mark.parametrize("case", IGEMM2_TAP_DG_S1_CASES)
def test_igemm2_gather_stride1_input_gradient_matches_torch(case):
    F = _F()
    from lightning_gan_zoo_amd._lib import lib
    N, C, H, K, k, pd = case
    geom = F.Geom(k, k, 1, pd)
    tile = lib.gz_conv2d_tile(1, N, C, H, H, K, H, H, k, k, 1)
    assert F._TILES[tile] == "256x128", F._TILES[tile]
    gy = rnd(N, K, H, H, seed=71)
    w = rnd(K, C, k, k, seed=72, scale=0.05)
    b = rnd(C, seed=73)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = torch.relu(torch.nn.grad.conv2d_input((N, C, H, H), w, gy, stride=1, padding=pd) + b.view(1, -1, 1, 1))
    out = F._conv_dgrad_raw(gy.cuda(), w.cuda(), b.cuda(), geom, (H, H), F.ACT_RELU, 0.0)
    assert out.shape == ref.shape
    err = rel(out, ref)
    assert err < TOL, err


IGEMM2WG_CASES = [
    # N, C, H (image side, square), K, k, stride, pad: the generic raw-row image of the LDS-DMA weight gradient
    (64, 64, 64, 128, 5, 2, 2),       # HoloGAN EXT-128 D.block1: rows of 32 pixels, the 128 x 256 tile, 1600 columns
    (64, 128, 32, 256, 5, 2, 2),      # D.block2: rows of 16 pixels, 3200 columns = 25 tiles
    (64, 256, 16, 512, 5, 2, 2),      # D.block3: rows of 8 pixels (two per chunk)
    (256, 256, 8, 512, 5, 2, 2),      # rows of 4 pixels (a whole 4 x 4 map per chunk)
    (128, 52, 32, 256, 5, 2, 2),      # 52 channels: 1300 columns = 10.2 tiles, the last one mostly empty
    (16, 64, 64, 256, 3, 1, 1),       # 3x3 s1 p1, rows of 64 pixels: interior chunks with both halo quads inside
    (64, 128, 16, 256, 3, 1, 1),      # 3x3 s1 p1, rows of 16 pixels
    (1024, 64, 8, 256, 3, 1, 1),      # 3x3 s1 p1, rows of 8 pixels
    (4096, 64, 4, 256, 3, 1, 1),      # 3x3 s1 p1, 4 x 4 maps
]


@pytest.mark.parametrize("case", IGEMM2WG_CASES)
def test_igemm2w_generic_geometry_weight_gradient(case):
    """igemm2w_kernel with WgImgBG (5x5 s2 p2 and 3x3 s1 p1): a 32-column block is not a whole number of channels, top
    halo of two rows, per-block lane bases -- against torch's CPU operator."""
    F = _F()
    from lightning_gan_zoo_amd._lib import lib
    N, C, H, K, k, st, pd = case
    geom = F.Geom(k, k, st, pd)
    OH = (H + 2 * pd - k) // st + 1
    tile = lib.gz_conv2d_tile(2, N, C, H, H, K, OH, OH, k, k, st)
    assert F._TILES[tile] == ("256x128" if K >= 256 else "128x256"), F._TILES[tile]
    x = rnd(N, C, H, H, seed=81)
    gy = rnd(N, K, OH, OH, seed=82)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = torch.nn.grad.conv2d_weight(x, (K, C, k, k), gy, stride=st, padding=pd)
    dw = F._conv_wgrad_raw(x.cuda(), gy.cuda(), geom)
    assert dw.shape == ref.shape
    err = rel(dw, ref)
    assert err < TOL, err
    assert torch.equal(dw, F._conv_wgrad_raw(x.cuda(), gy.cuda(), geom))


def _random_gather_cases(n, seed):
    """Shapes around the igemm2 plans' boundaries (tile counts near 224 / 256, ragged pixel tiles, channel counts that
    are not multiples of a chunk, column counts that are not multiples of a tile), seeded."""
    import random
    rng = random.Random(seed)
    out = []
    while len(out) < n:
        k, st, pd = rng.choice([(5, 2, 2), (3, 1, 1), (1, 1, 0)])
        H = rng.choice([8, 16, 32]) * (2 if st == 2 else 1)
        C = rng.choice([16, 24, 64, 72, 128, 200])
        K = rng.choice([128, 132, 160, 256, 320])
        OH = H // st
        pixels_wanted = rng.choice([150, 223, 224, 257, 300, 520]) * 256 // ((K + 127) // 128)
        N = max(1, pixels_wanted // (OH * OH) + rng.choice([0, 1]))
        if N * C * H * H > 40e6 or N * K * OH * OH > 40e6:
            continue
        out.append((N, C, H, K, k, st, pd))
    return out


@pytest.mark.parametrize("case", _random_gather_cases(14, 7))
def test_gather_paths_on_plan_boundaries(case):
    """Forward, input gradient and weight gradient of the tap-major geometries at seeded shapes around the plans'
    thresholds: whichever kernel the plan picks, the result matches torch (and repeats bit for bit)."""
    F = _F()
    N, C, H, K, k, st, pd = case
    geom = F.Geom(k, k, st, pd)
    OH = (H + 2 * pd - k) // st + 1
    x = rnd(N, C, H, H, seed=91)
    w = rnd(K, C, k, k, seed=92, scale=0.05)
    gy = rnd(N, K, OH, OH, seed=93)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    y = F._conv_fwd_raw(x.cuda(), w.cuda(), None, geom, F.ACT_NONE, 0.0)
    assert rel(y, TF.conv2d(x, w, None, st, pd)) < TOL
    dx = F._conv_dgrad_raw(gy.cuda(), w.cuda(), None, geom, (H, H), F.ACT_NONE, 0.0)
    assert rel(dx, torch.nn.grad.conv2d_input((N, C, H, H), w, gy, stride=st, padding=pd)) < TOL
    dw = F._conv_wgrad_raw(x.cuda(), gy.cuda(), geom)
    assert rel(dw, torch.nn.grad.conv2d_weight(x, (K, C, k, k), gy, stride=st, padding=pd)) < TOL
    assert torch.equal(dx, F._conv_dgrad_raw(gy.cuda(), w.cuda(), None, geom, (H, H), F.ACT_NONE, 0.0))
    assert torch.equal(dw, F._conv_wgrad_raw(x.cuda(), gy.cuda(), geom))


@pytest.mark.parametrize("case", [(3, 3, 16, 16), (2, 1, 32, 24), (4, 4, 64, 64), (64, 3, 128, 64), (130, 3, 64, 72),
                                  (5, 2, 8, 16)])
def test_small_channel_transposed_conv_5x5(case):
    """ConvTranspose2d k5 s2 p2 (output_padding 1) onto <= 4 image channels -- the gradient of HoloGAN's first critic
    convolution with respect to the image: the four-positions direct kernel with the tap-major weight pack (9 / 6 / 6 / 4
    taps per phase), both the channel-split (small maps) and the full-chip form; bias + tanh epilogue."""
    F = _F()
    N, C, H, K = case                      # image side [N, C, H, H], feature side [N, K, H/2, H/2]
    geom = F.Geom(5, 5, 2, 2)
    gy = rnd(N, K, H // 2, H // 2, seed=27)
    w = rnd(K, C, 5, 5, seed=28, scale=0.1)
    b = rnd(C, seed=29)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = torch.tanh(torch.nn.grad.conv2d_input((N, C, H, H), w, gy, stride=2, padding=2) + b.view(1, -1, 1, 1))
    out = F._conv_dgrad_raw(gy.cuda(), w.cuda(), b.cuda(), geom, (H, H), F.ACT_TANH, 0.0)
    assert out.shape == ref.shape and rel(out, ref) < TOL
