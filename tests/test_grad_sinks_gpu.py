"""Gradient sinks (round 4): weight gradients of a first-order backward go straight into ``p.grad`` -- split launches
leave their slabs unreduced (gz_conv2d_wgrad_partial), ONE gz_reduce_multi launch sums the slabs of every pending
parameter, writing a fresh gradient or accumulating into the existing one.  Checked against autograd's own path
(sinks off: per-layer reduction + AccumulateGrad) and against torch's CPU convolution gradients."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _layers(dev):
    from lightning_gan_zoo_amd import functional as F
    torch.manual_seed(3)
    w1 = torch.nn.Parameter(torch.randn(128, 64, 4, 4, device=dev) * 0.05)      # Conv2d 64 -> 128, k4 s2 p1
    w2 = torch.nn.Parameter(torch.randn(128, 32, 4, 4, device=dev) * 0.05)      # ConvTranspose2d 128 -> 32
    w3 = torch.nn.Parameter(torch.randn(16, 3, 4, 4, device=dev) * 0.05)        # 3-channel edge layer (narrow tile)

    def net(x, x3):
        h = F.conv2d(x, w1, None, F.K4S2P1, F.ACT_LRELU, 0.2)
        y = F.conv_transpose2d(h, w2, None, F.K4S2P1, F.ACT_NONE, 0.0)
        z = F.conv2d(x3, w3, None, F.K4S2P1, F.ACT_NONE, 0.0)
        return y.square().mean() + z.square().mean()

    return (w1, w2, w3), net


def _cpu_reference(ws, inputs):
    import torch.nn.functional as TF
    ws = [w.detach().cpu().clone().requires_grad_(True) for w in ws]
    total = 0.0
    for x, x3 in inputs:
        h = TF.leaky_relu(TF.conv2d(x.cpu(), ws[0], None, 2, 1), 0.2)
        y = TF.conv_transpose2d(h, ws[1], None, 2, 1)
        z = TF.conv2d(x3.cpu(), ws[2], None, 2, 1)
        total = total + y.square().mean() + z.square().mean()
    total.backward()
    return [w.grad for w in ws]


def _rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("batch", [16, 96])
def test_sunk_weight_gradients_match_autograd_and_torch(batch):
    from lightning_gan_zoo_amd import functional as F
    dev = "cuda"
    ws, net = _layers(dev)
    g = torch.Generator().manual_seed(11)
    inputs = [(torch.randn(batch, 64, 32, 32, generator=g).to(dev), torch.randn(batch, 3, 32, 32, generator=g).to(dev))
              for _ in range(2)]
    ref = _cpu_reference(ws, inputs)

    # autograd's path: two uses per parameter, AccumulateGrad adds them
    (net(*inputs[0]) + net(*inputs[1])).backward()
    plain = [w.grad.clone() for w in ws]
    for w in ws:
        w.grad = None

    prev = F.set_grad_sinks(True)
    try:
        (net(*inputs[0]) + net(*inputs[1])).backward()
        assert all(w.grad is None for w in ws[:2]), "the split launches must not have gone through AccumulateGrad"
        F.flush_grad_sinks()
    finally:
        F.set_grad_sinks(*prev)
    for w, p, r in zip(ws, plain, ref):
        assert w.grad is not None and w.grad.shape == w.shape
        assert _rel(w.grad, p) < 2e-6, _rel(w.grad, p)
        assert _rel(w.grad.cpu(), r) < 1e-3, _rel(w.grad.cpu(), r)      # (torch CPU: other summation order, LeakyReLU ties)

    # gradient accumulation: p.grad exists -> the launch accumulates into it (beta = 1)
    before = [w.grad.clone() for w in ws]
    prev = F.set_grad_sinks(True)
    try:
        net(*inputs[0]).backward()
        F.flush_grad_sinks()
    finally:
        F.set_grad_sinks(*prev)
    single = _cpu_reference(ws, inputs[:1])
    for w, b, s in zip(ws, before, single):
        assert _rel((w.grad - b).cpu(), s) < 1e-3, _rel((w.grad - b).cpu(), s)


def test_sinks_are_off_outside_a_trainer_step():
    """Plain ``loss.backward()`` users (the Lightning drop-in route) keep autograd's behaviour: after backward every
    gradient is in ``p.grad`` without any flush."""
    from lightning_gan_zoo_amd import functional as F
    ws, net = _layers("cuda")
    assert not F.grad_sinks_enabled()
    x = torch.randn(8, 64, 32, 32, device="cuda")
    x3 = torch.randn(8, 3, 32, 32, device="cuda")
    net(x, x3).backward()
    assert all(w.grad is not None for w in ws)


@pytest.mark.parametrize("expt", ["dc_gan", "wgan_gp", "hologan"])
def test_trainer_with_and_without_sinks_tracks(expt):
    """Three optimizer cycles with the sinks on (default) and off: the two paths sum the same slabs in a different
    order, so the first cycle's losses agree to rounding and the trajectory tracks (wgan_gp's beta1 = 0 Adam follows
    gradient signs and amplifies rounding differences after its first step: DESIGN.md section 5's trajectory bars)."""
    import numpy as np
    from helpers import synthetic_real
    from lightning_gan_zoo_amd.config import locate, make_cfg
    from lightning_gan_zoo_amd.harness import Trainer
    res = {}
    for sinks in (True, False):
        cfg = make_cfg(expt, batch_size=8, features=16, noise_dim=16)
        torch.manual_seed(42)
        module = locate(cfg.model.lm["_target_"])(cfg, None).to("cuda")
        trainer = Trainer(module, grad_sinks=sinks)
        torch.manual_seed(7)
        np.random.seed(7)
        labels = torch.zeros(8, dtype=torch.int64, device="cuda")
        res[sinks] = [float(trainer.step((synthetic_real(8, seed=700 + k).cuda(), labels))[0])
                      for k in range(3 * len(trainer.order))]
    for k, (a, b) in enumerate(zip(res[True], res[False])):
        bar = 1e-5 if k < 2 else (5e-2 if expt == "wgan_gp" else 2e-3)
        assert abs(a - b) <= bar * max(1.0, abs(b)), (k, res[True], res[False])


@pytest.mark.parametrize("norm", ["batch", "instance"])
def test_norm_affine_gradients_are_written_in_place(norm):
    """BatchNorm / InstanceNorm(affine) gamma and beta: under sinks the finalize kernel writes the first contribution
    into a fresh ``p.grad`` and ADDS the second (a discriminator applied twice) -- equal to autograd's accumulation."""
    from lightning_gan_zoo_amd import functional as F
    g = torch.Generator().manual_seed(2)
    C = 32
    xs = [torch.randn(8, C, 16, 16, generator=g).cuda() for _ in range(2)]
    gamma = torch.nn.Parameter((1 + 0.1 * torch.randn(C, generator=g)).cuda())
    beta = torch.nn.Parameter((0.1 * torch.randn(C, generator=g)).cuda())

    def loss():
        tot = 0.0
        for x in xs:
            if norm == "batch":
                rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
                nbt = torch.zeros((), dtype=torch.int64, device="cuda")
                y = F.batch_norm_act(x, gamma, beta, rm, rv, nbt, True, 0.1, 1e-5, F.ACT_LRELU, 0.2)
            else:
                y = F.instance_norm_act(x, gamma, beta, 1e-5, F.ACT_LRELU, 0.2)
            tot = tot + (y * y).mean()
        return tot

    loss().backward()
    plain = (gamma.grad.clone(), beta.grad.clone())
    gamma.grad = beta.grad = None
    prev = F.set_grad_sinks(True)
    try:
        loss().backward()
        F.flush_grad_sinks()
    finally:
        F.set_grad_sinks(*prev)
    assert _rel(gamma.grad, plain[0]) < 1e-5 and _rel(beta.grad, plain[1]) < 1e-5


@pytest.mark.parametrize("expt", ["dc_gan", "wgan_gp", "hologan"])
def test_adam_from_slabs_equals_reduce_then_step(expt):
    """Round 5: outside data parallelism the trainer hands the pending weight-gradient slabs to the fused Adam
    (functional.take_grad_sinks -> optim.Adam.step(sink_sources=...): gz_adam_step_from_slabs sums them with
    gz_reduce_multi's own code) instead of reducing them into p.grad first.  Same bits: four optimizer cycles with
    and without (an optimizer that does not accept sources) leave identical parameters and Adam moments."""
    import numpy as np
    from helpers import synthetic_real
    from lightning_gan_zoo_amd.config import locate, make_cfg
    from lightning_gan_zoo_amd.harness import Trainer
    res = {}
    for fused in (True, False):
        cfg = make_cfg(expt, batch_size=8, features=16, noise_dim=16)
        torch.manual_seed(42)
        module = locate(cfg.model.lm["_target_"])(cfg, None).to("cuda")
        trainer = Trainer(module)
        took = []
        for o in trainer.optim:
            opt = o["optimizer"]
            assert opt.accepts_sink_sources
            if not fused:
                opt.accepts_sink_sources = False
            else:
                inner = opt._step_from_slabs
                opt._step_from_slabs = lambda src, gs, inner=inner: (took.append(len(src)), inner(src, gs))[1]
        torch.manual_seed(7)
        np.random.seed(7)
        labels = torch.zeros(8, dtype=torch.int64, device="cuda")
        for k in range(4 * len(trainer.order)):
            trainer.step((synthetic_real(8, seed=900 + k).cuda(), labels))
        torch.cuda.synchronize()
        if fused:
            assert took and min(took) >= 3, took              # every step handed several conv weights over
        state = [module.state_dict()[k].detach().clone() for k in sorted(module.state_dict())]
        for o in trainer.optim:
            for p in o["optimizer"].param_groups[0]["params"]:
                st = o["optimizer"].state[p]
                state += [st["exp_avg"].clone(), st["exp_avg_sq"].clone(), st["step"].clone().cuda()]
        res[fused] = state
    assert len(res[True]) == len(res[False])
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)
