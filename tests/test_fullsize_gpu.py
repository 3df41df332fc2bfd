"""Parity at BASELINE.json's full sizes (features 64; bs 512 for dc_gan, bs 64 / in_planes 64 for hologan): the
cases the fixture tests are too small to reach -- train-mode BatchNorm statistics over 524,288 elements per
channel, a whole bs=512 discriminator step against the CPU oracle, the HoloGAN nets at the benchmarked shape."""
import numpy as np
import pytest
import torch

import scenario
from lightning_gan_zoo_amd.config import locate, make_cfg

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("shape,act", [((512, 128, 32, 32), "relu"), ((512, 64, 32, 32), "lrelu")])
def test_train_mode_batchnorm_at_bs512(shape, act):
    """G.block4 / D.block1-sized BatchNorm(train) + activation, forward, backward and the running buffers against
    float64 on the CPU (statistics over N*H*W = 524,288 values per channel)."""
    from lightning_gan_zoo_amd import functional as F
    N, C, H, W = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(shape, generator=g) * 1.7 + torch.linspace(-2, 2, C).view(1, C, 1, 1)
    go = torch.randn(shape, generator=g)
    gamma = 1 + 0.1 * torch.randn(C, generator=g)
    beta = 0.1 * torch.randn(C, generator=g)
    rm0, rv0 = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    # float64 reference
    bn = torch.nn.BatchNorm2d(C).double()
    with torch.no_grad():
        bn.weight.copy_(gamma), bn.bias.copy_(beta), bn.running_mean.copy_(rm0), bn.running_var.copy_(rv0)
    xr = x.double().requires_grad_()
    pre = bn(xr)
    ref = torch.relu(pre) if act == "relu" else torch.nn.functional.leaky_relu(pre, 0.2)
    ref.backward(go.double())
    xd = x.cuda().requires_grad_()
    gd, bd = gamma.cuda().requires_grad_(), beta.cuda().requires_grad_()
    rm, rv, nbt = rm0.cuda(), rv0.cuda(), torch.zeros((), dtype=torch.int64, device="cuda")
    out = F.batch_norm_act(xd, gd, bd, rm, rv, nbt, True, 0.1, 1e-5, F.ACT_RELU if act == "relu" else F.ACT_LRELU, 0.2)
    out.backward(go.cuda())
    errs = {"out": _rel(out, ref), "dx": _rel(xd.grad, xr.grad), "dgamma": _rel(gd.grad, bn.weight.grad),
            "dbeta": _rel(bd.grad, bn.bias.grad), "running_mean": _rel(rm, bn.running_mean),
            "running_var": _rel(rv, bn.running_var)}
    print(shape, act, {k: f"{v:.1e}" for k, v in errs.items()})
    assert max(errs.values()) < TOL, errs
    assert int(nbt) == int(bn.num_batches_tracked) == 1


@pytest.mark.parametrize("expt,bs,idx,sinks", [("dc_gan", 512, 0, False), ("dc_gan", 512, 1, False),
                                               ("wgan_gp", 256, 0, False), ("wgan_gp", 256, 1, False),
                                               # round 4: the metric string's own batch, and the Trainer's gradient
                                               # path (weight-gradient slabs summed into p.grad by gz_reduce_multi,
                                               # norm affine gradients written in place)
                                               ("dc_gan", 128, 0, True), ("dc_gan", 128, 1, True),
                                               ("dc_gan", 512, 0, True), ("wgan_gp", 256, 0, True)])
def test_baseline_size_step_matches_cpu_oracle(expt, bs, idx, sinks):
    """BASELINE configs 2 and 3 at their own sizes (features 64; dc_gan bs 512, wgan_gp bs 256), one whole
    ``training_step`` + backward per case against the CPU oracle on the same inputs, norm layers in TRAIN mode:

    * dc_gan / optimizer_idx 0 (reference lightning_module.py:112-121) and 1 (:124-128: BatchNorm through both
      networks, the generator's gradients arrive through the discriminator's input gradient);
    * wgan_gp / optimizer_idx 0 (:184-201 + utils.py:39-58): the gradient penalty's double backward at the batch
      where the launches take the 128x128 tiles and the split-K plans bs 8 never reaches -- loss, the penalty
      itself, every critic gradient; optimizer_idx 1 for completeness.

    Stable-mask scenario (ReLU / LeakyReLU pre-activations off the threshold), so the plain 1e-3 applies to every
    gradient (relative L2 per parameter) and every BatchNorm buffer."""
    from helpers import FixedNoise, synthetic_noise, synthetic_real
    torch.set_num_threads(min(16, torch.get_num_threads()))
    res = {}
    g = torch.Generator().manual_seed(4244)
    alpha = torch.rand(bs, 1, 1, 1, generator=g)
    for name, root, dev in (("hip", None, "cuda"), ("cpu", "oracle.reference_cpu", "cpu")):
        cfg = make_cfg(expt, **({"module_root": root} if root else {}), batch_size=bs)
        torch.manual_seed(42)
        step = locate(cfg.model.lm["_target_"])(cfg, None)
        scenario._prepare(step, True)
        step.to(dev)
        scenario._toggle(step, idx)
        step.noise_distn = FixedNoise(synthetic_noise(bs, 100, 4242))
        if expt == "wgan_gp":
            step.gp_alpha = alpha
        real = (synthetic_real(bs, seed=4243).abs() * 0.9 + 0.1).to(dev)
        use_sinks = sinks and root is None
        if use_sinks:
            from lightning_gan_zoo_amd import functional as F
            prev = F.set_grad_sinks(True)
        try:
            loss = step.training_step((real, torch.zeros(bs, dtype=torch.int64, device=dev)), 0, idx)
            loss.backward()
            if use_sinks:
                F.flush_grad_sinks()
        finally:
            if use_sinks:
                F.set_grad_sinks(*prev)
        net = step.discriminator if idx == 0 else step.generator
        extra = {}
        if expt == "wgan_gp" and idx == 0:      # the penalty on its own (the loss is lambda * gp - a difference of means)
            mod = "lightning_gan_zoo_amd.core.utils.utils" if root is None else "oracle.reference_cpu"
            gp_fn = getattr(__import__(mod, fromlist=["gradient_penalty"]), "gradient_penalty")
            with torch.no_grad():
                fake = step.generator(synthetic_noise(bs, 100, 4242).to(dev))
            extra["gp"] = float(gp_fn(step.discriminator, real, fake, device=dev, alpha=alpha).detach())
        res[name] = (float(loss.detach()),
                     {n: p.grad.detach().double().cpu() for n, p in net.named_parameters()},
                     {k: b.detach().double().cpu() for nn_ in ("generator", "discriminator")
                      for k, b in getattr(step, nn_).named_buffers(prefix=nn_)}, extra)
        other = step.generator if idx == 0 else step.discriminator
        assert all(p.grad is None for p in other.parameters()), "the frozen network received gradients"
    (lh, gh, bh, eh), (lc, gc, bc, ec) = res["hip"], res["cpu"]
    assert abs(lh - lc) <= TOL * max(1.0, abs(lc)), (lh, lc)
    for k in ec:
        assert abs(eh[k] - ec[k]) <= TOL * max(1.0, abs(ec[k])), (k, eh[k], ec[k])
    worst = {}
    for n, ref in gc.items():
        worst["grad " + n] = float((gh[n] - ref).norm() / ref.norm())
    for n, ref in bc.items():
        if ref.dtype == torch.float64 and "num_batches" not in n:
            worst["buffer " + n] = float((bh[n] - ref).abs().max() / ref.abs().max())
        else:
            assert torch.equal(bh[n], ref), n          # num_batches_tracked (D saw two batches in a D step, G one)
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
    print(f"{expt} bs{bs} optimizer_idx {idx} vs CPU oracle: loss {lh:.6f} / {lc:.6f} {eh} {ec}",
          [(k, f"{v:.1e}") for k, v in top])
    assert len(gc) >= 5 and top[0][1] <= TOL, top


@pytest.mark.parametrize("img", [64, 128])
def test_hologan_bs64_batch_consistency(img):
    """``img`` 128: EXT-128 at BASELINE config 5's size (128x128, bs 64 per GPU), forward and backward.

    The benchmarked HoloGAN shape (in_planes 64, z 128, bs 64).  Neither network shares a statistic across samples
    (AdaIN and InstanceNorm are per sample), so one bs=64 pass -- large tiles, split-K weight gradients -- must equal
    eight bs=8 passes: images, logits, latent predictions, input gradients and summed parameter gradients.  The bs=8
    pieces are pinned to the CPU oracle by test_hologan_step_gradients_with_pinned_masks."""
    from helpers import fill_closed_form
    cfg = make_cfg("hologan", batch_size=64, features=64, noise_dim=128, img_size=img)
    torch.manual_seed(42)
    step = locate(cfg.model.lm["_target_"])(cfg, None)
    scenario._prepare(step, True)          # masks off the threshold wherever they can be
    step.to("cuda")
    step.eval()                            # spectral norm: no power iteration between the passes (same sigma)
    G, D = step.generator, step.discriminator
    g = torch.Generator().manual_seed(77)
    z = (torch.rand(64, 128, generator=g) * 2 - 1).cuda()
    w = torch.randn(64, generator=g).cuda()
    scenario.seed_views(step, 9)
    view = G.sample_view(64)

    up = torch.randn(64, 3, img, img, generator=g).cuda()   # a fixed upstream gradient on the image
    with torch.no_grad():
        fake64 = G(z, view)

    def run(idx):
        # generator: forward + backward from a fixed image gradient (its own masks are off the threshold); the
        # discriminator sees identical images in both tilings, so only ITS masks can differ between them
        step.zero_grad(set_to_none=True)
        fake = G(z[idx], view[idx])
        (fake * up[idx]).sum().backward()
        x = fake64[idx].clone().requires_grad_()
        logit, zp = D(x)
        ((logit.reshape(-1) * w[idx]).sum() + (zp * zp).sum()).backward()
        grads = {"G." + n: p.grad.detach().clone() for n, p in G.named_parameters()}
        grads.update({"D." + n: p.grad.detach().clone() for n, p in D.named_parameters() if p.grad is not None})
        grads["D.input"] = x.grad.detach().clone()
        return fake.detach(), logit.detach().reshape(-1), zp.detach(), grads

    full = run(slice(0, 64))
    parts = [run(slice(i, i + 8)) for i in range(0, 64, 8)]
    worst = {}
    for k, name in enumerate(("images", "logits", "z prediction")):
        a, b = full[k], torch.cat([p[k] for p in parts])
        worst[name] = float((a - b).abs().max() / b.abs().max())
    for n, gf in full[3].items():
        gs = torch.cat([p[3][n] for p in parts]) if n == "D.input" else sum(p[3][n] for p in parts)
        if n.endswith(("convTranspose.bias", "conv2d.bias")) and ".block" in "." + n:
            continue                        # exact gradient 0 (constant in front of AdaIN / InstanceNorm)
        worst["grad " + n] = float((gf - gs).norm() / gs.norm().clamp_min(1e-30))
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
    print(f"hologan {img}x{img} bs64 vs 8 x bs8:", [(k, f"{v:.1e}") for k, v in top])
    # LeakyReLU entries behind InstanceNorm2d(affine=False) within rounding of zero can land on either side in the
    # two tilings (tests/mask_pinning.py): 2.5e-3 = one such entry in a discriminator layer
    # (the input gradient is per sample -- nothing averages a flipped entry out: 4e-3 observed over the 64 samples)
    assert max(v for k, v in worst.items() if k.startswith("grad D.") and k != "grad D.input") <= 2.5e-3, top
    assert worst["grad D.input"] <= 1e-2, top
    assert max(v for k, v in worst.items() if not k.startswith("grad D.")) <= TOL, top


@pytest.mark.parametrize("img", [64, 128])
def test_hologan_bs64_stacked_critic_step_matches_two_calls(img):
    """The benchmarked HoloGAN D step (in_planes 64, z 128, bs 64; ``img`` 128 = EXT-128) in training mode: ONE critic
    pass over [real; fake] with one sigma per half (the default) against two calls, both through the fused
    spectral-norm blocks -- the same power iterations, the same per-sample InstanceNorm, the pair loss against the two
    BCE means.

    Round 6 (ADVICE r5 / VERDICT r5 item 3): the comparison is PINNED instead of flip-tolerant.  The two-call step runs
    first and its ReLU / LeakyReLU decisions are taped; the stacked step then takes exactly those decisions (the two
    calls' masks of a layer concatenated: tests/mask_pinning.py), so that no LeakyReLU entry behind the zero-mean
    InstanceNorm can land on another side in the 128-row launches -- and every gradient is held to a FIXED 1e-3
    (round 5 scaled the bar with the number of differing entries, up to 2e-2).  The entries the stacked pass would have
    decided differently by itself are counted and must be rounding-level pre-activations, a few in 10^6."""
    import numpy as np
    from helpers import FixedNoise
    from mask_pinning import MaskTape, pinned_product_masks, record_product_masks, stack_discriminator_decisions
    res = {}
    tape = MaskTape()
    for stacked in (False, True):
        cfg = make_cfg("hologan", batch_size=64, features=64, noise_dim=128, img_size=img)
        torch.manual_seed(42)
        step = locate(cfg.model.lm["_target_"])(cfg, None)
        scenario._prepare(step, True)
        step.to("cuda")
        step.stack_d_passes = stacked
        step.real_first = False
        g = torch.Generator().manual_seed(77)
        step.noise_distn = FixedNoise(torch.rand(64, 128, generator=g) * 2 - 1)
        real = (torch.rand(64, 3, img, img, generator=g) * 1.8 - 0.9).cuda()
        labels = torch.zeros(64, dtype=torch.int64, device="cuda")
        np.random.seed(5)
        scenario._toggle(step, 0)
        step.zero_grad(set_to_none=True)
        if not stacked:
            with record_product_masks(tape):
                loss = step.training_step((real, labels), 0, 0)
            assert len(tape.masks) == 11 + 5 + 5       # the generator's decisions, then the critic's five per call
        else:
            one = stack_discriminator_decisions(tape, "hologan")
            with pinned_product_masks(one.rewind()):
                loss = step.training_step((real, labels), 0, 0)
            one._skip_reserved()
            assert one.cursor == len(one.masks) == 11 + 5
        loss.backward()
        res[stacked] = (float(loss.detach()), {n: p.grad.detach().clone() for n, p in step.discriminator.named_parameters()},
                        {n: b.detach().clone() for n, b in step.discriminator.named_buffers()})
    flips = sum(m[1] for m in one.mismatches)
    entries = sum(m.numel() for m in one.masks[11:])
    (la, ga, ba), (lb, gb, bb) = res[True], res[False]
    assert abs(la - lb) <= 1e-5 * max(1.0, abs(lb)), (la, lb)
    for n in bb:
        assert float((ba[n] - bb[n]).norm() / bb[n].norm().clamp_min(1e-30)) <= 1e-5, n
    worst = {}
    for n in gb:
        if n.startswith("blocks.") and n.endswith("conv2d.bias"):
            assert float(ga[n].abs().max()) == 0.0 and float(gb[n].abs().max()) == 0.0
            continue
        worst[n] = float((ga[n] - gb[n]).norm() / gb[n].norm().clamp_min(1e-30))
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:3]
    print(f"hologan {img}x{img} bs64 stacked (two-call decisions replayed) vs two calls: the stacked pass alone would decide "
          f"{flips} of {entries} critic entries differently;", [(k, f"{v:.1e}") for k, v in top])
    assert flips <= 2e-5 * entries and all(m[3] <= 1e-4 for m in one.mismatches), one.mismatches
    assert top[0][1] <= 1e-3, top
