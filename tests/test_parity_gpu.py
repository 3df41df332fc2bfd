"""Parity of the HIP product path against the fixtures generated from the unmodified reference
(tests/golden/*.npz): two G+D pairs with Lightning's toggle semantics -- losses, D(G(z)) logits,
per-parameter gradients, norm buffers, post-optimizer parameters.  Bar: 1e-3 relative (fp32),
integer counters bit-exact (north star)."""
import numpy as np
import pytest
import torch

import scenario
from lightning_gan_zoo_amd.config import locate, make_cfg
from test_oracle_golden import build_oracle_step, compare, load_golden, set_alpha, update_agreement

pytestmark = pytest.mark.gpu

TOL = 1e-3


def build_product_step(expt, size, stable=False):
    cfg = make_cfg(expt, **scenario.cfg_kwargs(expt, size, stable))
    torch.manual_seed(42)
    return locate(cfg.model.lm["_target_"])(cfg, logging_dir=None)


LR = {"dc_gan": 2e-4, "wgan": 5e-5, "wgan_gp": 1e-4, "hologan": 1e-4, "gan_stability_r1": 1e-4}


@pytest.mark.parametrize("size", ["tiny", "full"])
@pytest.mark.parametrize("expt", scenario.ALL_EXPTS)
def test_regression_guard_product_vs_unpinned_reference_fixture(expt, size):
    """REGRESSION GUARD, not the parity statement for gradients.  Forward quantities at 1e-3 max-norm
    -- that part IS the parity bar -- and gradient-side quantities at max(5e-3, 10 x the reference's own
    fp32-vs-fp64 discrepancy) in relative L2 (see compare()): on these un-pinned fixtures a ReLU / LeakyReLU decision
    on a pre-activation that is zero up to rounding lands on either side in two fp32 implementations.  The plain
    1e-3 statements on every gradient are test_product_with_reference_mask_decisions_every_gradient_at_1e3 (the
    reference's own decisions replayed) and test_product_matches_stable_mask_fixture below."""
    inputs, golden, cond = load_golden(expt, size)
    step = build_product_step(expt, size)
    full = size == "tiny"
    out = scenario.run_scenario(step, inputs, "cuda", full=full, set_alpha=set_alpha,
                                shadow=build_oracle_step(expt, size))
    scale = float(np.abs(golden["probe/logits"]).max())
    if size == "tiny":
        # features 8 / bs 4: ONE ReLU decision on a pre-activation that is zero up to rounding moves a generator gradient
        # by ~1e-2 (4 samples x 8 channels average little out), and every change of a summation order re-rolls which
        # entries those are -- round 5 answered by loosening this guard's gradient floor to 1e-2.  Round 6: the tiny
        # nets' gradients are held to the plain 1e-3 where that is well defined, on the reference's own decisions
        # (*_tiny_pinned.npz: test_product_with_reference_mask_decisions... and test_default_path_gpu.py), and this
        # un-pinned guard keeps what does not depend on a decision: forward quantities, buffers, first losses (1e-3),
        # the optimizer plumbing below.
        drop = lambda d: {k: v for k, v in d.items() if not k.startswith("grad")}      # noqa: E731
        out, golden = drop(out), drop(golden)
    worst = compare(out, golden, TOL, f"hip {expt}/{size}", atol_scale=scale, cond=cond,
                    final_abs=2 * 2 * LR[expt], grad_floor=5e-3, report=True)
    print(f"{expt}/{size}: worst {worst[3]} err {worst[1]:.2e} (bar {worst[2]:.2e})")
    # second pair: HIP loss vs the CPU oracle evaluated on the SAME (HIP-trained) parameters
    for tag in ("d", "g"):
        got, ref = out[f"loss_{tag}1"], out[f"shadow_loss_{tag}1"]
        assert abs(got - ref) <= TOL * max(abs(ref), scale), (tag, got, ref)
    # optimizer plumbing: updates agree with the reference's wherever the gradient is above rounding
    frac, n = update_agreement(out, golden, scenario.initial_params(build_oracle_step(expt, size), full),
                               LR[expt])
    print(f"{expt}/{size}: {frac:.3f} of {n} parameter updates agree")
    # HoloGAN: conv biases in front of AdaIN have an exactly-zero gradient and the second-pair gradients
    # inherit the first pair's +-lr noise through AdaIN's 1/sigma, so fewer entries agree
    assert frac >= (0.6 if expt == "hologan" else 0.9) and n > 100


@pytest.mark.parametrize("size", ["tiny", "full"])
def test_r1_resnet_path_matches_reference_fixture(size):
    """GANStabilityR1.training_step on the ResNet G/D (full = the shipped nfilter 16 at
    128x128) against the fixture generated from the unmodified reference.  No ReLU here (LeakyReLU masks only,
    cond <= 3e-5 in the fixture), so gradients are held to the plain 1e-3 bar."""
    expt = scenario.R1_EXPT
    inputs, golden, cond = load_golden(expt, size)
    step = build_product_step(expt, size)
    full = size == "tiny"
    out = scenario.run_scenario(step, inputs, "cuda", full=full, shadow=build_oracle_step(expt, size))
    scale = float(np.abs(golden["probe/logits"]).max())
    worst = compare(out, golden, TOL, f"hip {expt}/{size}", atol_scale=scale, cond=cond,
                    final_abs=2 * 2 * 10 * LR[expt], grad_floor=TOL)   # RMSprop's first steps are ~10 lr
    print(f"{expt}/{size}: worst {worst[3]} err {worst[1]:.2e} (bar {worst[2]:.2e})")
    for tag in ("d", "g"):
        got, ref = out[f"loss_{tag}1"], out[f"shadow_loss_{tag}1"]
        assert abs(got - ref) <= TOL * max(abs(ref), scale), (tag, got, ref)
    frac, n = update_agreement(out, golden, scenario.initial_params(build_oracle_step(expt, size), full),
                               LR[expt])
    print(f"{expt}/{size}: {frac:.3f} of {n} parameter updates agree")
    assert frac >= 0.9 and n > 100


@pytest.mark.parametrize("expt", ["dc_gan", "wgan_gp", "hologan"])
def test_product_matches_stable_mask_fixture(expt):
    """Masks away from the threshold: every loss, logit and GRADIENT at the plain 1e-3 bar
    (cond <= 3e-4 everywhere in the dc_gan / wgan_gp fixtures, so the conditioning slack is at most 3e-3).

    hologan (the reference's default in_planes 64, z 128, bs 8; both pairs' gradients recorded, spectral-norm
    u / v included): the generator's masks and the discriminator's first / head masks are stable, the three
    InstanceNorm2d(affine=False) -> LeakyReLU stages cannot be (scenario.stabilise_hologan).  Of the 3.2 M
    pre-activations they see, the closest is 3e-7 from zero in the reference run (``margin/`` in the fixture), and
    ONE element landing on the other side moves that layer's weight gradient by ~0.8 / sqrt(positions * channels)
    = 1.6e-3 -- the reference's own fp32-vs-fp64 pair shows exactly that (cond 5e-4 .. 7e-3 on the discriminator's
    second-pair gradients, <= 1e-4 elsewhere).  Hence max(1e-3, 10 cond) per quantity, as everywhere else.

    Second-pair GENERATOR gradients (``grad1_g/*``), hologan only: not compared here (round 6; rounds 2-5 carried a 5e-2
    floor).  The discriminator's input gradient is a sum of nearly cancelling contributions (|d loss / d image| = 0.07
    against 15 inside the blocks), so ONE LeakyReLU decision of the G step taken the other way moves the generator's
    gradients by up to 3.8e-2 (final_layer) -- measured in round 2 by running this product with two split-K plans of the
    same ConvTranspose3d, i.e. a 1e-9 relative perturbation of the volume: one plan lands on the reference's side of that
    3e-7 margin, the other does not, every forward quantity and every first-pair gradient agreeing to 1e-6 in both.
    Which side an fp32 implementation lands on is not a property the reference defines; the 1e-3 statements about the
    generator step's gradients are the reference-decision tests (``hologan_full_pinned.npz``, both call orders)."""
    inputs, golden, cond = load_golden(expt, "full", stable=True)
    step = build_product_step(expt, "full", stable=True)
    out = scenario.run_scenario(step, inputs, "cuda", full=False, set_alpha=set_alpha, stable=True)
    scale = float(np.abs(golden["probe/logits"]).max())
    if expt == "hologan":
        # round 6: the 5e-2 floor on these is gone.  One LeakyReLU decision of the G step (|pre-activation| 3e-7 in the
        # reference) moves them by 3.8e-2 whichever implementation takes it; a bar that wide guards nothing, and the
        # 1e-3 statement on the generator step's gradients exists on the reference's own decisions
        # (test_hologan_with_reference_mask_decisions_every_gradient_at_1e3, test_default_path_gpu.py)
        drop = lambda d: {k: v for k, v in d.items() if not k.startswith("grad1_g/")}      # noqa: E731
        out, golden = drop(out), drop(golden)
    worst = compare(out, golden, TOL, f"hip {expt}/full/stable", atol_scale=scale, cond=cond, report=True)
    print(f"{expt}/full/stable: worst {worst[3]} err {worst[1]:.2e} (bar {worst[2]:.2e})")


@pytest.mark.parametrize("init,img,bs", [("closed_form", 64, 8), ("default_init", 64, 8), ("stable", 64, 8),
                                         ("closed_form", 128, 4), ("stable", 128, 4)])
def test_hologan_step_gradients_with_pinned_masks(init, img, bs):
    _hologan_pinned(init, img, bs, 0)


@pytest.mark.parametrize("off", [1, 2, 3, 4, 5])
def test_hologan_pinned_gradients_do_not_depend_on_the_seed(off):
    """The same statement for five more draws of parameters, latents, images and views (the stable fixture
    is the best of six input seeds -- this check has no fixture to select)."""
    _hologan_pinned("default_init", 64, 8, 1000 * off)


def _hologan_pinned(init, img, bs, off):
    """``img`` 128 = EXT-128 (BASELINE config 5's image size; the reference cannot run there, the
    oracle carries the same stride-2 extension): the whole training step -- forward AND backward of both
    optimizer indices -- at in_planes 64, as below.

    HOLOGAN.training_step at the reference's default width (in_planes 64, z 128) and bs 8, HIP vs the CPU oracle
    holding the same state and taking the SAME ReLU / LeakyReLU decisions (tests/mask_pinning.py): every parameter
    gradient of the D step and of the G step within 1e-3 relative L2, the spectral-norm buffers and losses within
    1e-3.  The handful of mask entries on which the two implementations would disagree by themselves are counted
    and must be rounding-level (|pre-activation| <= 1e-4, fewer than 1 % of all entries; with the default
    initialisation whole AdaIN rows have scale = shift = 0, i.e. pre-activations of +-0 / 1e-12, and account for
    almost all of them)."""
    from helpers import FixedNoise, synthetic_noise, synthetic_real
    from mask_pinning import MaskTape, pinned_oracle_masks, record_product_masks
    kw = dict(batch_size=bs, features=64, noise_dim=128, img_size=img)
    steps = {}
    for name, root in (("hip", None), ("cpu", "oracle.reference_cpu")):
        cfg = make_cfg("hologan", **({"module_root": root} if root else {}), **kw)
        torch.manual_seed(1234 + off)
        steps[name] = locate(cfg.model.lm["_target_"])(cfg, None)
    hip, cpu = steps["hip"], steps["cpu"]
    hip.real_first = False        # the tape is replayed in call order: keep the reference's (G(z), D(real), D(fake))
    hip.stack_d_passes = False    # ... and its two discriminator calls
    if init != "default_init":
        scenario._prepare(hip, init == "stable")
    for net in ("generator", "discriminator"):
        getattr(cpu, net).load_state_dict(getattr(hip, net).state_dict())
    hip.to("cuda")
    worst, flips, entries = {}, 0, 0
    for idx, tag in ((0, "d"), (1, "g")):
        z = synthetic_noise(bs, 128, 41 + idx + off, uniform=True)
        real = synthetic_real(bs, size=img, seed=51 + idx + off)
        if init == "stable":
            real = real.abs() * 0.9 + 0.1
        labels = torch.zeros(bs, dtype=torch.int64)
        res = {}
        tape = MaskTape()
        for name, step, dev in (("hip", hip, "cuda"), ("cpu", cpu, "cpu")):
            scenario._toggle(step, idx)
            step.zero_grad(set_to_none=True)
            step.noise_distn = FixedNoise(z)
            scenario.seed_views(step, 61 + idx + off)
            batch = (real.clone().to(dev), labels.to(dev))
            if name == "hip":
                with record_product_masks(tape):
                    loss = step.training_step(batch, idx, idx)
            else:
                with pinned_oracle_masks(step, tape):
                    loss = step.training_step(batch, idx, idx)
                assert tape.cursor == len(tape.masks), "the two forward passes took different numbers of mask decisions"
            loss.backward()
            net = step.discriminator if idx == 0 else step.generator
            res[name] = (float(loss.detach()), {n: p.grad.detach().double().cpu() for n, p in net.named_parameters()},
                         {n: b.detach().double().cpu() for n, b in step.discriminator.named_buffers()})
        flips += sum(m[1] for m in tape.mismatches)
        if tape.mismatches:
            print(f"  {tag}-step mask decisions that differ (tape position, entries, of, max |x|):", tape.mismatches)
        entries += sum(m.numel() for m in tape.masks)
        assert all(m[3] <= 1e-4 for m in tape.mismatches), tape.mismatches     # only rounding-level pre-activations
        (lh, gh, bh), (lc, gc, bc) = res["hip"], res["cpu"]
        assert abs(lh - lc) <= TOL * max(1.0, abs(lc)), (tag, lh, lc)
        for n in bc:
            assert float((bh[n] - bc[n]).abs().max() / bc[n].abs().max()) <= TOL, (tag, n)
        for n, ref in gc.items():
            if n.endswith(("convTranspose.bias", "conv2d.bias")) and n.startswith("block"):
                # a per-channel constant in front of AdaIN / InstanceNorm is removed by the mean subtraction: the
                # exact gradient is 0 and both implementations hold rounding noise
                wn = n.rsplit(".", 1)[0] + (".weight_orig" if tag == "d" else ".weight")
                assert float(gh[n].norm()) <= 1e-3 * float(gh[wn].norm()), (tag, n)
                continue
            worst[f"{tag}/{n}"] = float((gh[n] - ref).norm() / ref.norm())
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:3]
    print(f"hologan {init} {img}x{img} bs {bs}: {flips} of {entries} mask entries differ by themselves; worst gradients (rel L2):",
          [(k, f"{v:.1e}") for k, v in top])
    assert flips <= 1e-2 * entries
    assert top[0][1] <= TOL, top


def test_logits_within_1e3_of_cpu_reference():
    """north star: D(G(z)) logits within 1e-3 of the CPU reference (features 64 nets)."""
    inputs, golden, _ = load_golden("dc_gan", "full")
    step = build_product_step("dc_gan", "full")
    from helpers import fill_closed_form
    fill_closed_form(step.generator, 1)
    fill_closed_form(step.discriminator, 2)
    step.to("cuda")
    with torch.no_grad():
        logits = step.discriminator(step.generator(inputs["z_d0"].cuda())).reshape(-1).cpu().numpy()
    ref = golden["probe/logits"]
    assert np.abs(logits - ref).max() <= 1e-3 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("cfg", [
    dict(img_size=32, channels_img=1, features=8, bs=3, nz=7, norm="batch_norm"),      # MNIST-like, ragged everything
    dict(img_size=128, channels_img=3, features=4, bs=2, nz=10, norm="batch_norm"),    # deeper stacks (5 blocks)
    dict(img_size=64, channels_img=2, features=12, bs=5, nz=33, norm="instance_norm2d"),
    dict(img_size=64, channels_img=3, features=8, bs=1, nz=16, norm="identity"),       # batch of one, no norm in D
])
def test_standard_networks_other_shapes_match_oracle(cfg):
    """Shapes outside the benchmark configs (image sizes 32 / 128, 1-2 image channels, odd batch sizes,
    batch 1, identity norm): forward, input gradient and every parameter gradient vs the CPU oracle."""
    from helpers import fill_closed_form, rel_err
    from lightning_gan_zoo_amd.core.models import standard_networks as P
    from oracle import reference_cpu as O
    c = cfg
    torch.manual_seed(1)
    gp, go = (m.Generator(c["nz"], c["channels_img"], c["features"], c["img_size"]) for m in (P, O))
    dp, do = (m.Discriminator(c["channels_img"], c["features"], c["norm"], c["img_size"], False) for m in (P, O))
    for a, b in ((gp, go), (dp, do)):
        fill_closed_form(a, 5)
        with torch.no_grad():      # keep the ReLU / LeakyReLU masks behind the norms away from the threshold
            for n, p in a.named_parameters():
                if n.endswith(("batch_norm.bias", "instance_norm2d.bias")):
                    p.add_(8.0)
        b.load_state_dict(a.state_dict())
    gp.cuda(), dp.cuda()
    g = torch.Generator().manual_seed(9)
    z = torch.randn(c["bs"], c["nz"], generator=g)
    zp, zo = z.cuda().requires_grad_(), z.clone().requires_grad_()
    outs = []
    for G, D, zz in ((gp, dp, zp), (go, do, zo)):
        fake = G(zz)
        score = D(fake)
        (score.sum() + 0.1 * fake.pow(2).sum()).backward()
        outs.append((fake.detach().cpu(), score.detach().cpu(), zz.grad.cpu(),
                     {n: p.grad.cpu() for n, p in list(G.named_parameters()) + list(D.named_parameters())}))
    (fp, sp, gzp, pgp), (fo, so, gzo, pgo) = outs
    assert fp.shape == (c["bs"], c["channels_img"], c["img_size"], c["img_size"])
    assert rel_err(fp.numpy(), fo.numpy()) < TOL and rel_err(sp.numpy(), so.numpy()) < TOL
    # gradients: relative L2 (ReLU masks at rounding level, see test_oracle_golden.compare)
    def l2(a, b):
        return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    assert l2(gzp, gzo) < 5e-3
    worst = max((l2(pgp[n], pgo[n]), n) for n in pgo)
    assert worst[0] < 5e-3, worst


@pytest.mark.parametrize("expt", ["dc_gan", "wgan_gp", scenario.R1_EXPT])
def test_loss_trajectory_tracks_oracle(expt):
    """SURVEY 8-f3's stand-in for FID parity (Inception weights are not reachable offline): K = 12 optimizer cycles
    from the same default initialisation (same seed -> bit-identical parameters) on the same batches and latents,
    product on the GPU vs oracle on the CPU.  The first losses must agree to 1e-3; afterwards the optimizers
    turn rounding-level gradient differences into +-lr parameter differences, so the trajectories are held to
    2e-2 (observed over the pool's boxes: 3e-7 .. 9e-3 dc_gan, 1e-3 .. 1e-2 wgan_gp with a 5e-2 bar, 2e-7 R1), and the generator's eval-mode output (BatchNorm running statistics) is compared at
    the end with the oracle carrying the product's state."""
    from helpers import FixedNoise, synthetic_noise, synthetic_real
    # the CPU oracle's summation order depends on the intra-op thread count (oneDNN partitions its reductions by
    # thread): pinned, so that a box with another core count does not move the oracle's trajectory
    torch.set_num_threads(8)
    kw = dict(batch_size=8, features=8, noise_dim=16)
    if expt == scenario.R1_EXPT:
        kw.update(features=4, img_size=32)
    img = kw.get("img_size", 64)
    steps = {}
    for name, root, dev in (("hip", None, "cuda"), ("cpu", "oracle.reference_cpu", "cpu")):
        cfg = make_cfg(expt, **({"module_root": root} if root else {}), **kw)
        torch.manual_seed(42)
        steps[name] = (locate(cfg.model.lm["_target_"])(cfg, None).to(dev), dev)
    for a, b in zip(steps["hip"][0].state_dict().values(), steps["cpu"][0].state_dict().values()):
        assert torch.equal(a.cpu(), b), "default initialisation differs"
    traj = {}
    for name, (step, dev) in steps.items():
        opts = step.configure_optimizers()
        labels = torch.zeros(8, dtype=torch.int64, device=dev)
        out = []
        for k in range(12):
            for idx in (0, 1):
                real = synthetic_real(8, size=img, seed=900 + 2 * k + idx).to(dev)
                step.noise_distn = FixedNoise(synthetic_noise(8, kw["noise_dim"], 950 + 2 * k + idx))
                if expt == "wgan_gp":
                    set_alpha(step, torch.rand(8, 1, 1, 1, generator=torch.Generator().manual_seed(990 + k)))
                scenario._toggle(step, idx)
                loss = step.training_step((real, labels), 2 * k + idx, idx)
                loss.backward()
                opts[idx]["optimizer"].step()
                opts[idx]["optimizer"].zero_grad()
                out.append(float(loss.item()))
        traj[name] = np.array(out)
    d = np.abs(traj["hip"] - traj["cpu"]) / np.maximum(1.0, np.abs(traj["cpu"]))
    print(f"{expt}: trajectory deviation first {d[:2].max():.1e}, max {d.max():.1e}; losses {traj['cpu'][:2]} -> {traj['cpu'][-2:]}")
    # wgan_gp: Adam with beta1 = 0 follows the sign of every gradient entry and the penalty (|g| - 1)^2 is steep: a
    # different (equally valid) summation order -- new tile shapes in round 2 -- moved one of its 24 losses by 2.8e-2.
    # The bars are NOT tightened to one box's observation (round 3 tried 1e-4 for dc_gan after seeing 3e-7: the next
    # box's CPU oracle, with another thread count and hence another summation order, gave 5.9e-3).
    assert d[:2].max() < TOL and d.max() < (5e-2 if expt == "wgan_gp" else 2e-2)
    assert np.abs(traj["cpu"][-2:] - traj["cpu"][:2]).max() > 1e-4, "the scenario did not train"
    # eval-mode generator (running statistics) with the product's trained state in the oracle
    hip, cpu = steps["hip"][0], steps["cpu"][0]
    cpu.generator.load_state_dict({k: v.cpu() for k, v in hip.generator.state_dict().items()})
    hip.generator.eval(), cpu.generator.eval()
    z = synthetic_noise(8, kw["noise_dim"], 999)
    with torch.no_grad():
        a, b = hip.generator(z.cuda()).cpu(), cpu.generator(z)
    assert float((a - b).abs().max()) < TOL * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("expt", ["dc_gan", "wgan", "wgan_gp", scenario.R1_EXPT, "hologan"])
def test_graphed_trainer_matches_eager_trainer(expt):
    """harness.GraphedTrainer (each optimizer step captured once in a HIP graph and replayed) against the eager
    Trainer: same seed, same host RNG stream, same batches -> the same kernels in the same order, so the loss
    trajectory and the final parameters must agree to rounding (bit-identical in practice).  hologan: the view
    matrices are staged like the noise, and its LambdaLR changes the learning rate at every ``end_epoch()`` of this
    run (num_epochs 4), which must reach the replayed optimizer launches (round-1 advisor finding)."""
    from helpers import synthetic_real
    from lightning_gan_zoo_amd.harness import GraphedTrainer, Trainer
    kw = dict(batch_size=8, features=8, noise_dim=16)
    if expt == scenario.R1_EXPT:
        kw.update(features=4, img_size=32)
    if expt == "hologan":
        kw["dotted"] = {"train.num_epochs": 4}
    img = kw.get("img_size", 64)
    results = {}
    for name, cls in (("eager", Trainer), ("graph", GraphedTrainer)):
        cfg = make_cfg(expt, **kw)
        torch.manual_seed(42)
        module = locate(cfg.model.lm["_target_"])(cfg, None).to("cuda")
        trainer = cls(module)
        torch.manual_seed(7)                      # host generator: latent noise / alpha draws
        np.random.seed(7)                         # HoloGAN's views
        labels = torch.zeros(8, dtype=torch.int64, device="cuda")
        losses = []
        for k in range(6 * len(trainer.order)):
            real = synthetic_real(8, size=img, seed=600 + k).cuda()
            loss, idx = trainer.step((real, labels))
            losses.append(float(loss.item()))
            if (k + 1) % (2 * len(trainer.order)) == 0:
                trainer.end_epoch()
        if expt == "hologan":
            assert trainer.optim[0]["optimizer"].param_groups[0]["lr"] < 1e-4       # the schedule did move
        results[name] = (np.array(losses), {k: v.detach().cpu().clone() for k, v in module.state_dict().items()})
    (le, se), (lg, sg) = results["eager"], results["graph"]
    if expt == "hologan":
        # the resampling adjoint accumulates with LDS float atomics (order not fixed), so two runs agree to rounding,
        # not bit for bit; parameters whose exact gradient is 0 (conv biases in front of InstanceNorm / AdaIN) then
        # random-walk by +-lr under Adam in both runs and are left out
        assert np.abs(le - lg).max() <= 2e-3 * max(1.0, np.abs(le).max()), (le, lg)     # observed 6e-4 after 18 steps
        for k in se:
            if k.endswith(("conv2d.bias", "conv2d_spec_norm.bias", "convTranspose.bias")) and \
                    ("blocks." in k or ".block" in k):
                continue
            assert torch.allclose(se[k].float(), sg[k].float(), rtol=1e-3, atol=2e-5), k
        return
    assert np.abs(le - lg).max() <= 1e-6 * max(1.0, np.abs(le).max()), (le, lg)
    for k in se:
        assert torch.allclose(se[k].float(), sg[k].float(), rtol=1e-6, atol=1e-7), k


def test_full_size_batch_consistency():
    """BASELINE size (features 64, bs 512), size-independent property: a sample's output does not depend on its
    batch when no statistic is shared across samples -- the generator with BatchNorm in eval mode, the WGAN-GP
    critic with its per-sample InstanceNorm -- and a sum-reduced loss makes gradients additive over the batch.
    The bs=512 pass runs the 128x128 / 128x64 tiles and the split-K weight gradients, the eight bs=64 passes the
    small tiles: both must give the same images, scores, input gradients and critic parameter gradients (1e-3);
    the small-batch pieces are themselves pinned to the CPU oracle by the fixture tests above."""
    from helpers import fill_closed_form
    step = build_product_step("wgan_gp", "full")
    fill_closed_form(step.generator, 1)
    fill_closed_form(step.discriminator, 2)
    scenario.stabilise(step)            # LeakyReLU masks away from the threshold: gradients comparable at 1e-3
    step.to("cuda")
    step.generator.eval()
    g = torch.Generator().manual_seed(77)
    z = torch.randn(512, 100, generator=g).cuda()
    w = torch.randn(512, generator=g).cuda()            # per-sample loss weights: a non-trivial upstream gradient
    with torch.no_grad():
        fake512 = step.generator(z)

    def run(idx):
        step.zero_grad(set_to_none=True)
        with torch.no_grad():
            fake = step.generator(z[idx])
        images = fake
        fake = fake512[idx].clone().requires_grad_()     # the critic sees identical inputs in both tilings
        scores = step.discriminator(fake).reshape(-1)
        (scores * w[idx]).sum().backward()
        grads = {n: p.grad.detach().clone() for n, p in step.discriminator.named_parameters()}
        return images, scores.detach(), fake.grad.detach(), grads

    full = run(slice(0, 512))
    parts = [run(slice(i, i + 64)) for i in range(0, 512, 64)]
    worst = {}
    for k, name in enumerate(("images", "scores", "input gradient")):
        a, b = full[k], torch.cat([p[k] for p in parts])
        # forward quantities in max norm; gradients in relative L2 (a LeakyReLU mask entry whose pre-activation is
        # zero up to rounding may differ between two tilings: a local O(1) change, see test_oracle_golden.compare)
        worst[name] = float((a - b).abs().max() / b.abs().max()) if k < 2 else float((a - b).norm() / b.norm())
        assert worst[name] <= TOL, (name, worst[name])
    for n, gf in full[3].items():
        gs = sum(p[3][n] for p in parts)
        err = float((gf - gs).norm() / gs.norm().clamp_min(1e-30))
        worst["parameter gradients"] = max(worst.get("parameter gradients", 0.0), err)
        assert err <= TOL, (n, err)
    print("bs512 vs 8 x bs64:", {k: f"{v:.1e}" for k, v in worst.items()})


@pytest.mark.parametrize("fixture", ["tiny", "full"])
@pytest.mark.parametrize("expt", ["dc_gan", "wgan", "wgan_gp"])
def test_product_with_reference_mask_decisions_every_gradient_at_1e3(expt, fixture):
    """The un-stabilised features-64 scenario with the ReLU / LeakyReLU decisions of the UNMODIFIED reference pinned
    (``*_full_pinned.npz``: ~7 M packed mask bits recorded by tests/golden/make_golden.py): the product takes exactly
    those decisions (tests/mask_pinning.py: fused op with ACT_NONE + where(mask, y, slope y) on the device), so no
    conditioning slack is needed and EVERY quantity is held to the plain 1e-3:

      * vs the fixture (the reference's own numbers): losses, BatchNorm buffers, and per gradient its norm, sum and
        16 sampled entries (what a features-64 fixture stores);
      * vs the CPU oracle run live with the same pinned decisions (pinned == reference at 1e-5, CPU suite:
        test_oracle_takes_the_reference_mask_decisions): every gradient of the D step and of the G step -- the
        WGAN-GP double backward included -- in full relative L2, every buffer in max norm.

    The decisions the product would have taken by itself are counted: they may differ from the reference's only on
    pre-activations that are zero up to rounding."""
    from mask_pinning import pinned_module_masks, pinned_product_masks
    from test_oracle_golden import PINNED_KW, load_pinned, pinned_scale
    inputs, golden, tape = load_pinned(expt, fixture)      # tiny: *_tiny_pinned.npz (features 8 / bs 4, round 6)
    scale = pinned_scale(expt, fixture)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    with pinned_module_masks(tape.rewind()):
        cpu = scenario.run_scenario(build_oracle_step(expt, fixture), inputs, "cpu", full=True, set_alpha=set_alpha,
                                    **PINNED_KW)
    assert tape.cursor == len(tape.masks)
    product = build_product_step(expt, fixture)
    product.real_first = False    # the reference's decisions are taped in ITS call order (G(z), D(real), D(fake))
    product.stack_d_passes = False
    with pinned_product_masks(tape.rewind()):
        hip = scenario.run_scenario(product, inputs, "cuda", full=True, set_alpha=set_alpha, **PINNED_KW)
    assert tape.cursor == len(tape.masks), "product and reference took different numbers of mask decisions"
    total = sum(m.numel() for m in tape.masks)
    flips = sum(m[1] for m in tape.mismatches)
    print(f"{expt}: the product alone would decide {flips} of {total} mask entries differently "
          f"(largest |pre-activation| among them {max([m[3] for m in tape.mismatches], default=0.0):.1e})")
    # a sanity bound on the diagnostic, not the parity bar (those follow): the entries the product alone would decide
    # differently are pre-activations within the 1e-3 tolerance of zero -- observed <= 7.8e-5 in round 3, 1.2e-4 once
    # the BatchNorm statistics of split launches come from the finish pass (other summation order), activations O(1)
    assert flips <= max(4, 1e-4 * total) and all(m[3] <= 3e-4 for m in tape.mismatches), tape.mismatches
    assert set(hip) == set(cpu) == set(golden)
    worst = []
    for k, ref in cpu.items():
        got = np.asarray(hip[k], dtype=np.float64)
        ref = np.asarray(ref, dtype=np.float64)
        if np.asarray(cpu[k]).dtype.kind in "iu":
            assert np.array_equal(hip[k], cpu[k]) and np.array_equal(hip[k], golden[k]), k
            continue
        if ref.ndim == 0:
            e = abs(float(got) - float(ref)) / max(abs(float(ref)), scale)
        elif k.startswith("grad"):
            e = float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30))        # full relative L2
        else:
            e = float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30))
        worst.append((e, k))
    worst.sort(reverse=True)
    print(f"{expt}: product vs pinned oracle, worst of {len(worst)}:", [(k, f"{e:.1e}") for e, k in worst[:4]])
    assert worst[0][0] <= TOL, worst[:4]
    # and against the reference's own numbers (summaries)
    summ = {k: (v if np.asarray(v).ndim == 0 or not k.startswith(("grad", "final/")) else
                scenario.summarize(torch.from_numpy(np.asarray(v)))) for k, v in hip.items()}
    if fixture == "tiny":
        summ = hip                  # the tiny fixture stores full tensors
    w = compare(summ, golden, TOL, f"hip {expt}/{fixture}/pinned vs reference", atol_scale=scale)
    print(f"{expt}: product vs reference fixture, worst {w[3]} {w[1]:.1e}")


@pytest.mark.parametrize("fixture", ["tiny", "full"])
def test_hologan_with_reference_mask_decisions_every_gradient_at_1e3(fixture):
    """HoloGAN's counterpart of the test above -- ``hologan_full_pinned.npz`` holds the ReLU / LeakyReLU
    decisions of the UNMODIFIED reference (in_planes 64, z 128, bs 8, plain closed-form parameters, non-right-angle
    views; one D step and one G step; 21 M decisions).  The product takes exactly those decisions: every fused op runs
    with ACT_NONE and the activation is where(mask, y, slope y) on the device; the five ZMapping ReLUs, which the
    product evaluates in one launch up front, take the tape positions at which the reference decides them
    (tests/mask_pinning.py: replay_at).  Every loss, spectral-norm buffer and gradient of both steps at the plain 1e-3:
    in full relative L2 against the CPU oracle run live with the same decisions (oracle + these decisions == reference
    at 1e-5: test_oracle_takes_the_reference_mask_decisions_hologan), and against the reference's own recorded numbers.
    No conditioning slack, no second-pair floor."""
    from mask_pinning import pinned_oracle_masks, pinned_product_masks
    from test_oracle_golden import (PINNED_KW, drop_exact_zero_gradients, load_pinned, pinned_scale,
                                    pinned_scenario_size)
    expt = "hologan"
    PINNED_SIZE = pinned_scenario_size(expt, fixture)      # tiny: hologan_tiny_pinned.npz (in_planes 8, bs 4; round 6)
    inputs, golden, tape = load_pinned(expt, fixture)
    scale = pinned_scale(expt, fixture)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    oracle = build_oracle_step(expt, PINNED_SIZE)
    with pinned_oracle_masks(oracle, tape.rewind()):
        cpu = scenario.run_scenario(oracle, inputs, "cpu", full=True, **PINNED_KW)
    # (on the host the fixture was made on the oracle takes every one of these decisions by itself -- CPU suite; another
    # CPU model / thread count moves a handful of pre-activations of ~1e-6 across zero: the replay pins them)
    assert tape.cursor == len(tape.masks)
    assert sum(m[1] for m in tape.mismatches) <= 64 and all(m[3] <= 1e-4 for m in tape.mismatches), tape.mismatches
    product = build_product_step(expt, PINNED_SIZE)
    product.real_first = False    # the reference's decisions are taped in ITS call order (G(z), D(real), D(fake))
    product.stack_d_passes = False
    with pinned_product_masks(tape.rewind()):
        hip = scenario.run_scenario(product, inputs, "cuda", full=True, **PINNED_KW)
    tape._skip_reserved()
    assert tape.cursor == len(tape.masks), "product and reference took different numbers of mask decisions"
    total = sum(m.numel() for m in tape.masks)
    flips = sum(m[1] for m in tape.mismatches)
    print(f"hologan: the product alone would decide {flips} of {total} mask entries differently "
          f"(largest |pre-activation| among them {max([m[3] for m in tape.mismatches], default=0.0):.1e})")
    assert flips <= max(4, 1e-4 * total) and all(m[3] <= 3e-4 for m in tape.mismatches), tape.mismatches
    cpu, hip, golden = (drop_exact_zero_gradients(d) for d in (cpu, hip, golden))
    assert set(hip) == set(cpu) == set(golden)
    worst = []
    for k, ref in cpu.items():
        got = np.asarray(hip[k], dtype=np.float64)
        ref = np.asarray(ref, dtype=np.float64)
        if np.asarray(cpu[k]).dtype.kind in "iu":
            assert np.array_equal(hip[k], cpu[k]) and np.array_equal(hip[k], golden[k]), k
            continue
        if ref.ndim == 0:
            e = abs(float(got) - float(ref)) / max(abs(float(ref)), scale)
        elif k.startswith("grad"):
            e = float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30))        # full relative L2
        else:
            e = float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30))
        worst.append((e, k))
    worst.sort(reverse=True)
    print(f"hologan: product vs pinned oracle, worst of {len(worst)}:", [(k, f"{e:.1e}") for e, k in worst[:4]])
    assert worst[0][0] <= TOL, worst[:4]
    summ = {k: (v if np.asarray(v).ndim == 0 or not k.startswith(("grad", "final/")) else
                scenario.summarize(torch.from_numpy(np.asarray(v)))) for k, v in hip.items()}
    if fixture == "tiny":
        summ = hip                  # the tiny fixture stores full tensors
    w = compare(summ, golden, TOL, f"hip hologan/{fixture}/pinned vs reference", atol_scale=scale)
    print(f"hologan: product vs reference fixture, worst {w[3]} {w[1]:.1e}")


@pytest.mark.parametrize("expt", ["dc_gan", "wgan", "wgan_gp", "hologan"])
def test_real_first_order_is_bit_identical_to_the_reference_order(expt):
    """Discriminator steps launch D(real) before G(z) (``BaseGAN.real_first``: under data parallelism the generator's
    last gradient bucket and optimizer step hide behind it).  D(real) does not read the generator, the host noise is
    drawn first either way and D's norm buffers still see real, then fake -- so losses, every parameter and every
    buffer after three optimizer cycles are BIT-identical to the reference's order (G(z) first)."""
    from helpers import synthetic_real
    from lightning_gan_zoo_amd.harness import Trainer
    res = {}
    for real_first in (True, False):
        cfg = make_cfg(expt, batch_size=8, features=8, noise_dim=16)
        torch.manual_seed(42)
        module = locate(cfg.model.lm["_target_"])(cfg, None).to("cuda")
        module.real_first = real_first
        module.stack_d_passes = False     # (the stacked discriminator pass has no D(real) / G(z) order to compare)
        trainer = Trainer(module)
        torch.manual_seed(7)
        np.random.seed(7)
        labels = torch.zeros(8, dtype=torch.int64, device="cuda")
        losses = [float(trainer.step((synthetic_real(8, seed=700 + k).cuda(), labels))[0])
                  for k in range(3 * len(trainer.order))]
        res[real_first] = (losses, {k: v.detach().clone() for k, v in module.state_dict().items()})
    (la, sa), (lb, sb) = res[True], res[False]
    assert la == lb, (la, lb)
    if expt == "hologan":
        return      # (its resampling adjoint is order-stable too, but keep the state comparison to the exact experiments)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
