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


def build_product_step(expt, size):
    feats, bs, zdim = scenario.sizes(expt, size)
    cfg = make_cfg(expt, batch_size=bs, features=feats, noise_dim=zdim)
    torch.manual_seed(42)
    return locate(cfg.model.lm["_target_"])(cfg, logging_dir=None)


LR = {"dc_gan": 2e-4, "wgan": 5e-5, "wgan_gp": 1e-4, "hologan": 1e-4}


@pytest.mark.parametrize("size", ["tiny", "full"])
@pytest.mark.parametrize("expt", scenario.ALL_EXPTS)
def test_product_matches_reference_fixture(expt, size):
    """Forward quantities at 1e-3 max-norm; gradient-side quantities at max(5e-3, 10 x the
    reference's own fp32-vs-fp64 discrepancy) in relative L2 -- see compare().  The plain 1e-3 bar
    on gradients is asserted on the stable-mask fixtures below."""
    inputs, golden, cond = load_golden(expt, size)
    step = build_product_step(expt, size)
    full = size == "tiny"
    out = scenario.run_scenario(step, inputs, "cuda", full=full, set_alpha=set_alpha,
                                shadow=build_oracle_step(expt, size))
    scale = float(np.abs(golden["probe/logits"]).max())
    worst = compare(out, golden, TOL, f"hip {expt}/{size}", atol_scale=scale, cond=cond,
                    final_abs=2 * 2 * LR[expt], grad_floor=5e-3)
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


@pytest.mark.parametrize("expt", ["dc_gan", "wgan_gp"])
def test_product_matches_stable_mask_fixture(expt):
    """Masks away from the threshold: every loss, logit and GRADIENT at the plain 1e-3 bar
    (cond <= 3e-4 everywhere in these fixtures, so the conditioning slack is at most 3e-3)."""
    inputs, golden, cond = load_golden(expt, "full", stable=True)
    step = build_product_step(expt, "full")
    out = scenario.run_scenario(step, inputs, "cuda", full=False, set_alpha=set_alpha, stable=True)
    scale = float(np.abs(golden["probe/logits"]).max())
    worst = compare(out, golden, TOL, f"hip {expt}/full/stable", atol_scale=scale, cond=cond)
    print(f"{expt}/full/stable: worst {worst[3]} err {worst[1]:.2e} (bar {worst[2]:.2e})")


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
