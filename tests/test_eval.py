"""Evaluation path: FID / KID arithmetic against values produced by the reference's own functions
(tests/golden/make_eval_golden.py) and, on the GPU, the eval-mode sample dump against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN_DIR


def activations(seed, n, d, shift):          # same construction as tests/golden/make_eval_golden.py
    rng = np.random.RandomState(seed)
    basis = rng.randn(d, d) / np.sqrt(d)
    return rng.randn(n, d).dot(basis) + shift * rng.rand(d)


def test_fid_kid_arithmetic_matches_reference_values():
    from lightning_gan_zoo_amd import eval as E
    gold = np.load(os.path.join(GOLDEN_DIR, "eval_metrics.npz"))
    real, fake = activations(1, 400, 48, 0.0), activations(2, 360, 48, 0.3)
    fid = E.frechet_distance(*E.activation_statistics(real), *E.activation_statistics(fake))
    assert abs(fid - float(gold["fid"])) <= 1e-9 * abs(float(gold["fid"]))
    assert abs(E.frechet_distance(*E.activation_statistics(real), *E.activation_statistics(real))) < 1e-9
    np.random.seed(123)
    mmds, variances = E.polynomial_mmd_averages(real, fake, n_subsets=7, subset_size=150)
    assert np.allclose(mmds, gold["kid_mmds"], rtol=1e-10, atol=0)
    assert np.allclose(variances, gold["kid_vars"], rtol=1e-8, atol=0)
    full = np.array(E.polynomial_mmd(fake[:300], real[:300]))
    assert np.allclose(full, gold["mmd_full"], rtol=1e-9, atol=0)


@pytest.mark.gpu
def test_sample_dump_matches_oracle_generator():
    """Fixed latents from the host generator, eval-mode generator on the HIP path, uint8 images as the callback
    writes them: against the CPU oracle with the same parameters (at most one grey level apart at rounding ties)."""
    from lightning_gan_zoo_amd import eval as E
    from lightning_gan_zoo_amd.config import locate, make_cfg
    steps = {}
    for name, root, dev in (("hip", None, "cuda"), ("cpu", "oracle.reference_cpu", "cpu")):
        cfg = make_cfg("dc_gan", **({"module_root": root} if root else {}), batch_size=8, features=8, noise_dim=16)
        torch.manual_seed(42)
        steps[name] = locate(cfg.model.lm["_target_"])(cfg, None).to(dev)
    torch.manual_seed(5)
    dump = E.SampleDump(steps["hip"], n_samples=20, batch_size=8)
    assert [len(z) for z in dump.z_samples] == [8, 8, 4]
    hip = np.concatenate(list(dump.images(steps["hip"])))
    assert hip.shape == (20, 64, 64, 3) and hip.dtype == np.uint8 and steps["hip"].training
    cpu_step = steps["cpu"]
    cpu_step.eval()
    with torch.no_grad():
        ref = torch.cat([cpu_step.generator(z) for z in dump.z_samples])
    ref = (torch.clamp(ref, 0, 1).permute(0, 2, 3, 1).numpy() * 255).astype(int)
    assert np.abs(hip.astype(int) - ref).max() <= 1 and (hip.astype(int) != ref).mean() < 1e-3
    out = E.evaluate(steps["hip"], dump, lambda img: img.reshape(len(img), -1)[:, ::512].astype(np.float64),
                     np.random.RandomState(0).rand(30, 24) * 255, n_subsets=3)
    assert np.isfinite(out["fid"]) and np.isfinite(out["kid"])


def test_make_grid_layout_and_normalisation():
    """torchvision.utils.make_grid(normalize=True) semantics used by validation_epoch_end (reference
    core/lightning_module.py:64-73): global min/max normalisation, nrow 8, 2-pixel padding of zeros."""
    from lightning_gan_zoo_amd.eval import make_grid
    g = torch.Generator().manual_seed(0)
    x = torch.randn(11, 3, 5, 4, generator=g) * 3 + 1
    grid = make_grid(x, normalize=True)
    assert grid.shape == (3, 2 * 7 + 2, 8 * 6 + 2)
    lo, hi = x.min(), x.max()
    want = (x - lo) / (hi - lo)
    assert torch.allclose(grid[:, 2:7, 2:6], want[0], atol=1e-6)
    assert torch.allclose(grid[:, 9:14, 2 + 2 * 6:2 + 2 * 6 + 4], want[10], atol=1e-6)      # 11th image: row 1, column 2
    assert float(grid[:, :2].abs().max()) == 0 and float(grid[:, 9:14, 20:].abs().max()) == 0   # padding / empty cells
    assert float(grid.min()) == 0.0 and abs(float(grid.max()) - 1.0) < 1e-6
    mono = make_grid(torch.rand(2, 1, 4, 4, generator=g))
    assert mono.shape == (3, 8, 14)
    assert make_grid(torch.rand(1, 3, 4, 4, generator=g)).shape == (3, 4, 4)


@pytest.mark.gpu
def test_validation_epoch_end_grids():
    from lightning_gan_zoo_amd.config import locate, make_cfg
    cfg = make_cfg("dc_gan", batch_size=4, features=8, noise_dim=16)
    torch.manual_seed(42)
    step = locate(cfg.model.lm["_target_"])(cfg, None).cuda()
    step.eval()
    logged = []

    class Exp:
        def add_image(self, name, img, epoch):
            logged.append((name, tuple(img.shape), epoch))

    step.logger = type("L", (), {"experiment": Exp()})()
    real = torch.rand(16, 3, 64, 64, device="cuda") * 2 - 1
    gr, gf = step.validation_epoch_end([step.validation_step((real, None), 0)])
    assert gr.shape == gf.shape == (3, 68, 8 * 66 + 2)
    assert logged == [("Real", (3, 68, 530), 0), ("Fake", (3, 68, 530), 0)]
    assert 0.0 <= float(gf.min()) and float(gf.max()) <= 1.0 + 1e-6
