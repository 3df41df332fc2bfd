"""INTEGRATION.md route 1 as a test: the step class is built through ``dropin.install()`` from the reference's OWN
dotted names (``core.lightning_module.DCGAN`` ... as its conf/expt/*.yaml spell them) and driven by the hooks
``pl.Trainer.fit(model)`` calls (reference run_network.py:72; tests/fake_lightning.py lists them): data hooks
(reference core/lightning_module.py:89-102), the ``verbose_shape`` hook (:53-54), optimizer dicts, training /
validation steps.  The CPU leg goes as far as a machine without a GPU can (everything but the arithmetic, which must
fail loudly); the GPU leg runs the whole ``fit``."""
import os

import pytest
import torch

from fake_lightning import FakeTrainer, TinyImages
from lightning_gan_zoo_amd import config as C
from lightning_gan_zoo_amd import dropin

REF_CONF = "/root/reference/conf"
SMALL = ["train.features_gen=8", "train.features_disc=8", "model.noise_dim=16", "train.batch_size=4",
         "train.num_workers=0", "calc_fid=False"]


def _compose(expt, extra=()):
    """The reference's yaml tree when it exists (build container), else this package's restatement of it -- both
    spell the hot-path targets ``core.*`` here, and only ``dropin`` makes those names importable."""
    overrides = ["+expt=" + expt] + SMALL + list(extra)
    if os.path.isdir(REF_CONF):
        cfg = C.compose_tree(REF_CONF, overrides + ["filepaths=example"])
    else:
        dotted = {k: C.parse_value(v) for k, v in (o.split("=", 1) for o in SMALL + list(extra)) if k != "calc_fid"}
        cfg = C.make_cfg(expt, module_root="core", dotted=dotted)
    data = {"_target_": "fake_lightning.TinyImages", "root": "/nowhere", "n": 10}
    cfg["dataset"] = C.to_cfg({"n_channels": 3, "train": dict(data), "val": dict(data, n=6), "test": dict(data, n=3)})
    return cfg


@pytest.fixture
def installed():
    dropin.install()
    yield
    dropin.uninstall()


def _build(expt, extra=()):
    cfg = _compose(expt, extra)
    assert cfg.model.lm["_target_"].startswith("core.lightning_module.")
    torch.manual_seed(42)
    model = C.instantiate(cfg.model.lm, cfg, logging_dir=None)
    assert type(model).__module__ == "lightning_gan_zoo_amd.core.lightning_module"      # the HIP step class
    return model, cfg


def test_data_hooks_follow_the_reference(installed):
    """train / val / test loaders: the dataset node instantiated with ``transform=self.transform``, the configured
    batch size, no shuffling, incomplete last batch kept; the transform is Resize -> ToTensor -> Normalize."""
    model, cfg = _build("dc_gan")
    for hook, n in ((model.train_dataloader, 10), (model.val_dataloader, 6), (model.test_dataloader, 3)):
        loader = hook()
        assert isinstance(loader.dataset, TinyImages) and len(loader.dataset) == n
        assert loader.dataset.transform is model.transform and loader.batch_size == 4 and loader.num_workers == 0
        assert isinstance(loader.sampler, torch.utils.data.SequentialSampler) and not loader.drop_last
    batches = list(model.train_dataloader())
    assert [tuple(x.shape) for x, _ in batches] == [(4, 3, 64, 64), (4, 3, 64, 64), (2, 3, 64, 64)]
    assert [y.tolist() for _, y in batches] == [[0, 1, 2, 0], [1, 2, 0, 1], [2, 0]]
    x = batches[0][0]
    assert x.dtype == torch.float32 and -1.0 <= float(x.min()) < -0.9 and 0.9 < float(x.max()) <= 1.0
    again = next(iter(model.train_dataloader()))[0]
    assert torch.equal(again, x)                                     # same order every epoch


def test_verbose_shape_hook_prints_every_layer(installed, capsys):
    """``debug.verbose_shape=true`` (reference lightning_module.py:53-54, utils.py:13-27): every child module
    reports ``name: input shape --> output shape`` on forward.  Shown on a module tree that runs on the CPU
    (the hook is plain ``nn.Module`` machinery); the step class registers it at construction."""
    from core.utils.utils import VerboseShapeExecution, interpolate_sphere, init_weights    # noqa: F401 - surface
    net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Sequential(torch.nn.ReLU(), torch.nn.Linear(3, 2)))
    net.apply(VerboseShapeExecution)
    net(torch.zeros(5, 4))
    lines = capsys.readouterr().out.strip().splitlines()
    assert lines == ["0: torch.Size([5, 4]) --> torch.Size([5, 3])", "0: torch.Size([5, 3]) --> torch.Size([5, 3])",
                     "1: torch.Size([5, 3]) --> torch.Size([5, 2])", "1: torch.Size([5, 3]) --> torch.Size([5, 2])"]
    model, _ = _build("dc_gan", ["debug.verbose_shape=true"])
    hooked = [m for m in model.modules() if m._forward_hooks]
    assert model.generator in hooked and model.discriminator in hooked and len(hooked) > 10
    quiet, _ = _build("dc_gan")
    assert not [m for m in quiet.modules() if m._forward_hooks]
    # slerp keeps the end points and the norm of unit vectors
    a, b = torch.nn.functional.normalize(torch.randn(3, 8), dim=1), torch.nn.functional.normalize(torch.randn(3, 8), dim=1)
    assert torch.allclose(interpolate_sphere(a, b, 0.0), a, atol=1e-6) and torch.allclose(interpolate_sphere(a, b, 1.0), b, atol=1e-6)
    assert torch.allclose(interpolate_sphere(a, b, 0.3).norm(dim=1), torch.ones(3), atol=1e-5)


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU leg")
def test_fit_reaches_the_arithmetic_and_fails_loudly_without_a_gpu(installed):
    model, _ = _build("dc_gan")
    trainer = FakeTrainer(max_epochs=1, num_sanity_val_steps=0)
    with pytest.raises(RuntimeError, match="no CPU fallback|HIP|cuda"):
        trainer.fit(model)
    assert [c[0] for c in trainer.calls] == ["configure_optimizers", "train_dataloader", "val_dataloader", "training_step"]


@pytest.mark.gpu
@pytest.mark.parametrize("expt", ["dc_gan", "wgan", "wgan_gp", "hologan"])
def test_fit_through_dropin_on_the_gpu(installed, expt):
    """Two epochs of the fake ``Trainer.fit`` over the step class built from the reference's names: optimizer
    alternation by ``frequency`` (wgan 5:1, hologan 1:2), a loss with a graph from every ``training_step``, the
    validation hooks logging the 'Real' / 'Fake' grids, finite parameters afterwards."""
    model, cfg = _build(expt, ["train.num_epochs=4"] if expt == "hologan" else [])
    model.to("cuda")
    trainer = FakeTrainer(max_epochs=2).fit(model)
    names = [c[0] for c in trainer.calls]
    assert names[:3] == ["configure_optimizers", "train_dataloader", "val_dataloader"]
    assert names.count("training_step") == 6 and names.count("validation_epoch_end") == 3
    freqs = (cfg.optimisation.disc_freq, cfg.optimisation.gen_freq)
    order = [d[1] for n, d in trainer.calls if n == "training_step"]
    want = ([0] * freqs[0] + [1] * freqs[1]) * 6
    assert order == want[:6], (order, freqs)
    assert all(torch.isfinite(torch.tensor(trainer.losses)))
    tags = [t for t, _, _ in trainer.logger.experiment.images]
    assert tags == ["Real", "Fake"] * 3
    assert all(torch.isfinite(p).all() for p in model.parameters())
    assert all(p.requires_grad for p in model.parameters())          # untoggled after every step
