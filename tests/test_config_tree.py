"""The drop-in boundary on the host side: composing a ``conf/``-shaped tree (the reference's own, when present),
resolving its ``core.*`` ``_target_`` strings to the HIP modules, instantiating through them."""
import os
import subprocess
import sys
import textwrap

import pytest
import torch

from lightning_gan_zoo_amd import config as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CONF = "/root/reference/conf"
needs_reference = pytest.mark.skipif(not os.path.isdir(REF_CONF), reason="reference tree not present (build container only)")


def _strip(d, keys):
    return {k: v for k, v in d.items() if k not in keys}


@needs_reference
@pytest.mark.parametrize("expt", ["dc_gan", "wgan", "wgan_gp", "hologan", "gan_stability_r1"])
def test_reference_tree_composes_to_the_builtin_config(expt):
    """compose_tree over the reference's yaml == config.make_cfg's restatement, on every key make_cfg produces.
    Known, documented differences: precision (reference default 16 = AMP; the parity target is fp32), ``loss_weight``
    (the literal string MISSING where an experiment sets none), and the root config's ``img_size`` / ``final_sigmoid``
    keys inside the HoloGAN / ResNet network nodes (which the reference's own classes reject)."""
    tree = C.compose_tree(REF_CONF, ["+expt=" + expt, "filepaths=example"])
    built = C.make_cfg(expt, module_root="core")
    for key, want in built.items():
        got = tree[key]
        if key == "precision":
            assert got == 16 and want == 32
        elif key == "loss_weight" and want == {}:
            assert got == "MISSING"
        elif key in ("discriminator", "generator") and expt in ("hologan", "gan_stability_r1"):
            assert _strip(got, ("img_size", "final_sigmoid")) == _strip(want, ("img_size", "final_sigmoid")), key
        else:
            assert got == want, (key, got, want)
    # the rest of the tree is there too
    assert tree.dataset._target_ == "torchvision.datasets.ImageFolder" and tree.dataset.train.root.endswith("/train")
    assert tree.val.inception_stats_filepath.endswith("/val_inception_stats.pkl") and tree.calc_fid is True
    assert "sample_grid" in tree.figures and tree.figure_details.img_size == tree.train.img_size


@needs_reference
def test_reference_tree_overrides():
    # BASELINE config 1's literal command line (+ the filepaths file the reference asks the user to create)
    t = C.compose_tree(REF_CONF, ["+expt=dc_gan", "dataset=celeb_a", "filepaths=example"])
    assert t.name == "dc_gan" and t.train.channels_img == 3
    t = C.compose_tree(REF_CONF, ["+expt=dc_gan", "dataset=mnist", "filepaths=example", "train.batch_size=64",
                                  "optimisation.lr=1e-3", "+machine=big", "+train.weight_clip=0.5", "~version"])
    assert t.train.channels_img == 1 and t.discriminator.channels_img == 1 and t.generator.channels_img == 1
    assert t.train.batch_size == 64 and t.disc_optimiser.lr == 0.001 and t.gen_optimiser.betas == [0.5, 0.999]
    assert t.num_gpus == 8 and t.train.img_size == 128 and t.generator.img_size == 128        # conf/machine/big.yaml
    assert t.train.weight_clip == 0.5 and "version" not in t
    # nested ``override /group`` directives (conf/expt/hologan.yaml:55-58)
    h = C.compose_tree(REF_CONF, ["+expt=hologan", "filepaths=example"])
    assert h.model.noise_distn._target_.endswith("Uniform") and h.optimisation.lr_scheduler.total_epochs == 25
    with pytest.raises(C.ConfigError):
        C.compose_tree(REF_CONF, ["+expt=dc_gan", "filepaths=example", "train.no_such=1"])      # needs a "+"
    with pytest.raises(C.ConfigError):
        C.compose_tree(REF_CONF, ["+expt=dc_gan"])       # conf/filepaths/local.yaml is for the user to create


def test_compose_tree_on_a_synthetic_tree(tmp_path):
    """The same grammar on a tree written here (runs on the GPU box too, where the reference is absent)."""
    conf = tmp_path / "conf"
    (conf / "expt").mkdir(parents=True)
    (conf / "opt").mkdir()
    (conf / "figs").mkdir()
    (conf / "config.yaml").write_text(textwrap.dedent("""\
        # @package _global_
        name: MISSING
        train:
          batch_size: 128
          lr: 2e-4
          decay: 1.
        net:
          width: "${train.width}"
          tag: "w${train.width}-${name}"
        opt_d: "${opt}"
        defaults:
          - opt: adam
          - override hydra/job_logging: disabled
        """))
    (conf / "opt" / "adam.yaml").write_text("_target_: torch.optim.Adam\nlr: \"${train.lr}\"\nbetas: [\"${train.b1}\", 0.999]\n")
    (conf / "opt" / "rms.yaml").write_text("_target_: torch.optim.RMSprop\nlr: \"${train.lr}\"\n")
    (conf / "figs" / "grid.yaml").write_text("_target_: x.Grid\nncol: 3\n")
    (conf / "expt" / "a.yaml").write_text(textwrap.dedent("""\
        # @package _global_
        name: a
        train:
          width: 8
          b1: 0.5
        defaults:
          - /figs@figures.grid: grid
          - override /opt: rms
        """))
    cfg = C.compose_tree(str(conf), ["+expt=a", "train.batch_size=4"])
    assert cfg.name == "a" and cfg.train == {"batch_size": 4, "lr": 0.0002, "decay": 1.0, "width": 8, "b1": 0.5}
    assert isinstance(cfg.train.lr, float) and cfg.net == {"width": 8, "tag": "w8-a"}
    assert cfg.opt._target_ == "torch.optim.RMSprop" and cfg.opt_d == cfg.opt and cfg.opt.lr == 0.0002
    assert cfg.figures.grid == {"_target_": "x.Grid", "ncol": 3} and "hydra" not in cfg and "defaults" not in cfg
    cfg = C.compose_tree(str(conf), ["+expt=a", "opt=adam"])                 # the command line beats the nested override
    assert cfg.opt.betas == [0.5, 0.999]
    opt = C.instantiate(cfg.opt, [torch.nn.Parameter(torch.zeros(2))])
    assert isinstance(opt, torch.optim.Adam) and opt.param_groups[0]["lr"] == 0.0002


def _run(code, extra_path=()):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT] + list(extra_path)))
    return subprocess.run([sys.executable, "-c", textwrap.dedent(code)], capture_output=True, text=True, env=env, cwd="/tmp")


def test_dropin_alias_without_a_reference_tree():
    r = _run("""
        import lightning_gan_zoo_amd.dropin as d
        d.install()
        import core.lightning_module, core.models.standard_networks as m, core.utils.utils, core.utils.hologan
        import core.models.hologan_generator, core.models.hologan_discriminator
        import core.submodules.gan_stability.models.resnet
        import lightning_gan_zoo_amd.core.models.standard_networks as p
        assert m is p and core.lightning_module.DCGAN.__module__ == "lightning_gan_zoo_amd.core.lightning_module"
        from lightning_gan_zoo_amd.config import locate
        assert locate("core.models.standard_networks.Generator") is p.Generator
        """)
    assert r.returncode == 0, r.stderr


@needs_reference
def test_reference_targets_instantiate_the_hip_modules():
    """The reference's own yaml, its own ``_target_`` strings, this package's classes: what a maintainer gets by
    adding ``lightning_gan_zoo_amd.dropin.install()`` to run_network.py.  Modules outside the hot path still come
    from the reference's ``core`` package."""
    r = _run("""
        import sys, torch
        import lightning_gan_zoo_amd.dropin as d
        d.install()
        from lightning_gan_zoo_amd import config as C
        import core.utils.coordconv as cc                   # reference-only module, untouched
        assert cc.__file__.startswith("/root/reference/"), cc.__file__
        for expt in ("dc_gan", "wgan", "wgan_gp", "hologan", "gan_stability_r1"):
            cfg = C.compose_tree("/root/reference/conf", ["+expt=" + expt, "filepaths=example", "train.features_gen=8",
                                                           "train.features_disc=8"] +
                                 (["generator.in_planes=8", "discriminator.out_planes=8"] if expt == "hologan" else []) +
                                 (["generator.nfilter=4", "discriminator.nfilter=4", "train.img_size=32"]
                                  if expt == "gan_stability_r1" else []))
            torch.manual_seed(42)
            step = C.instantiate(cfg.model.lm, cfg, logging_dir=None)
            assert type(step).__module__ == "lightning_gan_zoo_amd.core.lightning_module", type(step)
            assert type(step.generator).__module__.startswith("lightning_gan_zoo_amd.core."), type(step.generator)
            assert type(step.discriminator).__module__.startswith("lightning_gan_zoo_amd.core.")
            d_opt, g_opt = step.configure_optimizers()
            assert d_opt["frequency"] == cfg.optimisation.disc_freq and "lr_scheduler" in g_opt
        print("ok")
        """, extra_path=["/root/reference"])
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
