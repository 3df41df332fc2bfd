"""Host logic of the thin runner on CPU (using the oracle networks as the model): Hydra-style override
grammar, derived config values, Lightning-style checkpoint keys, save / resume round trip."""
import os

import pytest
import torch

from lightning_gan_zoo_amd import run_network as R


def test_override_grammar_and_derived_values():
    expt, ov, run = R.parse_overrides(["+expt=wgan_gp", "train.batch_size=16", "train.features_gen=8",
                                       "train.features_disc=8", "model.noise_dim=16", "max_steps=3",
                                       "optimisation.lr=0.001"])
    assert expt == "wgan_gp" and run["max_steps"] == 3 and ov["train.batch_size"] == 16
    cfg = R.compose(expt, ov)
    assert cfg.generator.features_g == 8 and cfg.discriminator.features_d == 8
    assert cfg.generator.channels_noise == 16 and cfg.discriminator.norm == "instance_norm2d"
    assert cfg.disc_optimiser.lr == 0.001 and cfg.loss_weight.lambda_gp == 10
    expt, ov, _ = R.parse_overrides(["+expt=gan_stability_r1", "train.img_size=32", "model.noise_dim=24",
                                     "loss_weight.reg=3.5"])
    cfg = R.compose(expt, ov)
    assert cfg.generator.size == 32 and cfg.discriminator.size == 32 and cfg.generator.z_dim == 24
    assert cfg.loss_weight.reg == 3.5 and cfg.model.lm["_target_"].endswith("GANStabilityR1")
    with pytest.raises(SystemExit):
        R.compose("dc_gan", {"train.no_such_key": 1})
    with pytest.raises(SystemExit):
        R.parse_overrides(["train.batch_size=4"])          # +expt is mandatory


def test_train_checkpoint_resume_roundtrip(tmp_path):
    ck = str(tmp_path / "ckpt")
    args = ["+expt=dc_gan", "module_root=oracle.reference_cpu", "device=cpu", "train.batch_size=4",
            "train.features_gen=8", "train.features_disc=8", "model.noise_dim=16", "train.ckpt_dir=" + ck,
            "log_every=1000"]
    torch.set_num_threads(2)
    m1, t1, s1 = R.main(args + ["max_steps=4"])
    files = os.listdir(ck)
    assert files == ["step=4.ckpt"]
    blob = torch.load(os.path.join(ck, files[0]), weights_only=False)
    keys = set(blob["state_dict"])
    assert "generator.net.block1.transpose_conv.weight" in keys            # Lightning-style names
    assert "discriminator.disc.block3.batch_norm.running_var" in keys
    # resume: picks the single *.ckpt, continues the optimizer alternation where it stopped
    m2, t2, s2 = R.main(args + ["max_steps=6"])
    assert s2 == 6 and os.listdir(ck) == ["step=6.ckpt"]
    assert t2.optim[0]["optimizer"].state_dict()["state"][0]["step"] >= 3  # Adam state was restored, then advanced


def test_runner_trains_from_an_image_folder(tmp_path):
    """`dataset=image_folder`: the reference's ImageFolder -> Resize -> ToTensor -> Normalize input step feeding
    the step classes (here the CPU oracle's, so no GPU is needed)."""
    from test_input_step import make_folder
    root = str(tmp_path / "imgs")
    make_folder(root)
    torch.set_num_threads(2)
    module, trainer, step = R.main(["+expt=dc_gan", "module_root=oracle.reference_cpu", "device=cpu",
                                    "dataset=image_folder", "dataset_path=" + root, "train.batch_size=3",
                                    "train.features_gen=8", "train.features_disc=8", "model.noise_dim=16",
                                    "max_steps=4", "log_every=1000"])
    assert step == 4 and all(torch.isfinite(p).all() for p in module.parameters())
