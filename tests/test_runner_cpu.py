"""Host logic of the thin runner on CPU (tests/cpu_harness.py supplies the oracle networks as the model): Hydra-style
override grammar, derived config values, epochs / LR schedule / stop conditions, Lightning's checkpoint envelope,
save / resume round trip, best-score bookkeeping, per-rank sharding."""
import os

import pytest
import torch

from cpu_harness import run_on_cpu
from lightning_gan_zoo_amd import run_network as R

SMALL = ["train.features_gen=8", "train.features_disc=8", "model.noise_dim=16", "log_every=1000"]


def test_override_grammar_and_derived_values():
    conf, expt, ov, run = R.parse_overrides(["+expt=wgan_gp", "train.batch_size=16", "train.features_gen=8",
                                             "train.features_disc=8", "model.noise_dim=16", "+max_steps=3",
                                             "optimisation.lr=1e-3"])
    assert conf is None and expt == "wgan_gp" and run["max_steps"] == 3
    cfg = R.compose(conf, expt, ov, run)
    assert cfg.generator.features_g == 8 and cfg.discriminator.features_d == 8 and cfg.train.batch_size == 16
    assert cfg.generator.channels_noise == 16 and cfg.discriminator.norm == "instance_norm2d"
    assert cfg.disc_optimiser.lr == 0.001 and cfg.gen_optimiser.lr == 0.001 and cfg.loss_weight.lambda_gp == 10
    conf, expt, ov, run = R.parse_overrides(["+expt=gan_stability_r1", "train.img_size=32", "model.noise_dim=24",
                                             "loss_weight.reg=3.5"])
    cfg = R.compose(conf, expt, ov, run)
    assert cfg.generator.size == 32 and cfg.discriminator.size == 32 and cfg.generator.z_dim == 24
    assert cfg.loss_weight.reg == 3.5 and cfg.model.lm["_target_"].endswith("GANStabilityR1")
    cfg = R.compose(*R.parse_overrides(["+expt=hologan", "train.num_epochs=10", "optimisation.beta1=0.5"])[:3])
    assert cfg.optimisation.lr_scheduler.total_epochs == 10 and cfg.disc_optimiser.betas == [0.5, 0.999]
    with pytest.raises(SystemExit):
        R.compose(None, "dc_gan", ["train.no_such_key=1"])
    with pytest.raises(SystemExit):
        R.parse_overrides(["train.batch_size=4"])          # +expt is mandatory
    # BASELINE config 1's literal command line parses (built-in tree): +expt=dc_gan dataset=celeb_a
    cfg = R.compose(*R.parse_overrides(["+expt=dc_gan", "dataset=celeb_a", "filepaths.celeb_a_root=/data/celeba"])[:3])
    assert cfg.dataset._target_ == "torchvision.datasets.ImageFolder" and cfg.dataset.train.root == "/data/celeba/train"


def test_product_command_line_cannot_reach_the_oracle_or_a_cpu_device():
    """oracle-vs-kernels is a TEST switch.  The runner has no module_root / device keys, its main()
    refuses to run without a GPU, and the real-data input step has no CPU arithmetic."""
    with pytest.raises(SystemExit):
        R.compose(*R.parse_overrides(["+expt=dc_gan", "module_root=oracle.reference_cpu"])[:3])
    with pytest.raises(SystemExit):
        R.compose(*R.parse_overrides(["+expt=dc_gan", "device=cpu"])[:3])
    if not torch.cuda.is_available():
        with pytest.raises(SystemExit, match="no CPU fallback"):
            R.main(["+expt=dc_gan", "+max_steps=1"])


def test_train_checkpoint_resume_roundtrip(tmp_path):
    ck = str(tmp_path / "ckpt")
    args = SMALL + ["train.batch_size=4", "train.ckpt_dir=" + ck]
    torch.set_num_threads(2)
    m1, t1, s1, _ = run_on_cpu("dc_gan", args + ["max_steps=4"])
    files = os.listdir(ck)
    assert files == ["step=4.ckpt"]
    blob = torch.load(os.path.join(ck, files[0]), weights_only=False)
    # Lightning's envelope (pl.Trainer.save_checkpoint of the 1.1 / 1.2 generation)
    assert {"epoch", "global_step", "pytorch-lightning_version", "state_dict", "optimizer_states", "lr_schedulers",
            "callbacks"} <= set(blob)
    assert blob["global_step"] == 4 and blob["callbacks"]["ModelCheckpoint"]["monitor"] == "fid"
    keys = set(blob["state_dict"])
    assert "generator.net.block1.transpose_conv.weight" in keys            # Lightning-style names
    assert "discriminator.disc.block3.batch_norm.running_var" in keys
    assert len(blob["optimizer_states"]) == 2 and "param_groups" in blob["optimizer_states"][0]
    # resume: picks the single *.ckpt, continues the optimizer alternation where it stopped
    m2, t2, s2, _ = run_on_cpu("dc_gan", args + ["max_steps=6"])
    assert s2 == 6 and os.listdir(ck) == ["step=6.ckpt"]
    assert t2.optim[0]["optimizer"].state_dict()["state"][0]["step"] >= 3  # Adam state was restored, then advanced
    open(os.path.join(ck, "second.ckpt"), "w").write("x")
    with pytest.raises(AssertionError, match="Multiple ckpts"):            # reference run_network.py:21
        R.find_ckpt(ck)


def test_epochs_scheduler_and_stop_at_num_epochs(tmp_path):
    """An epoch is ceil(len(dataset) / batch) steps (synthetic: steps_per_epoch); the LR schedulers step per epoch and
    training stops at train.num_epochs -- HoloGAN's LambdaLR reaches 0 there and would turn negative beyond."""
    torch.set_num_threads(2)
    m, tr, steps, cfg = run_on_cpu("hologan", ["train.features_gen=8", "train.features_disc=8", "log_every=1000",
                                               "train.batch_size=2", "train.num_epochs=4", "steps_per_epoch=3",
                                               "model.noise_dim=16"])
    assert steps == 12                                             # 4 epochs x 3 steps, no max_steps given
    lrs = [o["optimizer"].param_groups[0]["lr"] for o in tr.optim]
    assert all(lr >= 0 for lr in lrs) and lrs[0] == pytest.approx(0.0)      # 1 - (4 - 2) / 2


def test_best_score_bookkeeping_follows_model_checkpoint(tmp_path):
    """ModelCheckpoint(monitor='fid', filename='model_best-{fid:.2f}'), mode min, save_top_k 1."""
    ck = str(tmp_path / "ckpt")
    scores = iter([31.237, 40.0, 12.5])
    torch.set_num_threads(2)
    seen = []

    def evaluate(module, epoch):
        s = next(scores)
        seen.append(sorted(os.listdir(ck)) if os.path.isdir(ck) else [])
        return {"fid": s}

    run_on_cpu("dc_gan", SMALL + ["train.batch_size=2", "train.ckpt_dir=" + ck, "steps_per_epoch=2", "max_steps=6"],
               evaluate=evaluate)
    assert seen == [[], ["model_best-fid=31.24.ckpt"], ["model_best-fid=31.24.ckpt"]]      # 40.0 did not replace it
    assert os.listdir(ck) == ["model_best-fid=12.50.ckpt"]
    st = torch.load(os.path.join(ck, "model_best-fid=12.50.ckpt"), weights_only=False)["callbacks"]["ModelCheckpoint"]
    assert st["best_model_score"] == 12.5 and st["best_model_path"].endswith("model_best-fid=12.50.ckpt")


def test_shard_indices_match_distributed_sampler():
    """Unshuffled and -- what Lightning's DDP actually runs: ``auto_add_sampler(shuffle=True)`` + ``set_epoch`` --
    reshuffled per epoch, against torch's own sampler."""
    from torch.utils.data.distributed import DistributedSampler
    for n, world in ((10, 4), (7, 2), (8, 8), (3, 4), (5, 1)):
        for rank in range(world):
            ref = list(DistributedSampler(range(n), num_replicas=world, rank=rank, shuffle=False))
            assert R.shard_indices(n, rank, world) == ref, (n, world, rank)
            if world > 1:
                sampler = DistributedSampler(range(n), num_replicas=world, rank=rank, shuffle=True, seed=0)
                for epoch in range(3):
                    sampler.set_epoch(epoch)
                    assert R.shard_indices(n, rank, world, epoch) == list(sampler), (n, world, rank, epoch)


def test_image_folder_reshuffles_per_epoch_under_data_parallelism(tmp_path):
    from test_input_step import make_folder
    from lightning_gan_zoo_amd.run_network import ImageFolderImages
    root = str(tmp_path / "imgs")
    make_folder(root)                                   # 6 images
    labels = {}
    for rank in range(2):
        it = ImageFolderImages(root, 3, 8, 3, 0.5, 0.5, "cpu", rank=rank, world=2).host_batches()
        labels[rank] = [next(it)[1].tolist() for _ in range(3)]         # three epochs of one batch each
    full = sorted(c for _, c in R.image_folder_samples(root)[0])
    for epoch in range(3):
        assert sorted(labels[0][epoch] + labels[1][epoch]) == full      # the two ranks partition every epoch
    assert len({tuple(labels[0][e]) for e in range(3)}) > 1             # ... in a different order each time
    one = ImageFolderImages(root, 6, 8, 3, 0.5, 0.5, "cpu").host_batches()
    assert next(one)[1].tolist() == next(one)[1].tolist() == [c for _, c in R.image_folder_samples(root)[0]]


def test_resumed_run_continues_the_sampler_at_the_checkpoints_epoch(tmp_path):
    """advisor finding (round 3): Lightning calls ``sampler.set_epoch(trainer.current_epoch)``, so a DDP run resumed
    from a checkpoint of epoch E draws ``randperm(seed + E)`` next -- not ``seed + 0`` again.  ``fit`` hands the
    checkpoint's epoch to the data object; the order the resumed rank then consumes is shard_indices(n, r, w, E)."""
    from test_input_step import make_folder
    from lightning_gan_zoo_amd.run_network import ImageFolderImages
    root = str(tmp_path / "imgs")
    make_folder(root)                                   # 6 images, 2 ranks -> 3 per rank and epoch
    samples = R.image_folder_samples(root)[0]
    for rank in range(2):
        f = ImageFolderImages(root, 3, 8, 3, 0.5, 0.5, "cpu", rank=rank, world=2)
        f.set_epoch(2)
        it = f.host_batches()
        for epoch in (2, 3):
            want = [samples[j][1] for j in R.shard_indices(len(samples), rank, 2, epoch)]
            assert next(it)[1].tolist() == want, (rank, epoch)
    # through fit(): a run of 2 epochs, then a resumed one -- the data object is told to start at epoch 2
    seen = []

    class Recorder(ImageFolderImages):
        def set_epoch(self, epoch):
            seen.append(epoch)
            super().set_epoch(epoch)

    from cpu_harness import HostNormalisedFolder
    ck = str(tmp_path / "ck")
    torch.set_num_threads(2)
    args = SMALL + ["train.batch_size=3", "train.ckpt_dir=" + ck, "train.img_size=8"]
    data = lambda: HostNormalisedFolder(Recorder(root, 3, 8, 3, 0.5, 0.5, "cpu", rank=0, world=2))  # noqa: E731
    run_on_cpu("dc_gan", args + ["max_steps=2"], data=data())          # one batch per epoch: two epochs
    assert seen == []                                   # a fresh run starts at epoch 0 by itself
    run_on_cpu("dc_gan", args + ["max_steps=3"], data=data())
    assert seen == [2]                                  # resumed: told the checkpoint's epoch


def test_a_run_cut_short_keeps_the_best_checkpoint(tmp_path):
    """advisor finding (round 2): ``+max_steps`` ending mid-epoch wrote step=N.ckpt and swept model_best-fid=X.ckpt
    away.  ModelCheckpoint(save_top_k=1) never removes its best file for a state without a better metric."""
    ck = str(tmp_path / "ck")
    torch.set_num_threads(2)
    run_on_cpu("dc_gan", SMALL + ["train.batch_size=2", "train.ckpt_dir=" + ck, "steps_per_epoch=2", "max_steps=3"],
               evaluate=lambda module, epoch: {"fid": 20.0})
    assert os.listdir(ck) == ["model_best-fid=20.00.ckpt"]              # the tail save at step 3 did not replace it
    blob = torch.load(os.path.join(ck, "model_best-fid=20.00.ckpt"), weights_only=False)
    assert blob["global_step"] == 2 and blob["callbacks"]["ModelCheckpoint"]["best_model_score"] == 20.0


def test_real_activations_native_resolution_and_cache(tmp_path):
    """advisor finding (round 2): the callback's real statistics come from the val images at their NATIVE size
    (pytorch_fid: ToTensor only) in batches of 16 equal-sized files, are cached as <val_root>/inception_cache.npz
    (mu / sigma / act) and an existing *.npz is used instead of recomputing."""
    import numpy as np
    from PIL import Image
    root = str(tmp_path / "val")
    os.makedirs(os.path.join(root, "a"))
    rng = np.random.RandomState(0)
    sizes = [(20, 30)] * 18 + [(24, 24)] * 3 + [(20, 30)]                 # 18 equal (-> 16 + 2), 3 of another size, 1
    for i, (h, w) in enumerate(sizes):
        Image.fromarray(rng.randint(0, 256, size=(h, w, 3), dtype=np.uint8)).save(os.path.join(root, "a", "%03d.png" % i))
    open(os.path.join(root, "a", "notes.txt"), "w").write("x")
    seen = []

    def features(u8):
        seen.append(u8.shape)
        return u8.reshape(len(u8), -1).astype(np.float64)[:, :5]

    act = R.real_activations(root, features)
    assert seen == [(16, 20, 30, 3), (2, 20, 30, 3), (3, 24, 24, 3), (1, 20, 30, 3)] and act.shape == (22, 5)
    with np.load(os.path.join(root, "inception_cache.npz")) as d:
        assert set(d.files) == {"mu", "sigma", "act"} and np.array_equal(d["act"], act) and d["sigma"].shape == (5, 5)
    seen.clear()
    again = R.real_activations(root, features)                            # second launch / resume: from the cache
    assert not seen and np.array_equal(again, act)


def test_runner_trains_from_an_image_folder(tmp_path):
    """`dataset=image_folder`: the reference's ImageFolder -> Resize -> ToTensor -> Normalize input step feeding
    the step classes (here the CPU oracle's, so no GPU is needed)."""
    from test_input_step import make_folder
    root = str(tmp_path / "imgs")
    make_folder(root)
    torch.set_num_threads(2)
    module, trainer, step, cfg = run_on_cpu("dc_gan", SMALL + ["dataset=image_folder", "dataset_path=" + root,
                                                               "train.batch_size=3", "max_steps=4"])
    assert step == 4 and all(torch.isfinite(p).all() for p in module.parameters())
    # 6 images, batch 3 -> 2 steps per epoch -> the 4 steps were 2 epochs
    assert trainer.optim[0]["lr_scheduler"].last_epoch == 2


def test_accumulate_grad_batches_reaches_the_trainer_in_both_forms():
    """reference run_network.py:61-68: ``accumulate_grad_batches`` is the int 1 (conf/config.yaml:57) or a ``{start_epoch,
    accumulation_factor}`` node (conf/machine/*.yaml) that becomes Lightning's ``{start_epoch: factor}`` scheduler.  Both
    forms go through the runner's command line into harness.Trainer, and change the trajectory as they should: factor 2
    from epoch 0 differs from no accumulation from the first epoch on, factor 2 from epoch 1 equals no accumulation for
    the whole first epoch."""
    torch.set_num_threads(2)
    base = SMALL + ["train.batch_size=2", "steps_per_epoch=4"]

    def params(extra, steps):
        module, trainer, step, cfg = run_on_cpu("dc_gan", base + ["max_steps=%d" % steps] + extra)
        assert step == steps
        return torch.cat([p.detach().reshape(-1) for p in module.parameters()]), trainer

    plain4, _ = params([], 4)
    plain8, tr = params([], 8)
    assert tr.accumulate_grad_batches == 1
    acc4, tr = params(["accumulate_grad_batches=2"], 4)
    assert tr.accumulate_grad_batches == 2 and not torch.equal(acc4, plain4)
    late4, tr = params(["accumulate_grad_batches={start_epoch: 1, accumulation_factor: 2}"], 4)
    assert tr.accumulate_grad_batches == {1: 2} and torch.equal(late4, plain4)       # epoch 0: factor 1
    late8, _ = params(["accumulate_grad_batches={start_epoch: 1, accumulation_factor: 2}"], 8)
    assert not torch.equal(late8, plain8)                                            # epoch 1: factor 2
