"""Thin runner with the reference's command-line grammar (run_network.py:25-75 + Hydra overrides):

    python -m lightning_gan_zoo_amd.run_network +expt=dc_gan train.batch_size=64 +max_steps=200 \\
           dataset=synthetic train.ckpt_dir=output/ckpt
    python -m lightning_gan_zoo_amd.run_network --config-dir /path/to/lightning_gan_zoo/conf \\
           +expt=dc_gan dataset=celeb_a filepaths=local

Without ``--config-dir`` the experiment is composed from the built-in restatement of the reference's tree
(config.make_cfg); with it, from the user's own ``conf/``-shaped directory (config.compose_tree: defaults lists,
``+expt=``, ``group=option``, ``${a.b}``), whose ``core.*`` ``_target_`` strings are served by this package
(dropin.install).  ``num_gpus`` > 1 expects a torch.distributed launch (one process per GPU).

What the reference delegates to pytorch_lightning is done by harness.Trainer: per-batch optimizer alternation, toggle,
backward, step, per-epoch LR scheduler, ``max_epochs=cfg.train.num_epochs``, checkpoint save / resume
(``find_ckpt``: the single ``*.ckpt`` under ``train.ckpt_dir``; ``ModelCheckpoint(monitor='fid',
filename='model_best-{fid:.2f}')`` keeps the best-scoring file when the metric is available, else the latest), the
DistributedSampler's per-rank sharding of the dataset.  Checkpoints are ``torch.save`` dicts in Lightning's envelope
(``epoch, global_step, pytorch-lightning_version, state_dict, optimizer_states, lr_schedulers, callbacks``) whose
``state_dict`` keys are ``generator.<...>`` / ``discriminator.<...>`` exactly as the reference's LightningModule
produces them.

The product runner only ever builds the HIP modules on a GPU; tests drive ``fit()`` with their own module / data.
"""
import glob
import math
import os
import random
import sys
import time

import numpy as np
import torch

from .config import ConfigError, compose_tree, locate, make_cfg, parse_value

# keys of this runner, not of the reference's config (accepted with or without Hydra's "+")
RUNNER_KEYS = {"max_steps": None, "log_every": 10, "dataset_path": None, "seed": 42, "steps_per_epoch": 100,
               "inception_weights": None, "inception_check_hash": True}
BUILTIN_DATASETS = ("synthetic", "image_folder", "tensor_file", "celeb_a")
LIGHTNING_VERSION_TAG = "1.2.0"       # envelope layout written below (Lightning 1.1 / 1.2 generation)


def parse_overrides(argv):
    """-> (conf_dir or None, expt, [override strings for the composer], runner keys)."""
    argv = list(argv)
    conf_dir, rest, runner = None, [], dict(RUNNER_KEYS)
    i = 0
    while i < len(argv):
        arg = argv[i]
        if arg in ("--config-dir", "-cd", "--config-path", "-cp"):
            conf_dir = argv[i + 1]
            i += 2
            continue
        if arg.startswith(("--config-dir=", "--config-path=")):
            conf_dir = arg.split("=", 1)[1]
            i += 1
            continue
        i += 1
        if "=" not in arg and not arg.startswith("~"):
            raise SystemExit("cannot parse %r (expected key=value or +expt=name)" % arg)
        key = arg.split("=", 1)[0].lstrip("+")
        if key in runner:
            runner[key] = parse_value(arg.split("=", 1)[1])
        else:
            rest.append(arg)
    expt = next((parse_value(a.split("=", 1)[1]) for a in rest if a.split("=", 1)[0] in ("+expt", "expt")), None)
    if expt is None:
        raise SystemExit("missing +expt=<dc_gan|wgan|wgan_gp|hologan|gan_stability_r1>")
    return conf_dir, expt, rest, runner


def _builtin_dataset(name, run, filepaths):
    if name == "synthetic":
        return {"_target_": "lightning_gan_zoo_amd.run_network.SyntheticImages", "n_channels": 3}
    if name == "tensor_file":
        return {"_target_": "lightning_gan_zoo_amd.run_network.TensorFileImages", "n_channels": 3,
                "root": run["dataset_path"], "train": {"root": run["dataset_path"]}}
    if name == "image_folder":
        root = run["dataset_path"]
        return {"_target_": "torchvision.datasets.ImageFolder", "n_channels": 3, "root": root, "train": {"root": root}}
    if name == "celeb_a":                 # conf/dataset/celeb_a.yaml:1-12
        root = filepaths.get("celeb_a_root")
        if not root:
            raise SystemExit("dataset=celeb_a needs filepaths.celeb_a_root=<dir> (conf/filepaths/example.yaml)")
        return {"_target_": "torchvision.datasets.ImageFolder", "n_channels": 3, "root": root,
                "train": {"root": root + "/train"}, "val": {"root": root + "/train"}, "test": {"root": root + "/train"}}
    raise SystemExit("unknown dataset %r (built in: %s)" % (name, ", ".join(BUILTIN_DATASETS)))


def compose(conf_dir, expt, overrides, run=None):
    """The experiment's config: from ``conf_dir`` when given, else from the built-in tree."""
    run = run or dict(RUNNER_KEYS)
    if conf_dir:
        from . import dropin
        dropin.install()
        try:
            return compose_tree(conf_dir, overrides)
        except ConfigError as e:
            raise SystemExit("config error: %s" % e) from e
    dotted, dataset, filepaths = {}, "synthetic", {}
    for arg in overrides:
        key, text = arg.split("=", 1)
        key = key.lstrip("+")
        if key == "expt":
            continue
        if key == "dataset":
            dataset = parse_value(text)
        elif key.startswith("filepaths."):
            filepaths[key.split(".", 1)[1]] = parse_value(text)
        elif key == "filepaths":
            continue                      # a group choice of the yaml tree; the built-in tree has no file to pick
        else:
            dotted[key] = parse_value(text)
    probe = make_cfg(expt)
    for key in dotted:                    # Hydra refuses to override a key that does not exist
        node = probe
        parts = key.split(".")
        ok = True
        for k in parts[:-1]:
            if not isinstance(node, dict) or k not in node:
                ok = False
                break
            node = node[k]
        if not ok or (parts[-1] not in node and key not in ("train.weight_clip", "train.ckpt_dir", "save_ckpts")
                      and not key.startswith("loss_weight.")):
            raise SystemExit("unknown config key %r" % key)
    cfg = make_cfg(expt, dotted=dotted)
    cfg["dataset"] = _to_cfg(_builtin_dataset(dataset, run, filepaths))
    cfg.train["channels_img"] = cfg.dataset.get("n_channels", 3) if "train.channels_img" not in dotted else \
        cfg.train["channels_img"]
    return cfg


def _to_cfg(obj):
    from .config import to_cfg
    return to_cfg(obj)


def seed_everything(seed):
    """pytorch_lightning.seed_everything (reference run_network.py:27): python, numpy and torch generators -- the
    SAME seed on every rank, as the reference does."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


# ---------------------------------------------------------------------------------------------------------
# data
# ---------------------------------------------------------------------------------------------------------
def shard_indices(n, rank, world, epoch=None, seed=0):
    """torch.utils.data.DistributedSampler(drop_last=False): pad to a multiple of ``world`` by wrapping around, then
    every ``world``-th index from ``rank``.  Under ``accelerator="ddp"`` (reference run_network.py:66) Lightning puts a
    DistributedSampler with shuffle=True in front of the train loader and calls ``set_epoch`` every epoch, so each
    epoch is a fresh permutation ``randperm(n, generator=manual_seed(seed + epoch))`` shared by all ranks (seed 0,
    the sampler's default): pass ``epoch`` to get that order; ``epoch=None`` is the unshuffled order (one process:
    the reference's loader has no shuffle, core/lightning_module.py:89-92)."""
    if world <= 1:
        return list(range(n))
    if epoch is None:
        idx = list(range(n))
    else:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(n, generator=g).tolist()
    total = int(math.ceil(n / world)) * world
    idx += idx[:total - n]
    return idx[rank:total:world]


class SyntheticImages:
    """Endless batches of uniform[-1, 1] images (the range ToTensor+Normalize(0.5, 0.5) produces)."""

    def __init__(self, batch, channels, size, device, seed):
        g = torch.Generator().manual_seed(seed)
        self.real = (torch.rand(batch, channels, size, size, generator=g) * 2 - 1).to(device)
        self.labels = torch.zeros(batch, dtype=torch.int64, device=device)

    def __len__(self):
        return 0          # no epoch structure of its own: the runner's steps_per_epoch applies

    def __iter__(self):
        while True:
            yield self.real, self.labels


class TensorFileImages:
    """A ``.pt`` / ``.npy`` file holding [M, C, H, W] floats already normalised to [-1, 1]."""

    def __init__(self, path, batch, device, rank=0, world=1):
        data = torch.load(path) if path.endswith(".pt") else torch.from_numpy(np.load(path))
        self.data, self.batch, self.device = data.float(), batch, device
        self.rank, self.world, self.epoch = rank, world, 0
        self.order = shard_indices(len(self.data), rank, world)

    def set_epoch(self, epoch):
        """The epoch the next ``iter()`` starts with (Lightning: ``sampler.set_epoch(trainer.current_epoch)``)."""
        self.epoch = int(epoch)

    def __len__(self):
        return len(self.order)

    def __iter__(self):
        epoch = self.epoch
        while True:
            # one process: no shuffling, as the reference's train_dataloader (:89-92); data parallel: the
            # DistributedSampler(shuffle=True) permutation of this epoch (shard_indices)
            order = torch.tensor(self.order if self.world <= 1
                                 else shard_indices(len(self.data), self.rank, self.world, epoch))
            epoch += 1
            for i in range(0, len(order), self.batch):
                real = self.data[order[i:i + self.batch]]
                if torch.device(self.device).type == "cuda":
                    real = real.pin_memory().to(self.device, non_blocking=True)
                yield real, torch.zeros(len(real), dtype=torch.int64, device=self.device)


IMG_EXTENSIONS = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif", ".tiff", ".webp")


def image_folder_samples(root):
    """(path, class_index) list with torchvision.datasets.ImageFolder's ordering: classes = sorted sub-directories,
    inside a class a sorted walk, file names sorted, image extensions only (reference lightning_module.py:89-91)."""
    classes = sorted(d.name for d in os.scandir(root) if d.is_dir())
    if not classes:
        raise FileNotFoundError("no class folders under %r" % root)
    samples = []
    for idx, cls in enumerate(classes):
        for dirpath, _, fnames in sorted(os.walk(os.path.join(root, cls), followlinks=True)):
            for fname in sorted(fnames):
                if fname.lower().endswith(IMG_EXTENSIONS):
                    samples.append((os.path.join(dirpath, fname), idx))
    return samples, classes


class ImageFolderImages:
    """The reference's real-data input step -- ImageFolder -> Resize((S, S)) -> ToTensor -> Normalize(mean, std),
    DataLoader without shuffling, incomplete last batch kept (core/lightning_module.py:42-47,89-92), sharded over the
    data-parallel ranks like DistributedSampler(shuffle=False) -- with the decode + resize on a background host thread
    and ToTensor / Normalize / HWC->CHW on the device (functional.normalize_u8_images): the host hands over 1 byte per
    element through a ring of pinned buffers.  ``host_batches()`` is the host half (uint8, testable anywhere);
    iterating the object is the device half and needs the GPU."""

    def __init__(self, root, batch, img_size, channels, mean, std, device, prefetch=3, rank=0, world=1):
        self.samples, self.classes = image_folder_samples(root)
        if not self.samples:
            raise FileNotFoundError("no images under %r" % root)
        self.rank, self.world, self.epoch = rank, world, 0
        self.order = shard_indices(len(self.samples), rank, world)          # (length / unshuffled view; see host_batches)
        self.batch, self.size, self.channels = batch, img_size, channels
        self.mean, self.std, self.device, self.prefetch = mean, std, torch.device(device), prefetch

    def __len__(self):
        return len(self.order)

    def set_epoch(self, epoch):
        """The epoch the next ``host_batches()`` / ``iter()`` starts with.  Lightning calls
        ``sampler.set_epoch(trainer.current_epoch)`` at every epoch start, so a run resumed from a checkpoint of epoch
        E continues with the permutation ``randperm(seed + E)``, not with ``seed + 0`` again (``fit`` passes the
        checkpoint's epoch here)."""
        self.epoch = int(epoch)

    def decode(self, path):
        from PIL import Image
        with open(path, "rb") as f:
            img = Image.open(f).convert("RGB" if self.channels == 3 else "L")    # ImageFolder's pil_loader gives RGB
        img = img.resize((self.size, self.size), Image.BILINEAR)                # transforms.Resize on a PIL image
        a = np.asarray(img, dtype=np.uint8)
        return a if a.ndim == 3 else a[:, :, None]

    def host_batches(self):
        """uint8 [b, S, S, C] arrays and int64 labels, epoch after epoch: dataset order in one process, a fresh
        DistributedSampler(shuffle=True) permutation per epoch under data parallelism (shard_indices)."""
        epoch = self.epoch
        while True:
            order = self.order if self.world <= 1 else shard_indices(len(self.samples), self.rank, self.world, epoch)
            epoch += 1
            for i in range(0, len(order), self.batch):
                chunk = [self.samples[j] for j in order[i:i + self.batch]]
                yield (np.stack([self.decode(p) for p, _ in chunk]), np.array([c for _, c in chunk], dtype=np.int64))

    def __iter__(self):
        import queue
        import threading
        from . import functional as F
        if self.device.type != "cuda":
            raise RuntimeError("lightning_gan_zoo_amd: the input step normalises on the GPU (no CPU fallback); "
                               "host_batches() yields the decoded uint8 batches")
        q = queue.Queue(maxsize=self.prefetch)

        def producer():
            for item in self.host_batches():
                q.put(item)

        threading.Thread(target=producer, daemon=True).start()
        ring, cursor = {}, 0
        while True:
            imgs, labels = q.get()
            key = imgs.shape
            slots = ring.setdefault(key, [[torch.empty(key, dtype=torch.uint8).pin_memory(), None]
                                          for _ in range(self.prefetch + 1)])
            buf, ev = slots[cursor % len(slots)]
            if ev is not None:
                ev.synchronize()
            buf.copy_(torch.from_numpy(imgs))
            u8 = buf.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            slots[cursor % len(slots)][1] = ev
            cursor += 1
            yield (F.normalize_u8_images(u8, self.mean, self.std),
                   torch.from_numpy(labels).to(self.device, non_blocking=True))


def build_data(cfg, run, device, rank=0, world=1):
    """The training set of ``cfg.dataset`` (reference ``instantiate(cfg.dataset.train, transform=...)``, :89-92)."""
    t = cfg.train
    node = cfg.get("dataset") or {}
    target = str(node.get("_target_", ""))
    if run.get("synthetic") or target.endswith("SyntheticImages") or not target:
        return SyntheticImages(t.batch_size, t.channels_img, t.img_size, device, 1234 + rank)
    train = node.get("train", node)
    if target.endswith("TensorFileImages"):
        return TensorFileImages(train["root"], t.batch_size, device, rank, world)
    if target.endswith("ImageFolder"):
        return ImageFolderImages(train["root"], t.batch_size, t.img_size, t.channels_img, t.data_mean, t.data_std,
                                 device, rank=rank, world=world)
    raise SystemExit("dataset target %r is not supported by this runner (ImageFolder-shaped datasets, tensor files "
                     "and synthetic batches are)" % target)


# ---------------------------------------------------------------------------------------------------------
# checkpoints
# ---------------------------------------------------------------------------------------------------------
def find_ckpt(ckpt_dir):
    """reference run_network.py:19-23: every *.ckpt below the directory; more than one is an error."""
    if not ckpt_dir:
        return None
    hits = [y for x in os.walk(ckpt_dir) for y in glob.glob(os.path.join(x[0], "*.ckpt"))]
    assert len(hits) <= 1, "Multiple ckpts found!"
    return hits[0] if hits else None


def checkpoint_blob(module, trainer, step, epoch, keeper=None):
    """Lightning's checkpoint envelope (``trainer.checkpoint_connector.dump_checkpoint`` of the 1.1 / 1.2 generation)."""
    state = {k: v.detach().cpu() for k, v in module.state_dict().items()}       # generator.* / discriminator.*
    blob = {"epoch": epoch, "global_step": step, "pytorch-lightning_version": LIGHTNING_VERSION_TAG,
            "state_dict": state,
            "optimizer_states": [o["optimizer"].state_dict() for o in trainer.optim],
            "lr_schedulers": [o["lr_scheduler"].state_dict() for o in trainer.optim],
            "callbacks": {}}
    if keeper is not None:
        blob["callbacks"]["ModelCheckpoint"] = keeper.state()
    return blob


def save_checkpoint(path, module, trainer, step, epoch, keeper=None):
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    tmp = path + ".tmp"
    torch.save(checkpoint_blob(module, trainer, step, epoch, keeper), tmp)
    os.replace(tmp, path)


def load_checkpoint(path, module, trainer):
    blob = torch.load(path, map_location="cpu", weights_only=False)
    sd = blob["state_dict"]
    for prefix, net in (("discriminator.", module.discriminator), ("generator.", module.generator)):
        net.load_state_dict({k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)})
    for o, s in zip(trainer.optim, blob.get("optimizer_states", [])):
        o["optimizer"].load_state_dict(s)
    for o, s in zip(trainer.optim, blob.get("lr_schedulers", [])):
        o["lr_scheduler"].load_state_dict(s)
    if blob.get("global_step", 0) > 0:
        # the optimizers HAVE stepped -- in the process that wrote the checkpoint; torch's "lr_scheduler.step() before
        # optimizer.step()" bookkeeping is per process and would otherwise warn at the first epoch end after a resume
        for o in trainer.optim:
            o["optimizer"]._opt_called = True
    return blob.get("global_step", 0), blob.get("epoch", 0), blob.get("callbacks", {}).get("ModelCheckpoint")


class CheckpointKeeper:
    """``ModelCheckpoint(monitor='fid', filename='model_best-{fid:.2f}')`` with Lightning's defaults (mode min,
    save_top_k 1), reference run_network.py:48-50: a new file when the monitored metric improves, the previous best
    removed AFTER the new one is on disk.  While the metric is not being produced (no InceptionV3 weights offline)
    the latest state is kept instead, as ``step=<n>.ckpt`` -- either way the directory holds exactly one ``*.ckpt``,
    which is what ``find_ckpt`` resumes from."""

    def __init__(self, dirpath, monitor="fid", filename="model_best-{fid:.2f}"):
        self.dirpath, self.monitor, self.filename = dirpath, monitor, filename
        self.best_score, self.best_path = None, None

    def state(self):
        return {"monitor": self.monitor, "best_model_score": self.best_score, "best_model_path": self.best_path}

    def load_state(self, st, current_path=None):
        if st:
            self.best_score, self.best_path = st.get("best_model_score"), st.get("best_model_path")
        if current_path:
            self.best_path = current_path

    def update(self, metrics, step, save_fn):
        """``save_fn(path)`` writes the checkpoint; returns the path written or None."""
        score = metrics.get(self.monitor) if metrics else None
        if score is None and self.best_score is not None and self.best_path and os.path.exists(self.best_path):
            # a state without a metric (a run cut short mid-epoch) never displaces the monitored best:
            # ModelCheckpoint(save_top_k=1) keeps its best file until something better arrives
            return None
        if score is not None:
            score = float(score)
            if self.best_score is not None and not score < self.best_score:
                return None
            name = self.filename.replace("{%s:.2f}" % self.monitor, "%s=%.2f" % (self.monitor, score)) + ".ckpt"
            self.best_score = score
        else:
            name = "step=%d.ckpt" % step
        path = os.path.join(self.dirpath, name)
        old, self.best_path = self.best_path, path
        save_fn(path)
        for stale in glob.glob(os.path.join(self.dirpath, "*.ckpt")):
            if os.path.abspath(stale) != os.path.abspath(path):
                os.remove(stale)
        del old
        return path


# ---------------------------------------------------------------------------------------------------------
# the loop
# ---------------------------------------------------------------------------------------------------------
def fit(module, cfg, data, run, sync=None, rank=0, world=1, evaluate=None, slow_group=None):
    """``pl.Trainer(max_epochs=cfg.train.num_epochs, resume_from_checkpoint=find_ckpt(...)).fit(model)`` for the
    step classes of this package (reference run_network.py:61-72).  Returns (module, trainer, global_step)."""
    from .harness import Trainer
    t = cfg.train
    # reference run_network.py:61-68: ``1`` or a ``{start_epoch, accumulation_factor}`` node -> Lightning's
    # ``accumulate_grad_batches={start_epoch: factor}`` (GradientAccumulationScheduler)
    acc = cfg.get("accumulate_grad_batches", 1)
    if not isinstance(acc, int):
        acc = {int(acc["start_epoch"]): int(acc["accumulation_factor"])}
    trainer = Trainer(module, grad_sync=sync, accumulate_grad_batches=acc)
    n = len(data) if hasattr(data, "__len__") else 0
    steps_per_epoch = int(math.ceil(n / t.batch_size)) if n else int(run["steps_per_epoch"])
    step = epoch = 0
    skip = 0
    ckpt_dir = t.get("ckpt_dir")
    keeper = CheckpointKeeper(ckpt_dir) if (ckpt_dir and cfg.get("save_ckpts", True)) else None
    ckpt = find_ckpt(ckpt_dir)
    if ckpt:
        step, epoch, kst = load_checkpoint(ckpt, module, trainer)
        trainer.batch_idx = step
        trainer.epoch, trainer.epoch_batch_idx = epoch, step % steps_per_epoch
        if keeper is not None:
            keeper.load_state(kst, ckpt)
        if hasattr(data, "set_epoch"):
            data.set_epoch(epoch)         # the sampler's permutation continues at seed + epoch (Lightning: set_epoch)
            skip = step % steps_per_epoch if n else 0     # a checkpoint cut by max_steps: the epoch continues where
                                                          # it stopped instead of replaying its first batches
        if rank == 0:
            print("resumed from %s at step %d (epoch %d)" % (ckpt, step, epoch))
    # a full Python garbage collection walks every object torch has created (~70 ms, several training steps):
    # park the set-up's survivors in the permanent generation so that later collections stay short
    import gc
    gc.collect()
    gc.freeze()
    t0 = time.time()
    last, first_step = {}, step
    max_steps, max_epochs = run.get("max_steps"), int(t.get("num_epochs", 99999))

    def checkpoint(metrics=None):
        if keeper is None:
            return
        if sync is not None:
            sync.flush()
            sync.sync_buffers()           # a collective: every rank takes part, rank 0 writes
        if rank == 0:
            keeper.update(metrics, step, lambda path: save_checkpoint(path, module, trainer, step, epoch, keeper))
        if world > 1:
            torch.distributed.barrier()

    if rank == 0 and n:
        #  the order the data is visited in is part of the run's definition: say it
        print("data order: %s" % ("dataset order, no shuffling (reference train_dataloader, one process)" if world <= 1
                                  else "DistributedSampler(shuffle=True, seed=0) permutation per epoch, %d ranks" % world))
    for batch in data:
        if skip:
            skip -= 1
            continue
        if (max_steps is not None and step >= max_steps) or epoch >= max_epochs:
            break
        loss, idx = trainer.step(batch, last_in_epoch=(step + 1) % steps_per_epoch == 0)
        last[("d_loss", "g_loss")[idx]] = loss
        step += 1
        if step % run["log_every"] == 0 and rank == 0:
            msg = " ".join("%s=%.4f" % (k, float(v)) for k, v in sorted(last.items()))
            rate = (step - first_step) * t.batch_size * world / (time.time() - t0)
            print("step %d epoch %d %s (%.1f img/s)" % (step, epoch, msg, rate))
        if step % steps_per_epoch == 0:
            trainer.end_epoch()           # lr_scheduler.step(), current_epoch += 1
            epoch += 1
            metrics = evaluate(module, epoch) if evaluate is not None else None
            if slow_group is not None:    # ranks > 0 wait for rank 0's Inception pass on the long-timeout group
                torch.distributed.barrier(group=slow_group)
            checkpoint(metrics)
    trainer.finish()
    if step % steps_per_epoch:            # a run cut short by max_steps still leaves a resumable state
        checkpoint(None)
    return module, trainer, step


FID_IMAGE_EXTENSIONS = ("bmp", "jpg", "jpeg", "pgm", "png", "ppm", "tif", "tiff", "webp")


def real_image_files(root):
    """The files ``compute_activations_of_path`` feeds to pytorch_fid (core/callback_inception_metrics.py:141-147):
    every image under ``root``, recursively, sorted."""
    out = []
    for d, _, names in os.walk(root):
        out += [os.path.join(d, f) for f in names if f.lower().endswith(FID_IMAGE_EXTENSIONS)]
    return sorted(out)


def real_activations(root, features, batch_size=16):
    """Inception activations of the real images as the reference's callback obtains them (:159-163, :211-221): the
    first ``*.npz`` in ``root`` if there is one (keys mu / sigma / act), else every image at its NATIVE resolution --
    pytorch_fid's loader applies ToTensor only, the one resize is InceptionV3's own to 299x299 -- in batches of
    ``batch_size`` consecutive files of equal size, and the result written to ``<root>/inception_cache.npz`` for the
    next launch / resume.  Returns the activations [n, 2048]."""
    cached = sorted(f for f in os.listdir(root) if ".npz" in f)
    if cached:
        with np.load(os.path.join(root, cached[0])) as data:
            return np.asarray(data["act"])
    from PIL import Image
    from . import eval as E
    acts, batch = [], []

    def flush():
        if batch:
            acts.append(features(np.stack(batch)))
            del batch[:]

    for path in real_image_files(root):
        with open(path, "rb") as f:
            img = np.asarray(Image.open(f).convert("RGB"), dtype=np.uint8)
        if batch and (img.shape != batch[0].shape or len(batch) == batch_size):
            flush()
        batch.append(img)
    flush()
    if not acts:
        raise FileNotFoundError("no images under %r" % root)
    act = np.concatenate(acts, axis=0)
    mu, sigma = E.activation_statistics(act)
    try:
        np.savez(os.path.join(root, "inception_cache.npz"), mu=mu, sigma=sigma, act=act)
    except OSError as e:            # read-only data set directory: recomputed on the next launch
        print("could not write %s/inception_cache.npz (%s)" % (root, e))
    return act


def wants_fid(cfg, run):
    node = cfg.get("dataset") or {}
    return bool(cfg.get("calc_fid", True) and run.get("inception_weights") and (node.get("val") or {}).get("root")
                and str(node.get("_target_", "")).endswith("ImageFolder"))


def make_fid_evaluator(cfg, run, module, device):
    """The reference's InceptionMetrics callback (run_network.py:51-56, core/callback_inception_metrics.py:136-246) as
    the ``evaluate`` hook of fit(): FID / KID of ``val.fid_n_samples`` generated images against the validation images,
    both through the InceptionV3 pool features on the HIP kernels (inception.py).  Needs the FID weight file the
    reference downloads (``inception_weights=<path to pt_inception-2015-12-05-6726825d.pth>``; no network here) and an
    ImageFolder-shaped validation set; returns None otherwise, and the run keeps its latest checkpoint instead of
    the best-FID one."""
    node = cfg.get("dataset") or {}
    if not (cfg.get("calc_fid", True) and run.get("inception_weights")):
        return None
    val_root = (node.get("val") or {}).get("root")
    if not val_root or not str(node.get("_target_", "")).endswith("ImageFolder"):
        return None
    from . import eval as E
    from .inception import InceptionFeatures, load_fid_weights
    features = InceptionFeatures(load_fid_weights(run["inception_weights"], device,
                                                  check_hash=bool(run.get("inception_check_hash", True))))
    n = int((cfg.get("val") or {}).get("fid_n_samples", 5000))
    dump = E.SampleDump(module, n_samples=n, batch_size=16)          # host RNG draw, right after the module is built
    # before training starts, persisted next to the images: a resume or the next launch loads it, and under data
    # parallelism the other ranks wait for it ONCE, at start-up (main() gives the process group a long timeout), not
    # inside a collective in the middle of the run
    real_act = real_activations(val_root, features)

    def evaluate(mod, epoch):
        m = E.evaluate(mod, dump, features, real_act)
        print("epoch %d FID: %.4f KID mean: %.6f KID stddev: %.6f" % (epoch, m["fid"], m["kid"], m["kid_std"]))
        return m

    return evaluate


def main(argv=None):
    conf_dir, expt, overrides, run = parse_overrides(sys.argv[1:] if argv is None else argv)
    cfg = compose(conf_dir, expt, overrides, run)
    seed_everything(run["seed"])                    # seed_everything(42), reference :27
    if not torch.cuda.is_available():
        raise SystemExit("lightning_gan_zoo_amd.run_network needs an MI355X: the step runs on the HIP kernels and has "
                         "no CPU fallback")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # GZ_REHEARSE_ONE_GPU=1 (tests, 1-GPU boxes): all ranks on cuda:0, collectives on gloo -- RCCL refuses two ranks on
    # one device; everything else of the multi-rank path runs as it would on N GPUs
    rehearsal = bool(os.environ.get("GZ_REHEARSE_ONE_GPU")) and world > 1
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    sync = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    module = locate(cfg.model.lm["_target_"])(cfg, logging_dir="output").to(device)
    slow_group = None
    if world > 1:
        from .ddp import GradSync
        sync = GradSync(module)
        if wants_fid(cfg, run):
            # with the FID evaluator on, rank 0 alone generates val.fid_n_samples images and runs InceptionV3 over them
            # at start-up and at every epoch end -- far longer than the default watchdog's 10 minutes.  Only THOSE waits
            # get a long timeout (their own group); every training collective keeps the default, so a rank that dies
            # mid-training is still noticed in minutes.
            import datetime
            # (gloo: the wait is a host-side one, no GPU stream sits behind it)
            slow_group = torch.distributed.new_group(timeout=datetime.timedelta(hours=6), backend="gloo")
    evaluate, failure = None, None
    if rank == 0:
        try:
            evaluate = make_fid_evaluator(cfg, run, module, device)
        except BaseException as e:  # noqa: BLE001  (SystemExit from a missing weight file included)
            failure = e
            if slow_group is None:
                raise
    if slow_group is not None:
        # rank 0 has the real activations (computed or loaded) before step 0 -- or failed, and then says so: the
        # other ranks leave with it instead of sitting in a barrier for the long timeout
        flag = torch.tensor([1 if failure is not None else 0])
        torch.distributed.broadcast(flag, 0, group=slow_group)
        if int(flag.item()):
            if rank == 0:
                print("FID evaluator could not be built on rank 0: %r" % (failure,), file=sys.stderr)
            torch.distributed.destroy_process_group()
            raise SystemExit(2)
    data = build_data(cfg, run, device, rank, world)
    out = fit(module, cfg, data, run, sync=sync, rank=rank, world=world, evaluate=evaluate, slow_group=slow_group)
    if world > 1:
        torch.distributed.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
