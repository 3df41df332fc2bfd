"""Thin runner with the reference's command-line grammar (run_network.py:25-75 + Hydra overrides):

    python -m lightning_gan_zoo_amd.run_network +expt=dc_gan train.batch_size=64 max_steps=200 \\
           dataset=synthetic save_ckpts=true train.ckpt_dir=output/ckpt

``+expt=<name>`` selects the experiment (dc_gan | wgan | wgan_gp | hologan); ``a.b.c=value`` overrides
any config leaf (values parsed as YAML scalars); ``num_gpus`` > 1 expects a torch.distributed launch
(one process per GPU).  What the reference delegates to pytorch_lightning is done by harness.Trainer:
per-batch optimizer alternation, toggle, backward, step, per-epoch LR scheduler, checkpoint save /
resume (``find_ckpt`` semantics: the single ``*.ckpt`` under ``train.ckpt_dir``).

Checkpoints are plain ``torch.save`` dicts with a Lightning-style ``state_dict`` whose keys are
``generator.<...>`` / ``discriminator.<...>`` exactly as the reference's LightningModule would
produce, so the weights interchange with the reference.
"""
import glob
import os
import sys
import time

import torch
import yaml

from .config import locate, make_cfg

RUNNER_KEYS = {"max_steps": 100, "log_every": 10, "dataset": "synthetic", "dataset_path": None, "seed": 42,
               "device": "cuda", "module_root": None, "steps_per_epoch": 100}


def parse_overrides(argv):
    expt, overrides, runner = None, {}, dict(RUNNER_KEYS)
    for arg in argv:
        if "=" not in arg:
            raise SystemExit("cannot parse %r (expected key=value or +expt=name)" % arg)
        key, val = arg.split("=", 1)
        val = yaml.safe_load(val)
        if key in ("+expt", "expt"):
            expt = val
        elif key in runner:
            runner[key] = val
        else:
            overrides[key] = val
    if expt is None:
        raise SystemExit("missing +expt=<dc_gan|wgan|wgan_gp|hologan>")
    return expt, overrides, runner


def compose(expt, overrides, module_root=None):
    kw = {}
    if module_root:
        kw["module_root"] = module_root
    cfg = make_cfg(expt, **kw)
    derived = {"train.features_disc": ("discriminator", "features_d"), "train.features_gen": ("generator", "features_g"),
               "train.img_size": None, "model.noise_dim": ("generator", "channels_noise")}
    for dotted, v in overrides.items():
        node = cfg
        keys = dotted.split(".")
        for k in keys[:-1]:
            node = node[k]
        if keys[-1] not in node and dotted not in ("train.weight_clip", "train.ckpt_dir"):
            raise SystemExit("unknown config key %r" % dotted)
        node[keys[-1]] = v
        tgt = derived.get(dotted)
        if tgt and tgt[1] in cfg[tgt[0]]:        # the ${...} interpolations of the reference's yaml
            cfg[tgt[0]][tgt[1]] = v
        if dotted == "train.img_size":
            for net in ("discriminator", "generator"):
                for key in ("img_size", "size"):       # `size: ${train.img_size}` in gan_stability_r1.yaml
                    if key in cfg[net]:
                        cfg[net][key] = v
        if dotted == "model.noise_dim":
            for net in ("discriminator", "generator"):
                if "z_dim" in cfg[net]:                # `z_dim: ${model.noise_dim}`
                    cfg[net]["z_dim"] = v
        if dotted.startswith("optimisation.lr"):
            for o in ("disc_optimiser", "gen_optimiser", "optimiser"):
                cfg[o]["lr"] = v
    return cfg


class SyntheticImages:
    """Endless batches of uniform[-1, 1] images (the range ToTensor+Normalize(0.5, 0.5) produces)."""

    def __init__(self, batch, channels, size, device, seed):
        g = torch.Generator().manual_seed(seed)
        self.real = (torch.rand(batch, channels, size, size, generator=g) * 2 - 1).to(device)
        self.labels = torch.zeros(batch, dtype=torch.int64, device=device)

    def __iter__(self):
        while True:
            yield self.real, self.labels


class TensorFileImages:
    """A ``.pt`` / ``.npy`` file holding [M, C, H, W] floats already normalised to [-1, 1]."""

    def __init__(self, path, batch, device):
        import numpy as np
        data = torch.load(path) if path.endswith(".pt") else torch.from_numpy(np.load(path))
        self.data, self.batch, self.device = data.float(), batch, device

    def __iter__(self):
        n = len(self.data) // self.batch * self.batch
        while True:
            for i in range(0, n, self.batch):        # no shuffling, as the reference's train_dataloader (:89-92)
                real = self.data[i:i + self.batch].pin_memory().to(self.device, non_blocking=True)
                yield real, torch.zeros(self.batch, dtype=torch.int64, device=self.device)


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
    DataLoader without shuffling, incomplete last batch kept (core/lightning_module.py:42-47,89-92) -- with the
    decode + resize on a background host thread and ToTensor / Normalize / HWC->CHW on the device
    (functional.normalize_u8_images): the host hands over 1 byte per element through a ring of pinned buffers."""

    def __init__(self, root, batch, img_size, channels, mean, std, device, prefetch=3):
        self.samples, self.classes = image_folder_samples(root)
        if not self.samples:
            raise FileNotFoundError("no images under %r" % root)
        self.batch, self.size, self.channels = batch, img_size, channels
        self.mean, self.std, self.device, self.prefetch = mean, std, torch.device(device), prefetch

    def decode(self, path):
        import numpy as np
        from PIL import Image
        with open(path, "rb") as f:
            img = Image.open(f).convert("RGB" if self.channels == 3 else "L")    # ImageFolder's pil_loader gives RGB
        img = img.resize((self.size, self.size), Image.BILINEAR)                # transforms.Resize on a PIL image
        a = np.asarray(img, dtype=np.uint8)
        return a if a.ndim == 3 else a[:, :, None]

    def host_batches(self):
        """uint8 [b, S, S, C] arrays and int64 labels, epoch after epoch, in dataset order."""
        import numpy as np
        while True:
            for i in range(0, len(self.samples), self.batch):
                chunk = self.samples[i:i + self.batch]
                yield (np.stack([self.decode(p) for p, _ in chunk]), np.array([c for _, c in chunk], dtype=np.int64))

    def __iter__(self):
        import queue
        import threading
        from . import functional as F
        q = queue.Queue(maxsize=self.prefetch)

        def producer():
            for item in self.host_batches():
                q.put(item)

        threading.Thread(target=producer, daemon=True).start()
        ring, cursor = {}, 0
        while True:
            imgs, labels = q.get()
            if self.device.type != "cuda":        # CPU oracle runs: same arithmetic with torch
                x = torch.from_numpy(imgs).permute(0, 3, 1, 2).float().div(255).sub(self.mean).div(self.std)
                yield x, torch.from_numpy(labels)
                continue
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


def find_ckpt(ckpt_dir):
    """reference run_network.py:19-23: exactly one *.ckpt in the directory, else none."""
    if not ckpt_dir:
        return None
    hits = glob.glob(os.path.join(ckpt_dir, "*.ckpt"))
    return hits[0] if len(hits) == 1 else None


def save_checkpoint(path, module, trainer, step, epoch):
    state = {}
    for prefix, net in (("discriminator.", module.discriminator), ("generator.", module.generator)):
        for k, v in net.state_dict().items():
            state[prefix + k] = v.detach().cpu()
    blob = {"state_dict": state, "global_step": step, "epoch": epoch,
            "optimizer_states": [o["optimizer"].state_dict() for o in trainer.optim],
            "lr_schedulers": [o["lr_scheduler"].state_dict() for o in trainer.optim]}
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    torch.save(blob, path)


def load_checkpoint(path, module, trainer):
    blob = torch.load(path, map_location="cpu", weights_only=False)
    sd = blob["state_dict"]
    for prefix, net in (("discriminator.", module.discriminator), ("generator.", module.generator)):
        net.load_state_dict({k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)})
    for o, s in zip(trainer.optim, blob.get("optimizer_states", [])):
        o["optimizer"].load_state_dict(s)
    for o, s in zip(trainer.optim, blob.get("lr_schedulers", [])):
        o["lr_scheduler"].load_state_dict(s)
    return blob.get("global_step", 0), blob.get("epoch", 0)


def main(argv=None):
    from .harness import Trainer
    expt, overrides, run = parse_overrides(sys.argv[1:] if argv is None else argv)
    cfg = compose(expt, overrides, run["module_root"])
    torch.manual_seed(run["seed"])                  # seed_everything(42), reference :27
    device = torch.device(run["device"])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    sync = None
    if world > 1:
        import torch.distributed as dist
        from .ddp import GradSync
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    module = locate(cfg.model.lm["_target_"])(cfg, logging_dir="output").to(device)
    if world > 1:
        sync = GradSync(module)
    trainer = Trainer(module, grad_sync=sync)
    t = cfg.train
    if run["dataset"] == "synthetic":
        data = SyntheticImages(t.batch_size, t.channels_img, t.img_size, device, 1234 + int(os.environ.get("RANK", "0")))
    elif run["dataset"] == "image_folder":          # the reference's ImageFolder(root=cfg.dataset.root) (:89-91)
        data = ImageFolderImages(run["dataset_path"], t.batch_size, t.img_size, t.channels_img, t.data_mean, t.data_std,
                                 device)
    else:
        data = TensorFileImages(run["dataset_path"], t.batch_size, device)
    step = epoch = 0
    ckpt = find_ckpt(t.get("ckpt_dir"))
    if ckpt:
        step, epoch = load_checkpoint(ckpt, module, trainer)
        trainer.batch_idx = step
        print("resumed from %s at step %d" % (ckpt, step))
    # a full Python garbage collection walks every object torch has created (~70 ms, several training steps):
    # park the set-up's survivors in the permanent generation so that later collections stay short
    import gc
    gc.collect()
    gc.freeze()
    t0 = time.time()
    last = {}
    for batch in data:
        if step >= run["max_steps"]:
            break
        loss, idx = trainer.step(batch)
        last[("d_loss", "g_loss")[idx]] = loss
        step += 1
        if step % run["steps_per_epoch"] == 0:
            trainer.end_epoch()
            epoch += 1
        if step % run["log_every"] == 0 and int(os.environ.get("RANK", "0")) == 0:
            msg = " ".join("%s=%.4f" % (k, float(v)) for k, v in sorted(last.items()))
            print("step %d epoch %d %s (%.1f img/s)" % (step, epoch, msg, step * t.batch_size * world / (time.time() - t0)))
    trainer.finish()
    if cfg.get("save_ckpts", True) and t.get("ckpt_dir") and int(os.environ.get("RANK", "0")) == 0:
        if sync is not None:
            sync.sync_buffers()
        for old in glob.glob(os.path.join(t["ckpt_dir"], "*.ckpt")):
            os.remove(old)
        save_checkpoint(os.path.join(t["ckpt_dir"], "step=%d.ckpt" % step), module, trainer, step, epoch)
    return module, trainer, step


if __name__ == "__main__":
    main()
