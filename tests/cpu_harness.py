"""Test infrastructure: drive the product runner's host logic (run_network.fit: alternation, epochs, schedulers,
checkpoints, sharding) on the CPU with the ORACLE's step classes as the model.  The product command line cannot be
pointed at the oracle or at a CPU device (a test switch, not a product backend) -- this module can."""
import torch

from lightning_gan_zoo_amd import run_network as R
from lightning_gan_zoo_amd.config import locate, make_cfg

ORACLE_ROOT = "oracle.reference_cpu"


class HostNormalisedFolder:
    """ImageFolderImages' host half (decode, resize, order, sharding) + the normalisation arithmetic in torch, for
    runs without a GPU: x / 255 -> (x - mean) / std, HWC -> CHW."""

    def __init__(self, folder):
        self.folder = folder

    def __len__(self):
        return len(self.folder)

    def set_epoch(self, epoch):
        self.folder.set_epoch(epoch)

    def __iter__(self):
        f = self.folder
        for imgs, labels in f.host_batches():
            x = torch.from_numpy(imgs).permute(0, 3, 1, 2).float().div(255).sub(f.mean).div(f.std)
            yield x, torch.from_numpy(labels)


def oracle_cfg(expt, overrides):
    """Same composition as the product runner, with the ``_target_`` strings pointed at the oracle."""
    conf_dir, expt, rest, run = R.parse_overrides(["+expt=" + expt] + list(overrides))
    cfg = R.compose(None, expt, rest, run)
    ref = make_cfg(expt, module_root=ORACLE_ROOT)
    for node, sub in (("model", "lm"), ("discriminator", None), ("generator", None)):
        tgt = (cfg[node][sub] if sub else cfg[node])
        tgt["_target_"] = (ref[node][sub] if sub else ref[node])["_target_"]
    sch = cfg.optimisation.lr_scheduler
    if "lightning_gan_zoo_amd" in sch["_target_"]:
        sch["_target_"] = ref.optimisation.lr_scheduler["_target_"]
    return cfg, run


def run_on_cpu(expt, overrides, data=None, sync_factory=None, rank=0, world=1, evaluate=None):
    cfg, run = oracle_cfg(expt, overrides)
    R.seed_everything(run["seed"])
    module = locate(cfg.model.lm["_target_"])(cfg, logging_dir=None)
    sync = sync_factory(module) if sync_factory else None
    if data is None:
        data = R.build_data(cfg, run, "cpu", rank, world)
        if isinstance(data, R.ImageFolderImages):
            data = HostNormalisedFolder(data)
    return R.fit(module, cfg, data, run, sync=sync, rank=rank, world=world, evaluate=evaluate) + (cfg,)
