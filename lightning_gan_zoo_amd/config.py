"""Hot-path configuration: the values the reference composes with Hydra from
``conf/config.yaml`` + ``conf/expt/<name>.yaml`` (citations inline), as plain
nested dicts with attribute access.  Only the keys the step classes read
(core/lightning_module.py:38-54,76-87,161,197) are produced.

``_target_`` strings keep the reference's grammar; ``module_root`` chooses whose
classes they point at (this package by default).
"""
import copy
import importlib


class Cfg(dict):
    """dict with attribute access (stands in for an OmegaConf DictConfig)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return Cfg({k: copy.deepcopy(v, memo) for k, v in self.items()})


def to_cfg(obj):
    if isinstance(obj, dict):
        return Cfg({k: to_cfg(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return [to_cfg(v) for v in obj]
    return obj


def locate(path):
    """Resolve a dotted ``_target_``: import the longest importable module
    prefix, then walk the remaining attributes (as hydra's locate does)."""
    parts = path.split(".")
    for n in range(len(parts) - 1, 0, -1):
        try:
            obj = importlib.import_module(".".join(parts[:n]))
        except ModuleNotFoundError:
            continue
        for name in parts[n:]:
            obj = getattr(obj, name)
        return obj
    raise ImportError("cannot locate %r" % path)


def instantiate(node, *args, **kwargs):
    """``hydra.utils.instantiate`` for the subset the hot path uses: import
    ``_target_`` and call it with the node's other keys merged with kwargs."""
    node = dict(node)
    target = locate(node.pop("_target_"))
    return target(*args, **{**node, **kwargs})


PRODUCT_ROOT = "lightning_gan_zoo_amd.core"

_NORMAL = {"_target_": "torch.distributions.normal.Normal", "loc": 0, "scale": 1}    # conf/noise_distn/normal.yaml
_UNIFORM = {"_target_": "torch.distributions.uniform.Uniform", "low": -1, "high": 1}  # conf/noise_distn/uniform.yaml
_STEP_LR = {"_target_": "torch.optim.lr_scheduler.StepLR", "step_size": -1, "gamma": 1}  # conf/lr_scheduler/step_lr.yaml


def _base(root):
    # conf/config.yaml:1-40
    return {
        "name": None,
        "num_gpus": 1,
        "model": {"lm": {"_target_": root + ".lightning_module.BaseGAN"},
                  "noise_distn": dict(_NORMAL), "noise_dim": 100},
        "train": {"batch_size": 128, "img_size": 64, "num_workers": 4, "channels_img": 3,
                  "num_epochs": 99999, "features_disc": 64, "features_gen": 64, "ckpt_dir": None,
                  "criterion": {"_target_": "torch.nn.BCEWithLogitsLoss"},
                  "data_mean": 0.5, "data_std": 0.5},
        "optimisation": {"disc_freq": 1, "gen_freq": 1, "lr_scheduler": dict(_STEP_LR)},
        "loss_weight": {},
        "debug": {"verbose_shape": False, "fast_dev_run": False},
        "precision": 32,   # reference default is 16 (AMP); the parity target is fp32
    }


def _std_nets(cfg, root, **disc_extra):
    t = cfg["train"]
    cfg["discriminator"] = {
        "_target_": root + ".models.standard_networks.Discriminator",
        "channels_img": t["channels_img"], "features_d": t["features_disc"],
        "img_size": t["img_size"], "final_sigmoid": False, **disc_extra}   # config.yaml:36-38
    cfg["generator"] = {
        "_target_": root + ".models.standard_networks.Generator",
        "channels_noise": cfg["model"]["noise_dim"], "channels_img": t["channels_img"],
        "features_g": t["features_gen"], "img_size": t["img_size"]}


def make_cfg(expt, module_root=PRODUCT_ROOT, batch_size=None, features=None, img_size=None,
             noise_dim=None, **overrides):
    """Compose the config of ``+expt=<expt>``.  ``features`` sets both
    ``train.features_disc`` and ``train.features_gen``."""
    cfg = _base(module_root)
    cfg["name"] = expt
    t = cfg["train"]
    if features is not None:
        t["features_disc"] = t["features_gen"] = features
    if img_size is not None:
        t["img_size"] = img_size
    if noise_dim is not None:
        cfg["model"]["noise_dim"] = noise_dim
    lm = module_root + ".lightning_module."
    if expt == "dc_gan":                                  # conf/expt/dc_gan.yaml
        cfg["model"]["lm"]["_target_"] = lm + "DCGAN"
        cfg["optimisation"].update(lr=2e-4, beta1=0.5, beta2=0.999)
        cfg["optimiser"] = {"_target_": "torch.optim.Adam", "lr": 2e-4, "betas": [0.5, 0.999]}
        _std_nets(cfg, module_root)
    elif expt == "wgan":                                  # conf/expt/wgan.yaml
        cfg["model"]["lm"]["_target_"] = lm + "WGAN"
        t.update(batch_size=64, weight_clip=1e-2)
        cfg["optimisation"].update(lr=5e-5, disc_freq=5, gen_freq=1)
        cfg["optimiser"] = {"_target_": "torch.optim.RMSprop", "lr": 5e-5}
        _std_nets(cfg, module_root)
    elif expt == "wgan_gp":                               # conf/expt/wgan_gp.yaml
        cfg["model"]["lm"]["_target_"] = lm + "WGANGP"
        t.update(batch_size=64)
        cfg["optimisation"].update(lr=1e-4, beta1=0.0, beta2=0.9)
        cfg["loss_weight"] = {"lambda_gp": 10}
        cfg["optimiser"] = {"_target_": "torch.optim.Adam", "lr": 1e-4, "betas": [0.0, 0.9]}
        _std_nets(cfg, module_root, norm="instance_norm2d")
    elif expt == "hologan":                               # conf/expt/hologan.yaml
        cfg["model"]["lm"]["_target_"] = lm + "HOLOGAN"
        cfg["model"]["noise_dim"] = noise_dim or 128
        cfg["model"]["noise_distn"] = dict(_UNIFORM)
        t.update(batch_size=32, num_epochs=25)
        cfg["optimisation"].update(lr=1e-4, disc_freq=1, gen_freq=2, beta1=0.9, beta2=0.999,
                                   lr_scheduler={"_target_": module_root + ".utils.hologan.create_hologan_lr_scheduler",
                                                 "total_epochs": t["num_epochs"]})
        cfg["optimiser"] = {"_target_": "torch.optim.Adam", "lr": 1e-4, "betas": [0.9, 0.999]}
        cfg["discriminator"] = {"_target_": module_root + ".models.hologan_discriminator.Discriminator",
                                "in_planes": t["channels_img"], "out_planes": features or 64,
                                "z_planes": cfg["model"]["noise_dim"]}
        cfg["generator"] = {"_target_": module_root + ".models.hologan_generator.Generator",
                            "in_planes": features or 64, "out_planes": t["channels_img"],
                            "z_planes": cfg["model"]["noise_dim"], "gpu": True,
                            "img_size": t["img_size"],
                            "view_args": {"elevation_low": 70, "elevation_high": 110,
                                          "azimuth_low": 220, "azimuth_high": 320,
                                          "scale_low": 1, "scale_high": 1,
                                          "transX_low": 0, "transX_high": 0,
                                          "transY_low": 0, "transY_high": 0,
                                          "transZ_low": 0, "transZ_high": 0,
                                          "batch_size": t["batch_size"]}}
        if t["img_size"] == 128:     # EXT-128: the reference itself cannot run at 128
            cfg["generator"]["ext128"] = True
            cfg["discriminator"]["img_size"] = 128
    elif expt == "gan_stability_r1":                      # conf/expt/gan_stability_r1.yaml
        cfg["model"]["lm"]["_target_"] = lm + "GANStabilityR1"
        cfg["model"]["noise_dim"] = noise_dim or 256
        t.update(batch_size=64, img_size=img_size or 128)
        cfg["optimisation"].update(lr=1e-4, disc_freq=1, lr_anneal=1.0, anneal_every=150000)
        cfg["optimiser"] = {"_target_": "torch.optim.RMSprop", "lr": 1e-4}
        cfg["loss_weight"] = {"reg": 10}
        # Only the keys the ResNet classes accept: the root config would also merge `img_size` (and
        # `final_sigmoid` for D, conf/config.yaml:35-40), which resnet.Discriminator.__init__ (resnet.py:55)
        # rejects -- the reference's own composition of this experiment cannot construct its discriminator.
        for net in ("generator", "discriminator"):
            cfg[net] = {"_target_": module_root + ".submodules.gan_stability.models.resnet." + net.capitalize(),
                        "z_dim": cfg["model"]["noise_dim"], "nlabels": 1, "size": t["img_size"],
                        "nfilter": features or 16, "nfilter_max": 512, "embed_size": 1}
    else:
        raise ValueError("unknown expt %r (covered: dc_gan, wgan, wgan_gp, hologan, gan_stability_r1)" % expt)
    if batch_size is not None:
        t["batch_size"] = batch_size
        if expt == "hologan":
            cfg["generator"]["view_args"]["batch_size"] = batch_size
    cfg["disc_optimiser"] = copy.deepcopy(cfg["optimiser"])   # config.yaml:30-31
    cfg["gen_optimiser"] = copy.deepcopy(cfg["optimiser"])
    cfg = to_cfg(cfg)
    for dotted, v in overrides.items():
        node = cfg
        keys = dotted.split("__")
        for k in keys[:-1]:
            node = node[k]
        node[keys[-1]] = to_cfg(v)
    return cfg
