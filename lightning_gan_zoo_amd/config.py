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
    node = _plain(node)
    target = locate(node.pop("_target_"))
    recursive = node.pop("_recursive_", True)
    for k, v in list(node.items()):
        if recursive and isinstance(v, dict) and "_target_" in v:
            node[k] = instantiate(v)                       # Hydra 1.1 instantiates nested nodes
        elif isinstance(v, list) and v and all(isinstance(x, (int, float)) and not isinstance(x, bool) for x in v) \
                and any(isinstance(x, float) for x in v):
            node[k] = [float(x) for x in v]                # ``betas: [0, 0.9]`` (conf/expt/wgan_gp.yaml): torch wants floats
    return target(*args, **{**node, **kwargs})


def _plain(node):
    """A Hydra / OmegaConf node (the reference's own harness hands the step classes a DictConfig) as a resolved
    plain dict; plain mappings are shallow-copied."""
    if type(node).__module__.startswith("omegaconf"):
        from omegaconf import OmegaConf
        return OmegaConf.to_container(node, resolve=True)
    return dict(node)


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
        "accumulate_grad_batches": 1,      # conf/config.yaml:57; conf/machine/*.yaml: {start_epoch, accumulation_factor}
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


PRIMARY = ("train", "model", "optimisation")      # nodes other nodes are derived from (the yaml's ${...} sources)


def _assign(cfg, dotted, value, sep="."):
    node = cfg
    keys = dotted.split(sep)
    for k in keys[:-1]:
        node = node.setdefault(k, {})
    node[keys[-1]] = value


def make_cfg(expt, module_root=PRODUCT_ROOT, batch_size=None, features=None, img_size=None,
             noise_dim=None, dotted=None, **overrides):
    """The config of ``+expt=<expt>`` as the reference's tree composes it (built in: no yaml needed).  ``features``
    sets both ``train.features_disc`` and ``train.features_gen``.  ``dotted``: {"a.b.c": value} overrides with the
    command-line semantics -- those under train / model / optimisation are applied BEFORE the nodes that interpolate
    them are derived (``features_d: ${train.features_disc}``, ``lr: ${optimisation.lr}``, ...), the rest after."""
    cfg = _base(module_root)
    cfg["name"] = expt
    t = cfg["train"]
    dotted = dict(dotted or {})
    explicit = {k: dotted[k] for k in ("train.batch_size", "train.img_size", "train.num_epochs", "model.noise_dim",
                                       "optimisation.lr", "optimisation.beta1", "optimisation.beta2",
                                       "optimisation.disc_freq", "optimisation.gen_freq", "train.weight_clip")
                if k in dotted}
    if features is not None:
        t["features_disc"] = t["features_gen"] = features
    if img_size is not None:
        t["img_size"] = img_size
    if noise_dim is not None:
        cfg["model"]["noise_dim"] = noise_dim
    for k, v in dotted.items():
        if k.split(".")[0] in PRIMARY and k not in explicit:
            _assign(cfg, k, v)
    if "train.img_size" in explicit:
        img_size = t["img_size"] = explicit["train.img_size"]
    if "model.noise_dim" in explicit:
        noise_dim = cfg["model"]["noise_dim"] = explicit["model.noise_dim"]
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
    if "train.batch_size" in explicit:
        batch_size = explicit["train.batch_size"]
    if batch_size is not None:
        t["batch_size"] = batch_size
        if expt == "hologan":
            cfg["generator"]["view_args"]["batch_size"] = batch_size
    # the experiment files set some primary keys themselves; an explicit override wins over them, and the nodes
    # that interpolate them follow (optimiser.lr: ${optimisation.lr}, betas, total_epochs: ${train.num_epochs})
    for k, v in explicit.items():
        _assign(cfg, k, v)
    o = cfg["optimisation"]
    cfg["optimiser"]["lr"] = o["lr"]
    if "betas" in cfg["optimiser"]:
        cfg["optimiser"]["betas"] = [o["beta1"], o["beta2"]]
    if "total_epochs" in o["lr_scheduler"]:
        o["lr_scheduler"]["total_epochs"] = t["num_epochs"]
    cfg["disc_optimiser"] = copy.deepcopy(cfg["optimiser"])   # config.yaml:30-31
    cfg["gen_optimiser"] = copy.deepcopy(cfg["optimiser"])
    for k, v in dotted.items():
        if k.split(".")[0] not in PRIMARY:
            _assign(cfg, k, v)
    cfg = to_cfg(cfg)
    for key, v in overrides.items():           # keyword form used by the tests: loss_weight__reg=...
        _assign(cfg, key, to_cfg(v), sep="__")
    return cfg


# ---------------------------------------------------------------------------------------------------------
# Composition over a ``conf/``-shaped tree (reference conf/config.yaml:64-69, run_network.py:25): the subset of
# Hydra 1.1 the reference's tree uses -- defaults lists (``- group: option``, ``- /group@package: option``,
# ``- override /group: option``), ``# @package _global_`` headers, ``+group=option`` / ``group=option`` /
# ``a.b=value`` / ``+a.b=value`` / ``~a.b`` command-line overrides, ``${a.b}`` interpolation.  hydra-core and
# omegaconf are not in this image (and cannot be assumed on a GPU box), PyYAML is.
# ---------------------------------------------------------------------------------------------------------
import os      # noqa: E402
import re      # noqa: E402

import yaml    # noqa: E402


class _Loader(yaml.SafeLoader):
    """SafeLoader with OmegaConf's float rule: ``2e-4`` / ``1.`` are floats (plain YAML 1.1 reads ``2e-4`` as a string)."""


_Loader.add_implicit_resolver(
    "tag:yaml.org,2002:float",
    re.compile(r"""^(?:[-+]?(?:[0-9][0-9_]*)\.[0-9_]*(?:[eE][-+]?[0-9]+)?
                    |[-+]?(?:[0-9][0-9_]*)(?:[eE][-+]?[0-9]+)
                    |\.[0-9_]+(?:[eE][-+][0-9]+)?
                    |[-+]?\.(?:inf|Inf|INF)
                    |\.(?:nan|NaN|NAN))$""", re.X),
    list("-+0123456789."))


def parse_value(text):
    """A command-line override value, parsed as a YAML scalar / flow collection with the loader above."""
    return yaml.load(text, Loader=_Loader)


class ConfigError(Exception):
    pass


def _load_file(conf_dir, rel):
    path = os.path.join(conf_dir, rel + ".yaml")
    if not os.path.isfile(path):
        raise ConfigError("config file %r not found under %s" % (rel + ".yaml", conf_dir))
    with open(path) as f:
        text = f.read()
    m = re.search(r"^#\s*@package\s+(\S+)", text, re.M)
    body = yaml.load(text, Loader=_Loader) or {}
    defaults = body.pop("defaults", []) or []
    return body, defaults, (m.group(1) if m else None)


def _parse_default(entry):
    """-> (group, option, package, is_override); group None for ``_self_`` / a bare file name."""
    if isinstance(entry, str):
        return (None, entry, None, False)
    (key, option), = entry.items()
    key = key.strip()
    override = key.startswith("override ")
    if override:
        key = key[len("override "):].strip()
    group, _, package = key.partition("@")
    return group.strip("/"), option, (package or None), override


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)


def _set_path(root, dotted, value, create=True):
    node = root
    keys = dotted.split(".")
    for k in keys[:-1]:
        if k not in node or not isinstance(node[k], dict):
            if not create:
                raise ConfigError("no such config node %r" % dotted)
            node[k] = {}
        node = node[k]
    return node, keys[-1]


def _place(root, package, body):
    if package in (None, "", "_global_"):
        _merge(root, body)
    else:
        node, last = _set_path(root, package, None)
        if not isinstance(node.get(last), dict):
            node[last] = {}
        _merge(node[last], body)


_INTERP = re.compile(r"\$\{([^${}]+)\}")


def _lookup(root, dotted):
    node = root
    for k in dotted.strip().split("."):
        if isinstance(node, dict):
            node = node[k]
        elif isinstance(node, list):
            node = node[int(k)]
        else:
            raise KeyError(dotted)
    return node


def resolve(root):
    """Replace every ``${a.b}`` by the value it points at (whole-value references keep their type; references inside
    a longer string are formatted in).  OmegaConf resolves lazily, so a dangling reference in a branch nobody reads
    is not an error there: such strings are left as they are here."""
    def res(value, stack):
        if isinstance(value, dict):
            return {k: res(v, stack) for k, v in value.items()}
        if isinstance(value, list):
            return [res(v, stack) for v in value]
        if not isinstance(value, str) or "${" not in value:
            return value
        whole = _INTERP.fullmatch(value.strip())
        try:
            if whole:
                key = whole.group(1)
                if key in stack:
                    raise ConfigError("interpolation cycle through %r" % key)
                return res(copy.deepcopy(_lookup(root, key)), stack + (key,))

            def sub(m):
                key = m.group(1)
                if key in stack:
                    raise ConfigError("interpolation cycle through %r" % key)
                return str(res(_lookup(root, key), stack + (key,)))
            return _INTERP.sub(sub, value)
        except (KeyError, IndexError, ValueError, TypeError):
            return value
    return res(root, ())


def compose_tree(conf_dir, overrides=(), config_name="config"):
    """Compose ``conf_dir/config.yaml`` with Hydra-style command-line ``overrides`` -> resolved ``Cfg``."""
    conf_dir = os.path.abspath(conf_dir)
    primary, primary_defaults, _ = _load_file(conf_dir, config_name)
    choices, appended, values = {}, [], []
    for arg in overrides:
        if arg.startswith("~"):
            values.append(("~", arg[1:].split("=", 1)[0], None))
            continue
        if "=" not in arg:
            raise ConfigError("cannot parse override %r (expected key=value)" % arg)
        key, text = arg.split("=", 1)
        add = key.startswith("+")
        key = key.lstrip("+")
        group_dir = os.path.isdir(os.path.join(conf_dir, key.replace(".", "/")))
        if group_dir and "." not in key:
            if add:
                appended.append({key: parse_value(text)})
            else:
                choices[key] = parse_value(text)
        else:
            values.append(("+" if add else "=", key, parse_value(text)))

    top = list(primary_defaults) + appended
    known_groups = {_parse_default(e)[0] for e in top}
    for g in choices:
        if g not in known_groups:
            raise ConfigError("could not override %r: no such group in the defaults list (use +%s=... to add it)" % (g, g))

    # phase 1: ``override /group: option`` directives anywhere in the tree (the command line wins)
    def collect(entries, seen):
        for e in entries:
            group, option, _, is_override = _parse_default(e)
            if group is None or group.startswith("hydra"):
                continue
            if is_override:
                seen.setdefault(group, option)
                continue
            opt = choices.get(group, seen.get(group, option))
            if opt is None:
                continue
            _, sub, _ = _load_file(conf_dir, "%s/%s" % (group, opt))
            collect(sub, seen)
    nested = {}
    for _ in range(2):            # an override may select a file that carries further overrides
        collect(top, nested)
    final = dict(nested)
    final.update(choices)

    # phase 2: merge.  The primary config's own content goes first (Hydra 1.0 / 1.1 order for a primary config
    # without ``_self_``), then the defaults in list order -- ``+expt=...`` is appended last and wins
    root = {}
    _merge(root, primary)

    def add(entries, parent_pkg):
        for e in entries:
            group, option, package, is_override = _parse_default(e)
            if group is None or is_override or group.startswith("hydra"):
                continue
            opt = final.get(group, option)
            if opt is None:
                continue
            body, sub, header = _load_file(conf_dir, "%s/%s" % (group, opt))
            pkg = package or header or group.replace("/", ".")
            if pkg != "_global_" and parent_pkg not in (None, "_global_") and package is not None:
                pkg = parent_pkg + "." + pkg
            _place(root, pkg, body)
            add(sub, pkg)
    add(top, None)
    root.pop("hydra", None)

    for op, key, value in values:
        if op == "~":
            node, last = _set_path(root, key, None, create=False)
            node.pop(last, None)
            continue
        try:
            _lookup(root, key)
            exists = True
        except (KeyError, IndexError, ValueError):
            exists = False
        if op == "=" and not exists:
            raise ConfigError("could not override %r: key not in the config (use +%s=... to add it)" % (key, key))
        node, last = _set_path(root, key, None, create=True)
        node[last] = value
    return to_cfg(resolve(root))
