"""TEST INFRASTRUCTURE ONLY -- import the *unmodified* reference (read-only at
/root/reference) on CPU so that golden vectors can be generated from it.

The reference's step classes (core/lightning_module.py:1-33) import
pytorch_lightning, torchvision, hydra, pytorch3d and an un-vendored
``core.submodules.tps_deformation`` -- none of which exist in this image.  This
module registers minimal stand-ins in ``sys.modules`` so
that ``DCGAN/WGAN/WGANGP/HOLOGAN.training_step`` and ``configure_optimizers``
run exactly as written.  Nothing here is product code and nothing here travels
to the GPU box in a usable form (``/root/reference`` does not exist there);
``available()`` tells callers whether the reference can be imported at all.
"""
import importlib
import os
import sys
import types

import torch
from torch import nn

REFERENCE_ROOT = os.environ.get("GZ_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "core", "lightning_module.py"))


class AttrDict(dict):
    """dict with attribute access, standing in for an OmegaConf node."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:  # pragma: no cover
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def to_attr(obj):
    if isinstance(obj, dict):
        return AttrDict({k: to_attr(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return [to_attr(v) for v in obj]
    return obj


def _locate(path):
    mod, _, name = path.rpartition(".")
    return getattr(importlib.import_module(mod), name)


def instantiate(cfg, *args, **kwargs):
    """hydra.utils.instantiate stand-in: locate ``_target_`` and call it with
    the node's remaining keys merged with the call-site kwargs."""
    cfg = dict(cfg)
    target = _locate(cfg.pop("_target_"))
    merged = {**cfg, **kwargs}
    return target(*args, **merged)


class _LightningModule(nn.Module):
    """pl.LightningModule stand-in: records ``self.log`` calls, device = cpu."""

    def __init__(self):
        super().__init__()
        self.logged = {}

    def log(self, key, value, *a, **k):
        self.logged[key] = value.detach().clone() if torch.is_tensor(value) else value

    @property
    def device(self):
        return torch.device("cpu")


class _Anything:
    def __init__(self, *a, **k):
        pass

    def __call__(self, x):
        return x


def _module(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_LOADED = None


def load_reference():
    """Returns a namespace with the reference's own modules:
    ``.lightning_module``, ``.standard_networks``, ``.utils``,
    ``.hologan_generator``, ``.hologan_discriminator``, ``.hologan_utils``."""
    global _LOADED
    if _LOADED is not None:
        return _LOADED
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    pl = _module("pytorch_lightning", LightningModule=_LightningModule,
                 seed_everything=lambda s: torch.manual_seed(s))
    tv_t = _module("torchvision.transforms", Compose=_Anything, Resize=_Anything,
                   ToTensor=_Anything, Normalize=_Anything)
    tv_u = _module("torchvision.utils", make_grid=lambda x, **k: x)
    _module("torchvision", transforms=tv_t, utils=tv_u)
    hu = _module("hydra.utils", instantiate=instantiate, call=instantiate)
    _module("hydra", utils=hu, main=lambda **k: (lambda f: f))
    r3d_names = ["look_at_view_transform", "OpenGLPerspectiveCameras", "PointLights",
                 "DirectionalLights", "Materials", "RasterizationSettings", "MeshRenderer",
                 "MeshRasterizer", "SoftPhongShader", "SoftSilhouetteShader",
                 "TexturesVertex", "FoVOrthographicCameras", "FoVPerspectiveCameras"]
    r3d = _module("pytorch3d.renderer", **{n: _Anything for n in r3d_names})
    s3d = _module("pytorch3d.structures", Meshes=_Anything, Pointclouds=_Anything)
    t3d = _module("pytorch3d.transforms", quaternion_to_matrix=_Anything,
                  euler_angles_to_matrix=_Anything)
    _module("pytorch3d", renderer=r3d, structures=s3d, transforms=t3d)
    tps = _module("core.submodules.tps_deformation.tps", functions=_Anything)
    # the parent packages exist in the reference tree except tps_deformation
    importlib.import_module("core.submodules")
    _module("core.submodules.tps_deformation", tps=tps)

    import numpy as np
    if not hasattr(np, "float"):  # hologan_generator.py:88 uses the removed alias
        np.float = float

    ns = types.SimpleNamespace()
    ns.standard_networks = importlib.import_module("core.models.standard_networks")
    ns.utils = importlib.import_module("core.utils.utils")
    ns.hologan_generator = importlib.import_module("core.models.hologan_generator")
    ns.hologan_discriminator = importlib.import_module("core.models.hologan_discriminator")
    ns.hologan_utils = importlib.import_module("core.utils.hologan")
    ns.lightning_module = importlib.import_module("core.lightning_module")
    ns.pl = pl
    _LOADED = ns
    return ns
