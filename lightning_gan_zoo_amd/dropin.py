"""Make the reference's ``_target_`` strings resolve to the HIP path.

The reference reaches its hot path only through dotted names in its config tree (``core.lightning_module.DCGAN``,
``core.models.standard_networks.Generator``, ... -- conf/expt/*.yaml), resolved by ``hydra.utils.instantiate`` with an
ordinary import.  ``install()`` puts a finder in front of ``sys.meta_path`` that answers exactly the hot-path module
names with this package's modules (the SAME module objects as ``lightning_gan_zoo_amd.core...`` -- no second copy)
and leaves every other ``core.*`` name alone, so inside the reference's tree its callbacks, figures and metrics keep
importing from the reference's own ``core`` package.  Where there is no reference tree (this repo's own runner, the
GPU box) a second finder at the END of ``sys.meta_path`` serves the rest of the ``core`` namespace from this package.

    # in the reference: one line at the top of run_network.py
    import lightning_gan_zoo_amd.dropin; lightning_gan_zoo_amd.dropin.install()
    # or without touching it
    python -m lightning_gan_zoo_amd.dropin run_network.py +expt=dc_gan dataset=celeb_a
"""
import importlib
import importlib.abc
import importlib.machinery
import sys

PACKAGE = "lightning_gan_zoo_amd"
HOT_PATH_MODULES = (
    "core.lightning_module",                                   # DCGAN / WGAN / WGANGP / HOLOGAN / GANStabilityR1
    "core.models.standard_networks",
    "core.models.hologan_generator",
    "core.models.hologan_discriminator",
    "core.utils.utils",                                        # gradient_penalty, compute_grad2
    "core.utils.hologan",                                      # create_hologan_lr_scheduler
    "core.submodules.gan_stability.models.resnet",
)


class _Alias(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def __init__(self, names=None):
        self.names = names          # None: the whole ``core`` namespace (fallback finder)

    def find_spec(self, fullname, path=None, target=None):
        if self.names is not None:
            if fullname not in self.names:
                return None
        elif fullname != "core" and not fullname.startswith("core."):
            return None
        real = PACKAGE + "." + fullname
        try:
            mod = importlib.import_module(real)
        except ModuleNotFoundError as e:
            if e.name and real.startswith(e.name):
                return None         # this package has no such module: not ours to answer
            raise
        return importlib.machinery.ModuleSpec(fullname, self, is_package=hasattr(mod, "__path__"))

    def create_module(self, spec):
        return sys.modules[PACKAGE + "." + spec.name]       # the existing module object, not a copy

    def exec_module(self, module):
        pass


_installed = []


def install():
    """Idempotent.  Returns the list of redirected module names."""
    if not _installed:
        front, back = _Alias(frozenset(HOT_PATH_MODULES)), _Alias(None)
        sys.meta_path.insert(0, front)
        sys.meta_path.append(back)
        _installed.extend([front, back])
        for name in HOT_PATH_MODULES:           # a reference module imported earlier must not win
            sys.modules.pop(name, None)
    return list(HOT_PATH_MODULES)


def uninstall():
    for f in _installed:
        if f in sys.meta_path:
            sys.meta_path.remove(f)
    del _installed[:]
    for name in list(sys.modules):
        if name == "core" or name.startswith("core."):
            mod = sys.modules[name]
            if getattr(mod, "__name__", "").startswith(PACKAGE + "."):
                del sys.modules[name]


if __name__ == "__main__":
    import runpy
    if len(sys.argv) < 2:
        raise SystemExit("usage: python -m lightning_gan_zoo_amd.dropin <script.py> [args...]")
    install()
    script = sys.argv[1]
    sys.argv = sys.argv[1:]
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(script)))
    runpy.run_path(script, run_name="__main__")
