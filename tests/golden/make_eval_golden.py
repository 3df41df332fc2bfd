#!/usr/bin/env python3
"""tests/golden/eval_metrics.npz: FID / KID arithmetic of the UNMODIFIED reference on seeded activations.

Build container only.  Imports core/callback_inception_metrics.py (its own polynomial_mmd_averages, :15-133)
and core/submodules/gan_stability/metrics/fid_score.py (calculate_frechet_distance, :25-80) with stand-ins for
the packages that are not installed (pytorch_fid, imageio, torchvision.models); neither function touches them.
"""
import importlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_import  # noqa: E402


def activations(seed, n, d, shift):
    rng = np.random.RandomState(seed)
    basis = rng.randn(d, d) / np.sqrt(d)
    return rng.randn(n, d).dot(basis) + shift * rng.rand(d)


def main():
    ref_import.load_reference()

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    anything = type("Anything", (), {"__init__": lambda self, *a, **k: None})
    mod("pytorch_fid")
    mod("pytorch_fid.fid_score", get_activations=anything, calculate_frechet_distance=anything)
    mod("pytorch_fid.inception", InceptionV3=anything)
    mod("imageio")
    mod("tqdm", tqdm=lambda it, **k: _Bar(it))
    sys.modules["pytorch_lightning"].callbacks = mod("pytorch_lightning.callbacks")
    sys.modules["pytorch_lightning.callbacks"].base = mod("pytorch_lightning.callbacks.base", Callback=object)
    import torch
    blocks = {n: type(n, (torch.nn.Module,), {}) for n in ("InceptionA", "InceptionC", "InceptionE")}
    inc = mod("torchvision.models.inception", inception_v3=anything, **blocks)
    tvm = mod("torchvision.models", inception_v3=anything, inception=inc)
    mod("torchvision.models.utils", load_state_dict_from_url=anything)
    sys.modules["torchvision"].models = tvm
    cb = importlib.import_module("core.callback_inception_metrics")
    fs = importlib.import_module("core.submodules.gan_stability.metrics.fid_score")

    real = activations(1, 400, 48, 0.0)
    fake = activations(2, 360, 48, 0.3)
    out = {"fid": fs.calculate_frechet_distance(np.mean(real, 0), np.cov(real, rowvar=False),
                                                np.mean(fake, 0), np.cov(fake, rowvar=False))}
    out["fid_same"] = fs.calculate_frechet_distance(np.mean(real, 0), np.cov(real, rowvar=False),
                                                    np.mean(real, 0), np.cov(real, rowvar=False))
    np.random.seed(123)
    mmds, variances = cb.polynomial_mmd_averages(real, fake, n_subsets=7, subset_size=150, output=open(os.devnull, "w"))
    out["kid_mmds"], out["kid_vars"] = mmds, variances
    out["mmd_full"] = np.array(cb.polynomial_mmd(fake[:300], real[:300]))
    np.savez(os.path.join(HERE, "eval_metrics.npz"), **out)
    print({k: np.asarray(v).ravel()[:3] for k, v in out.items()})


class _Bar:
    def __init__(self, it):
        self.it = it

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def __iter__(self):
        return iter(self.it)

    def set_postfix(self, *a, **k):
        pass


if __name__ == "__main__":
    main()
