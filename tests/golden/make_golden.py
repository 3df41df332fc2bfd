#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the UNMODIFIED reference
(/root/reference, imported through oracle/ref_import.py) on CPU, fp32.

Run in the build container only (the reference does not exist on the GPU box):

    python tests/golden/make_golden.py            # all fixtures
    python tests/golden/make_golden.py dc_gan     # one experiment

Each fixture holds the scenario inputs (``in/...``) and everything
tests/scenario.py records (``out/...``).  'tiny' fixtures store full tensors,
'full' fixtures (features 64) store (norm, sum, 16 samples) summaries.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))            # tests/
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))  # repo root

from oracle import ref_import                        # noqa: E402
from lightning_gan_zoo_amd.config import make_cfg    # noqa: E402
import scenario                                      # noqa: E402


class _TorchProxy:
    """Forwards to torch but serves a queued tensor from ``rand`` -- pins the
    gradient-penalty alpha the reference draws at core/utils/utils.py:41."""

    def __init__(self):
        self.alpha = None

    def rand(self, shape, *a, **k):
        assert self.alpha is not None and tuple(self.alpha.shape) == tuple(shape)
        return self.alpha.clone()

    def __getattr__(self, name):
        return getattr(torch, name)


def build_reference_step(expt, size):
    ns = ref_import.load_reference()
    feats, bs, zdim = scenario.SIZES[size]
    cfg = make_cfg(expt, module_root="core", batch_size=bs, features=feats, noise_dim=zdim)
    cfg = ref_import.to_attr(cfg)
    if expt == "hologan":
        cfg.generator["gpu"] = False
    torch.manual_seed(42)
    cls = ref_import._locate(cfg.model.lm["_target_"])
    return cls(cfg, logging_dir=None), ns


def main(argv):
    torch.set_num_threads(4)
    expts = argv or list(scenario.STD_EXPTS)
    for expt in expts:
        for size in ("tiny", "full"):
            step, ns = build_reference_step(expt, size)
            proxy = _TorchProxy()
            ns.utils.torch = proxy

            def set_alpha(_step, alpha, proxy=proxy):
                proxy.alpha = alpha

            inputs = scenario.make_inputs(expt, size)
            out = scenario.run_scenario(step, inputs, "cpu", full=(size == "tiny"),
                                        set_alpha=set_alpha)
            ns.utils.torch = torch
            blob = {"in/" + k: v.numpy() for k, v in inputs.items()}
            blob.update({"out/" + k: np.asarray(v) for k, v in out.items()})
            path = os.path.join(HERE, f"{expt}_{size}.npz")
            np.savez_compressed(path, **blob)
            print(f"{path}: {len(blob)} arrays, {os.path.getsize(path) / 1e6:.2f} MB, "
                  f"loss_d0={out['loss_d0']:.6f} loss_g0={out['loss_g0']:.6f} "
                  f"loss_d1={out['loss_d1']:.6f} loss_g1={out['loss_g1']:.6f}")


if __name__ == "__main__":
    main(sys.argv[1:])
