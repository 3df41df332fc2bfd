#!/usr/bin/env python3
"""Algorithmic conv / GEMM FLOP (2 x MAC) of one optimizer cycle per sample, traced on the CPU oracle with
torch's FlopCounterMode as PyTorch autograd executes it (the way SURVEY.md 8-d's figures were obtained from the
reference).  Used for the configurations SURVEY has no figure for (HoloGAN EXT-128).

    python tests/diagnostics/flop_trace.py hologan 64      # -> 29.06 G
    python tests/diagnostics/flop_trace.py hologan 128
"""
import os
import sys

import torch
from torch.utils.flop_counter import FlopCounterMode

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from lightning_gan_zoo_amd.config import locate, make_cfg      # noqa: E402
from lightning_gan_zoo_amd.harness import optimizer_schedule, toggle_optimizer      # noqa: E402


def main():
    expt, img = sys.argv[1], int(sys.argv[2])
    bs = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    cfg = make_cfg(expt, module_root="oracle.reference_cpu", batch_size=bs, img_size=img)
    torch.manual_seed(0)
    step = locate(cfg.model.lm["_target_"])(cfg, None)
    opts = step.configure_optimizers()
    order = optimizer_schedule([o["frequency"] for o in opts])
    real = torch.rand(bs, 3, img, img) * 2 - 1
    labels = torch.zeros(bs, dtype=torch.int64)
    total = 0
    for i, idx in enumerate(order):
        toggle_optimizer(step, idx)
        with FlopCounterMode(display=False) as fc:
            loss = step.training_step((real, labels), i, idx)
            loss.backward()
        n = fc.get_total_flops()
        print("optimizer_idx %d: %.4f GFLOP per sample" % (idx, n / bs / 1e9))
        total += n
        step.zero_grad()
    print("%s %dx%d cycle of %d batches: %.4f GFLOP per sample (%d FLOP)" % (expt, img, img, len(order), total / bs / 1e9, total // bs))


if __name__ == "__main__":
    main()
