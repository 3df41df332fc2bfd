"""Per-op forward / backward norms of the stable HoloGAN scenario (tests/scenario.py), written as JSON: run it under
two builds / environment switches and diff the files to find the first op at which two runs part (round 2: a
3e-7-margin LeakyReLU decision in the discriminator, taken differently under two split-K plans of the same
ConvTranspose3d).      python tools/hologan_op_trace.py out.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenario                                                     # noqa: E402
from test_oracle_golden import load_golden                          # noqa: E402
from test_parity_gpu import build_product_step, set_alpha           # noqa: E402
from lightning_gan_zoo_amd import functional as F                   # noqa: E402

log = []


def wrap(name):
    orig = getattr(F, name)

    def traced(*a, **k):
        out = orig(*a, **k)
        t = out if torch.is_tensor(out) else out[0]
        rec = {"op": name, "i": len(log), "shape": list(t.shape), "fwd": float(t.detach().double().norm()), "bwd": None}
        log.append(rec)
        if t.requires_grad:
            t.register_hook(lambda g, rec=rec: rec.__setitem__("bwd", float(g.double().norm())))
        return out
    setattr(F, name, traced)


for n in ("conv2d", "conv_transpose2d", "conv_transpose3d", "adain_act_packed", "adain_const_act", "instance_norm_act", "linear_act",
          "rigid_resample", "spectral_normalize", "bce_logits_mean", "mse_mean"):
    wrap(n)
inputs, golden, cond = load_golden("hologan", "full", stable=True)
step = build_product_step("hologan", "full", stable=True)
scenario.run_scenario(step, inputs, "cuda", full=False, set_alpha=set_alpha, stable=True)
torch.cuda.synchronize()
json.dump(log, open(sys.argv[1], "w"))
