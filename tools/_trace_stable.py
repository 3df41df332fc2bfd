import os, sys, json
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenario
from test_oracle_golden import load_golden
from test_parity_gpu import build_product_step, set_alpha
from lightning_gan_zoo_amd import functional as F
log = []
def wrap(name):
    orig = getattr(F, name)
    def f(*a, **k):
        out = orig(*a, **k)
        i = len(log)
        t = out if torch.is_tensor(out) else out[0]
        rec = {"op": name, "i": i, "shape": list(t.shape), "fwd": float(t.detach().double().norm()), "bwd": None}
        log.append(rec)
        if t.requires_grad:
            t.register_hook(lambda g, rec=rec: rec.__setitem__("bwd", float(g.double().norm())))
        return out
    setattr(F, name, f)
for n in ("conv2d", "conv_transpose2d", "conv_transpose3d", "adain_act_packed", "instance_norm_act", "linear_act",
          "rigid_resample", "spectral_normalize", "bce_logits_mean", "mse_mean"):
    wrap(n)
inputs, golden, cond = load_golden("hologan", "full", stable=True)
step = build_product_step("hologan", "full", stable=True)
out = scenario.run_scenario(step, inputs, "cuda", full=False, set_alpha=set_alpha, stable=True)
torch.cuda.synchronize()
json.dump(log, open(sys.argv[1], "w"))
