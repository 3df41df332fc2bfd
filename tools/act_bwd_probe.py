"""Which layers still launch a stand-alone activation backward (gz::act_bwd_kernel) in one optimizer cycle: shape and
Python caller of every functional._act_bwd_raw call.     python tools/act_bwd_probe.py hologan [img_size]"""
import os
import sys
import traceback
from collections import Counter

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench      # noqa: E402
from lightning_gan_zoo_amd.functional import _base, _conv, _hologan, _norm      # noqa: E402

args = sys.argv[1:]
expt = args[0] if args else "hologan"
img = int(args[1]) if len(args) > 1 else bench.NATIVE_IMG_SIZE.get(expt, 64)
batch = {"dc_gan": 128, "hologan": 64, "wgan_gp": 256, "wgan": 512, "gan_stability_r1": 64}[expt]
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
seen, on = Counter(), [False]
real = _base._act_bwd_raw


def spy(g, out, act, slope):
    if on[0]:
        st = [f"{os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in traceback.extract_stack()[:-1]
              if "lightning_gan_zoo_amd" in f.filename][-3:]
        seen[(tuple(g.shape), act, " <- ".join(reversed(st)))] += 1
    return real(g, out, act, slope)


for m in (_base, _conv, _hologan, _norm):
    m._act_bwd_raw = spy
module, trainer = bench.build_trainer(expt, batch, dev, 1, img_size=img)
b = bench.synthetic_batch(batch, dev, 0, img)
n = len(trainer.order)
for _ in range(n):
    trainer.step(b)
on[0] = True
for _ in range(n):
    trainer.step(b)
trainer.finish()
torch.cuda.synchronize()
for (shape, act, who), c in sorted(seen.items(), key=lambda kv: -kv[1] * torch.Size(kv[0][0]).numel()):
    print("%3d  act %d  %-24s %s" % (c, act, shape, who))
print("total", sum(seen.values()))
