"""Which framework (aten) operators still launch kernels inside one optimizer cycle: torch.profiler over a few
eager cycles of an experiment, grouped by input shape.     python tools/aten_ops_probe.py hologan"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench      # noqa: E402

expt = sys.argv[1] if len(sys.argv) > 1 else "hologan"
batch = {"dc_gan": 512, "hologan": 64, "wgan_gp": 256, "wgan": 512, "gan_stability_r1": 64}[expt]
dev = torch.device("cuda", 0)
module, trainer = bench.build_trainer(expt, batch, dev, 1, img_size=bench.NATIVE_IMG_SIZE.get(expt, 64))
b = bench.synthetic_batch(batch, dev, 0, bench.NATIVE_IMG_SIZE.get(expt, 64))
n = len(trainer.order)
for _ in range(2 * n):
    trainer.step(b)
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], record_shapes=True, with_stack=False) as prof:
    for _ in range(n):
        trainer.step(b)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True)
        if e.key.startswith("aten::") and e.key.split("::")[1] in
        ("fill_", "zero_", "sum", "add", "add_", "copy_", "mul", "mul_", "div", "repeat", "cat", "clone", "neg", "mean")]
rows.sort(key=lambda e: -e.count)
for e in rows[:40]:
    print("%4d  %-14s %s" % (e.count, e.key, str(e.input_shapes)[:140]))
