"""Where do the remaining scalar fill_ / copy_ launches of a cycle come from?  python tools/fill_probe.py [expt] [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench      # noqa: E402

expt = sys.argv[1] if len(sys.argv) > 1 else "dc_gan"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda", 0)
module, trainer = bench.build_trainer(expt, batch, dev, 1, img_size=bench.NATIVE_IMG_SIZE.get(expt, 64))
b = bench.synthetic_batch(batch, dev, 0, bench.NATIVE_IMG_SIZE.get(expt, 64))
n = len(trainer.order)
for _ in range(2 * n):
    trainer.step(b)
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA],
                            record_shapes=True, with_stack=True) as prof:
    for _ in range(n):
        trainer.step(b)
    torch.cuda.synchronize()
seen = {}
for e in prof.events():
    if e.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::cat", "aten::clone") and e.device_time_total > 0:
        stack = [s for s in (e.stack or []) if "lightning_gan_zoo_amd" in s or "bench.py" in s or "torch/autograd" in s][:4]
        key = (e.name, str(e.input_shapes)[:60], tuple(stack))
        seen[key] = seen.get(key, 0) + 1
for (name, shapes, stack), cnt in sorted(seen.items(), key=lambda kv: -kv[1]):
    print("%3d %-12s %s" % (cnt, name, shapes))
    for s in stack:
        print("        ", s[:150])
