"""Which framework (aten) operators still launch kernels inside one optimizer cycle: torch.profiler over a few
eager cycles of an experiment, grouped by input shape and (for the memset / memcpy / add family) by Python caller.
    python tools/aten_ops_probe.py hologan [img_size] [--sync]      (--sync: under ddp.GradSync, single-rank RCCL)"""
import os
import sys
from collections import Counter

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench      # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
expt = args[0] if args else "hologan"
img = int(args[1]) if len(args) > 1 else bench.NATIVE_IMG_SIZE.get(expt, 64)
sync = "--sync" in sys.argv
batch = {"dc_gan": 128, "hologan": 64, "wgan_gp": 256, "wgan": 512, "gan_stability_r1": 64}[expt]
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)


def run():
    module, trainer = bench.build_trainer(expt, batch, dev, 1, force_sync=sync, img_size=img)
    b = bench.synthetic_batch(batch, dev, 0, img)
    n = len(trainer.order)
    for _ in range(2 * n):
        trainer.step(b)
    trainer.finish()
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], record_shapes=True,
                                with_stack=True) as prof:
        for _ in range(n):
            trainer.step(b)
        trainer.finish()
        torch.cuda.synchronize()
    names = ("fill_", "zero_", "zeros", "sum", "add", "add_", "copy_", "mul", "mul_", "div", "repeat", "cat", "clone",
             "neg", "mean", "empty_like", "zeros_like")
    rows = [e for e in prof.key_averages(group_by_input_shape=True)
            if e.key.startswith("aten::") and e.key.split("::")[1] in names]
    rows.sort(key=lambda e: -e.count)
    for e in rows[:40]:
        print("%4d  %-14s %s" % (e.count, e.key, str(e.input_shapes)[:140]))
    who = Counter()
    for ev in prof.events():
        if ev.name in ("aten::zero_", "aten::fill_", "aten::copy_", "aten::add_", "aten::add", "c10d::allreduce_"):
            stack = [s for s in (ev.stack or []) if "lightning_gan_zoo_amd" in s or "torch/distributed" in s
                     or "autograd" in s]
            who[(ev.name, " <- ".join(s.split("/")[-1][:60] for s in stack[:3]))] += 1
    print("--- callers")
    for (name, st), c in who.most_common(40):
        print("%4d  %-12s %s" % (c, name, st))


if sync:
    with bench.single_rank_rccl(dev):
        run()
else:
    run()
